#!/usr/bin/env python3
"""Diagnostics of the group-of-8 HAMT kernels on the GPU box: how many queries each kernel settles, per-kernel times
(run under `rocprofv3 --kernel-trace --stats`)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import ipc_filecoin_proofs_amd as ipcfp  # noqa: E402

torch.cuda.init()
eng = ipcfp.Engine(0)
T = bench._state_tipset() if len(sys.argv) < 2 else None
w = eng.witness(T.data, T.off, T.lens, T.cids)
keys = [bench._idaddr(int(i)) for i in T.query_ids]
for _ in range(3):
    t0 = time.perf_counter()
    st, loc = w.hamt_get(T.actors_root, 5, "actor_state", keys)
    print("hamt_get", time.perf_counter() - t0, np.bincount(st, minlength=256).nonzero()[0], np.bincount(st)[np.bincount(st).nonzero()[0]])
n = len(T.sc_actor)
cl = ipcfp.pack_storage_claims(T.child_cid, T.state_root, T.child_epoch, T.sc_actor, T.sc_actor_state, T.sc_storage_root, T.sc_slot, T.sc_value)
d_cl = torch.from_numpy(cl.view(np.uint8).reshape(-1)).cuda()
d_st = torch.zeros(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    w.verify_storage_claims_device(d_cl.data_ptr(), n, d_st.data_ptr())
    print("storage", time.perf_counter() - t0)
g = d_st.cpu().numpy()
print(np.bincount(g)[np.bincount(g).nonzero()[0]], np.bincount(g).nonzero()[0])
