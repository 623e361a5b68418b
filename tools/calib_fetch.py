#!/usr/bin/env python3
"""Calibration of rocprofv3's FETCH_SIZE for K1's access pattern (per-lane 16-byte loads, one block per
lane): run K1 over N x 1 KiB blocks (known byte count, far beyond the 256 MiB Infinity Cache) under
`rocprofv3 --pmc FETCH_SIZE` and compare the counter with the bytes actually read."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ipc_filecoin_proofs_amd as ipcfp  # noqa: E402
from bench import make_cfg2  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
eng = ipcfp.Engine(0)
data, off, lens = make_cfg2(n, 99)
cids = np.zeros((n, 40), dtype=np.uint8)
cids[:, :6] = np.frombuffer(bytes.fromhex("0171a0e40220"), dtype=np.uint8)
w = eng.witness(data, off, lens, cids)
for _ in range(3):
    w.verify_cids_async()
eng.sync()
print("K1 launches: 3, blocks:", n, "payload bytes per launch:", int(lens.astype(np.int64).sum()),
      "algorithmic bytes per launch:", int(lens.astype(np.int64).sum()) + n * 56)
