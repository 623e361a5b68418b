#!/usr/bin/env python3
"""GPU-side A/B of the K1 (Blake2b-256 CID) kernel variants.  Prints one line per
(mode, workgroup, n) with the HIP-event kernel time and the algorithmic GB/s."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ipc_filecoin_proofs_amd as ipcfp  # noqa: E402
from bench import make_cfg2  # noqa: E402


def run(mode, wg, n, reps=10, block=1024, extra_env=None):
    os.environ["IPCFP_B2B_MODE"] = str(mode)
    os.environ["IPCFP_B2B_WG"] = str(wg)
    for k, v in (extra_env or {}).items():
        os.environ[k] = str(v)
    eng = ipcfp.Engine(0)
    data, off, lens = make_cfg2(n, 1234)
    if block != 1024:
        lens = np.full(n, block, dtype=np.uint32)
    cids = np.zeros((n, 40), dtype=np.uint8)
    cids[:, :6] = np.frombuffer(bytes.fromhex("0171a0e40220"), dtype=np.uint8)
    w = eng.witness(data, off, lens, cids)
    for _ in range(2):
        w.verify_cids_async()
    eng.sync()
    eng.profile_reset()
    eng.profile_enable(True)
    for _ in range(reps):
        w.verify_cids_async()
    eng.profile_enable(False)
    cnt, ms = eng.profile_read("blake2b_cid")
    avg = ms / cnt
    algo = float(lens.astype(np.float64).sum()) + n * 56
    print(f"mode={mode} wg={wg:3d} n={n:8d} block={block:5d} env={extra_env}: {avg*1e3:9.1f} us  "
          f"{algo/avg/1e6:8.1f} GB/s  ({n/avg/1e3:.2f} M CIDs/s)", flush=True)
    w.close()
    eng.close()


if __name__ == "__main__":
    # modes: 0 = hipcc's u64 adds + v_alignbit rotates (default), 1 = explicit carry adds, 2 = rot63 as shift + add,
    #        3 = message words staged in LDS (5 waves/SIMD)
    for n in (100_000, 1_000_000, 4_000_000):
        for mode in (0, 2, 3, 1):
            run(mode, 64, n, reps=5 if n > 1_000_000 else 10)
    for mode in (0, 2, 3):
        run(mode, 64, 1_000_000, block=341)
