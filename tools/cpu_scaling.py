#!/usr/bin/env python3
"""tools/cpu_scaling.py [receipts] [threads,threads,...] — per-phase seconds of the CPU baseline (BASELINE.md variant B2: the
C++ oracle, -O3 -march=native, OpenMP) at several thread counts on THIS host, one line per count.  Run once per OpenMP
placement (OMP_PLACES / OMP_PROC_BIND are read when libgomp starts): tools/gpu_cpu_scaling.sh does the sweep.  This is
how the `cpu_baseline` leg of bench.py was tuned; it measures the checker, never the product."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import claims as claims_mod  # noqa: E402
import oracle_lib  # noqa: E402
from tools.synth import SEED_BASE, Tipset  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    counts = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 16, 32, 64, 128]
    sample = int(os.environ.get("CPU_SCALING_SAMPLE", "200000"))
    orc, march = oracle_lib.load_native()
    tip = Tipset(seed=SEED_BASE + 3, n_receipts=n, n_parents=5, dup_permille=20, n_planted=max(1, n // 1000), max_events=4,
                 no_events_permille=0, variety=0)
    expect32 = np.ascontiguousarray(tip.cids[:, 6:38])
    ec1 = claims_mod.EventClaims(tip, indices=np.arange(1))
    ec = claims_mod.EventClaims(tip, indices=np.arange(min(n, sample)))
    tag = "places=%s bind=%s" % (os.environ.get("OMP_PLACES", "-"), os.environ.get("OMP_PROC_BIND", "-"))
    for threads in counts:
        orc.use_threads(threads)
        sec = {}
        st = None
        for _ in range(2):
            if st is not None:
                st.close()
            t0 = time.perf_counter()
            st = orc.store(tip.data, tip.off, tip.lens, tip.cids, threads=threads)
            dt = time.perf_counter() - t0
            sec["store"] = min(sec.get("store", dt), dt)
        t0 = time.perf_counter()
        orc.blake2b256_verify(tip.data, tip.off, tip.lens, expect32, threads=threads)
        sec["cid"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        st.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor, want_touched=False, threads=threads)
        sec["scan"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        st.verify_event_proofs(ec1, mode=2, threads=threads)
        sec["exec"] = time.perf_counter() - t0
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            st.verify_event_proofs(ec, mode=2, threads=threads)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        sec["verify"] = best * (n / ec.n)
        st.close()
        step = sum(sec.values())
        print("%s threads=%3d step=%.4f s  %.2f M proofs/s  %s" % (tag, threads, step, n / step / 1e6,
                                                                   " ".join("%s=%.4f" % kv for kv in sec.items())), flush=True)


if __name__ == "__main__":
    main()
