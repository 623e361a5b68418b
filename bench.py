#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X Merkle-witness engine.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches
one rank per GPU through torch.distributed.run (RCCL).  W untimed warm-up steps, then exactly K
timed steps bracketed by barrier + torch.cuda.synchronize() on both sides, MAX over ranks; rank 0
prints ONE JSON line.

Workload (BASELINE.json `metric`: "Merkle proofs verified/sec + HBM GB/s, 1M-receipt synthetic
tipset"): configs[2]'s tipset — 1 000 000 receipts under an Amtv0<Receipt>, one events AMT
(v3, bit width 5) per receipt, 5 parent headers with TxMeta + BLS/secp message AMTs, ≈1.29 M witness
blocks ≈ 0.44 GB — and one EventProof claim per receipt.  A "step" is one full verification pass of
that tipset with every input already resident in HBM:
    K4  rebuild the CID → block index            (load_witness_store)
    K1  Blake2b-256 CID check of every block     (the kernel the roofline is quoted on)
    K6  two-pass event-filter scan of all receipts (topic0 + topic1 + actor_id_filter)
        exec-order reconstruction (TxMeta re-hash + message AMT walks + first-seen dedupe)
        verify_event_proof for every claim (receipt AMT walk + events AMT walk + event compare)
`value` = claims verified per second.  Multi-GPU: the proof batch is sharded by receipt index — rank r
owns the receipts, events and claims of its own 1M-receipt shard (weak scaling) — with no data-path
collective; one RCCL all-gather of the per-shard verdict bytes + CID bitmaps closes each step.

torch is plumbing only (HBM residency of inputs, device sync, torch.distributed); every timed
kernel is launched by libipcfp.so through its C ABI.  DESIGN.md §Measurement has the byte accounting.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md §Chip-level parameters)
METRIC = "Merkle proofs verified/sec + HBM GB/s, 1M-receipt synthetic tipset, 1/2/4/8 GPU"


class DevView:
    """Expose a raw device pointer to torch through __cuda_array_interface__."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}


def splitmix64(seed: int, n: int) -> np.ndarray:
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def make_cfg2(n: int, seed: int):
    """BASELINE.json configs[1]: n blocks of exactly 1024 B = DAG-CBOR byte-string header 59 03 FD +
    1021 PRNG bytes (used by tools/tune_b2b.py and `--workload cid`)."""
    data = splitmix64(seed, n * 128).view(np.uint8).reshape(n, 1024).copy()
    data[:, 0] = 0x59
    data[:, 1] = 0x03
    data[:, 2] = 0xFD
    off = np.arange(n, dtype=np.uint64) * np.uint64(1024)
    lens = np.full(n, 1024, dtype=np.uint32)
    return data.reshape(-1), off, lens


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--receipts", type=int, default=1_000_000, help="receipts per GPU shard")
    ap.add_argument("--cpu-sample", type=int, default=50_000, help="claims the 1-thread cpu_baseline leg verifies")
    ap.add_argument("--cpu-sample-mt", type=int, default=200_000, help="claims the all-cores cpu_baseline leg verifies")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--t2-reps", type=int, default=3, help="repetitions of the PCIe-inclusive window (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import ipc_filecoin_proofs_amd as ipcfp
    from tools.synth import SEED_BASE, Tipset

    eng = ipcfp.Engine(local_rank)
    info = eng.device_info()

    # ---- this rank's shard of the tipset (weak scaling: a fixed number of receipts per GPU) ----
    t_gen = time.perf_counter()
    tip = Tipset(seed=SEED_BASE + 3 + (rank << 24), n_receipts=args.receipts, n_parents=5, dup_permille=20,
                 n_planted=max(1, args.receipts // 1000), max_events=4, no_events_permille=0, variety=0)
    n_claims = len(tip.claim_exec)
    ts, cl, blob, blob_len = ipcfp.pack_event_claims(
        tip.parent_cids, tip.child_cid, tip.parent_epoch, tip.child_epoch, tip.claim_exec, tip.claim_event,
        tip.claim_emitter, tip.exec_order[tip.claim_exec.astype(np.int64)], tip.claim_ntopics, tip.claim_topics,
        tip.claim_datalen, tip.claim_data)
    t_gen = time.perf_counter() - t_gen

    # ---- inputs resident in HBM before the clock starts ----
    t_bytes = torch.from_numpy(tip.data).to(dev)
    t_off = torch.from_numpy(tip.off.view(np.int64)).to(dev)
    t_len = torch.from_numpy(tip.lens.view(np.int32)).to(dev)
    t_cids = torch.from_numpy(tip.cids.reshape(-1)).to(dev)
    t_claims = torch.from_numpy(cl.view(np.uint8).reshape(-1)).to(dev)
    t_blob = torch.from_numpy(blob).to(dev)
    t_status = torch.zeros(n_claims, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    w = eng.witness_device(t_bytes.data_ptr(), tip.data.size, t_off.data_ptr(), t_len.data_ptr(), t_cids.data_ptr(),
                           tip.n_blocks)
    bitmap_bytes = ((tip.n_blocks + 31) // 32) * 4
    t_bitmap = torch.as_tensor(DevView(w.cid_bitmap_ptr, bitmap_bytes), device=dev)
    gather = None
    if world > 1:
        # shards are generated from different seeds, so their block counts differ by a few: every rank pads its
        # message to the longest one (an all-gather needs equal sizes) — shard.PaddedGather, tested on gloo
        from ipc_filecoin_proofs_amd.shard import PaddedGather

        gather = PaddedGather(bitmap_bytes + n_claims, dist, device=dev)
    scan_result = {}

    def step():
        w.rebuild_index()                                                               # K4
        w.verify_cids_async()                                                           # K1
        st, _, m, _ = w.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor,
                                    want_touched=False, counts_only=True)               # K6
        scan_result["status"], scan_result["matches"] = st, m
        w.verify_event_claims_device(ts, t_claims.data_ptr(), n_claims, t_blob.data_ptr(), blob_len,
                                     t_status.data_ptr())                               # exec order + verify
        if world > 1:
            eng.sync()  # K1 runs on the engine's second stream: its bitmap is complete after ctx_sync
            gather.payload[:bitmap_bytes].copy_(t_bitmap)
            gather.payload[bitmap_bytes:bitmap_bytes + n_claims].copy_(t_status)
            gather.run()                                                                 # the one collective

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.profile_reset()
    eng.profile_enable(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    t1 = time.perf_counter()
    eng.profile_enable(False)
    elapsed = t1 - t0
    total_claims = n_claims
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tc = torch.tensor([n_claims], dtype=torch.int64, device=dev)
        dist.all_reduce(tc, op=dist.ReduceOp.SUM)
        total_claims = int(tc.item())

    # ---- what was timed must be right (self-check; parity proper lives in tests/) ----
    status = t_status.cpu().numpy()
    cid_status, n_bad = w.verify_cids()
    if n_bad or not (cid_status == 1).all():
        raise SystemExit("bench self-check failed: a witness CID did not verify")
    if not (status == 1).all():
        raise SystemExit(f"bench self-check failed: {int((status != 1).sum())} honest claims did not verify")
    if scan_result["status"] != 1 or scan_result["matches"] < len(tip.planted):
        raise SystemExit("bench self-check failed: the scan missed planted matches")

    # ---- window T2: the same pass end to end from HOST memory — witness upload over PCIe, repack,
    # CID index, K1, scan, claim upload, verify, status bytes and CID verdicts back.  `value` above is window T3
    # (inputs resident in HBM); T2 is what the ≥10x-over-CPU claim is judged on.
    t2 = None
    if world == 1 and args.t2_reps > 0:
        reps = []
        for _ in range(args.t2_reps + 1):  # the first repetition warms the pinned ring and the allocator
            torch.cuda.synchronize()
            ta = time.perf_counter()
            w2 = eng.witness(tip.data, tip.off, tip.lens, tip.cids)
            tb = time.perf_counter()
            w2.verify_cids_async()
            st2, _, nm2, _ = w2.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor,
                                            want_touched=False, counts_only=True)
            tc_ = time.perf_counter()
            status2 = w2.verify_event_claims(ts, cl, blob, blob_len)
            td = time.perf_counter()
            cs2, nbad2 = w2.cid_results()
            te = time.perf_counter()
            w2.close()
            if st2 != 1 or nbad2 or not np.array_equal(status2, status) or nm2 != scan_result["matches"]:
                raise SystemExit("bench self-check failed: the from-host pass differs from the resident one")
            reps.append({"total": te - ta, "witness_create_h2d_repack_index": tb - ta, "k1_launch_scan": tc_ - tb,
                         "claims_h2d_verify_status_d2h": td - tc_, "cid_verdicts_d2h": te - td})
        reps = reps[1:]
        best = min(reps, key=lambda r: r["total"])
        h2d_bytes = int(tip.data.size + tip.off.nbytes + tip.lens.nbytes + tip.cids.nbytes + cl.nbytes + blob_len)
        t2 = {"value": n_claims / best["total"], "unit": "proofs/s", "ms_per_tipset": best["total"] * 1e3,
              "ms_phases": {k: round(v * 1e3, 3) for k, v in best.items() if k != "total"},
              "h2d_bytes": h2d_bytes, "h2d_GBps_if_all_transfer": h2d_bytes / best["total"] / 1e9, "reps": len(reps),
              "ms_all_reps": [round(r["total"] * 1e3, 3) for r in reps],
              "note": "pageable host numpy buffers in, host status bytes out; claims in packed binary form"}

    # ---- the full scan result, untimed, for the oracle cross-check of the cpu_baseline leg ----
    gpu_scan = None
    if world == 1 and not args.no_cpu_baseline:
        gs, ghas, gm, _ = w.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor,
                                        want_touched=False)
        gpu_scan = (gs, ghas, gm)

    kern = {}
    for k in ("blake2b_cid", "cid_index", "event_scan", "replay", "event_verify", "exec_order"):
        cnt, ms = eng.profile_read(k)
        kern[k] = {"launches": cnt, "ms_per_step": ms / args.steps}
    k_launches, k_ms = eng.profile_read("blake2b_cid")
    k_avg_ms = k_ms / max(k_launches, 1)
    # algorithmic bytes of one K1 launch: every block's payload + its 40-byte claimed CID + 16 B (offset, len, id)
    algo_bytes = float(tip.lens.astype(np.float64).sum() + tip.n_blocks * (40 + 16))
    achieved = algo_bytes / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
    # HBM traffic of one K1 launch: FETCH_SIZE needs a rocprofv3 --pmc pass of its own, so the figure is
    # the committed measurement of this very command (profiles/r01_k1_traffic.json) and only reported
    # when the workload is the one that was profiled
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_k1_traffic.json")) as f:
            tr = json.load(f)
        if tr["workload"]["witness_blocks"] == tip.n_blocks and tr["workload"]["payload_bytes"] == tip.stats["payload_bytes"]:
            traffic = tr["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass

    if rank == 0:
        out = {
            "metric": METRIC,
            "value": total_claims * args.steps / elapsed,
            "unit": "proofs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[2] (the 1M-receipt tipset the metric is quoted on): %d receipts per GPU (Amtv0<Receipt> + one Amt<StampedEvent> each, 5 parent "
                            "headers with TxMeta and message AMTs), %d witness blocks, %.3f GB; one EventProof claim per "
                            "receipt; step = CID index + Blake2b-256 CID check of every block + event-filter scan + "
                            "exec-order reconstruction + verify_event_proof of every claim" %
                            (args.receipts, tip.n_blocks, tip.stats["payload_bytes"] / 1e9),
                "receipts_per_gpu": args.receipts,
                "claims_per_gpu": n_claims,
                "witness_blocks_per_gpu": tip.n_blocks,
                "witness_bytes_per_gpu": tip.stats["payload_bytes"],
                "scan_matches": int(scan_result["matches"]),
                "sharding": ("receipt-index shard per rank; one RCCL all-gather of verdict bytes + CID bitmaps per step"
                             if world > 1 else "single GPU"),
                "device": info["name"],
                "setup_seconds_untimed": round(t_gen, 2),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_blake2b256_cid",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel_avg_ms": k_avg_ms,
                "launches": k_launches,
                "algorithmic_bytes_per_launch": algo_bytes,
                "note": "VALU-bound on gfx950 (≈19 int ops/byte; profiles/r01_ubench_valu_rates.log): see DESIGN.md §K1",
            },
            "kernels_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in kern.items()},
        }
        if t2 is not None:
            out["value_T2"] = t2["value"]
            out["window_T2"] = t2
        out["window"] = "T3 (inputs resident in HBM; index rebuilt and every cached enumeration dropped each step)"
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(tip, status, args.cpu_sample, args.cpu_sample_mt, cid_status, gpu_scan)
            cb = out["cpu_baseline"]
            out["speedup_vs_cpu_all_cores"] = {"T3": out["value"] / cb["value"],
                                               "T2": (t2["value"] / cb["value"]) if t2 else None}
        print(json.dumps(out))
    w.close()
    eng.close()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(tip, gpu_status, sample, sample_mt=None, gpu_cid_status=None, gpu_scan=None):
    """The C++ oracle — a restatement of the reference path; the Rust reference cannot be built in this image —
    compiled on THIS box with -O3 -march=native and timed on its host cores.  Variant B2 of BASELINE.md §2
    (witness store and execution order built once per tipset), single thread AND all cores; `value` is the
    all-cores number, the strongest CPU figure.  Variant B1 (as written: execution order rebuilt per proof,
    quadratic) is timed on a reduced tipset and labelled as such.  The fixed parts of the step run on the FULL
    tipset; the per-claim verifier runs on a bounded sample and is scaled linearly.  Everything the oracle
    computes on the way doubles as a check of what the GPU reported."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import claims as claims_mod
    import oracle_lib
    from tools.synth import Tipset

    orc, march = oracle_lib.load_native()
    cores = orc.num_procs()
    n = len(tip.claim_exec)
    sample = min(sample, n)
    sample_mt = min(sample_mt or sample, n)
    expect32 = np.ascontiguousarray(tip.cids[:, 6:38])
    ecs = {1: claims_mod.EventClaims(tip, indices=np.arange(sample))}
    ecs[0] = ecs[1] if sample_mt == sample else claims_mod.EventClaims(tip, indices=np.arange(sample_mt))
    ec1 = claims_mod.EventClaims(tip, indices=np.arange(1))
    legs = {}
    for threads in (1, 0):
        sec = {}
        st = None
        for _ in range(2):  # the second build finds the allocator's arenas grown; keep the better of the two
            if st is not None:
                st.close()
            t0 = time.perf_counter()
            st = orc.store(tip.data, tip.off, tip.lens, tip.cids, threads=threads)
            dt = time.perf_counter() - t0
            sec["store_build"] = min(sec.get("store_build", dt), dt)
        t0 = time.perf_counter()
        ok, good = orc.blake2b256_verify(tip.data, tip.off, tip.lens, expect32, threads=threads)
        sec["cid_check"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        s, has, trip, _ = st.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor,
                                         want_touched=False, threads=threads)
        sec["event_scan"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        st.verify_event_proofs(ec1, mode=2, threads=threads)  # builds the execution order kept in the store
        sec["exec_order"] = time.perf_counter() - t0
        e = ecs[threads]
        t0 = time.perf_counter()
        got = st.verify_event_proofs(e, mode=2, threads=threads)
        sec["verify_sample"] = time.perf_counter() - t0
        st.close()
        # ---- the oracle's results against the GPU's ----
        if good != int((gpu_cid_status == 1).sum() if gpu_cid_status is not None else tip.n_blocks) or s != 1:
            raise SystemExit("cpu_baseline: the oracle's CID check / scan status differs from the GPU's")
        if gpu_cid_status is not None and not np.array_equal(ok, (gpu_cid_status == 1).astype(np.uint8)):
            raise SystemExit("cpu_baseline: per-block CID verdicts differ from the GPU's")
        if not np.array_equal(got, gpu_status[:e.n]):
            raise SystemExit("cpu_baseline: the oracle's verdicts differ from the GPU's on the sample")
        if gpu_scan is not None:
            gs, ghas, gm = gpu_scan
            same = (gs == s and np.array_equal(ghas, has) and len(gm) == len(trip) and
                    np.array_equal(gm["exec_index"], trip[:, 0]) and np.array_equal(gm["event_index"], trip[:, 1]) and
                    np.array_equal(gm["emitter"], trip[:, 2]))
            if not same:
                raise SystemExit("cpu_baseline: the oracle's scan (has-match map / match list) differs from the GPU's")
        fixed = sec["store_build"] + sec["cid_check"] + sec["event_scan"] + sec["exec_order"]
        sec["step"] = fixed + sec["verify_sample"] * (n / e.n)
        sec["verify_sample_claims"] = e.n
        legs[threads] = sec
    # ---- variant B1, as written, on a reduced tipset (never extrapolated) ----
    nb1 = 1500
    tb = Tipset(n_receipts=nb1, n_parents=5, dup_permille=20, n_planted=3, max_events=4, no_events_permille=0, variety=0)
    eb = claims_mod.EventClaims(tb)
    sb = orc.store(tb.data, tb.off, tb.lens, tb.cids)
    t0 = time.perf_counter()
    r0 = sb.verify_event_proofs(eb, mode=0)
    t_b1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    r1 = sb.verify_event_proofs(eb, mode=1, threads=1)
    t_b2_small = time.perf_counter() - t0
    sb.close()
    if not np.array_equal(r0, r1) or not (r0 == 1).all():
        raise SystemExit("cpu_baseline: B1 and B2 disagree on the reduced tipset")
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": n / legs[0]["step"],
        "unit": "proofs/s",
        "cores": cores,
        "kind": "port",
        "sample": "C++ oracle (restatement of the reference; the Rust crate cannot be built here), g++ -O3 -march=%s, "
                  "OpenMP on all %d host processors: sharded store build + Blake2b CID check + two-pass event scan + "
                  "exec-order on the FULL %d-receipt tipset, verify_event_proof on the first %d of %d claims scaled "
                  "linearly; store and exec order built once per tipset (BASELINE.md variant B2 all-cores)"
                  % (march, cores, tip.params["n_receipts"], legs[0]["verify_sample_claims"], n),
        "seconds": {k: v for k, v in legs[0].items()},
        "value_1_thread": n / legs[1]["step"],
        "seconds_1_thread": {k: v for k, v in legs[1].items()},
        "scaling_1_to_all": legs[1]["step"] / legs[0]["step"],
        "b1_as_written": {"receipts": nb1, "proofs_per_s": eb.n / t_b1, "seconds": t_b1,
                          "b2_1_thread_same_tipset_proofs_per_s": eb.n / t_b2_small,
                          "note": "reference semantics incl. per-proof execution-order rebuild "
                                  "(events/verifier.rs:190): quadratic, measured at %d receipts, NOT extrapolated" % nb1},
        "host_cpus": os.cpu_count(),
        "cpu_model": cpu_model,
        "march": march,
        "checked_against_gpu": "every block's CID verdict, scan status + has-match map + (exec, event, emitter) match "
                               "list, and the status byte of every sampled claim",
    }


if __name__ == "__main__":
    main()
