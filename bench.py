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
`value` = claims verified per second.  Multi-GPU (`--gpus N`, one rank per GPU), two partitionings of a proof batch
(DESIGN.md §6 has the per-rank projection of both):
  --shard tipsets  (default)  the batch holds one 1M-receipt tipset PER RANK (a chain of epochs: tipsets are independent
                   objects): rank r runs the single-GPU step on tipset r and ONE ncclAllGather (RCCL called directly
                   by libipcfp.so) of [header | status bytes | has-match map | CID bitmap] closes the step — no other
                   data-path collective; per-GPU work is fixed, `scaling: weak`.
  --shard receipts ONE tipset cut into N receipt-range shards (`scaling: strong`): rank r plans and places its shard
                   once (untimed) — its receipts' events AMTs and receipts-AMT paths, plus the replicated headers,
                   TxMeta blocks and message AMTs — and a step is CID index + K1 + range-restricted scan + verify of
                   the claims routed to it, closed by the same all-gather.
`--workload cid|hamt|storage` run BASELINE.json configs[1], [3], [4]; with `--gpus N` the block / query / claim index
range is cut with ipcfp_shard_range (state tree replicated), each rank runs the ordinary entry point on its range and
ONE ncclAllGather of the status bytes (CID bitmap) closes the step (`scaling: strong`).  The default single-GPU line
also carries compact records of those three configs (`configs`), each with its own `roofline` and `cpu_baseline`.

torch is plumbing only (HBM residency of inputs, device sync, torch.distributed); every timed
kernel is launched by libipcfp.so through its C ABI.  DESIGN.md §Measurement has the byte accounting.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # the cpu_baseline leg's OpenMP team sleeps between phases

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md §Chip-level parameters)
VALU_PEAK_TOPS = 78.6  # 32-bit integer lane operations: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz
K1_CYCLES_PER_WAVE_CHUNK = 6800.0  # DESIGN.md §K1, from tools/ubench/valu_rates
K1_SUSTAINED_GHZ = 2.18  # GRBM_GUI_ACTIVE under K1 (profiles/r01_final_pmc.txt)
VALU_INSTS_PER_CHUNK = 2083.0  # K1: VALU instructions per wavefront per 128-byte chunk (profiles/r01_final_pmc.txt)
MATCH_CAP = 1 << 16   # match records the step's HBM buffer holds (ipcfp_event_match_t; the tipset plants receipts / 1000)
MATCH_BYTES = 40
METRIC = "Merkle proofs verified/sec + HBM GB/s, 1M-receipt synthetic tipset, 1/2/4/8 GPU"

# ---- the line the driver keeps -------------------------------------------------------------------------------------
# The bench contract asks for ONE JSON line on stdout.  Round 5's line carried the whole report (24.5 KB, 14 per-shard
# dicts) and the driver could not parse it (BENCH_r05.json: parsed = null).  Since round 6 the report goes to
# `bench_detail.json` (next to this file, a copy under gpurun_out/ when that directory exists) and stdout gets
# `driver_line(report)`: the contract's keys, `roofline`, `cpu_baseline`, and one compact record per extra figure.
# tests/test_bench_line.py holds the line to this shape.
LINE_MAX_BYTES = 6000   # well inside the 8 KB of stdout the driver keeps; VERDICT r5 asked for <= 12 KB
STR_MAX = 120           # the driver cuts strings at 128 characters

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _short(s, n=STR_MAX):
    if not isinstance(s, str):
        return s
    s = " ".join(s.split())
    return s if len(s) <= n else s[: n - 3] + "..."


def _num(x, sig=6):
    """Numbers of the line at 6 significant digits (the contract's own `value` / `ms_per_step` stay exact)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        if x == int(x) and abs(x) < 2 ** 53:
            return int(x)  # byte and launch counts kept as floats by the arithmetic that made them
        return float("%.*g" % (sig, x))
    if isinstance(x, dict):
        return {k: _num(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v, sig) for v in x]
    if hasattr(x, "item"):
        return _num(x.item(), sig)
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


ROOFLINE_KEYS = ("bound", "limiter", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                 "traffic_over_algorithmic", "kernel_avg_ms", "launches", "algorithmic_bytes_per_launch", "bytes_basis",
                 "frac_of_valu_ceiling")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "value_1_thread", "host_cpus", "cpu_quota_cpus", "physical_cores",
            "value_if_linear_in_host_cores", "cpu_model", "checked_against_gpu")


def _line_roofline(r):
    if not isinstance(r, dict):
        return r
    out = _pick(r, ROOFLINE_KEYS)
    out.setdefault("traffic", None)
    for k in ("traffic_source", "bytes_basis", "kernel"):
        if k in out:
            out[k] = _short(out[k])
    if isinstance(r.get("alone"), dict):
        out["alone"] = _pick(r["alone"], ("kernel_avg_ms", "achieved", "frac"))
    return out


def _line_cpu(c):
    if not isinstance(c, dict):
        return c
    out = _pick(c, CPU_KEYS)
    for k in ("sample", "checked_against_gpu", "cpu_model"):
        if k in out:
            out[k] = _short(out[k])
    if isinstance(c.get("gpu_over_cpu"), dict):
        out["gpu_over_cpu"] = _pick(c["gpu_over_cpu"], ("T3_resident", "T2_pcie_inclusive", "T3_vs_linear_in_host_cores",
                                                        "T2_vs_linear_in_host_cores"))
    return out


def _line_config(c):
    out = {}
    for k, v in c.items():
        if isinstance(v, str):
            out[k] = _short(v)
        elif isinstance(v, dict):
            out[k] = {kk: (_short(vv) if isinstance(vv, str) else vv) for kk, vv in v.items() if not isinstance(vv, (dict, list))}
        elif isinstance(v, list):
            if len(v) <= 8 and all(not isinstance(x, (dict, list)) for x in v):
                out[k] = v
        else:
            out[k] = v
    return out


def _line_sub(rec):
    """configs[1]/[3]/[4] in the line: {value, unit, ms_per_step, roofline.frac, traffic_over_algorithmic, cpu_baseline.value}."""
    out = _pick(rec, ("value", "unit", "ms_per_step", "steps"))
    r = rec.get("roofline") or {}
    out["roofline"] = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "kernel_avg_ms", "launches"))
    c = rec.get("cpu_baseline") or {}
    out["cpu_baseline"] = _pick(c, ("value", "unit", "cores", "kind"))
    return out


def _line_projection(p):
    """scaling_projection in the line: max over shards per G, nothing per shard."""
    out = {"one_gpu_ms": p.get("one_gpu_ms"), "host_memory_read_GBps": p.get("host_memory_read_GBps"), "G": {}}
    for g, rec in (p.get("shards") or {}).items():
        out["G"][g] = {k: v for k, v in rec.items() if not isinstance(v, (list, dict))}
    return out


def driver_line(report: dict) -> dict:
    """The compact record printed as the bench's ONE stdout line (see the block comment above)."""
    line = {}
    for k in CONTRACT_KEYS:
        if k in report:
            line[k] = report[k]
    if line.get("scaling") not in ("weak", "strong"):
        raise ValueError("bench line: `scaling` must be weak or strong, not %r" % (line.get("scaling"),))
    line["config"] = _line_config(report.get("config", {}))
    line["roofline"] = _line_roofline(report.get("roofline"))
    if "cpu_baseline" in report:
        line["cpu_baseline"] = _line_cpu(report["cpu_baseline"])
    for k in ("window", "value_T2", "kernels_ms_per_step", "ms_per_step_by_order", "ms_per_step_separate_calls",
              "ms_per_step_counts_only", "ms_per_step_with_gather_message", "detail"):
        if k in report:
            line[k] = _short(report[k]) if isinstance(report[k], str) else report[k]
    if isinstance(report.get("window_T2"), dict):
        line["window_T2"] = _pick(report["window_T2"], ("value", "unit", "ms_per_tipset", "h2d_bytes", "ms_phases"))
    if isinstance(report.get("scaling_projection"), dict):
        line["scaling_projection"] = _line_projection(report["scaling_projection"])
    if isinstance(report.get("configs"), dict):
        line["configs"] = {k: _line_sub(v) for k, v in report["configs"].items()}
    if isinstance(report.get("weak_scaling_batch"), dict):
        line["weak_scaling_batch"] = _pick(report["weak_scaling_batch"], ("value", "unit", "ms_per_step", "scaling"))
    exact = {k: line.get(k) for k in ("value", "ms_per_step")}
    line = _num(line)
    line.update(exact)
    return line


def line_text(report: dict) -> str:
    """Strict JSON (no NaN / Infinity), one line, bounded."""
    text = json.dumps(driver_line(report), allow_nan=False, separators=(",", ":"))
    if "\n" in text or len(text) > LINE_MAX_BYTES:
        raise ValueError("bench line is %d bytes (limit %d)" % (len(text), LINE_MAX_BYTES))
    return text


def emit(report: dict) -> None:
    """Write the whole report to bench_detail.json, print the driver's line LAST on stdout."""
    report = dict(report)
    written = []
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if not os.path.isdir(d):
            continue
        try:
            with open(os.path.join(d, "bench_detail.json"), "w") as f:
                json.dump(report, f, indent=1, default=lambda o: o.item() if hasattr(o, "item") else str(o))
            written.append(os.path.relpath(os.path.join(d, "bench_detail.json"), ROOT))
        except OSError:
            pass
    report["detail"] = (", ".join(written) + " (whole report: kernel groups, per-shard records, thread sweeps)") if written else "not written"
    sys.stdout.flush()
    print(line_text(report), flush=True)


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
    ap.add_argument("--receipts", type=int, default=1_000_000, help="receipts of the tipset (cut into --gpus shards)")
    ap.add_argument("--workload", choices=["tipset", "cid", "hamt", "storage"], default="tipset",
                    help="tipset = BASELINE.json configs[2] (the metric's); cid/hamt/storage = configs[1]/[3]/[4]")
    ap.add_argument("--blocks", type=int, default=100_000, help="--workload cid: number of 1 KiB blocks")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU code path (plan, sub-witness, RCCL all-gather) with a single rank")
    ap.add_argument("--cpu-sample", type=int, default=50_000, help="claims the 1-thread cpu_baseline leg verifies")
    ap.add_argument("--cpu-sample-mt", type=int, default=200_000, help="claims the all-cores cpu_baseline leg verifies")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--order", default="K,V,S",
                    help="order of the three calls of a tipset step after the index rebuild: K = CID check (K1, asynchronous), "
                         "V = verify_event_proof batch (incl. execution order), S = event-filter scan")
    ap.add_argument("--t2-reps", type=int, default=3, help="repetitions of the PCIe-inclusive window (0 = skip)")
    ap.add_argument("--shard", choices=["both", "tipsets", "receipts"], default="both",
                    help="--workload tipset with --gpus N > 1: `receipts` = ONE tipset cut into N receipt-range shards (strong "
                         "scaling: the metric's config, and the line's `value`), `tipsets` = one tipset per rank (weak scaling), "
                         "`both` (default) = the strong line with the weak record under `weak_scaling_batch`")
    ap.add_argument("--logical-shards", default="2,4,8",
                    help="single GPU: receipt-range shard counts G whose shards are timed one by one in windows T3 and T2 "
                         "(`scaling_projection` in the line: strong scaling of ONE tipset); empty = skip")
    ap.add_argument("--separate-calls", action="store_true",
                    help="the step's verify and scan as two ABI calls (rounds 1-3) instead of ipcfp_verify_and_scan_device")
    ap.add_argument("--no-sub-records", action="store_true",
                    help="skip the compact configs[1]/[3]/[4] records the default single-GPU line carries")
    ap.add_argument("--plain", action="store_true",
                    help="only the timed steps: no per-kernel pass, no `alone` K1 launches, no other call order, no "
                         "gather-message step (what the profiler scripts trace)")
    ap.add_argument("--sub-steps", type=int, default=5, help="timed steps of each compact sub-record")
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
    if world > 1 or args.force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import ipc_filecoin_proofs_amd as ipcfp
    from tools.synth import SEED_BASE, Tipset

    eng = ipcfp.Engine(local_rank)
    info = eng.device_info()
    ranks = Ranks(torch, dist, eng, world, rank, dev)
    if args.workload != "tipset":
        out = {"cid": run_cid, "hamt": run_hamt, "storage": run_storage}[args.workload](args, eng, info, torch, ranks)
        if rank == 0:
            emit(out)
        ranks.close()
        eng.close()
        if world > 1:
            dist.destroy_process_group()
        return
    if world > 1 or args.force_sharded:
        # BASELINE.json's metric is ONE 1M-receipt tipset at 1/2/4/8 GPUs: the receipt cut (strong scaling) is the line's
        # `value`; the proof batch of one tipset per rank (weak scaling) rides along as a sub-record
        out = None
        if args.shard in ("receipts", "both") or args.force_sharded:
            out = run_tipset_sharded(args, eng, info, torch, dist, world, rank, dev, ranks)
        if args.shard in ("tipsets", "both") and not args.force_sharded:
            weak = run_tipset_batch(args, eng, info, torch, ranks)
            if rank == 0:
                if out is None:
                    out = weak
                else:
                    out["weak_scaling_batch"] = {k: weak[k] for k in ("value", "unit", "ms_per_step", "scaling", "config", "kernels_ms_per_step", "window")}
        if rank == 0:
            emit(out)
        ranks.close()
        eng.close()
        dist.destroy_process_group()
        return

    # ---- the tipset ----
    t_gen = time.perf_counter()
    tip = Tipset(seed=SEED_BASE + 3, n_receipts=args.receipts, n_parents=5, dup_permille=20,
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
    # the scan's product (find_matching_events, events/generator.rs:242-301) stays in caller HBM: the has-match map,
    # one byte per receipt, and the match records (exec_index, event_index, emitter, event location)
    t_has = torch.zeros(args.receipts, dtype=torch.uint8, device=dev)
    t_matches = torch.zeros(MATCH_CAP * MATCH_BYTES, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    w = eng.witness_device(t_bytes.data_ptr(), tip.data.size, t_off.data_ptr(), t_len.data_ptr(), t_cids.data_ptr(),
                           tip.n_blocks)
    scan_result = {}
    scan_mode = {"counts_only": False, "combined": not args.separate_calls}

    order = args.order.split(",")

    def step():
        w.rebuild_index()                                                               # K4
        if scan_mode["combined"] and order == ["K", "V", "S"]:
            w.verify_cids_async()                                                       # K1
            # verify_event_proof of every claim + the event-filter scan in ONE call: the scan's tail rides on the verify
            # call's synchronisation (ipcfp_verify_and_scan_device; same outputs as the two calls below)
            st, _, m = w.verify_and_scan_device(ts, t_claims.data_ptr(), n_claims, t_blob.data_ptr(), blob_len,
                                                t_status.data_ptr(), tip.topic0, tip.topic1, tip.filter_actor,
                                                t_has.data_ptr(), args.receipts, t_matches.data_ptr(), MATCH_CAP)
            scan_result["status"], scan_result["matches"] = st, m
            return
        for what in order:
            if what == "S" and scan_mode["counts_only"]:
                st, _, m, _ = w.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor,
                                            want_touched=False, counts_only=True)       # K6, the ABI's sizing call
                scan_result["status"], scan_result["matches"] = st, m
            elif what == "S":
                st, _, m = w.scan_events_device(tip.receipts_root, tip.topic0, tip.topic1, tip.filter_actor,
                                                t_has.data_ptr(), args.receipts, matches_ptr=t_matches.data_ptr(),
                                                cap_matches=MATCH_CAP)                  # K6: map + match records in HBM
                scan_result["status"], scan_result["matches"] = st, m
            elif what == "K":
                # K1 is independent of everything else in the step (own stream); where it is queued only decides
                # which kernels it runs beside
                w.verify_cids_async()                                                   # K1
            elif what == "V":
                w.verify_event_claims_device(ts, t_claims.data_ptr(), n_claims, t_blob.data_ptr(), blob_len,
                                             t_status.data_ptr())                       # exec order + verify

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.profile_reset()
    # over the timed region only K1's launches carry an event pair (the roofline's kernel); every other kernel of the
    # step stays back to back on its stream.  The per-kernel table comes from a second, untimed pass below
    eng.profile_enable(True, only="blake2b_cid")
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    t1 = time.perf_counter()
    eng.profile_enable(False)
    elapsed = t1 - t0
    k1_timed = eng.profile_read("blake2b_cid")
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
    # ... and the product the timed step left in HBM: the match records name every planted receipt, the map marks
    # exactly the receipts the records name
    step_has = t_has.cpu().numpy()
    step_matches = t_matches.cpu().numpy()[: min(scan_result["matches"], MATCH_CAP) * MATCH_BYTES].view(ipcfp.MATCH_DTYPE)
    if scan_result["matches"] > MATCH_CAP:
        raise SystemExit("bench self-check failed: more matches than the step's record buffer holds")
    rec_receipts = np.unique(step_matches["exec_index"].astype(np.int64))
    if not np.isin(tip.planted.astype(np.int64), rec_receipts).all() or not np.array_equal(np.nonzero(step_has)[0], rec_receipts):
        raise SystemExit("bench self-check failed: the timed step's match records / has-map differ from the planted matches")

    # ---- window T2: the same pass end to end from HOST memory — witness upload over PCIe, repack,
    # CID index, K1, scan, claim upload, verify, status bytes and CID verdicts back.  `value` above is window T3
    # (inputs resident in HBM); T2 is what the ≥10x-over-CPU claim is judged on.
    t2 = None
    if world == 1 and args.t2_reps > 0:
        # the bundle's tables and claims in TRANSPORT form (ipcfp_witness_create_packed / ipcfp_verify_event_claims_compact:
        # no offset table, 32-byte digests + one prefix, 56-byte claim records without per-topic framing) — built once,
        # untimed, like the plain forms above: they are what the caller holds
        pk = ipcfp.PackedWitnessTables(tip.data, tip.off, tip.lens, tip.cids)
        groups, cc, cblob, cblob_len = ipcfp.compact_event_claims(cl, blob, blob_len)

        def t2_pass(transport):
            torch.cuda.synchronize()
            ta = time.perf_counter()
            w2 = eng.witness_packed(pk) if transport else eng.witness(tip.data, tip.off, tip.lens, tip.cids)
            tb = time.perf_counter()
            # the calls in the order of the resident step (K, V, S): K1 is queued, the claims cross PCIe beside the
            # verify call's own AMT walk, the scan finds the receipts enumerated and the events tabulated
            w2.verify_cids_async()
            tc_ = time.perf_counter()
            status2 = (w2.verify_event_claims_compact(ts, groups, cc, cblob, cblob_len) if transport
                       else w2.verify_event_claims(ts, cl, blob, blob_len))
            td = time.perf_counter()
            st2, has2, m2, _ = w2.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor,
                                              want_touched=False, caps=(n_claims, MATCH_CAP))
            nm2 = len(m2)
            cs2, nbad2 = w2.cid_results()
            te = time.perf_counter()
            w2.close()
            if st2 != 1 or nbad2 or not np.array_equal(status2, status) or nm2 != scan_result["matches"] or \
                    not np.array_equal(has2, step_has) or not np.array_equal(m2["exec_index"], step_matches["exec_index"]):
                raise SystemExit("bench self-check failed: the from-host pass differs from the resident one")
            return {"total": te - ta, "witness_create_h2d_repack_index": tb - ta, "k1_launch": tc_ - tb,
                    "claims_h2d_verify_status_d2h": td - tc_, "scan_cid_verdicts_d2h": te - td}

        def t2_window(transport, h2d_bytes, note):
            reps = [t2_pass(transport) for _ in range(args.t2_reps + 1)][1:]  # the first repetition warms the allocator
            best = min(reps, key=lambda r: r["total"])
            return {"value": n_claims / best["total"], "unit": "proofs/s", "ms_per_tipset": best["total"] * 1e3,
                    "ms_phases": {k: round(v * 1e3, 3) for k, v in best.items() if k != "total"},
                    "h2d_bytes": h2d_bytes, "h2d_GBps_if_all_transfer": h2d_bytes / best["total"] / 1e9, "reps": len(reps),
                    "ms_all_reps": [round(r["total"] * 1e3, 3) for r in reps], "note": note}

        plain_bytes = int(tip.data.size + tip.off.nbytes + tip.lens.nbytes + tip.cids.nbytes + cl.nbytes + blob_len)
        transport_bytes = int(pk.h2d_bytes + groups.nbytes + cc.nbytes + cblob_len)
        t2 = t2_window(True, transport_bytes,
                       "pageable host numpy buffers in; host status bytes, CID verdicts, the scan's has-match map and match records out; "
                       "witness tables and claims in TRANSPORT form (ipcfp_witness_create_packed: block lengths + 32-byte digests + one CID "
                       "prefix, offsets and 40-byte slots rebuilt on the device; ipcfp_verify_event_claims_compact: 56-byte claim records + "
                       "unframed topics, expanded on the device); calls in the order of the resident step (K, V, S); uploads are the runtime's "
                       "blocking copies (56 GB/s measured, tools/ubench/h2d_paths), the claims cross beside the verify call's AMT walk; "
                       "verify and scan are separate calls here and the context remembers the last scan's filter (ctx scan hint), so the "
                       "verify call counts this filter's matches while it tabulates the events: a first-contact bundle scanned with ANOTHER "
                       "filter pays one more counting pass over the event records (k_count_from_table, ~27 us; INTEGRATION.md §4)")
        t2["plain_forms"] = t2_window(False, plain_bytes,
                                      "the same pass with the full tables (off[], 40-byte CID slots) and ipcfp_event_claim_t + framed blob: "
                                      "round 3's T2")

    # ---- the full scan result, untimed, for the oracle cross-check of the cpu_baseline leg ----
    gpu_scan = None
    if world == 1 and not args.no_cpu_baseline:
        gs, ghas, gm, _ = w.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor,
                                        want_touched=False)
        gpu_scan = (gs, ghas, gm)

    roof = k1_roofline(eng, tip.lens, tip.n_blocks, with_traffic=True, timed=k1_timed)
    # ---- every kernel group of the step, each bracketed by its own event pair (second pass, outside `value`) ----
    eng.profile_reset()
    eng.profile_enable(True)
    fence()
    ta = time.perf_counter()
    for _ in range(0 if args.plain else args.steps):
        step()
    fence()
    bracketed_ms = (time.perf_counter() - ta) / args.steps * 1e3
    eng.profile_enable(False)
    kern = {}
    for k in STEP_KERNEL_GROUPS:
        cnt, ms = eng.profile_read(k)
        kern[k] = {"launches": cnt, "ms_per_step": ms / args.steps}
    kernels = tipset_kernels(tip, kern, args.steps, n_claims, int(cl.nbytes) + int(blob_len), bracketed_ms)
    # In the step K1 shares the chip with the block-order event parse (both on side streams).  The same launch with the
    # chip to itself, untimed and outside `value`: what the kernel does when nothing runs beside it.
    eng.sync()
    eng.profile_reset()
    eng.profile_enable(True)
    for _ in range(0 if args.plain else 5):
        w.verify_cids_async()
        eng.sync()
    eng.profile_enable(False)
    a_cnt, a_ms = eng.profile_read("blake2b_cid")
    if a_cnt:
        alone_ms = a_ms / a_cnt
        roof["alone"] = {"kernel_avg_ms": alone_ms, "launches": a_cnt,
                         "achieved": roof["algorithmic_bytes_per_launch"] / (alone_ms * 1e-3) / 1e9,
                         "frac": roof["algorithmic_bytes_per_launch"] / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "note": "the same launch with nothing beside it (5 launches after the timed region); `achieved` / "
                                 "`frac` above are the in-step figures, K1 sharing the chip with k_block_events"}

    # ---- the other call order, and the step with the multi-GPU message on top (a few extra steps, outside `value`) ----
    extras = {}
    if world == 1 and not args.plain:
        other = "S,K,V" if args.order != "S,K,V" else "K,V,S"
        saved = order[:]
        order[:] = other.split(",")
        for _ in range(2):
            step()
        fence()
        ta = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        extras["ms_per_step_by_order"] = {args.order: elapsed / args.steps * 1e3, other: (time.perf_counter() - ta) / args.steps * 1e3}
        order[:] = saved
        # verify and scan as two ABI calls (each with its own synchronisation), the scan's product delivered
        was_combined = scan_mode["combined"]
        scan_mode["combined"] = False
        for _ in range(2):
            step()
        fence()
        ta = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        extras["ms_per_step_separate_calls"] = (time.perf_counter() - ta) / args.steps * 1e3
        # the step as rounds 1-3 timed it: the scan as the ABI's sizing call (no map, no records delivered)
        scan_mode["counts_only"] = True
        for _ in range(2):
            step()
        fence()
        ta = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        extras["ms_per_step_counts_only"] = (time.perf_counter() - ta) / args.steps * 1e3
        scan_mode["counts_only"] = False
        scan_mode["combined"] = was_combined
        extras["ms_per_step_with_gather_message"] = tipset_gather_step_ms(args, eng, torch, w, tip, ts, t_claims, t_blob, blob_len, t_status, n_claims)
        if args.logical_shards:
            extras["scaling_projection"] = shard_projection(
                args, eng, torch, dev, tip, ts, cl, blob, blob_len, w, status, step_has,
                extras["ms_per_step_with_gather_message"], t2["ms_per_tipset"] if t2 else None,
                shard_counts=tuple(int(x) for x in args.logical_shards.split(",")), steps=args.sub_steps)
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
            "scaling": "weak",  # (N = 1: neither cut applies; config.sharding says what --gpus N does)
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[2] (the 1M-receipt tipset the metric is quoted on): %d receipts (Amtv0<Receipt> + one Amt<StampedEvent> each, 5 parent "
                            "headers with TxMeta and message AMTs), %d witness blocks, %.3f GB; one EventProof claim per "
                            "receipt; step = CID index + Blake2b-256 CID check of every block + exec-order reconstruction + "
                            "verify_event_proof of every claim + event-filter scan (has-match map and match records delivered in "
                            "HBM); verify and scan in one ABI call (ipcfp_verify_and_scan_device) unless --separate-calls" %
                            (args.receipts, tip.n_blocks, tip.stats["payload_bytes"] / 1e9),
                "receipts_per_gpu": args.receipts,
                "claims_per_gpu": n_claims,
                "witness_blocks_per_gpu": tip.n_blocks,
                "witness_bytes_per_gpu": tip.stats["payload_bytes"],
                "scan_matches": int(scan_result["matches"]),
                "sharding": "single GPU.  --gpus N cuts THIS ONE tipset by receipt range over N ranks (strong scaling: the line's `value`), every "
                            "rank a self-planned shard (ipcfp_witness_create_shard_pull: the device follows the links and reads its blocks out of "
                            "the bundle in registered host memory), one ncclAllGather of [header | status bytes | has-match map | CID bitmap]; "
                            "`scaling_projection` below times every such shard by itself on this GPU with plan, cut, claims, upload, verify and scan "
                            "inside ONE T2 figure; --shard tipsets runs one tipset per rank instead (weak scaling, a sub-record)",
                "device": info["name"],
                "setup_seconds_untimed": round(t_gen, 2),
            },
            "roofline": roof,
            "kernels": kernels,
            "kernels_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in kern.items()},
        }
        if t2 is not None:
            out["value_T2"] = t2["value"]
            out["window_T2"] = t2
            # (a driver that keeps only the contract's keys keeps nested dicts: the window the >= 10x claim is judged on rides in `config`)
            out["config"]["window_T2"] = {"proofs_per_s": t2["value"], "ms_per_tipset": t2["ms_per_tipset"], "h2d_bytes": t2.get("h2d_bytes"),
                                          "what": "the same pass from pageable HOST memory (PCIe-inclusive: upload, index, K1, verify, scan, results back)"}
        out["window"] = "T3 (inputs resident in HBM; index rebuilt and every cached enumeration dropped each step)"
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(tip, status, args.cpu_sample, args.cpu_sample_mt, cid_status, gpu_scan)
            cb = out["cpu_baseline"]
            out["speedup_vs_cpu_all_cores"] = {"T3": out["value"] / cb["value"],
                                               "T2": (t2["value"] / cb["value"]) if t2 else None}
            if cb.get("value_if_linear_in_host_cores"):  # against the whole host, were the port linear in its cores (an upper bound)
                lin = cb["value_if_linear_in_host_cores"]
                out["speedup_vs_cpu_if_linear_in_host_cores"] = {"T3": out["value"] / lin, "T2": (t2["value"] / lin) if t2 else None}
            # the same facts INSIDE cpu_baseline (nested dicts survive a driver that keeps only the contract's keys)
            cb["gpu_over_cpu"] = {"T3_resident": out["value"] / cb["value"], "T2_pcie_inclusive": (t2["value"] / cb["value"]) if t2 else None,
                                  "T3_vs_linear_in_host_cores": (out["value"] / cb["value_if_linear_in_host_cores"]) if cb.get("value_if_linear_in_host_cores") else None,
                                  "T2_vs_linear_in_host_cores": (t2["value"] / cb["value_if_linear_in_host_cores"]) if (t2 and cb.get("value_if_linear_in_host_cores")) else None,
                                  "gpu_proofs_per_s_T3": out["value"], "gpu_proofs_per_s_T2": t2["value"] if t2 else None,
                                  "gpu_ms_per_tipset_T2": t2["ms_per_tipset"] if t2 else None, "h2d_bytes_T2": t2.get("h2d_bytes") if t2 else None}
        out.update(extras)
    w.close()
    del t_bytes, t_off, t_len, t_cids, t_claims, t_blob, t_status
    if rank == 0:
        if world == 1 and not args.no_sub_records:
            # BASELINE.json configs[1], [3], [4] in the SAME driver-visible line: compact records, each with its own
            # `roofline` and `cpu_baseline` (bounded samples)
            sub_args = argparse.Namespace(**vars(args))
            sub_args.steps, sub_args.warmup = args.sub_steps, 2
            state = _state_tipset()
            out["configs"] = {
                "configs[1] cid": compact(run_cid(sub_args, eng, info, torch, ranks)),
                "configs[3] hamt": compact(run_hamt(sub_args, eng, info, torch, ranks, state)),
                "configs[4] storage": compact(run_storage(sub_args, eng, info, torch, ranks, state)),
            }
        emit(out)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


class Ranks:
    """One process per GPU: the host's own channel (torch.distributed: rendezvous, barrier, the max-over-ranks clock,
    the 128-byte RCCL id) and the engine's communicator of the DATA path (RCCL called directly by libipcfp.so)."""

    def __init__(self, torch, dist, eng, world, rank, dev):
        self.torch, self.dist, self.eng, self.world, self.rank, self.dev = torch, dist, eng, world, rank, dev
        self.comm = None

    def make_comm(self):
        """The engine's RCCL communicator, or a loud failure: there is no torch.distributed fallback on the data path.
        ncclCommInitRank is a collective, so the ranks first AGREE (over the host channel) that every one of them can
        resolve librccl — a rank that cannot would otherwise leave the others hanging inside the init."""
        import ipc_filecoin_proofs_amd as ipcfp

        if self.comm is not None or self.world == 1:
            return self.comm
        try:
            uid_local, ok = ipcfp.comm_unique_id(), 1  # dlopen("librccl.so.1") + ncclGetUniqueId: the probe
        except ipcfp.EngineError as e:
            uid_local, ok = None, 0
            sys.stderr.write("rank %d: librccl cannot be resolved by libipcfp.so: %s\n" % (self.rank, e))
        flag = self.torch.tensor([ok], dtype=self.torch.int32, device=self.dev)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            raise SystemExit("bench.py: the engine's direct RCCL communicator cannot be made on every rank (see stderr); "
                             "there is no fallback collective")
        uid = [uid_local if self.rank == 0 else None]
        self.dist.broadcast_object_list(uid, src=0)
        self.comm = ipcfp.Comm(self.eng, uid[0], self.world, self.rank)  # raises EngineError (loud) on failure
        return self.comm

    def fence(self):
        self.eng.sync()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_seconds(self, seconds: float) -> float:
        if self.world == 1:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None


def timed(ranks, step, steps, warmup, only=None):
    """The contract's clock: W untimed steps, barrier + sync, K steps, barrier + sync, MAX over ranks.
    `only`: bracket that kernel group alone over the timed region (as the single-GPU tipset run does: an event pair
    around EVERY group costs a multi-kernel step about 5 %); None: every group."""
    for _ in range(warmup):
        step()
    ranks.fence()
    ranks.eng.profile_reset()
    ranks.eng.profile_enable(True, only=only)
    ranks.fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ranks.fence()
    elapsed = time.perf_counter() - t0
    ranks.eng.profile_enable(False)
    return ranks.max_seconds(elapsed)


def load_workload_traffic(wl):
    """L2→L1 traffic per step of a per-config bench from the newest profiles/rNN_traffic_workloads.json
    (tools/pmc_traffic_workload.py over the FETCH_SIZE pass of tools/gpu_pmc_workload.sh) → (bytes, source) or (None, None)."""
    try:
        names = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic_workloads.json") and f[0] == "r")
        for name in reversed(names):
            with open(os.path.join(ROOT, "profiles", name)) as f:
                tr = json.load(f)
            if wl in tr.get("workloads", {}):
                return tr["workloads"][wl]["traffic_bytes_per_step"], "profiles/" + name
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def with_workload_traffic(roof, wl, algorithmic_bytes_per_step):
    """Fill `traffic` (+ its ratio to the algorithmic bytes of a step) of a per-config roofline from the committed PMC pass."""
    t, src = load_workload_traffic(wl)
    if t is not None:
        roof["traffic"] = t
        roof["traffic_source"] = src + " (L2 memory-side read requests, TCC_EA0_RDREQ via FETCH_SIZE, x2.00 calibrated; all the step's kernels)"
        if algorithmic_bytes_per_step:
            roof["traffic_over_algorithmic"] = round(t / algorithmic_bytes_per_step, 3)
    return roof


def compact(rec):
    """A per-config record cut down to what the default line carries."""
    keep = ("value", "unit", "ms_per_step", "steps", "roofline", "cpu_baseline", "window")
    out = {k: rec[k] for k in keep if k in rec}
    out["workload"] = rec["config"]["workload"]
    if "roofline" in out:
        out["roofline"] = {k: v for k, v in out["roofline"].items() if k in
                           ("bound", "limiter", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_over_algorithmic",
                            "kernel_avg_ms", "launches", "algorithmic_bytes_per_launch", "bytes_basis")}
    return out


def range_layout(n, world):
    """[lo, hi) of every rank for n units and the common (16-byte aligned) width of a rank's status segment."""
    import ipc_filecoin_proofs_amd as ipcfp

    rng = [ipcfp.shard_range(n, world, r) for r in range(world)]
    width = (max(h - l for l, h in rng) + 15) & ~15
    return rng, max(width, 16)


def tipset_message(torch, dev, n_claims, n_receipts, n_blocks):
    """Device buffers of one rank's step message [header | status | has map | CID bitmap] (shard.Layout)."""
    from ipc_filecoin_proofs_amd import shard

    layout = shard.Layout(n_claims, n_receipts, n_blocks)
    hdr = np.array([n_claims, n_receipts, n_blocks, 0, 0, 0, 0, n_receipts], dtype=np.uint64)
    d_hdr = torch.from_numpy(hdr.view(np.uint8).copy()).to(dev)
    d_has = torch.zeros(layout.w_has, dtype=torch.uint8, device=dev)
    d_stage = torch.zeros(layout.bytes_per_rank, dtype=torch.uint8, device=dev)
    return layout, d_hdr, d_has, d_stage


def tipset_gather_step_ms(args, eng, torch, w, tip, ts, t_claims, t_blob, blob_len, t_status, n_claims):
    """N = 1 only: the single-GPU step PLUS what a rank of a multi-GPU run adds to it — the scan's has-match map kept
    in HBM, the step message packed, the (here degenerate) all-gather — so that the N = 1 point of a scaling curve can
    be read on the same definition as the N > 1 points."""
    import ipc_filecoin_proofs_amd as ipcfp
    from ipc_filecoin_proofs_amd import shard

    dev = t_status.device
    layout, d_hdr, d_has, d_stage = tipset_message(torch, dev, n_claims, args.receipts, tip.n_blocks)
    d_recv = torch.zeros(layout.bytes_per_rank, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    bits_bytes = (tip.n_blocks + 31) // 32 * 4

    def step():
        w.rebuild_index()
        w.verify_cids_async()
        w.verify_event_claims_device(ts, t_claims.data_ptr(), n_claims, t_blob.data_ptr(), blob_len, t_status.data_ptr())
        w.scan_events_device(tip.receipts_root, tip.topic0, tip.topic1, tip.filter_actor, d_has.data_ptr(), layout.w_has,
                             summary_ptr=d_hdr.data_ptr() + 24)
        ipcfp.allgather_segments(eng, None, [d_hdr.data_ptr(), t_status.data_ptr(), d_has.data_ptr(), w.cid_bitmap_ptr],
                                 [shard.HEADER_BYTES, n_claims, args.receipts, bits_bytes], d_stage.data_ptr(),
                                 d_recv.data_ptr(), layout.bytes_per_rank)

    for _ in range(2):
        step()
    eng.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps * 1e3


def host_read_bandwidth(buf, thread_counts=(4, 16, 32)):
    """What the host's memory system delivers to CPU threads streaming through `buf` (numpy reductions release the GIL),
    under whatever CPU quota the container has: a LOWER bound of what G PCIe links can read out of it by DMA."""
    from concurrent.futures import ThreadPoolExecutor

    words = buf[: buf.size // 8 * 8].view(np.uint64)
    res = {}
    for t in thread_counts:
        parts = np.array_split(words, t)
        with ThreadPoolExecutor(t) as ex:
            list(ex.map(lambda a: int(np.bitwise_xor.reduce(a)), parts))  # warm
            t0 = time.perf_counter()
            list(ex.map(lambda a: int(np.bitwise_xor.reduce(a)), parts))
            res[str(t)] = round(words.nbytes / (time.perf_counter() - t0) / 1e9, 1)
    return res


def shard_projection(args, eng, torch, dev, tip, ts, cl, blob, blob_len, w_full, status_full, has_full, t3_ms_1, t2_ms_1,
                     shard_counts=(2, 4, 8), steps=5, t2_reps=2):
    """STRONG scaling of ONE tipset, measured on the one GPU there is: for G in shard_counts every rank r = 0..G-1 of the
    receipt cut of north_star / SURVEY.md §8(e) is run BY ITSELF, one after the other, as a SELF-PLANNED shard
    (ipcfp_witness_create_shard_pull): the rank is given the bundle in registered host memory and the tipset key, nothing
    else — no plan from elsewhere, no whole witness in anybody's HBM, no block lists or claims cut on the host.
      T2  ONE figure per rank, everything inside: tables of the bundle up (36 B per block), the device walks the links level
          by level and reads the shard's blocks out of host memory itself, the shard's arena / schedule / index, K1, the
          rank's claims (two binary searches into the exec_index-ordered batch, uploaded as a slice, rebased on the device),
          verify, range-restricted scan, status bytes + CID verdicts + has-match map + match records back on the host.
      T3  the shard resident in HBM; step = shard.TipsetShard.step = CID index + K1 + verify of the shard's claims (the
          execution order is rebuilt on every rank) + range-restricted scan + the packed step message.
    A G-GPU run's step is max over ranks (+ the collective); `projected_speedup` = t(G = 1) / max_r t_r.  Every shard's
    verdicts are merged and compared with the unsharded run's before anything is reported."""
    from ipc_filecoin_proofs_amd import shard

    import ipc_filecoin_proofs_amd as ipcfp

    filt = (tip.topic0, tip.topic1, tip.filter_actor)
    out = {"one_gpu_ms": {"T3_with_gather_message": t3_ms_1, "T2": t2_ms_1}, "shards": {}}

    def dev_bytes(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)

    # the bundle as a rank finds it: transport form in an ingest buffer that was registered when it was made
    pk = ipcfp.PackedWitnessTables(tip.data, tip.off, tip.lens, tip.cids, ingest=True)
    t0 = time.perf_counter()
    ipcfp.host_register(pk.data)
    ipcfp.host_register(pk.digests)
    out["host_register_ms_once_untimed"] = round((time.perf_counter() - t0) * 1e3, 2)
    if not np.all(cl["exec_index"][1:] >= cl["exec_index"][:-1]):
        raise SystemExit("bench: the claim batch is not in exec_index order")
    out["host_memory_read_GBps"] = host_read_bandwidth(pk.data)
    try:
        for G in shard_counts:
            status = np.full(len(cl), 255, dtype=np.uint8)
            has = np.zeros(len(has_full), dtype=np.uint8)
            per, rows = [], []
            for r in range(G):
                # ---- T2: everything a rank does, from the bundle in host memory to the verdicts on the host ----
                reps, stats_last = [], None
                for rep in range(t2_reps + 1):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    st, w2, lo, hi, n_receipts, stats = eng.witness_shard_pull(pk, tip.parent_cids, tip.child_cid, G, r)
                    if st != 1:
                        raise SystemExit("bench self-check failed: shard %d of %d could not be pulled (status %d)" % (r, G, st))
                    t1 = time.perf_counter()
                    w2.verify_cids_async()
                    a, st2 = w2.verify_event_claims_range(ts, cl, blob, blob_len, lo, hi, r == G - 1)
                    t2_ = time.perf_counter()
                    sst, has2, m2, _ = w2.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor,
                                                      want_touched=False, caps=(hi - lo, MATCH_CAP))
                    t3_ = time.perf_counter()
                    cs2, nbad2 = w2.cid_results()
                    t4_ = time.perf_counter()
                    reps.append(t4_ - t0)
                    stats_last = dict(stats, phases_ms={"shard_pull": round((t1 - t0) * 1e3, 3), "claims_verify": round((t2_ - t1) * 1e3, 3),
                                                        "scan": round((t3_ - t2_) * 1e3, 3), "cid_results": round((t4_ - t3_) * 1e3, 3)})
                    if sst != 1 or nbad2:
                        raise SystemExit("bench self-check failed: shard %d of %d: scan status %d, %d bad CIDs" % (r, G, sst, nbad2))
                    if rep < t2_reps:
                        w2.close()
                status[a: a + len(st2)] = st2
                has[lo:hi] = has2
                n_claims_r, n_blocks_r = len(st2), w2.n
                # ---- T3: the same shard, resident (the witness of the last T2 pass), claims in HBM ----
                s_obj = shard.TipsetShard.__new__(shard.TipsetShard)
                s_obj.eng, s_obj.n_shards, s_obj.shard = eng, G, r
                s_obj.lo, s_obj.hi, s_obj.n_receipts_total, s_obj.block_ids = lo, hi, n_receipts, None
                s_obj.witness, s_obj.receipts_root = w2, bytes(tip.receipts_root)
                s_obj.parent_cids, s_obj.child_cid = tip.parent_cids, tip.child_cid
                s_obj.route(ts, cl, blob, blob_len)
                rows.append((s_obj, a))
                per.append({"shard": r, "receipts": [lo, hi], "blocks": int(n_blocks_r), "claims": int(n_claims_r),
                            "h2d_bytes": int(stats_last["table_bytes"] + stats_last["block_bytes"] + n_claims_r * cl.dtype.itemsize +
                                             s_obj.blob_len),
                            "pull": {"rounds": stats_last["rounds"], "table_bytes": int(stats_last["table_bytes"]),
                                     "block_bytes": int(stats_last["block_bytes"]), "tables_ms": round(stats_last["tables_ms"], 3),
                                     "pull_ms": round(stats_last["pull_ms"], 3), "create_ms": round(stats_last["create_ms"], 3)},
                            "T2_phases_ms_last_pass": stats_last["phases_ms"],
                            "T2_ms": round(min(reps[1:]) * 1e3, 3)})
            layout = shard.Layout(max(s.n_claims for s, _ in rows), max(s.hi - s.lo for s, _ in rows), max(s.witness.n for s, _ in rows))
            for (s_obj, a), rec in zip(rows, per):
                d_cl, d_blob = dev_bytes(s_obj.claims), dev_bytes(s_obj.blob)
                d_status = torch.zeros(layout.w_status, dtype=torch.uint8, device=dev)
                d_has = torch.zeros(layout.w_has, dtype=torch.uint8, device=dev)
                d_stage = torch.zeros(layout.bytes_per_rank, dtype=torch.uint8, device=dev)
                d_recv = torch.zeros(layout.bytes_per_rank, dtype=torch.uint8, device=dev)
                d_hdr = dev_bytes(s_obj.header())
                torch.cuda.synchronize()

                def step():
                    s_obj.step(layout, None, filt, d_cl.data_ptr(), d_blob.data_ptr(), d_status.data_ptr(), d_has.data_ptr(),
                               d_hdr.data_ptr(), d_stage.data_ptr(), d_recv.data_ptr())

                for _ in range(2):
                    step()
                eng.sync()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    step()
                eng.sync()
                torch.cuda.synchronize()
                rec["T3_ms_per_step"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
                msg = d_recv.cpu().numpy()
                hdr = msg[:shard.HEADER_BYTES].view(np.uint64)
                pos = s_obj.positions.astype(np.int64)
                if int(hdr[0]) != len(pos) or int(hdr[3]) != 1:
                    raise SystemExit("bench self-check failed: shard %d of %d reports claims %d / scan status %d" % (rec["shard"], G, int(hdr[0]), int(hdr[3])))
                if not np.array_equal(msg[layout.off_status: layout.off_status + len(pos)], status[pos]) or \
                        not np.array_equal(msg[layout.off_has: layout.off_has + (s_obj.hi - s_obj.lo)], has[s_obj.lo: s_obj.hi]):
                    raise SystemExit("bench self-check failed: shard %d of %d resident differs from its PCIe-inclusive pass" % (rec["shard"], G))
                s_obj.close()
                del d_cl, d_blob, d_status, d_has, d_stage, d_recv, d_hdr
            if not np.array_equal(status, status_full) or not np.array_equal(has, has_full):
                raise SystemExit("bench self-check failed: the merged verdicts of %d shards differ from the unsharded run" % G)
            t3_max, t2_max = max(p["T3_ms_per_step"] for p in per), max(p["T2_ms"] for p in per)
            # The same T2 under the OTHER host ceiling: G ranks reading one host's memory at the rate that host measurably
            # delivers (`host_memory_read_GBps`: CPU threads under the container's quota — a lower bound of its DMA paths)
            # instead of every link at the rate one link reaches alone.  Neither is measured with G links at once.
            host_gbps = max(out["host_memory_read_GBps"].values())
            t2_host = 0.0
            for p in per:
                link_ms = p["pull"]["tables_ms"] + p["pull"]["pull_ms"]
                shared_ms = G * (p["pull"]["table_bytes"] + p["pull"]["block_bytes"]) / (host_gbps * 1e9) * 1e3
                t2_host = max(t2_host, p["T2_ms"] - link_ms + max(link_ms, shared_ms))
            replicated = int(per[0]["pull"]["table_bytes"] + tip.stats["message_amt_bytes"])
            out["shards"][str(G)] = {
                "T2_ms_if_host_memory_bound": round(t2_host, 3),
                "projected_speedup_T2_if_host_memory_bound": round(t2_ms_1 / t2_host, 3) if t2_ms_1 else None,
                "host_GBps_assumed": host_gbps,
                "pcie_bytes_identical_on_every_rank": replicated,  # the bundle's tables + the message AMTs (the execution order is global)
                "allgather_bytes_per_rank": layout.bytes_per_rank,
                "T3_ms_max_over_shards": t3_max, "T2_ms_max_over_shards": t2_max,
                "projected_speedup_T3": round(t3_ms_1 / t3_max, 3), "projected_speedup_T2": round(t2_ms_1 / t2_max, 3) if t2_ms_1 else None,
                "projected_proofs_per_s_T3": len(cl) / (t3_max * 1e-3), "projected_proofs_per_s_T2": len(cl) / (t2_max * 1e-3),
                "h2d_bytes_max": max(p["h2d_bytes"] for p in per), "merged_equals_unsharded": True, "per_shard": per,
            }
    finally:
        ipcfp.host_unregister(pk.data)
        ipcfp.host_unregister(pk.digests)
    out["note"] = ("ONE 1M-receipt tipset cut by receipt range (north_star's cut); every rank is SELF-PLANNED "
                   "(ipcfp_witness_create_shard_pull) and timed by itself on this GPU.  T2 is ONE figure per rank with everything "
                   "inside: plan (the device follows the links), cut (the device reads its blocks out of the registered host buffer: "
                   "no host memcpy, no CPU in the data path — the box's 16-CPU quota does not bound it), upload, shard build, K1, "
                   "claims (a slice of the exec_index-ordered batch), verify, scan, results on the host.  Outside: registering the "
                   "ingest buffer (`host_register_ms_once_untimed`: done when the buffer is made, not per bundle) and the xGMI "
                   "all-gather of `allgather_bytes_per_rank` per rank (latency-bound; not measurable on one GPU).  The projection "
                   "assumes each rank's PCIe link reads host memory at the rate one link does alone: G x 50 GB/s of host DRAM reads "
                   "(`host_memory_read_GBps` is what the host's memory system delivers to CPU threads under the quota — a lower "
                   "bound of what its DMA paths can); `T2_ms_if_host_memory_bound` is the same T2 with the G pulls sharing THAT rate "
                   "instead.  `pcie_bytes_identical_on_every_rank` (tables + message AMTs) cross PCIe G times in this flow; pulling 1/G "
                   "each and all-gathering the blocks over xGMI is costed in DESIGN.md §13, not built.  T3 divides only as far as the "
                   "replicated execution order lets it: five parents' message AMTs are walked on every rank.")
    return out


def run_tipset_batch(args, eng, info, torch, ranks):
    """--gpus N > 1, --shard tipsets: the proof batch holds one 1M-receipt tipset per rank (rank r generates tipset
    seed + r: a chain of epochs).  Step = the single-GPU step on the rank's tipset + ONE ncclAllGather of
    [header | status bytes | has-match map | CID bitmap]; every rank then holds every tipset's verdicts."""
    import ipc_filecoin_proofs_amd as ipcfp
    from ipc_filecoin_proofs_amd import shard
    from tools.synth import SEED_BASE, Tipset

    world, rank, dev = ranks.world, ranks.rank, ranks.dev
    t_gen = time.perf_counter()
    tip = Tipset(seed=SEED_BASE + 3 + rank, n_receipts=args.receipts, n_parents=5, dup_permille=20,
                 n_planted=max(1, args.receipts // 1000), max_events=4, no_events_permille=0, variety=0)
    n_claims = len(tip.claim_exec)
    ts, cl, blob, blob_len = ipcfp.pack_event_claims(
        tip.parent_cids, tip.child_cid, tip.parent_epoch, tip.child_epoch, tip.claim_exec, tip.claim_event,
        tip.claim_emitter, tip.exec_order[tip.claim_exec.astype(np.int64)], tip.claim_ntopics, tip.claim_topics,
        tip.claim_datalen, tip.claim_data)
    t_gen = time.perf_counter() - t_gen
    t_bytes = torch.from_numpy(tip.data).to(dev)
    t_off = torch.from_numpy(tip.off.view(np.int64)).to(dev)
    t_len = torch.from_numpy(tip.lens.view(np.int32)).to(dev)
    t_cids = torch.from_numpy(tip.cids.reshape(-1)).to(dev)
    t_claims = torch.from_numpy(cl.view(np.uint8).reshape(-1)).to(dev)
    t_blob = torch.from_numpy(blob).to(dev)
    torch.cuda.synchronize()
    w = eng.witness_device(t_bytes.data_ptr(), tip.data.size, t_off.data_ptr(), t_len.data_ptr(), t_cids.data_ptr(), tip.n_blocks)
    # message widths: the maxima over ranks (tipsets differ a little in their block counts)
    counts = torch.tensor([n_claims, args.receipts, tip.n_blocks], dtype=torch.int64, device=dev)
    ranks.dist.all_reduce(counts, op=ranks.dist.ReduceOp.MAX)
    mc, mr, mb = [int(x) for x in counts.tolist()]
    layout = shard.Layout(mc, mr, mb)
    hdr = np.array([n_claims, args.receipts, tip.n_blocks, 0, 0, 0, 0, args.receipts], dtype=np.uint64)
    d_hdr = torch.from_numpy(hdr.view(np.uint8).copy()).to(dev)
    d_status = torch.zeros(layout.w_status, dtype=torch.uint8, device=dev)
    d_has = torch.zeros(layout.w_has, dtype=torch.uint8, device=dev)
    d_stage = torch.zeros(layout.bytes_per_rank, dtype=torch.uint8, device=dev)
    d_recv = torch.zeros(world * layout.bytes_per_rank, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    comm = ranks.make_comm()
    bits_bytes = (tip.n_blocks + 31) // 32 * 4

    def step():
        w.rebuild_index()                                                                    # K4
        w.verify_cids_async()                                                                # K1 (second stream)
        w.verify_event_claims_device(ts, t_claims.data_ptr(), n_claims, t_blob.data_ptr(), blob_len, d_status.data_ptr())
        w.scan_events_device(tip.receipts_root, tip.topic0, tip.topic1, tip.filter_actor, d_has.data_ptr(), layout.w_has,
                             summary_ptr=d_hdr.data_ptr() + 24)                              # K6
        ipcfp.allgather_segments(eng, comm, [d_hdr.data_ptr(), d_status.data_ptr(), d_has.data_ptr(), w.cid_bitmap_ptr],
                                 [shard.HEADER_BYTES, layout.w_status, layout.w_has, bits_bytes], d_stage.data_ptr(),
                                 d_recv.data_ptr(), layout.bytes_per_rank)                   # the ONE collective

    # the timed region brackets K1 alone — the N = 1 point of a scaling curve is measured the same way — and the per-kernel
    # table comes from a second, untimed pass of the same steps (every rank runs it: the step ends in a collective)
    elapsed = timed(ranks, step, args.steps, args.warmup, only="blake2b_cid")
    roof = k1_roofline(eng, tip.lens, tip.n_blocks)
    eng.profile_reset()
    eng.profile_enable(True)
    ranks.fence()
    for _ in range(args.steps):
        step()
    ranks.fence()
    eng.profile_enable(False)
    kern = {}
    for k in STEP_KERNEL_GROUPS + ("allgather",):
        cnt, ms = eng.profile_read(k)
        kern[k] = {"launches": cnt, "ms_per_step": ms / args.steps}
    # ---- what was timed must be right: EVERY rank checks EVERY tipset's gathered verdicts ----
    g = d_recv.cpu().numpy().reshape(world, layout.bytes_per_rank)
    total_claims, matches = 0, []
    for r in range(world):
        h = g[r, :shard.HEADER_BYTES].view(np.uint64)
        nc, nr, nb, sst, nm = int(h[0]), int(h[1]), int(h[2]), int(h[3]), int(h[4])
        st_r = g[r, layout.off_status: layout.off_status + nc]
        bits = np.unpackbits(g[r, layout.off_bits: layout.off_bits + (nb + 31) // 32 * 4], bitorder="little")[:nb]
        if not (st_r == 1).all() or sst != 1 or int(bits.sum()) != nb or nm < max(1, args.receipts // 1000):
            raise SystemExit("bench self-check failed (rank %d): tipset %d's gathered verdicts are not all TRUE" % (rank, r))
        total_claims += nc
        matches.append(nm)
    has_own = g[rank, layout.off_has: layout.off_has + args.receipts]
    if not has_own[tip.planted.astype(np.int64)].all():
        raise SystemExit("bench self-check failed (rank %d): the scan missed planted matches" % rank)
    per_rank = [None] * world
    ranks.dist.all_gather_object(per_rank, {"rank": rank, "blocks": tip.n_blocks, "claims": n_claims, "setup_seconds": round(t_gen, 2),
                                            "kernels_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in kern.items()}})
    if rank == 0:
        out = {
            "metric": METRIC, "value": total_claims * args.steps / elapsed, "unit": "proofs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[2] (the 1M-receipt tipset the metric is quoted on), one tipset PER RANK "
                            "(%d tipsets of %d receipts, seeds base+3+rank; %d claims in all): step = per-rank CID index + "
                            "Blake2b-256 CID check + exec-order reconstruction + verify_event_proof of every claim + event-filter "
                            "scan, closed by ONE ncclAllGather of %d bytes per rank [header | status | has map | CID bitmap]"
                            % (world, args.receipts, total_claims, layout.bytes_per_rank),
                "receipts_per_gpu": args.receipts, "claims_per_gpu": n_claims, "tipsets": world,
                "sharding": "--shard tipsets: independent tipsets, one per rank; no data-path collective besides the closing "
                            "all-gather (SURVEY.md §8e; `--shard receipts` cuts ONE tipset by receipt range instead)",
                "collective": "ncclAllGather called by libipcfp.so (RCCL resolved at run time)",
                "allgather_bytes_per_rank": layout.bytes_per_rank, "scan_matches_per_tipset": matches,
                "per_rank": per_rank, "device": info["name"],
            },
            "roofline": roof,
            "kernels_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in kern.items()},
            "window": "T3 (tipsets resident in HBM; index rebuilt and every cached enumeration dropped each step)",
        }
    w.close()
    return out if rank == 0 else None


def run_tipset_sharded(args, eng, info, torch, dist, world, rank, dev, ranks):
    """--gpus N > 1: ONE tipset, N receipt-range shards (strong scaling).  Setup (untimed, the analogue of the
    single-GPU upload): every rank holds the bundle in registered host memory and pulls ITS shard out of it by itself
    (ipcfp_witness_create_shard_pull); nobody holds the whole witness in HBM.  Timed step: shard.TipsetShard.step."""
    import ipc_filecoin_proofs_amd as ipcfp
    from ipc_filecoin_proofs_amd import shard
    from tools.synth import SEED_BASE, Tipset

    t_gen = time.perf_counter()
    tip = Tipset(seed=SEED_BASE + 3, n_receipts=args.receipts, n_parents=5, dup_permille=20,
                 n_planted=max(1, args.receipts // 1000), max_events=4, no_events_permille=0, variety=0)
    ts, cl, blob, blob_len = ipcfp.pack_event_claims(
        tip.parent_cids, tip.child_cid, tip.parent_epoch, tip.child_epoch, tip.claim_exec, tip.claim_event,
        tip.claim_emitter, tip.exec_order[tip.claim_exec.astype(np.int64)], tip.claim_ntopics, tip.claim_topics,
        tip.claim_datalen, tip.claim_data)
    # SELF-PLANNED: every rank is given the bundle in ITS host memory (transport form, an ingest buffer registered when it
    # was made) and the tipset key — nothing else; it finds and fetches its shard itself (ipcfp_witness_create_shard_pull)
    pk = ipcfp.PackedWitnessTables(tip.data, tip.off, tip.lens, tip.cids, ingest=True)
    ipcfp.host_register(pk.data)
    ipcfp.host_register(pk.digests)
    sh = shard.TipsetShard.from_pull(eng, pk, tip.parent_cids, tip.child_cid, tip.receipts_root, world, rank)
    sh.route(ts, cl, blob, blob_len)
    t_gen = time.perf_counter() - t_gen

    def allreduce_max(v):
        t = torch.from_numpy(v.copy()).to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.cpu().numpy()

    layout = shard.Layout.agree(sh.counts, allreduce_max)
    # the communicator of the data path: RCCL through libipcfp.so (Ranks.make_comm: agreed on beforehand, loud on failure)
    comm = ranks.make_comm() if world > 1 else None
    collective = "ncclAllGather called by libipcfp.so (RCCL resolved at run time)" if comm is not None else "single rank: the packed message is the result"

    def dev_bytes(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)

    d_cl, d_blob, d_hdr = dev_bytes(sh.claims), dev_bytes(sh.blob), dev_bytes(sh.header())
    d_status = torch.zeros(layout.w_status, dtype=torch.uint8, device=dev)
    d_has = torch.zeros(layout.w_has, dtype=torch.uint8, device=dev)
    d_stage = torch.zeros(layout.bytes_per_rank, dtype=torch.uint8, device=dev)
    d_recv = torch.zeros(world * layout.bytes_per_rank, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    filt = (tip.topic0, tip.topic1, tip.filter_actor)

    def step():
        sh.step(layout, comm, filt, d_cl.data_ptr(), d_blob.data_ptr(), d_status.data_ptr(), d_has.data_ptr(),
                d_hdr.data_ptr(), d_stage.data_ptr(), d_recv.data_ptr())

    def fence():
        eng.sync()
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
    elapsed = time.perf_counter() - t0
    eng.profile_enable(False)
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    kern = {}
    for k in STEP_KERNEL_GROUPS + ("allgather",):
        cnt, ms = eng.profile_read(k)
        kern[k] = {"launches": cnt, "ms_per_step": ms / args.steps}
    # ---- what was timed must be right: every rank merges the gathered messages and checks the WHOLE tipset ----
    gathered = d_recv.cpu().numpy()
    positions = shard.route_all(cl["exec_index"], sh.n_receipts_total, world)
    merged = shard.merge(gathered, layout, world, positions, len(cl), sh.n_receipts_total)
    if not (merged["status"] == 1).all() or merged["n_bad_cids"] or merged["scan_status"] != 1:
        raise SystemExit("bench self-check failed (rank %d): merged verdicts are not all TRUE" % rank)
    if merged["n_matches"] < len(tip.planted) or not merged["has"][tip.planted.astype(np.int64)].all():
        raise SystemExit("bench self-check failed (rank %d): the merged scan missed planted matches" % rank)
    k_launches, k_ms = kern["blake2b_cid"]["launches"], kern["blake2b_cid"]["ms_per_step"] * args.steps
    k_avg_ms = k_ms / max(k_launches, 1)
    algo_bytes = float(sh.pull_stats["payload_bytes"] + sh.witness.n * 44)
    achieved = algo_bytes / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
    # ---- window T2 of the cut: every rank plans, fetches and verifies ITS shard out of host memory at the same time ----
    t2_ms, h2d_bytes = None, None
    if args.t2_reps > 0:
        own = merged["status"][sh.positions.astype(np.int64)]
        reps = []
        for _ in range(args.t2_reps + 1):
            fence()
            t0 = time.perf_counter()
            st_p, w2, lo2, hi2, nr2, stats2 = eng.witness_shard_pull(pk, tip.parent_cids, tip.child_cid, world, rank)
            w2.verify_cids_async()
            a2, st2 = w2.verify_event_claims_range(ts, cl, blob, blob_len, lo2, hi2, rank == world - 1)
            sst, has2, m2, _ = w2.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor,
                                              want_touched=False, caps=(hi2 - lo2, MATCH_CAP))
            cs2, nbad2 = w2.cid_results()
            el = time.perf_counter() - t0
            w2.close()
            h2d_bytes = int(stats2["table_bytes"] + stats2["block_bytes"] + len(st2) * cl.dtype.itemsize + sh.blob_len)
            if st_p != 1 or sst != 1 or nbad2 or not np.array_equal(st2, own) or not np.array_equal(has2, merged["has"][sh.lo: sh.hi]):
                raise SystemExit("bench self-check failed (rank %d): the self-planned shard from host differs from the resident one" % rank)
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            reps.append(float(tt.item()))
        t2_ms = min(reps[1:]) * 1e3
    per_rank = [None] * world
    dist.all_gather_object(per_rank, {"rank": rank, "blocks": int(sh.witness.n), "claims": int(sh.n_claims),
                                      "receipts": [int(sh.lo), int(sh.hi)], "h2d_bytes": h2d_bytes,
                                      "kernels_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in kern.items()}})
    if rank == 0:
        total_claims = len(cl)
        out = {
            "metric": METRIC, "value": total_claims * args.steps / elapsed, "unit": "proofs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[2] (the 1M-receipt tipset the metric is quoted on): ONE tipset of %d "
                            "receipts (%d witness blocks, %.3f GB, one EventProof claim per receipt) cut into %d "
                            "receipt-range shards; step = per-rank CID index + Blake2b-256 CID check + range-restricted "
                            "event scan + exec-order reconstruction (replicated) + verify_event_proof of the rank's "
                            "claims, closed by one ncclAllGather of %d bytes per rank"
                            % (args.receipts, tip.n_blocks, tip.stats["payload_bytes"] / 1e9, world, layout.bytes_per_rank),
                "receipts": args.receipts, "claims": total_claims, "witness_blocks": tip.n_blocks,
                "sharding": "receipt-range shards of one tipset (SURVEY.md §8e): events AMTs + receipts-AMT paths per "
                            "rank, headers/TxMeta/message AMTs replicated; one RCCL all-gather per step",
                "allgather_bytes_per_rank": layout.bytes_per_rank, "collective": collective, "scan_matches": merged["n_matches"],
                "per_rank": per_rank, "device": info["name"], "setup_seconds_untimed": round(t_gen, 2),
            },
            "roofline": {"bound": "hbm", "limiter": "valu", "kernel": "k_blake2b256_cid (rank 0's shard)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel_avg_ms": k_avg_ms, "launches": k_launches, "algorithmic_bytes_per_launch": algo_bytes},
            "kernels_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in kern.items()},
            "window": "T3 (shards resident in HBM; index rebuilt and every cached enumeration dropped each step)",
        }
        if t2_ms is not None:
            out["value_T2"] = total_claims / (t2_ms * 1e-3)
            out["window_T2"] = {"value": out["value_T2"], "unit": "proofs/s", "ms_per_tipset": t2_ms, "h2d_bytes_this_rank": h2d_bytes,
                                "reps": args.t2_reps,
                                "note": "max over ranks of: the rank's OWN shard (witness blocks + routed claims) from pageable host "
                                        "memory over its own PCIe link, repack, index, K1, verify, range-restricted scan, status bytes / "
                                        "CID verdicts / has-map / match records back; every rank SELF-PLANNED (ipcfp_witness_create_shard_pull): plan, "
                                        "cut and claim slice are inside the window, registering the ingest buffer is not"}
    sh.close()
    ipcfp.host_unregister(pk.data)
    ipcfp.host_unregister(pk.digests)
    return out if rank == 0 else None


def load_traffic(n_blocks):
    """The newest `profiles/rNN_traffic.json` (tools/pmc_traffic.py: one `rocprofv3 --pmc FETCH_SIZE` pass of this
    command, every kernel's counter scaled by the factor calibrated for its access pattern with
    tools/ubench/fetch_calib) — used only when it was taken on this workload."""
    best = None
    try:
        names = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json") and f[0] == "r")
    except OSError:
        return None
    for name in reversed(names):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                tr = json.load(f)
            if tr["workload"]["witness_blocks"] == n_blocks and "groups" in tr:
                tr["file"] = "profiles/" + name
                best = tr
                break
        except (OSError, KeyError, ValueError):
            continue
    return best


def k1_roofline(eng, lens, n_blocks, with_traffic=False, extra_note="", timed=None):
    """`roofline` of K1 from the HIP events libipcfp.so recorded on the stream the kernel ran on.  Algorithmic
    bytes per launch follow SURVEY.md §8(d): len_i + 32 (expected digest) + 12 (offset u64 + len u32) per block.
    The kernel is bound by the issue rate of its own integer VALU stream (`bound`), so `frac` (of HBM peak, as the
    bench contract defines it) reads against `valu.ceiling_GBps`, not against 1.0."""
    k_launches, k_ms = timed if timed is not None else eng.profile_read("blake2b_cid")
    k_avg_ms = k_ms / max(k_launches, 1)
    lens64 = np.asarray(lens, dtype=np.int64)
    algo_bytes = float(lens64.sum() + n_blocks * 44)
    achieved = algo_bytes / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
    chunks = float(np.maximum(1, (lens64 + 127) // 128).sum())
    # 2 083 VALU instructions per wavefront per 128-byte chunk (SQ_INSTS_VALU of profiles/r01_final_pmc.txt over the
    # tipset witness's chunk count); one wavefront instruction = 64 lane operations
    lane_ops = chunks * VALU_INSTS_PER_CHUNK
    # the ubench's issue costs put one compression at ~6.8k cycles per wavefront: 1024 SIMDs x 64 lanes
    # x 128 B per 6.8k cycles at the 2.18 GHz the chip sustains under this kernel, scaled from compressed bytes
    # (chunks x 128) to algorithmic bytes
    chunk_rate = 1024 * 64 * 128 / (K1_CYCLES_PER_WAVE_CHUNK / K1_SUSTAINED_GHZ)  # bytes per ns = GB/s
    ceiling = chunk_rate * algo_bytes / (chunks * 128.0)
    traffic, src = None, None
    if with_traffic:
        tr = load_traffic(n_blocks)
        if tr and "blake2b_cid" in tr["groups"]:
            traffic = tr["groups"]["blake2b_cid"]["traffic_bytes_per_step"]
            src = "%s (r%s: L2 memory-side read requests, TCC_EA0_RDREQ via FETCH_SIZE, x%.2f calibrated)" % (
                tr["file"], tr.get("round"), tr["groups"]["blake2b_cid"].get("factor", 0.0))
    return {
        "bound": "hbm", "limiter": "valu", "kernel": "k_blake2b256_cid", "achieved": achieved, "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_6290": achieved / 6290.0,
        "frac_of_valu_ceiling": achieved / ceiling if ceiling > 0 else 0.0,
        "traffic": traffic, "traffic_source": src, "kernel_avg_ms": k_avg_ms, "launches": k_launches,
        "algorithmic_bytes_per_launch": algo_bytes,
        "valu": {"ceiling_GBps": ceiling,
                 "ceiling_basis": "tools/ubench/valu_rates (profiles/r01_ubench_valu_rates.log): a G function = 8 xor + 6 add64 "
                                  "+ 6 alignbit ~ 70 issue cycles, %.0f cycles per wavefront per 128-byte chunk, 1024 SIMDs at "
                                  "the %.2f GHz sustained under this kernel = %.0f GB/s of compressed bytes; x algorithmic / "
                                  "compressed bytes of THIS batch (final-chunk padding)" % (K1_CYCLES_PER_WAVE_CHUNK, K1_SUSTAINED_GHZ, chunk_rate),
                 "achieved_Tops": lane_ops / (k_avg_ms * 1e-3) / 1e12 if k_avg_ms > 0 else 0.0, "peak_Tops": VALU_PEAK_TOPS,
                 "frac": (lane_ops / (k_avg_ms * 1e-3) / 1e12 / VALU_PEAK_TOPS) if k_avg_ms > 0 else 0.0,
                 "lane_ops_per_launch": lane_ops,
                 "note": "32-bit integer lane operations; peak = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz. The G function's "
                         "64-bit adds, v_alignbit and v_perm issue at about half that rate on gfx950"},
        "note": "limited by integer VALU throughput, not by HBM (DESIGN.md §K1): `peak`/`frac` are the HBM figures the bench "
                "contract asks for, `valu.ceiling_GBps` is what the instruction stream allows" + extra_note,
    }


# kernel groups of a tipset step = the profile ids of include/ipcfp.h, in stream order
STEP_KERNEL_GROUPS = ("cid_index", "blake2b_cid", "tipset_prologue", "amt_walk", "exec_order", "event_scan", "event_verify", "replay")


def tipset_kernels(tip, kern, steps, n_claims, claim_bytes, bracketed_ms):
    """`kernels[]`: every kernel group of the step with its algorithmic bytes, its HIP-event time (second pass, each
    group bracketed), GB/s, fraction of HBM peak, and calibrated HBM traffic where the newest PMC file has it."""
    st = tip.stats
    n_msgs = int(len(tip.exec_order))  # (distinct messages; the lists carry dup_permille more)
    n_receipts = int(tip.params["n_receipts"])
    lens64 = np.asarray(tip.lens, dtype=np.int64)
    algo = {
        # 40-byte CID read, 8-byte slot written
        "cid_index": (48.0 * tip.n_blocks, "hbm", "latency (random 8-byte probes into a 64 MB table)",
                      "k_index_insert", "per block: 40-byte CID + 8-byte slot"),
        "blake2b_cid": (float(lens64.sum() + 44 * tip.n_blocks), "valu", "valu", "k_blake2b256_cid",
                        "per block: len + 32 (digest) + 12 (offset, len)"),
        "tipset_prologue": (None, "latency", "latency (5 headers, 10 TxMeta/AMT roots: a dependent chain of ~4 block reads)",
                            "k_tipset_prepare, k_enum_roots", "a few KB"),
        "amt_walk": (float(st["message_amt_bytes"] + st["receipts_amt_bytes"]) + 8.0 * n_msgs + 16.0 * n_receipts, "hbm",
                     "latency (the narrow levels in one single-workgroup launch, 3 dependent levels, then leaves)",
                     "k_dense_top, k_dense_level, k_dense_link_leaves, k_dense_leaves",
                     "message AMTs + receipts AMT read once; 8 B per message key and 16 B per receipt leaf written"),
        "exec_order": (28.0 * n_msgs, "hbm", "latency (hash-table insert, scan, scatter)",
                       "k_exec_insert_flags, k_exec_flag_sums, k_scan_tiles_u64, k_exec_apply_finish",
                       "per message: 8-byte key, 8-byte slot, first/pos/inv words"),
        # (the block-order parse cannot know which blocks are events AMTs before it has read them: EVERY block of the
        # witness goes through it once — VERDICT r3 weak #11: the events-AMT bytes alone understated what it must read)
        "event_scan": (float(lens64.sum()) + 24.0 * n_receipts, "hbm", "valu+latency (one CBOR parser per lane)",
                       "k_block_events, k_receipt_events, k_count_from_table (aux stream)",
                       "every witness block read once by the block-order parse (%.0f MB of them are events-AMT blocks); 24 B of records per receipt"
                       % (float(st["events_amt_bytes"]) / 1e6)),
        "event_verify": (float(claim_bytes) + 1.0 * n_claims + 112.0 * n_claims, "hbm", "latency (random record reads)",
                         "k_verify_events_table", "claim + claimed entry bytes, status byte, ~112 B of receipt/event records and event bytes per claim"),
        "replay": (None, "latency", "latency", "k_verify_events (fallback walkers; idle on the table path)", ""),
    }
    tr = load_traffic(tip.n_blocks)
    out = []
    for name in STEP_KERNEL_GROUPS:
        ms = kern[name]["ms_per_step"]
        if kern[name]["launches"] == 0:
            continue
        b, bound, limiter, hip, basis = algo[name]
        rec = {"group": name, "hip_kernels": hip, "ms_per_step": round(ms, 4), "brackets_per_step": kern[name]["launches"] / steps,
               "algorithmic_bytes": b, "basis": basis, "bound": bound, "limiter": limiter}
        if b and ms > 0:
            rec["achieved_GBps"] = round(b / (ms * 1e-3) / 1e9, 1)
            rec["frac_of_hbm_peak"] = round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if tr and name in tr["groups"]:
            g = tr["groups"][name]
            rec["traffic_bytes"] = g["traffic_bytes_per_step"]
            rec["traffic_factor"] = g.get("factor")
            if b:
                rec["traffic_over_algorithmic"] = round(g["traffic_bytes_per_step"] / b, 3)
        out.append(rec)
    return {"pass": "second pass of %d steps with EVERY group bracketed by its own HIP-event pair (%.3f ms per step; the timed "
                    "region brackets K1 alone).  blake2b_cid and event_scan run on side streams beside the main-stream chain "
                    "cid_index > tipset_prologue > amt_walk > exec_order > event_verify, so the times overlap and do not add up "
                    "to the step" % (steps, bracketed_ms),
            "traffic_source": (tr["file"] if tr else None), "groups": out}


def _gather_line(world, width):
    return ("; with --gpus %d: index ranges cut by ipcfp_shard_range, one ncclAllGather of %d status bytes per rank closes the step"
            % (world, width)) if world > 1 else ""


def run_cid(args, eng, info, torch, ranks):
    """BASELINE.json configs[1]: N x 1 KiB blocks (59 03 FD + 1021 PRNG bytes), digest bit flipped where i % 1024 == 7.
    --gpus N: rank r holds ONLY the blocks [lo_r, hi_r) and one ncclAllGather of the per-rank CID bitmaps closes the step."""
    from tools.synth import SEED_BASE

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ipc_filecoin_proofs_amd as ipcfp_mod
    import oracle_lib

    world, rank, dev = ranks.world, ranks.rank, ranks.dev
    n = args.blocks
    data, off, lens = make_cfg2(n, SEED_BASE + 2)
    rng, _ = range_layout(n, world)
    lo, hi = rng[rank]
    m = hi - lo
    d_lo = data[lo * 1024: hi * 1024]
    o_lo = off[lo:hi] - np.uint64(lo * 1024)
    l_lo = lens[lo:hi]
    dig = eng.blake2b256(d_lo, o_lo, l_lo)
    cids = np.zeros((m, 40), dtype=np.uint8)
    cids[:, :6] = np.frombuffer(bytes.fromhex("0171a0e40220"), dtype=np.uint8)
    cids[:, 6:38] = dig
    glob = np.arange(lo, hi)
    flipped_local = np.nonzero(glob % 1024 == 7)[0]
    cids[flipped_local, 6] ^= 1
    w = eng.witness(d_lo, o_lo, l_lo, cids)
    width = ((max(h - l for l, h in rng) + 31) // 32 * 4 + 15) & ~15  # bitmap bytes per rank
    comm = ranks.make_comm()
    d_recv = torch.zeros(world * width, dtype=torch.uint8, device=dev) if world > 1 else None
    d_send = torch.zeros(width, dtype=torch.uint8, device=dev) if world > 1 else None
    bits_bytes = (m + 31) // 32 * 4

    def step():
        w.verify_cids_async()
        if comm is not None:  # (the engine joins K1's stream before the collective)
            ipcfp_mod.allgather_segments(eng, comm, [w.cid_bitmap_ptr], [bits_bytes], d_send.data_ptr(), d_recv.data_ptr(), width)

    elapsed = timed(ranks, step, args.steps, args.warmup)
    st, nbad = w.cid_results()
    if nbad != len(flipped_local) or not (st[flipped_local] == 0).all() or int(st.sum()) != m - len(flipped_local):
        raise SystemExit("bench self-check failed: CID verdicts")
    if world > 1:  # every rank checks the WHOLE batch's gathered bitmap
        g = d_recv.cpu().numpy().reshape(world, width)
        for r, (l, h) in enumerate(rng):
            bits = np.unpackbits(g[r, : (h - l + 31) // 32 * 4], bitorder="little")[: h - l]
            want = (np.arange(l, h) % 1024 != 7).astype(np.uint8)
            if not np.array_equal(bits, want):
                raise SystemExit("bench self-check failed (rank %d): gathered CID bitmap of rank %d" % (rank, r))
    roof = k1_roofline(eng, l_lo, m, extra_note="; %d blocks = %d wavefronts on 1024 SIMDs" % (m, (m + 63) // 64))
    out = {"metric": METRIC, "value": n * args.steps / elapsed, "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": "BASELINE.json configs[1]: %d Blake2b-256 CID verifications over 1 KiB blocks; step = one "
                                  "K1 launch over the resident batch (1 proof = 1 CID check)%s" % (n, _gather_line(world, width)),
                      "blocks": n, "blocks_per_gpu": m, "device": info["name"]},
           "roofline": roof, "window": "T3"}
    with_workload_traffic(out["roofline"], "cid", roof.get("algorithmic_bytes_per_launch"))
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        orc, march = oracle_lib.load_native()
        exp = np.ascontiguousarray(cids[:, 6:38])
        best = None
        for t in [1] + thread_sweep(orc.num_procs()):
            orc.use_threads(t)
            t0 = time.perf_counter()
            ok, good = orc.blake2b256_verify(data, off, lens, exp, threads=t)
            dt = time.perf_counter() - t0
            if not np.array_equal(ok, st):
                raise SystemExit("cpu_baseline: CID verdicts differ")
            if best is None or n / dt > best[0]:
                best = (n / dt, t)
            if t == 1:
                one = n / dt
        out["cpu_baseline"] = {"value": best[0], "unit": "proofs/s", "cores": best[1], "kind": "port", "value_1_thread": one,
                               "cpu_quota_cpus": cpu_quota(),
                               "sample": "C++ oracle Blake2b-256 over all %d blocks, -march=%s, best thread count" % (n, march)}
    w.close()
    return out


def _state_tipset():
    from tools.synth import SEED_BASE, Tipset

    return Tipset(seed=SEED_BASE + 4, n_receipts=8, n_planted=0, n_actors=4_000_000, n_contracts=10_000,
                  slots_per_contract=256, keep_full_state=0, n_actor_queries=int(65536 * 1.01))


def _idaddr(i: int) -> bytes:
    b = bytearray([0])
    while True:
        c = i & 0x7F
        i >>= 7
        if i:
            b.append(c | 0x80)
        else:
            b.append(c)
            return bytes(b)


def run_hamt(args, eng, info, torch, ranks, state=None):
    """BASELINE.json configs[3]: 4M-actor state tree (HAMT v3, bit width 5), 65 536 present ids + 1 % absent.
    The state tree is replicated; --gpus N cuts the QUERY index range, each rank runs ipcfp_hamt_get_device on its keys
    (resident in HBM) and one ncclAllGather of the status bytes closes the step."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    world, rank, dev = ranks.world, ranks.rank, ranks.dev
    T = state if state is not None else _state_tipset()
    w = eng.witness(T.data, T.off, T.lens, T.cids)
    keys = [_idaddr(int(i)) for i in T.query_ids]
    n = len(keys)
    rng, width = range_layout(n, world)
    lo, hi = rng[rank]
    m = hi - lo
    kl = np.array([len(k) for k in keys[lo:hi]], dtype=np.uint32)
    ko = np.zeros(m, dtype=np.uint32)
    ko[1:] = np.cumsum(kl[:-1])
    kb = np.frombuffer(b"".join(keys[lo:hi]) + bytes(32), dtype=np.uint8).copy()
    d_kb, d_ko, d_kl = torch.from_numpy(kb).to(dev), torch.from_numpy(ko.view(np.int32)).to(dev), torch.from_numpy(kl.view(np.int32)).to(dev)
    d_st = torch.zeros(width, dtype=torch.uint8, device=dev)
    d_loc = torch.zeros(m * 12 + 16, dtype=torch.uint8, device=dev)
    d_recv = torch.zeros(world * width, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    comm = ranks.make_comm()

    def step():
        w.hamt_get_device(T.actors_root, 5, "actor_state", d_kb.data_ptr(), d_ko.data_ptr(), d_kl.data_ptr(), m,
                          d_st.data_ptr(), d_loc.data_ptr())
        if comm is not None:
            comm.allgather_device(d_st.data_ptr(), d_recv.data_ptr(), width)

    elapsed = timed(ranks, step, args.steps, args.warmup)
    cnt, ms = eng.profile_read("hamt_get")
    k_avg_ms = ms / max(cnt, 1)
    present = T.query_present.astype(bool)
    want = np.where(present, 1, 32).astype(np.uint8)
    gs = d_st.cpu().numpy()[:m]
    if not np.array_equal(gs, want[lo:hi]):
        raise SystemExit("bench self-check failed: actor gets")
    if world > 1:
        g = d_recv.cpu().numpy().reshape(world, width)
        for r, (l, h) in enumerate(rng):
            if not np.array_equal(g[r, : h - l], want[l:h]):
                raise SystemExit("bench self-check failed (rank %d): gathered statuses of rank %d" % (rank, r))
    # walk bytes per get (SURVEY.md §8d cfg 4 (ii)): per level 4 B bitfield + 43 B link, + <= 3 x 105 B bucket = 0.55 KB
    algo = m * 550.0
    out = {"metric": METRIC, "value": n * args.steps / elapsed, "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": "BASELINE.json configs[3]: HAMT state-tree actor lookup, %d actors, %d gets (%d absent), bit "
                                  "width 5; step = one ipcfp_hamt_get_device call (keys, statuses and locations resident in HBM, "
                                  "witness resident)%s" % (4_000_000, n, int((~present).sum()), _gather_line(world, width)),
                      "gets_per_gpu": m, "witness_blocks": T.n_blocks, "witness_bytes": T.stats["payload_bytes"], "device": info["name"]},
           "roofline": {"bound": "hbm", "limiter": "latency + instruction issue", "kernel": "k_hamt_lv_start + fused top (k_hamt_lv_advance_top) + per level k_hamt_lv_parse_actor (three size classes, 8 or 32 lanes per node) + k_hamt_lv_advance, k_hamt_get behind them (the K7 group: one HIP-event bracket)", "achieved": algo / (k_avg_ms * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algo / (k_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": None, "kernel_avg_ms": k_avg_ms, "launches": cnt, "algorithmic_bytes_per_launch": algo,
                        "bytes_basis": "SURVEY.md §8(d) cfg 4 (ii) WALK ONLY: 0.55 KB per get (4 B bitfield + 43 B link per level, <= 3 x 105 B bucket) — "
                                       "this call hashes no node, so (i) 'verified get' (1.07 KB per get: every path node's bytes once, for its CID check) "
                                       "is not what it moves; by (i) the figures are x 1.95",
                        "value_kernel_only_gets_per_s": m / (k_avg_ms * 1e-3),
                        "note": "0.55 KB walked per get (§8d cfg 4 (ii)); level by level: every VISITED node decoded once by 32 lanes (a visited node is decoded completely, as the reference does: the distinct nodes alone are 3.7x these bytes), a query = SHA-256 + one record per level"},
           "window": "T3"}
    with_workload_traffic(out["roofline"], "hamt", algo)
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        orc, march = oracle_lib.load_native()
        sweep = thread_sweep(orc.num_procs()) or [1]
        ost = orc.store(T.data, T.off, T.lens, T.cids, threads=sweep[0])
        best = None
        for t in sweep:
            orc.use_threads(t)
            os_, _ = ost.hamt_get(T.actors_root, 5, "actor_state", keys, want_values=False, threads=t)
            os_, _ = ost.hamt_get(T.actors_root, 5, "actor_state", keys, want_values=False, threads=t)  # (second call: arenas grown)
            dt = ost.last_call_seconds
            if not np.array_equal(os_, gs):
                raise SystemExit("cpu_baseline: actor-get statuses differ")
            if best is None or n / dt > best[0]:
                best = (n / dt, t)
        ost.close()
        out["cpu_baseline"] = {"value": best[0], "unit": "proofs/s", "cores": best[1], "kind": "port", "cpu_quota_cpus": cpu_quota(),
                               "sample": "C++ oracle Hamt::get of all %d keys (the C call alone, store built before the clock), "
                                         "-march=%s, OpenMP, best thread count" % (n, march)}
    w.close()
    return out


def run_storage(args, eng, info, torch, ranks, state=None):
    """BASELINE.json configs[4]: 10 000 contracts x 256 slots, every StorageProof claim, 0.1 % with a wrong value.
    The state tree and the contracts are replicated; --gpus N cuts the CLAIM index range (the loop of
    src/proofs/verifier.rs:19-28) and one ncclAllGather of the status bytes closes the step."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ipc_filecoin_proofs_amd as ipcfp
    import oracle_lib

    world, rank, dev = ranks.world, ranks.rank, ranks.dev
    T = state if state is not None else _state_tipset()
    w = eng.witness(T.data, T.off, T.lens, T.cids)
    n = len(T.sc_actor)
    cl = ipcfp.pack_storage_claims(T.child_cid, T.state_root, T.child_epoch, T.sc_actor, T.sc_actor_state,
                                   T.sc_storage_root, T.sc_slot, T.sc_value)
    wrong = np.arange(500, n, 1000)
    cl["value"][wrong, 31] ^= 1
    want = np.ones(n, dtype=np.uint8)
    want[wrong] = 21
    rng, width = range_layout(n, world)
    lo, hi = rng[rank]
    m = hi - lo
    d_cl = torch.from_numpy(cl[lo:hi].view(np.uint8).reshape(-1).copy()).to(dev)
    d_st = torch.zeros(width, dtype=torch.uint8, device=dev)
    d_recv = torch.zeros(world * width, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    comm = ranks.make_comm()

    def step():
        w.verify_storage_claims_device(d_cl.data_ptr(), m, d_st.data_ptr())
        if comm is not None:
            comm.allgather_device(d_st.data_ptr(), d_recv.data_ptr(), width)

    elapsed = timed(ranks, step, args.steps, args.warmup)
    cnt, ms = eng.profile_read("storage_verify")
    k_avg_ms = ms / max(cnt, 1)
    got = d_st.cpu().numpy()[:m]
    if not np.array_equal(got, want[lo:hi]):
        raise SystemExit("bench self-check failed: storage verdicts")
    if world > 1:
        g = d_recv.cpu().numpy().reshape(world, width)
        for r, (l, h) in enumerate(rng):
            if not np.array_equal(g[r, : h - l], want[l:h]):
                raise SystemExit("bench self-check failed (rank %d): gathered statuses of rank %d" % (rank, r))
    algo = float(T.stats["payload_bytes"]) + m * 760.0  # §8d cfg 5: unique witness bytes + 0.76 KB walked per proof
    out = {"metric": METRIC, "value": n * args.steps / elapsed, "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": "BASELINE.json configs[4]: EVM storage proofs, 10 000 contracts x 256 slots = %d claims "
                                  "(0.1 %% wrong), Keccak slot key + state-tree HAMT get + EVM state + storage HAMT get; "
                                  "step = one ipcfp_verify_storage_claims_device call, claims resident%s" % (n, _gather_line(world, width)),
                      "claims_per_gpu": m, "witness_blocks": T.n_blocks, "witness_bytes": T.stats["payload_bytes"], "device": info["name"]},
           "roofline": {"bound": "hbm", "limiter": "valu (node table) + L1 line lookups of divergent loads (claim kernel)", "kernel": "k_hamt_node_table_lane + k_hamt_lv_parse_actor (the per-call node table, side streams) + k_storage_run_* + k_verify_storage_table (the storage-proof group: one HIP-event bracket)", "achieved": algo / (k_avg_ms * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algo / (k_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": None, "kernel_avg_ms": k_avg_ms, "launches": cnt, "algorithmic_bytes_per_launch": algo,
                        "bytes_basis": "SURVEY.md §8(d) cfg 5: unique witness bytes read once + 0.76 KB per proof (64 B Keccak input + walk bytes)"},
           "window": "T3"}
    with_workload_traffic(out["roofline"], "storage", algo)
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        orc, march = oracle_lib.load_native()
        ost = orc.store(T.data, T.off, T.lens, T.cids, threads=0)
        sample = min(n, 400_000)
        best = None
        for t in [1] + thread_sweep(orc.num_procs()):
            k = sample if t > 1 else 20_000
            orc.use_threads(t)
            t0 = time.perf_counter()
            g = ost.verify_storage_claims_packed(cl[:k], threads=t)
            dt = time.perf_counter() - t0
            if not np.array_equal(g, want[:k]):
                raise SystemExit("cpu_baseline: storage verdicts differ")
            if best is None or k / dt > best[0]:
                best = (k / dt, t)
            if t == 1:
                one = k / dt
        ost.close()
        out["cpu_baseline"] = {"value": best[0], "unit": "proofs/s", "cores": best[1], "kind": "port", "value_1_thread": one,
                               "cpu_quota_cpus": cpu_quota(),
                               "sample": "C++ oracle verify_storage_proof (store built once) on the first %d claims, "
                                         "-march=%s, best thread count" % (sample, march)}
    w.close()
    return out


def cpu_quota():
    """CPUs the container may use on average (cgroup CFS quota), or None when unlimited / unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = float(f.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def thread_sweep(procs):
    """Thread counts a CPU-baseline leg tries beside 1: around the container's CPU quota when it has one (more threads
    than ≈ 2-4x the quota only get the process throttled: profiles/r04_cpu_scaling.txt), else up to every processor."""
    quota = cpu_quota()
    if procs > 1 and quota and quota < procs:
        q = max(2, int(round(quota)))
        return sorted({min(procs, q), min(procs, 2 * q), min(procs, 4 * q), min(procs, 6 * q)})
    return sorted({max(2, procs // 8), max(2, procs // 4), max(2, procs // 2), procs}) if procs > 1 else []


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
    procs = orc.num_procs()
    n = len(tip.claim_exec)
    sample = min(sample, n)
    sample_mt = min(sample_mt or sample, n)
    expect32 = np.ascontiguousarray(tip.cids[:, 6:38])
    ec_small = claims_mod.EventClaims(tip, indices=np.arange(sample))
    ec_big = ec_small if sample_mt == sample else claims_mod.EventClaims(tip, indices=np.arange(sample_mt))
    ec1 = claims_mod.EventClaims(tip, indices=np.arange(1))
    # thread counts: 1, then a sweep — the baseline is the BEST all-cores figure, not the one at the largest count.
    # "All cores" is what the box lets this process HAVE: the GPU boxes show 256 processors and give the container a CFS
    # quota of 16 CPUs (cpu.max "1600000 100000", profiles/r04_cpu_topology.txt) — threads beyond ≈ 2x the quota only
    # get the process throttled (round 3's sweep to 256 threads ran 10x slower there than at 32:
    # profiles/r04_cpu_scaling.txt).  With a quota the sweep brackets it; without one it goes up to every processor.
    quota = cpu_quota()
    sweep = thread_sweep(procs)
    legs = {}
    for threads in [1] + sweep:
        sec = {}
        if threads > 1:
            orc.use_threads(threads)  # team start-up and arena growth are not part of any phase
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
        e = ec_small if threads == 1 else ec_big
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
    best = min(sweep, key=lambda t: legs[t]["step"]) if sweep else 1
    cores = best
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
    phys_cores = None
    try:
        seen = set()
        phys, core = None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and not cpu_model:
                    cpu_model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        seen.add((phys, core))
                    phys, core = None, None
        phys_cores = len(seen) or None
    except OSError:
        pass
    return {
        "value": n / legs[best]["step"],
        "unit": "proofs/s",
        "cores": cores,
        "kind": "port",
        "sample": "C++ oracle (restatement of the reference; the Rust crate cannot be built here), g++ -O3 -march=%s, "
                  "OpenMP, best of %s threads on %d host processors (%d threads): sharded store build + Blake2b CID check "
                  "+ two-pass event scan + exec-order on the FULL %d-receipt tipset, verify_event_proof on the first %d of "
                  "%d claims scaled linearly; store and exec order built once per tipset (BASELINE.md variant B2 all-cores)"
                  % (march, "/".join(str(t) for t in sweep) or "1", procs, cores, tip.params["n_receipts"],
                     legs[best]["verify_sample_claims"], n),
        "seconds": {k: v for k, v in legs[best].items()},
        "thread_sweep_step_seconds": {str(t): legs[t]["step"] for t in sorted(legs)},
        "value_1_thread": n / legs[1]["step"],
        "seconds_1_thread": {k: v for k, v in legs[1].items()},
        "scaling_1_to_all": legs[1]["step"] / legs[best]["step"],
        "b1_as_written": {"receipts": nb1, "proofs_per_s": eb.n / t_b1, "seconds": t_b1,
                          "b2_1_thread_same_tipset_proofs_per_s": eb.n / t_b2_small,
                          "note": "reference semantics incl. per-proof execution-order rebuild "
                                  "(events/verifier.rs:190): quadratic, measured at %d receipts, NOT extrapolated" % nb1},
        "host_cpus": os.cpu_count(),
        "cpu_quota_cpus": quota,
        "physical_cores": phys_cores,
        "quota_note": (None if not quota else
                       "the container's CFS quota is %.1f CPUs of the %d processors it sees: `value` is the all-cores figure for THAT "
                       "allowance (scaling_1_to_all against it), not for the %s-core host.  `value_if_linear_in_host_cores` = value x "
                       "physical cores / quota: an UPPER bound for the whole host (the store lookups are memory-bound and the host "
                       "has two NUMA nodes), given so that the speed-ups can be read against it as well"
                       % (quota, os.cpu_count() or 0, phys_cores or "?")),
        "value_if_linear_in_host_cores": (n / legs[best]["step"]) * (phys_cores / quota) if quota and phys_cores and phys_cores > quota else None,
        "cpu_model": cpu_model,
        "march": march,
        "checked_against_gpu": "every block's CID verdict, scan status + has-match map + (exec, event, emitter) match "
                               "list, and the status byte of every sampled claim",
    }


if __name__ == "__main__":
    main()
