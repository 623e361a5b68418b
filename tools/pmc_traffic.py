#!/usr/bin/env python3
"""tools/pmc_traffic.py <pmc_FETCH_SIZE.txt> <fetch_calib_pmc.txt> <fetch_calib_stdout.txt> <bench.json> <round> > profiles/rNN_traffic.json

HBM traffic of every kernel of the tipset step from ONE `rocprofv3 --pmc FETCH_SIZE` pass of bench.py (tools/gpu_pmc.sh),
each kernel's counter scaled by the factor measured for ITS access pattern with tools/ubench/fetch_calib under the same
counter (guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE under-reports on gfx950 and the factor is only known for wide
streaming reads — "calibrate on a known byte count in your own access pattern").  bench.py reads the newest such file
for `roofline.traffic` and `kernels.groups[].traffic_bytes`."""
import json
import re
import sys

# HIP kernel (substring of the demangled name) -> (profile group of include/ipcfp.h, access pattern of fetch_calib)
KERNELS = [
    ("k_blake2b256_cid", "blake2b_cid", "lane_seq"),        # one block per lane, 16 B at a time
    ("k_index_insert", "cid_index", "rand16"),              # coalesced CID words + one random 8-byte slot per lane
    ("k_tipset_prepare", "tipset_prologue", "lane_seq"),
    ("k_enum_roots", "tipset_prologue", "lane_seq"),
    ("k_dense_top", "amt_walk", "lane_seq"),                # the narrow levels at the top, one workgroup
    ("k_dense_level", "amt_walk", "lane_seq"),              # one AMT node per lane, parsed front to back
    ("k_dense_link_leaves", "amt_walk", "lane_seq"),
    ("k_dense_leaves", "amt_walk", "lane_seq"),
    ("k_exec_insert", "exec_order", "rand16"),
    ("k_exec_first_sums", "exec_order", "rand16"),
    ("k_exec_apply_finish", "exec_order", "rand16"),
    ("k_block_events", "event_scan", "stream"),             # whole lines staged by the wavefront
    ("k_receipt_events", "event_scan", "rand64"),
    ("k_count_from_table", "event_scan", "stream"),
    ("k_scan_pass2", "event_scan", "stream"),
    ("k_verify_events_table", "event_verify", "rand64"),
    ("k_verify_events", "replay", "lane_seq"),
]
CALIB = {"k_stream": "stream", "k_lane_seq": "lane_seq", "k_rand<1>": "rand16", "k_rand<4>": "rand64", "k_rand<8>": "rand128"}


def rows(path, counter="FETCH_SIZE"):
    out = {}
    for line in open(path):
        if line.startswith("#"):
            continue
        f = [x.strip() for x in line.split("|")]
        if len(f) == 5 and f[1] == counter:
            out[f[0]] = (int(f[2]), float(f[3]), float(f[4]))
    return out


def main():
    pmc_path, calib_pmc, calib_out, bench_json, rnd = sys.argv[1:6]
    # ---- factors: bytes of distinct 128-byte lines touched / (FETCH_SIZE KB x 1024) ----
    touched = {}
    for line in open(calib_out):
        if line.startswith("lines_touched_x128"):
            for name, val in re.findall(r"(k_[a-z_]+(?:<\d>)?) (\d+)", line):
                touched[name] = int(val)
    factors, calib_rows = {}, {}
    for kname, (n, total, mean) in rows(calib_pmc).items():
        for cname, pattern in CALIB.items():
            if cname in kname and cname in touched:
                factors[pattern] = touched[cname] / (mean * 1024.0)
                calib_rows[pattern] = {"kernel": kname, "fetch_size_kb_per_launch": mean, "bytes_of_lines_touched": touched[cname],
                                       "factor": round(factors[pattern], 4)}
    # k_lane_seq (one parser per lane, 16 bytes at a time, EVERY wave slot of the chip occupied) is not a counter
    # calibration but a measurement of re-fetching: 64 lanes x 32 waves x 256 CUs x one 128-byte line each = 67 MB of lines
    # in flight against 32 MB of L2, so a line is evicted between two of its eight 16-byte reads.  Its row is kept as
    # `refetch_ratio`; kernels of that pattern take the streaming factor (K1, the same pattern at 3 waves/SIMD, was
    # calibrated at exactly 2.0 with tools/calib_fetch.py in round 1)
    if "lane_seq" in factors and "stream" in factors:
        calib_rows["lane_seq"]["refetch_ratio"] = round(factors["stream"] / factors["lane_seq"], 3)
        calib_rows["lane_seq"]["factor"] = round(factors["stream"], 4)
        calib_rows["lane_seq"]["factor_basis"] = "k_stream's; FETCH_SIZE x factor / bytes touched = refetch_ratio"
        factors["lane_seq"] = factors["stream"]
    # a lane that reads 16 or 64 bytes of a line does not necessarily move the whole line: the TRUE byte count of those two
    # launches is not known a priori, so they take the factor of the whole-line random read and what they imply per access
    # is recorded (the fetch granularity of a partial-line read)
    if "rand128" in factors:
        for pattern in ("rand16", "rand64"):
            if pattern in calib_rows:
                n_lines = calib_rows[pattern]["bytes_of_lines_touched"] // 128
                calib_rows[pattern]["implied_bytes_per_access"] = round(
                    calib_rows[pattern]["fetch_size_kb_per_launch"] * 1024.0 * factors["rand128"] / n_lines, 1)
                calib_rows[pattern]["factor"] = round(factors["rand128"], 4)
                calib_rows[pattern]["factor_basis"] = "rand128's (whole lines, known byte count)"
            factors[pattern] = factors["rand128"]
    bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
    pmc = rows(pmc_path)
    n_steps = None
    for kname, (n, total, mean) in pmc.items():
        if "k_tipset_prepare" in kname:
            n_steps = n
    if not n_steps:
        raise SystemExit("no k_tipset_prepare row: cannot tell how many steps the PMC pass ran")
    groups, kernels = {}, {}
    for kname, (n, total, mean) in sorted(pmc.items()):
        for sub, group, pattern in KERNELS:
            if sub in kname and not (sub == "k_verify_events" and "table" in kname):
                fac = factors.get(pattern)
                if fac is None:
                    continue
                # K1 also runs outside the steps (self-check, the `alone` launches): per launch, one launch per step
                per_step = mean * 1024.0 * fac * (1 if group == "blake2b_cid" else n / n_steps)
                kernels[kname] = {"group": group, "pattern": pattern, "factor": round(fac, 4), "dispatches": n,
                                  "fetch_size_kb_per_dispatch": mean, "traffic_bytes_per_step": per_step}
                g = groups.setdefault(group, {"traffic_bytes_per_step": 0.0, "kernels": [], "factor": round(fac, 4)})
                g["traffic_bytes_per_step"] += per_step
                g["kernels"].append(kname)
                break
    out = {
        "round": int(rnd),
        "command": "rocprofv3 --pmc FETCH_SIZE -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub-records --t2-reps 0  (own pass, no trace flags; tools/gpu_pmc.sh)",
        "workload": {"receipts": bench["config"]["receipts_per_gpu"], "witness_blocks": bench["config"]["witness_blocks_per_gpu"],
                     "payload_bytes": bench["config"]["witness_bytes_per_gpu"]},
        "steps_in_pass": n_steps,
        "calibration": {"tool": "rocprofv3 --pmc FETCH_SIZE -- tools/ubench/fetch_calib (2 GiB buffer, every 128-byte line touched exactly once per launch)",
                        "patterns": calib_rows},
        "groups": groups,
        "kernels": kernels,
    }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
