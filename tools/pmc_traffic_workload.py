#!/usr/bin/env python3
"""tools/pmc_traffic_workload.py <round> <workload>=<pmc1.txt> ... > profiles/rNN_traffic_workloads.json

L2→L1 traffic per step of the per-config benches (bench.py --workload cid|hamt|storage) from the FETCH_SIZE pass of
tools/gpu_pmc_workload.sh: every kernel's counter (KB per dispatch) x dispatches x 2.00 (the factor tools/ubench/fetch_calib
measures on gfx950 for every access pattern: the counter is half of the 128-byte lines that cross L2→L1,
profiles/r0N_fetch_calib_*.txt), divided by the steps of the pass (the dispatch count of a once-per-step kernel).  bench.py
reads the newest such file for the sub-records' `roofline.traffic`."""
import json
import sys

STEP_MARKER = {"cid": "k_blake2b256_cid", "hamt": "k_hamt_lv_start", "storage": "k_storage_run_flags"}
FACTOR = 2.0


def rows(path):
    out = {}
    for line in open(path):
        if line.startswith("#"):
            continue
        f = [x.strip() for x in line.split("|")]
        if len(f) == 5 and f[1] == "FETCH_SIZE":
            out[f[0]] = (int(f[2]), float(f[3]))
    return out


def main():
    rnd = int(sys.argv[1])
    out = {"round": rnd, "factor": FACTOR, "command": "rocprofv3 --pmc FETCH_SIZE -- python bench.py --workload W --steps 5 --warmup 2 --no-cpu-baseline "
           "(own pass, no trace flags; tools/gpu_pmc_workload.sh)", "workloads": {}}
    for arg in sys.argv[2:]:
        wl, path = arg.split("=", 1)
        r = rows(path)
        steps = next((n for k, (n, _) in r.items() if STEP_MARKER[wl] in k), None)
        if not steps:
            continue
        kern = {k: {"dispatches": n, "traffic_bytes_per_step": total * 1024.0 * FACTOR / steps} for k, (n, total) in r.items()
                if "ipcfp::" in k and total > 0}
        out["workloads"][wl] = {"steps_in_pass": steps, "traffic_bytes_per_step": sum(v["traffic_bytes_per_step"] for v in kern.values()),
                                "kernels": kern}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
