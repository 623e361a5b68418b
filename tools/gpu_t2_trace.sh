#!/bin/bash
# tools/gpu_t2_trace.sh <outdir> — where a from-host pass (window T2) spends its time on the HOST side: phase timestamps of
# ipcfp_witness_create_packed (IPCFP_TRACE_CREATE=1) for the bench's T2 repetitions.
out=${1:-gpurun_out/t2}
mkdir -p "$out"
IPCFP_TRACE_CREATE=1 timeout 200 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-sub-records --plain --logical-shards "" --t2-reps 3 > "$out/bench.json" 2> "$out/create_trace.txt"
grep "^\[create\]" "$out/create_trace.txt" | tail -24
python - "$out/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
t = d["window_T2"]
print("T2", round(t["ms_per_tipset"], 3), t["ms_phases"], "plain", round(t["plain_forms"]["ms_per_tipset"], 3), t["plain_forms"]["ms_phases"])
PY
