#!/bin/bash
# tools/gpu_api_trace.sh <outdir> — the HOST side of one bench step: HIP runtime API calls (rocprofv3 --runtime-trace, no
# counters) of the last timed step next to its kernels (tools/rocpd_api_tail.py).
out=${1:-gpurun_out/api}
mkdir -p "$out"
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --runtime-trace --kernel-trace -d /tmp/api_$$ -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub-records --plain --t2-reps 0 --logical-shards "" ) > "$out/api_run.log" 2>&1
db=$(find /tmp/api_$$ -name '*.db' | head -1)
python tools/rocpd_api_tail.py "$db" > "$out/api_tail.txt" 2>&1
python tools/rocpd_summary.py "$db" --timeline > "$out/api_timeline.txt" 2>&1
rm -rf /tmp/api_$$
tail -60 "$out/api_tail.txt"
