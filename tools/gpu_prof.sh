#!/bin/bash
# tools/gpu_prof.sh <outdir> — kernel trace of the default bench on the GPU box: per-kernel stats + one-step timeline.
out=${1:-gpurun_out/prof}
mkdir -p "$out"
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$$ -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub-records --plain --t2-reps 0 ) > "$out/prof_run.log" 2>&1
db=$(find /tmp/prof_$$ -name '*.db' | head -1)
python tools/rocpd_summary.py "$db" > "$out/stats.txt" 2>&1
python tools/rocpd_summary.py "$db" --timeline > "$out/timeline.txt" 2>&1
grep -o '"ms_per_step": [0-9.]*' "$out/prof_run.log" | head -1
tail -3 "$out/timeline.txt"
