#!/bin/bash
# tools/gpu_prof_workload.sh <outdir> <workload> — kernel trace of `bench.py --workload <cid|hamt|storage>`: stats + one-step timeline.
out=${1:-gpurun_out/prof}; wl=${2:-hamt}
mkdir -p "$out"
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/profw_$$ -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline ) > "$out/prof_${wl}_run.log" 2>&1
db=$(find /tmp/profw_$$ -name '*.db' | head -1)
python tools/rocpd_summary.py "$db" > "$out/${wl}_stats.txt" 2>&1
marker=k_index_insert; [ "$wl" = hamt ] && marker=k_hamt_lv_start; [ "$wl" = cid ] && marker=k_blake2b256_cid; python tools/rocpd_summary.py "$db" --timeline --step-marker $marker > "$out/${wl}_timeline.txt" 2>&1
grep -o '"ms_per_step": [0-9.]*' "$out/prof_${wl}_run.log" | head -1
tail -25 "$out/${wl}_timeline.txt"
rm -rf /tmp/profw_$$
