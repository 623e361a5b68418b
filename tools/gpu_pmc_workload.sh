#!/bin/bash
# tools/gpu_pmc_workload.sh <outdir> <workload: hamt|storage|cid> — kernel stats + PMC passes of one per-config bench
# (each rocprofv3 run is its own pass: --kernel-trace --stats alone, then --pmc alone)
out=$1; wl=$2
mkdir -p "$out"
export TMPDIR=/tmp
cmd="python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline"
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/wk_$$ -o k -- $cmd ) > "$out/${wl}_trace_run.log" 2>&1
db=$(find /tmp/wk_$$ -name '*.db' | head -1); python tools/rocpd_summary.py "$db" > "$out/${wl}_kernel_stats.txt" 2>&1
[ -n "$STEP_MARKER" ] && python tools/rocpd_summary.py "$db" --timeline --step-marker "$STEP_MARKER" > "$out/${wl}_timeline.txt" 2>&1
rm -rf /tmp/wk_$$
i=0
for ctrs in "FETCH_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --pmc $ctrs -d /tmp/wp_$$ -o p -- $cmd ) > "$out/${wl}_pmc${i}_run.log" 2>&1
  db=$(find /tmp/wp_$$ -name '*.db' | head -1); python tools/rocpd_summary.py "$db" --pmc > "$out/${wl}_pmc${i}.txt" 2>&1
  [ -n "$PMC_SEQ" ] && python tools/rocpd_summary.py "$db" --pmc --pmc-seq "$PMC_SEQ" > "$out/${wl}_pmc${i}_seq.txt" 2>&1
  rm -rf /tmp/wp_$$
done
head -8 "$out/${wl}_kernel_stats.txt"
