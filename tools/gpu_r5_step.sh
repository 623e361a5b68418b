#!/bin/bash
# tools/gpu_r5_step.sh <outdir> [notests] — round 5 working pass: the GPU suite, the plain step, its kernel trace (stats + one-step
# timeline), SQ counters and FETCH_SIZE of the step's kernels.
out=${1:-gpurun_out/r5_step}
mkdir -p "$out"
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
if [ "$2" != notests ]; then
  ( timeout 900 python -m pytest -m gpu -x -q tests ) > "$out/tests.log" 2>&1; tail -3 "$out/tests.log"
fi
( timeout 200 python bench.py --steps 30 --warmup 5 --plain --no-cpu-baseline --no-sub-records --t2-reps 0 ) > "$out/bench_plain.log" 2>&1
grep -o '"ms_per_step": [0-9.]*' "$out/bench_plain.log" | head -1
bash tools/gpu_prof.sh "$out"
bash tools/gpu_pmc.sh "$out" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
grep -E "block_events" "$out"/pmc_SQ*.txt
bash tools/gpu_pmc.sh "$out" "FETCH_SIZE"
grep -E "block_events|receipt_events|verify_events_table" "$out"/pmc_FETCH*.txt
