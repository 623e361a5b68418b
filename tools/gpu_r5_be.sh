#!/bin/bash
# tools/gpu_r5_be.sh <outdir> — round 5, k_block_events: parity subset, in-step A/B (ring vs line-staged reader, both with the
# scalar-slot event decode), every kernel ALONE (one stream: IPCFP_K1_STREAM=0 IPCFP_AUX_STREAM=0) under the kernel trace,
# SQ counters of the ring form.
out=${1:-gpurun_out/r5_be}
mkdir -p "$out"
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_events.py tests/test_gpu_event_table.py tests/test_golden.py tests/test_gpu_fuzz.py tests/test_gpu_generate.py ) > "$out/tests_ring.log" 2>&1; tail -2 "$out/tests_ring.log"
( IPCFP_BLOCK_EVENTS=l timeout 300 python -m pytest -m gpu -x -q tests/test_gpu_event_table.py tests/test_gpu_events.py ) > "$out/tests_line.log" 2>&1; tail -2 "$out/tests_line.log"
AB_STEPS=20 bash tools/gpu_ab.sh "$out/ab" 2 ring line:IPCFP_BLOCK_EVENTS=l
for v in ring line; do
  e=""; [ $v = line ] && e="IPCFP_BLOCK_EVENTS=l"
  ( cd /tmp && env $e IPCFP_K1_STREAM=0 IPCFP_AUX_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub-records --plain --t2-reps 0 ) > "$out/serial_$v.log" 2>&1
  db=$(find /tmp/prof_$v -name '*.db' | head -1)
  python tools/rocpd_summary.py "$db" > "$out/serial_stats_$v.txt" 2>&1
  grep -E "block_events|blake2b256_cid|verify_events_table|receipt_events|dense_leaves" "$out/serial_stats_$v.txt" | head -8
done
bash tools/gpu_pmc.sh "$out" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
grep -E "block_events" "$out"/pmc_SQ*.txt
bash tools/gpu_pmc.sh "$out" "FETCH_SIZE"
grep -E "block_events" "$out"/pmc_FETCH*.txt
