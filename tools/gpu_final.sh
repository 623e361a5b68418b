#!/bin/bash
# tools/gpu_final.sh <outdir> [round] — the measurement artefacts of a round, from ONE box:
#   1. a pre-pass: the FETCH_SIZE counter calibrated (tools/ubench/fetch_calib) and one FETCH_SIZE pass of the bench →
#      <outdir>/traffic.json (tools/pmc_traffic.py; copy it to profiles/rNN_traffic.json — bench.py reads the newest such
#      file for `roofline.traffic`; with TRAFFIC_INTO_PROFILES=1 the script places it there itself BEFORE step 2, so the
#      bench line of this very run carries this very run's traffic)
#   2. the default bench line (cpu_baseline, T2 window, sub-records), per-config lines, the receipt-cut path on one GPU,
#      the other call order (the S,K,V figure is also inside the default line)
#   3. kernel trace (stats + one-step timeline), two SQ PMC passes (own runs, no trace flags)
#   1b. configs[1] / [3] / [4]: kernel stats + PMC of their own kernels (tools/gpu_pmc_workload.sh) → traffic_workloads.json
# Every stage is bounded by `timeout`.  ≈ 6 minutes of box time.
out=${1:-gpurun_out/final}
rnd=${2:-0}
mkdir -p "$out"
export PYTHONUNBUFFERED=1
json_line() { grep '^{"metric"' "$1" | tail -1; }
# ---- 1 ----
if [ -z "$SKIP_FETCH" ]; then
  [ -x tools/ubench/fetch_calib ] && bash tools/gpu_calib.sh "$out/calib" > /dev/null 2>&1
  ( timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sub-records --t2-reps 0 --plain ) > "$out/bench_pre.log" 2>&1
  json_line "$out/bench_pre.log" > "$out/bench_pre.json"
  bash tools/gpu_pmc.sh "$out" "FETCH_SIZE"
  if [ -s "$out/calib/fetch_calib_pmc.txt" ] && [ -s "$out/pmc_FETCH_SIZE.txt" ]; then
    python tools/pmc_traffic.py "$out/pmc_FETCH_SIZE.txt" "$out/calib/fetch_calib_pmc.txt" "$out/calib/fetch_calib_stdout.txt" "$out/bench_pre.json" "$rnd" > "$out/traffic.json" 2> "$out/traffic.err"
    if [ -n "$TRAFFIC_INTO_PROFILES" ] && [ -s "$out/traffic.json" ] && [ "$rnd" != 0 ]; then cp "$out/traffic.json" "profiles/r$(printf %02d "$rnd")_traffic.json"; fi
  fi
fi
# ---- 1b: the per-config benches' FETCH_SIZE passes → profiles/rNN_traffic_workloads.json (the sub-records' roofline.traffic) ----
if [ -z "$SKIP_WORKLOAD_PMC" ]; then
  for wl in cid hamt storage; do bash tools/gpu_pmc_workload.sh "$out" $wl > /dev/null 2>&1; done
  python tools/pmc_traffic_workload.py "$rnd" cid="$out/cid_pmc1.txt" hamt="$out/hamt_pmc1.txt" storage="$out/storage_pmc1.txt" > "$out/traffic_workloads.json" 2> "$out/traffic_workloads.err"
  if [ -n "$TRAFFIC_INTO_PROFILES" ] && [ -s "$out/traffic_workloads.json" ] && [ "$rnd" != 0 ]; then cp "$out/traffic_workloads.json" "profiles/r$(printf %02d "$rnd")_traffic_workloads.json"; fi
fi
# ---- 2 ----
# (the line on stdout is the driver's compact record; the whole report is bench_detail.json, rewritten by every invocation)
( time timeout 400 python bench.py --steps 20 --warmup 5 ) > "$out/bench.log" 2>&1; json_line "$out/bench.log" > "$out/bench.json"; head -c 400 "$out/bench.json"; echo
cp bench_detail.json "$out/bench_detail.json" 2>/dev/null
for wl in cid hamt storage; do ( timeout 200 python bench.py --workload $wl --steps 10 --warmup 3 ) > "$out/bench_$wl.log" 2>&1; json_line "$out/bench_$wl.log" > "$out/bench_$wl.json"; cp bench_detail.json "$out/bench_detail_$wl.json" 2>/dev/null; head -c 300 "$out/bench_$wl.json"; echo; done
( timeout 200 python bench.py --steps 10 --warmup 3 --force-sharded ) > "$out/bench_sharded1.log" 2>&1; json_line "$out/bench_sharded1.log" > "$out/bench_sharded1.json"; head -c 300 "$out/bench_sharded1.json"; echo
# ---- 3 ----
bash tools/gpu_prof.sh "$out"
bash tools/gpu_pmc.sh "$out" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
bash tools/gpu_pmc.sh "$out" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_IFETCH GRBM_GUI_ACTIVE"
# the L2's own view of the step: hits and misses per kernel, a pass of its own
bash tools/gpu_pmc.sh "$out" "TCC_HIT_sum TCC_MISS_sum"
# (4: configs[3] / [4] kernel stats + PMC are stage 1b's files)
ls "$out"
