#!/bin/bash
# tools/gpu_final.sh <outdir> — the measurement artefacts of a round, from ONE box: default bench line (with cpu_baseline
# and the T2 window), per-config bench lines, kernel trace (stats + one-step timeline), PMC passes (own runs, no trace).
out=${1:-gpurun_out/final}
mkdir -p "$out"
export PYTHONUNBUFFERED=1
( time timeout 400 python bench.py --steps 20 --warmup 5 ) > "$out/bench.log" 2>&1; tail -1 "$out/bench.log" | head -c 400; echo
grep "^{\"metric\"" "$out/bench.log" | tail -1 > "$out/bench.json" 2>/dev/null
for wl in cid hamt storage; do ( timeout 200 python bench.py --workload $wl --steps 10 --warmup 3 ) > "$out/bench_$wl.log" 2>&1; tail -1 "$out/bench_$wl.log" | head -c 300; echo; done
( timeout 200 python bench.py --steps 10 --warmup 3 --force-sharded ) > "$out/bench_sharded1.log" 2>&1; tail -1 "$out/bench_sharded1.log" | head -c 300; echo
( timeout 200 python bench.py --order S,K,V --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records --t2-reps 0 ) > "$out/bench_order_SKV.log" 2>&1; grep -o '"ms_per_step": [0-9.]*' "$out/bench_order_SKV.log"
bash tools/gpu_prof.sh "$out"
[ -n "$SKIP_FETCH" ] || bash tools/gpu_pmc.sh "$out" "FETCH_SIZE"
bash tools/gpu_pmc.sh "$out" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
bash tools/gpu_pmc.sh "$out" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_IFETCH GRBM_GUI_ACTIVE"
ls "$out"
