#!/bin/bash
# tools/gpu_r5_storage.sh <outdir> — round 5, configs[4] storage proofs: parity tests, the node table in both forms (one block
# per lane with line staging / eight lanes per block with the ring reader), kernel stats of each.
out=${1:-gpurun_out/r5_storage}
mkdir -p "$out"
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_walks.py tests/test_gpu_hamt_routes.py tests/test_gpu_pyhamt.py tests/test_gpu_baseline_sizes.py tests/test_gpu_range_shards.py -k "storage or hamt or cfg5 or cfg4 or mutated or routes" ) > "$out/tests.log" 2>&1; tail -2 "$out/tests.log"
for v in ${VARIANTS:-lane ring}; do
  e="IPCFP_HAMT_TABLE_FORM=$v"
  ( env $e timeout 200 python bench.py --workload storage --steps 10 --warmup 3 --no-cpu-baseline ) > "$out/bench_$v.log" 2>&1
  grep -o '"ms_per_step": [0-9.]*' "$out/bench_$v.log" | head -1
  ( cd /tmp && env $e timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload storage --steps 5 --warmup 2 --no-cpu-baseline ) > "$out/prof_$v.log" 2>&1
  db=$(find /tmp/prof_$v -name '*.db' | head -1)
  python tools/rocpd_summary.py "$db" > "$out/stats_$v.txt" 2>&1
  grep -E "hamt_node_table|verify_storage|storage_run|scan" "$out/stats_$v.txt" | cut -c1-70,110-170 | head -10
done
