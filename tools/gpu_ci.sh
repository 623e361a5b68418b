#!/bin/bash
# tools/gpu_ci.sh <outdir> — the GPU-box routine of a development round: parity tests with per-test timestamps,
# then the bench lines.  Every stage is bounded by `timeout` so a hang cannot eat the box.
out=${1:-gpurun_out/ci}
mkdir -p "$out"
stamp() { while IFS= read -r l; do printf '%s %s\n' "$(date +%s)" "$l"; done; }
export PYTHONUNBUFFERED=1
( time timeout ${T_NEW:-420} python -m pytest tests/test_gpu_sharding.py tests/test_gpu_baseline_sizes.py -m gpu -x -v --timeout=300 -p no:cacheprovider 2>&1 | stamp ) > "$out/pytest_new.log" 2>&1
tail -5 "$out/pytest_new.log"
if [ -z "$SKIP_OLD" ]; then
( time timeout ${T_OLD:-420} python -m pytest tests -m gpu -x -v --timeout=200 -p no:cacheprovider --deselect tests/test_gpu_sharding.py --deselect tests/test_gpu_baseline_sizes.py 2>&1 | stamp ) > "$out/pytest_old.log" 2>&1
tail -5 "$out/pytest_old.log"
fi
( time timeout 300 python bench.py --steps 20 --warmup 5 ) > "$out/bench.log" 2>&1; tail -c 3000 "$out/bench.log"
( time timeout 200 python bench.py --steps 10 --warmup 3 --force-sharded ) > "$out/bench_sharded1.log" 2>&1; tail -c 1500 "$out/bench_sharded1.log"
( time timeout 120 python bench.py --workload cid --steps 20 --warmup 3 ) > "$out/bench_cid.log" 2>&1; tail -c 1200 "$out/bench_cid.log"
