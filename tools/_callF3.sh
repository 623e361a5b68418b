out=gpurun_out/final2; mkdir -p $out
( timeout 170 python -m pytest tests -x -q -m gpu -p no:cacheprovider ) > $out/tests.log 2>&1
grep -E "passed|failed" $out/tests.log | tail -1
( timeout 70 python bench.py --steps 20 --warmup 5 ) > $out/bench.log 2>&1
grep '^{"metric"' $out/bench.log | tail -1 > $out/bench.json
grep -o '"ms_per_step": [0-9.]*' $out/bench.json | head -1
