out=gpurun_out/final; mkdir -p $out
( time timeout 500 python -m pytest tests -x -q -m gpu ) > $out/tests.log 2>&1; grep -E "passed|failed" $out/tests.log | tail -1
bash tools/gpu_final.sh $out
