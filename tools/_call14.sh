out=gpurun_out/b13; mkdir -p $out
( timeout 300 python -m pytest tests/test_gpu_events.py tests/test_gpu_enum_shapes.py tests/test_gpu_event_table.py tests/test_gpu_sharding.py tests/test_gpu_boundary.py -x -q -m gpu ) > $out/tests.log 2>&1; grep -E "passed|failed" $out/tests.log | tail -1
bash tools/gpu_ab.sh $out 2 base
bash tools/gpu_prof.sh $out > /dev/null 2>&1; sed -n 8,12p; grep -n "rehash\|verify_events_table" $out/timeline.txt # $out/timeline.txt; tail -1 $out/timeline.txt
