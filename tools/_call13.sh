out=gpurun_out/b10; mkdir -p $out
( timeout 300 python -m pytest tests/test_gpu_events.py tests/test_gpu_enum_shapes.py tests/test_gpu_event_table.py tests/test_gpu_sharding.py -x -q -m gpu ) > $out/tests.log 2>&1; tail -2 $out/tests.log
bash tools/gpu_ab.sh $out 2 base
bash tools/gpu_prof.sh $out > /dev/null 2>&1; sed -n 14,34p $out/timeline.txt; tail -1 $out/timeline.txt
