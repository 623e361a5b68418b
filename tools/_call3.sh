out=gpurun_out/m3; mkdir -p $out
( timeout 600 python -m pytest tests/test_gpu_events.py tests/test_gpu_event_table.py tests/test_gpu_boundary.py tests/test_gpu_fuzz.py tests/test_gpu_enum_shapes.py tests/test_gpu_sharding.py tests/test_gpu_bundle.py -x -q -m gpu ) > $out/tests.log 2>&1; tail -5 $out/tests.log
( timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records --t2-reps 0 ) > $out/bench.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $out/bench.log | head -1; grep -o '"kernels_ms_per_step": {[^}]*}' $out/bench.log; grep -o '"ms_per_step_by_order": {[^}]*}' $out/bench.log
bash tools/gpu_prof.sh $out
cat $out/timeline.txt
