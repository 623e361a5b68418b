out=gpurun_out/b7; mkdir -p $out
( timeout 400 python -m pytest tests/test_gpu_events.py tests/test_gpu_event_table.py tests/test_gpu_boundary.py tests/test_gpu_enum_shapes.py tests/test_gpu_fuzz.py tests/test_gpu_generate.py tests/test_gpu_bundle.py -x -q -m gpu ) > $out/tests.log 2>&1; tail -3 $out/tests.log
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records ) > $out/bench.log 2>&1
python - <<'P'
import json
d=json.loads(open('gpurun_out/b7/bench.log').read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "by_order", d.get("ms_per_step_by_order"))
print("T2", json.dumps(d.get("window_T2")))
print("kernels", d.get("kernels_ms_per_step"))
P
bash tools/gpu_prof.sh $out > /dev/null 2>&1; cat $out/timeline.txt; grep -E "block_events|blake2b|verify_events_table|receipt_events" $out/stats.txt
