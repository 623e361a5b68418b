out=gpurun_out/b3; mkdir -p $out
( timeout 400 python -m pytest tests/test_gpu_events.py tests/test_gpu_event_table.py tests/test_gpu_boundary.py tests/test_gpu_enum_shapes.py tests/test_gpu_sharding.py tests/test_gpu_store.py -x -q -m gpu ) > $out/tests.log 2>&1; tail -5 $out/tests.log
bash tools/gpu_ab.sh $out 2 base nofuse:IPCFP_FUSE_INDEX=0 eager:IPCFP_LAZY_INDEX=0
bash tools/gpu_prof.sh $out/fused > /dev/null 2>&1; cat $out/fused/timeline.txt
