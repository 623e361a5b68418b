out=gpurun_out/b8; mkdir -p $out
( timeout 200 python -m pytest tests/test_gpu_events.py tests/test_gpu_enum_shapes.py tests/test_gpu_walks.py -x -q -m gpu ) > $out/tests.log 2>&1; tail -2 $out/tests.log
bash tools/gpu_ab.sh $out 2 base d3g:IPCFP_K1_DEFER=3,IPCFP_K1_GATE=1 d2g:IPCFP_K1_DEFER=2,IPCFP_K1_GATE=1
IPCFP_K1_DEFER=3 IPCFP_K1_GATE=1 bash tools/gpu_prof.sh $out/d3g > /dev/null 2>&1; cat $out/d3g/timeline.txt
