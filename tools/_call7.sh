out=gpurun_out/b4; mkdir -p $out
bash tools/gpu_ab.sh $out 2 base nohead:IPCFP_HEAD_STREAM=0
bash tools/gpu_prof.sh $out/head > /dev/null 2>&1; cat $out/head/timeline.txt
( time timeout 500 python -m pytest tests -x -q -m gpu ) > $out/tests.log 2>&1; tail -6 $out/tests.log
