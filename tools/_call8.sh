out=gpurun_out/b5; mkdir -p $out
timeout 120 tools/ubench/h2d_paths > $out/h2d_paths.txt 2>&1; cat $out/h2d_paths.txt
bash tools/gpu_ab.sh $out 2 base nohead:IPCFP_HEAD_STREAM=0
bash tools/gpu_prof.sh $out/head > /dev/null 2>&1; head -14 $out/head/timeline.txt
bash tools/gpu_calib.sh $out/calib
