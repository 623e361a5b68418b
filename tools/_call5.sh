out=gpurun_out/b2; mkdir -p $out
bash tools/gpu_ab.sh $out 2 base d1:IPCFP_K1_DEFER=1 d1g:IPCFP_K1_DEFER=1,IPCFP_K1_GATE=1 d2g:IPCFP_K1_DEFER=2,IPCFP_K1_GATE=1 d3g:IPCFP_K1_DEFER=3,IPCFP_K1_GATE=1
IPCFP_K1_DEFER=3 IPCFP_K1_GATE=1 bash tools/gpu_prof.sh $out/d3g > /dev/null 2>&1; cat $out/d3g/timeline.txt
IPCFP_K1_DEFER=2 IPCFP_K1_GATE=1 bash tools/gpu_prof.sh $out/d2g > /dev/null 2>&1; cat $out/d2g/timeline.txt
