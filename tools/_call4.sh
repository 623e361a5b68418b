out=gpurun_out/b1; mkdir -p $out
bash tools/gpu_ab.sh $out 2 base cas0:IPCFP_INDEX_CAS_FIRST=0 lv0:IPCFP_LEAVES_AUX=0 r4:IPCFP_RESERVE_CUS=4 r8:IPCFP_RESERVE_CUS=8 k1be:IPCFP_K1_AFTER_BE=1 r4k:IPCFP_RESERVE_CUS=4,IPCFP_K1_AFTER_BE=1 r8k:IPCFP_RESERVE_CUS=8,IPCFP_K1_AFTER_BE=1
IPCFP_RESERVE_CUS=4 IPCFP_K1_AFTER_BE=1 bash tools/gpu_prof.sh $out/r4k > /dev/null 2>&1; cat $out/r4k/timeline.txt
IPCFP_K1_AFTER_BE=1 bash tools/gpu_prof.sh $out/k1be > /dev/null 2>&1; cat $out/k1be/timeline.txt
