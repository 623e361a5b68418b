out=gpurun_out/f1; mkdir -p $out
( timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records --t2-reps 0 ) > $out/bench_pre.log 2>&1; tail -1 $out/bench_pre.log > $out/bench_pre.json; grep -o '"ms_per_step": [0-9.]*' $out/bench_pre.json | head -1
bash tools/gpu_pmc.sh $out "FETCH_SIZE"
bash tools/gpu_pmc_workload.sh $out hamt
bash tools/gpu_pmc_workload.sh $out storage
ls $out
