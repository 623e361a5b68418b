out=gpurun_out/b9; mkdir -p $out
( timeout 300 python -m pytest tests/test_gpu_pyhamt.py tests/test_gpu_walks.py "tests/test_gpu_baseline_sizes.py::test_cfg4_actor_gets_every_status_and_value" -x -q -m gpu ) > $out/tests.log 2>&1; tail -2 $out/tests.log
for L in 64 32 16 8 0; do
  ( IPCFP_HAMT_LANES=$L timeout 120 python bench.py --workload hamt --steps 10 --warmup 3 --no-cpu-baseline ) > $out/hamt_$L.log 2>&1
  echo "lanes $L $(grep -o '"ms_per_step": [0-9.]*' $out/hamt_$L.log | head -1)"
done
