out=gpurun_out/b6; mkdir -p $out
timeout 120 tools/ubench/h2d_paths > $out/h2d_paths.txt 2>&1; cat $out/h2d_paths.txt
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records ) > $out/bench.log 2>&1
python - <<'P'
import json
d=json.loads(open('gpurun_out/b6/bench.log').read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "by_order", d.get("ms_per_step_by_order"))
print("T2", json.dumps(d.get("window_T2")))
print("kernels", d.get("kernels_ms_per_step"))
P
tail -3 $out/bench.log | head -c 600
