#!/bin/bash
# tools/gpu_cpu_scaling.sh <outdir> — the CPU baseline's phases at several thread counts and OpenMP placements on the GPU box's
# host (tools/cpu_scaling.py), with the box's topology.  CPU work only.
out=${1:-gpurun_out/cpu}
mkdir -p "$out"
( lscpu | grep -E "Model name|Socket|Core|Thread|NUMA|L3|MHz" ; numactl -H 2>/dev/null | head -20; cat /sys/fs/cgroup/cpu.max 2>/dev/null ) > "$out/topology.txt" 2>&1
make -s -C oracle native > /dev/null 2>&1
T=${CPU_THREADS:-1,16,32,64,96,128,192,256}
( timeout 200 python tools/cpu_scaling.py 1000000 $T ) > "$out/scaling_default.txt" 2>&1
( OMP_PLACES=cores OMP_PROC_BIND=spread timeout 200 python tools/cpu_scaling.py 1000000 ${T#1,} ) > "$out/scaling_cores_spread.txt" 2>&1
( OMP_PLACES=cores OMP_PROC_BIND=close timeout 200 python tools/cpu_scaling.py 1000000 ${T#1,} ) > "$out/scaling_cores_close.txt" 2>&1
( OMP_PLACES=threads OMP_PROC_BIND=spread timeout 200 python tools/cpu_scaling.py 1000000 ${T#1,} ) > "$out/scaling_threads_spread.txt" 2>&1
cat "$out/topology.txt"; grep -h threads= "$out"/scaling_*.txt
