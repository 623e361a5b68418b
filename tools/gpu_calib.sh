#!/bin/bash
# tools/gpu_calib.sh <outdir> — FETCH_SIZE of the calibration kernels (tools/ubench/fetch_calib) in its own rocprofv3 pass
out=${1:-gpurun_out/calib}
mkdir -p "$out"
export TMPDIR=/tmp
bin=$GRAFT_REPO_ROOT/tools/ubench/fetch_calib
( cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/calib_$$ -o calib -- $bin ) > "$out/fetch_calib_stdout.txt" 2>&1
db=$(find /tmp/calib_$$ -name '*.db' | head -1)
python tools/rocpd_summary.py "$db" --pmc > "$out/fetch_calib_pmc.txt" 2>&1
rm -rf /tmp/calib_$$
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/calibk_$$ -o calib -- $bin ) > /dev/null 2>&1
db=$(find /tmp/calibk_$$ -name '*.db' | head -1)
python tools/rocpd_summary.py "$db" > "$out/fetch_calib_kernels.txt" 2>&1
rm -rf /tmp/calibk_$$
grep -h "k_\|bytes\|lines" "$out/fetch_calib_stdout.txt" "$out/fetch_calib_pmc.txt" | head -20
