#!/bin/bash
# tools/gpu_pmc.sh <outdir> "<counters>" [env...] — one rocprofv3 --pmc pass of the default bench (no trace flags)
out=$1; shift
ctrs=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
tag=$(echo "$ctrs" | tr ' ' '_' | cut -c1-60)
( cd /tmp && env "$@" timeout 300 rocprofv3 --pmc $ctrs -d /tmp/pmc_$$ -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub-records --plain --t2-reps 0 ) > "$out/pmc_run_$tag.log" 2>&1
db=$(find /tmp/pmc_$$ -name '*.db' | head -1)
python tools/rocpd_summary.py "$db" --pmc > "$out/pmc_$tag.txt" 2>&1
rm -rf /tmp/pmc_$$
