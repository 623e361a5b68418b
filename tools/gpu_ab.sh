#!/bin/bash
# tools/gpu_ab.sh <outdir> <rounds> <variant> [<variant> ...] — A/B runs of the default bench step on ONE box.
# A variant is NAME or NAME:ENV=VAL,ENV=VAL ; every round runs every variant once (interleaved, so drift hits all alike).
out=$1; rounds=$2; shift 2
mkdir -p "$out"
export PYTHONUNBUFFERED=1
for r in $(seq 1 "$rounds"); do
  for v in "$@"; do
    name=${v%%:*}; envs=""
    [ "$v" != "$name" ] && envs=$(echo "${v#*:}" | tr ',' ' ')
    ( env $envs timeout 120 python bench.py --steps ${AB_STEPS:-30} --warmup 5 --plain --no-cpu-baseline --no-sub-records --t2-reps 0 ) > "$out/ab_${name}_$r.log" 2>&1
    ms=$(grep -o '"ms_per_step": *[0-9.]*' "$out/ab_${name}_$r.log" | head -1)
    echo "AB round $r $name [$envs] $ms $(grep -c Traceback "$out/ab_${name}_$r.log") errors"
  done
done
