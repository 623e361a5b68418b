#!/bin/bash
# tools/gpu_storage_ab.sh <outdir> <variant> [<variant> ...] — configs[4]'s call (bench.py --workload storage) on one box: every
# variant (NAME or NAME:ENV=VAL,ENV=VAL) twice, interleaved; then per variant a kernel trace of the same command: the
# per-kernel averages (stats_<name>.txt) and the last call's timeline (timeline_<name>.txt).
out=$1; shift
mkdir -p "$out"
[ $# -eq 0 ] && set -- default
for r in 1 2; do
  for v in "$@"; do
    name=${v%%:*}; envs=""
    [ "$v" != "$name" ] && envs=$(echo "${v#*:}" | tr ',' ' ')
    env $envs python bench.py --workload storage --steps 10 --warmup 3 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4))" >> "$out/ab.txt"
  done
done
here=$(pwd)
for v in "$@"; do
  name=${v%%:*}; envs=""
  [ "$v" != "$name" ] && envs=$(echo "${v#*:}" | tr ',' ' ')
  rm -rf /tmp/kt_storage
  ( cd /tmp && export TMPDIR=/tmp && env $envs rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_storage -- python "$here/bench.py" --workload storage --steps 5 --warmup 2 > /dev/null 2>&1 )
  f=$(find /tmp/kt_storage -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$out/stats_$name.txt" > "$out/timeline_$name.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
with open(sys.argv[2], "w") as f:
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        f.write(f"{len(v):5d} calls  avg {sum(v)/len(v):9.1f} us  min {min(v):9.1f}  {k[:110]}\n")
last = max(i for i, r in enumerate(rows) if "k_storage_run_flags" in r["Kernel_Name"])
prev = max((i for i, r in enumerate(rows[:last]) if "k_verify_storage" in r["Kernel_Name"]), default=-1)
t0 = int(rows[prev + 1]["Start_Timestamp"])
for r in rows[prev + 1:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1000:9.1f} {e/1000:9.1f} {(e-s)/1000:8.1f} us  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:100]}")
PY
done
cat "$out/ab.txt"
