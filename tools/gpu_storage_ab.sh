#!/bin/bash
# tools/gpu_storage_ab.sh <outdir> [<IPCFP_STORAGE_RUN_CHILDREN value> ...] — configs[4]'s call (bench.py --workload storage) twice per
# value on one box, then a kernel trace of the same command (the per-kernel averages and one call's timeline).
out=$1; shift
mkdir -p "$out"
vals=${@:-1}
for r in 1 2; do
  for v in $vals; do
    IPCFP_STORAGE_RUN_CHILDREN=$v python bench.py --workload storage --steps 10 --warmup 3 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('children=$v', round(d['ms_per_step'],4))" >> "$out/ab.txt"
  done
done
here=$(pwd)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_storage -- python "$here/bench.py" --workload storage --steps 5 --warmup 2 > /dev/null 2>&1 )
f=$(find /tmp/kt_storage -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python - "$f" > "$out/timeline.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: from the last k_storage_run_flags on
last = max(i for i, r in enumerate(rows) if "k_storage_run_flags" in r["Kernel_Name"])
# the call may start a few launches earlier (table on the aux stream): back up to the previous k_verify_storage_table's end
prev = max((i for i, r in enumerate(rows[:last]) if "k_verify_storage" in r["Kernel_Name"]), default=-1)
t0 = int(rows[prev + 1]["Start_Timestamp"])
for r in rows[prev + 1:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1000:9.1f} {e/1000:9.1f} {(e-s)/1000:8.1f} us  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:100]}")
PY
cat "$out/ab.txt"
