#!/bin/bash
# kernel-only durations of one script (rocprofv3 --kernel-trace --stats): bash tools/kt.sh <tag> <script.py> [args]
# prints "average ns, calls, kernel" sorted by total time; the CSVs land in gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
cd $R
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kt_kernel_stats.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:${KT_TOP:-25}]:
    print(f'{float(r["AverageNs"])/1e3:9.1f} us x{int(r["Calls"]):4d}  {r["Name"][:110]}')
PY
