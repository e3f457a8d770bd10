#!/bin/bash
# one GPU round trip: parity tests, default bench (no CPU leg), kernel trace of a short bench run.
#   bash tools/gpu_cycle.sh <tag> [pytest -k expression]
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=$R/gpurun_out/$1; mkdir -p $T
K=${2:-}
cd $R
if [ -n "$K" ]; then timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "$K" > $T/pytest.log 2>&1; else timeout -k 10 800 python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1; fi
RC=$?
tail -4 $T/pytest.log
[ $RC -ne 0 ] && exit $RC
timeout -k 10 300 python bench.py --no-cpu-baseline > $T/bench.json 2> $T/bench.err || { tail -5 $T/bench.err; exit 1; }
python - <<PY
import json;d=json.load(open("$T/bench.json"));print("imgs/s",d["value"],"ms",d["ms_per_step"],"conv frac",d["roofline"]["frac"],"avg conv ms",d["roofline"]["avg_launch_ms"],"hbm frac",d["roofline_hbm_stages"]["frac"],d["stage_ms_per_step"])
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $T/kt -o kt -- python3 $R/bench.py --steps 6 --warmup 6 --no-cpu-baseline > $T/kt.log 2>&1 || { tail -5 $T/kt.log; exit 1; }
