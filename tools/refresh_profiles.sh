# Round artefacts for profiles/: default bench line, rocprofv3 kernel stats of the same command, PMC HBM traffic passes.
#   bash tools/refresh_profiles.sh <round tag, e.g. r02>       (on the GPU box, from the repo root)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd $R
timeout -k 10 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 1500 $OUT/bench_default.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline > $OUT/kt.log 2>&1
ls $OUT/kt | head
cd $R
# PMC: counters + kernel trace only, one counter per pass, python3 directly after `--` (tools/pmc_run.sh)
timeout -k 10 400 bash tools/pmc_run.sh final/pmc_fetch FETCH_SIZE $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline
timeout -k 10 400 bash tools/pmc_run.sh final/pmc_write WRITE_SIZE $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline
python tools/pmc_traffic.py $OUT/pmc_fetch/pmc_counter_collection.csv $OUT/pmc_write/pmc_counter_collection.csv $OUT/${TAG}_pmc_hbm_traffic.json "bench.py --steps 6 --warmup 4 --no-cpu-baseline"
