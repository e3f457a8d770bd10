set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/final
timeout -k 10 300 python bench.py > $R/gpurun_out/final/bench_default.json 2> $R/gpurun_out/final/bench_default.err
tail -c 3000 $R/gpurun_out/final/bench_default.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/kt -o kt -- python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline > $R/gpurun_out/final/kt.log 2>&1
ls $R/gpurun_out/final/kt | head
