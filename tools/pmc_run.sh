#!/bin/bash
# rocprofv3 PMC pass: counters + --kernel-trace only (the kernel trace names the dispatches; the pool refuses --pmc
# together with sys/runtime/hip/hsa/memory-copy/scratch/marker tracing, none of which is used); the program follows
# `--` directly (python3, no env/bash hop).  Run on the GPU box from the repo root:
#   bash tools/pmc_run.sh <outdir-under-gpurun_out> "<counter list>" <python script + args>
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
CTR="$1"; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT -o pmc -- python3 "$@" > $OUT/run.log 2>&1 || (tail -5 $OUT/run.log; exit 1)
ls $OUT
