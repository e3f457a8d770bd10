# Round artefacts for profiles/: default bench line, rocprofv3 kernel stats of the same command, PMC passes (HBM traffic,
# matrix-pipe busy, RoIAlign instruction mix), the fp16 line.
#   bash tools/refresh_profiles.sh <round tag, e.g. r03>       (on the GPU box, from the repo root; then copy
#   gpurun_out/final/<tag>_* into profiles/ in the build container)
# Every profiler run is "counters + kernel trace only", python3 directly after `--` (tools/pmc_run.sh).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd $R
BENCH="$R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-fp16-leg"
# 1. PMC passes first: the bench line below quotes them (same kernel sources)
timeout -k 10 300 bash tools/pmc_run.sh final/pmc_fetch FETCH_SIZE $BENCH
timeout -k 10 300 bash tools/pmc_run.sh final/pmc_write WRITE_SIZE $BENCH
timeout -k 10 300 bash tools/pmc_run.sh final/pmc_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_INSTS_VALU" $BENCH
python tools/pmc_traffic.py $OUT/pmc_fetch/pmc_counter_collection.csv $OUT/pmc_write/pmc_counter_collection.csv \
       $OUT/${TAG}_pmc_hbm_traffic.json "bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-fp16-leg" $OUT/pmc_mfma/pmc_counter_collection.csv
mkdir -p $R/profiles && cp $OUT/${TAG}_pmc_hbm_traffic.json $R/profiles/      # (on the box only: so that step 2 can quote it)
# 2. the default bench line (fp32 leg + fp16 leg + CPU baseline + parity)
timeout -k 10 500 python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/${TAG}_bench_default.json; echo
# 3. kernel stats of the same command
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-fp16-leg > $OUT/kt.log 2>&1
cp $OUT/kt/kt_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt16 -o kt -- python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --fp16 > $OUT/kt16.log 2>&1
cp $OUT/kt16/kt_kernel_stats.csv $OUT/${TAG}_kernel_stats_fp16.csv
cd $R
# 4. instruction mix of the two RoIAlign kernels (verdict item 6) and of the AR-FPN streaming kernels
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 300 bash tools/pmc_run.sh final/sq/$tag "$pass" $R/tools/probe/roi_bwd_probe.py || echo "pass $tag failed"
  timeout -k 10 300 bash tools/pmc_run.sh final/sqf/$tag "$pass" $R/tools/fpn_bench.py || echo "pass $tag failed"
done
{ echo "# rocprofv3 --pmc passes of tools/probe/roi_bwd_probe.py (per launch averages)"; for k in roialign_bwd_tiles roialign_fwd_sep; do echo "== $k"; python tools/pmc_show.py $OUT/sq $k; done; } > $OUT/${TAG}_pmc_roialign_sq.txt
{ echo "# rocprofv3 --pmc passes of tools/fpn_bench.py (per launch averages)"; for k in fpn_gather_gate fpn_scatter_fwd2 fpn_scatter_bwd_fused fpn_gate_wgrad_mfma fpn_input_bwd2; do echo "== $k"; python tools/pmc_show.py $OUT/sqf $k; done; } > $OUT/${TAG}_pmc_fpn_sq.txt
ls $OUT | grep $TAG
