# FC1-sized linear forward under forced tile / split-K plans (ARFE_GEMM_PLAN is read once per process)
for p in auto S2 S4 S8 L4 L8 L16; do
  if [ $p = auto ]; then unset ARFE_GEMM_PLAN; else export ARFE_GEMM_PLAN=$p; fi
  echo "plan $p: $(python tools/probe/gemm_small.py 2>/dev/null | grep 'in=12544')"
done
