# needs a probe build: bash tools/probe/build_variant.sh probes gemm_ops.hip -DARFE_PROBES && export ARFE_LIB=$PWD/arfe_amd/libarfe_var_probes.so
# time decomposition of the two stage-once conv kernels through their probe switches (ARFE_CONV_DBG bits:
# 1 = no epilogue, 2 = no MFMA phase, 4 = no activation staging, 8 = no weight DMA)
for d in 0 1; do echo "wgrad dbg=$d"; ARFE_CONV_DBG=$d python tools/wgrad_one.py 1024 2>&1 | grep "R=1024"; done
