#!/bin/bash
# Builds a variant of libarfe_hip.so with one translation unit recompiled under extra -D flags (kernel tuning sweeps):
#   bash tools/probe/build_variant.sh <tag> <source.hip> -DNAME=VALUE ...
# -> arfe_amd/libarfe_var_<tag>.so (git-ignored, travels to the GPU box); select it with ARFE_LIB=arfe_amd/libarfe_var_<tag>.so
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
TAG=$1; SRC=$2; shift 2
OBJ=$R/arfe_amd/csrc/_obj
mkdir -p $OBJ/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result "$@" -c $R/arfe_amd/csrc/$SRC -o $OBJ/var/$TAG.o
OBJS=""
for s in api.hip fpn.hip fpn_stream.hip roi.hip gemm_ops.hip fpn_base.hip; do
  if [ "$s" == "$SRC" ]; then OBJS="$OBJS $OBJ/var/$TAG.o"; else OBJS="$OBJS $OBJ/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o $R/arfe_amd/libarfe_var_$TAG.so $OBJS
echo $R/arfe_amd/libarfe_var_$TAG.so
