#!/bin/bash
# usage: tools/variant_lib.sh <out.so> <source.hip> [-DFLAG ...] -- libscae_hip.so with ONE source
# recompiled under extra flags (ablations, tuning sweeps); load it with SCAE_HIP_LIB=<out.so>
set -e
R=$(cd "$(dirname "$0")/.." && pwd); OUT=$1; SRC=$2; shift 2
B=$(basename $SRC .hip); O=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/torch_scae_amd/csrc "$@" \
  -c $R/torch_scae_amd/csrc/$B.hip -o $O/$B.o
OBJS=$(ls $R/torch_scae_amd/lib/obj/*.o | grep -v "/$B.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $O/$B.o -o $OUT
rm -rf $O
