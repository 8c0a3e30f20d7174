#!/bin/bash
# usage: tools/variants.sh <tag>:<-Dflags> ...   -- builds scratch/libscae_<tag>.so (whole library) per variant
cd "$(dirname "$0")/.."
mkdir -p scratch
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags -Iinclude -Itorch_scae_amd/csrc torch_scae_amd/csrc/*.hip -o scratch/libscae_$tag.so 2>&1 | grep -E "error" ) &
done
wait
ls -la scratch/libscae_*.so | awk '{print $5, $9}'
