#!/bin/bash
# build ablation variants of conv_mfma.hip into scratch/libconv_<tag>.so (run here, not on the GPU box)
cd "$(dirname "$0")/.."
for v in 0 1 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSCAE_PIPE_ABL=$v -Iinclude -Itorch_scae_amd/csrc torch_scae_amd/csrc/conv_mfma.hip torch_scae_amd/csrc/abi.hip -o scratch/libconv_abl$v.so &
done
wait
ls -la scratch/libconv_abl*.so
