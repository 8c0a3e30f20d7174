#!/bin/bash
# K1 backward A/B: the in-tree build against tools/ablibs/libold.so (build it from the commit to
# compare with: git stash / checkout, make, cp torch_scae_amd/lib/libscae_hip.so tools/ablibs/libold.so), three pose regimes
# (unit / init through bench.time_k1_kernels, trained = after NSTEPS bench steps).
out=${1:-gpurun_out/k1_ab}
mkdir -p $out
for lib in "" tools/ablibs/libold.so; do
  for regime in unit init; do
    SCAE_HIP_LIB=$lib python tools/k1_time.py mnist_24_24_bs128 $regime >> $out/k1_time.txt 2>&1
  done
  SCAE_HIP_LIB=$lib python tools/k1_time.py cifar_32_32_bs256 unit >> $out/k1_time.txt 2>&1
  SCAE_HIP_LIB=$lib python tools/k1_time.py mnist_48_64_bs1024 unit >> $out/k1_time.txt 2>&1
done
NSTEPS=600 SAVE_CAP=$out/cap.pt python tools/k1_model_pose.py > $out/pose_new.txt 2>&1
SCAE_HIP_LIB=tools/ablibs/libold.so LOAD_CAP=$out/cap.pt python tools/k1_model_pose.py > $out/pose_old.txt 2>&1
rm -f $out/cap.pt
