"""A few launches of the K8 forward / weight-gradient kernels of the cfg-2 layers
with the tile shape given by the environment -- for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
cfg = bench.CONFIGS["mnist_24_24_bs128"]
r = bench.time_k8_kernels(cfg, torch.device("cuda:0"), reps=2)
print({k: [round(l["seconds"] * 1e6, 1) for l in v] for k, v in r.items()})
