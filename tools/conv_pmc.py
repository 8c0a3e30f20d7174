"""Launch the K8 kernels of one cfg-2 step (B=128) a few times for PMC passes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
cfg = bench.CONFIGS["mnist_24_24_bs128"]
r = bench.time_k8_kernels(cfg, torch.device("cuda:0"), reps=4)
print({k: [round(l["seconds"] * 1e6, 1) for l in v] for k, v in r.items()})
