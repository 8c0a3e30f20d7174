"""Runs the K1 kernels (logprob fwd, render bwd, render fwd) a few times in isolation
(for PMC passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
cfg = bench.CONFIGS["mnist_24_24_bs128"]
bench.time_k1_kernels(cfg, torch.device("cuda", 0), reps=3)
