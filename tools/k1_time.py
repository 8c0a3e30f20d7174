"""K1 kernel timings (render_fwd / logprob_fwd / render_bwd) of a workload, for the
library named by SCAE_HIP_LIB (default: the in-tree build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "mnist_24_24_bs128"
regime = sys.argv[2] if len(sys.argv) > 2 else "unit"
k1 = bench.time_k1_kernels(bench.CONFIGS[wl], torch.device("cuda", 0), reps=50, pose_regime=regime)
print(os.environ.get("SCAE_HIP_LIB", "in-tree"), {k: round(v * 1e6, 1) for k, v in k1.items()}, flush=True)
