"""Yardstick for K8: the same three 128-channel encoder layers (cfg-2, B=128) on the vendor
library -- torch.nn.functional.conv2d forward and backward (MIOpen), NCHW and channels_last --
timed with HIP events, next to this repository's K8 launchers (bench.time_k8_kernels)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
import bench
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "mnist_24_24_bs128"]
dev = torch.device("cuda", 0)
B = cfg["batch"]
torch.backends.cudnn.benchmark = True     # MIOpen's find mode

def t(fn, reps=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best

tot = dict(nchw_fwd=0, nchw_bwd=0, cl_fwd=0, cl_bwd=0)
for li, (IH, Ci, Co, s) in enumerate(bench.conv_layers(cfg)):
    for fmt in ("nchw", "cl"):
        x = torch.randn(B, Ci, IH, IH, device=dev, requires_grad=True)
        w = torch.randn(Co, Ci, 3, 3, device=dev, requires_grad=True)
        if fmt == "cl":
            x = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_()
            w = w.detach().contiguous(memory_format=torch.channels_last).requires_grad_()
        y = F.conv2d(x, w, stride=s)
        gy = torch.randn_like(y)
        fwd = t(lambda: F.conv2d(x, w, stride=s))
        bwd = t(lambda: torch.autograd.grad(y, (x, w), gy, retain_graph=True))
        tot[fmt + "_fwd"] += fwd; tot[fmt + "_bwd"] += bwd
        print(f"layer {li + 2} ({IH}x{IH} s{s}) {fmt:4s}: fwd {fwd:7.1f} us   dgrad+wgrad {bwd:7.1f} us", flush=True)
print({k: round(v, 1) for k, v in tot.items()})
k8 = bench.time_k8_kernels(cfg, dev, reps=50)
print("K8 fwd", round(sum(l["seconds"] for l in k8["conv_fwd_kernel"]) * 1e6, 1),
      "us   K8 bwd pair", round(sum(l["seconds"] for l in k8["conv_bwd_pair_kernel"]) * 1e6, 1), "us")
