"""The capsule-MLP chain (K7b) alone at cfg-2's sizes: forward / backward launches, HIP-event timed."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch_scae_amd import ops
B, G, Kin, dims = 128, 24, 256, [128, 32, 128, 199]
if len(sys.argv) > 2 and sys.argv[2] == "cfg3":
    B, G, Kin, dims = 1024, 64, 256, [128, 32, 128, 391]
g = torch.Generator().manual_seed(0)
layers, K = [], Kin
for l, N in enumerate(dims):
    ones = l == 2
    w = (torch.randn(G, N, K + (1 if ones else 0), generator=g) / K ** 0.5).cuda().requires_grad_()
    b = None if l >= 2 else (torch.randn(G, N, generator=g) * 0.1).cuda().requires_grad_()
    layers.append((w, b, ones)); K = N
x = torch.randn(B, G, Kin, generator=g).cuda().requires_grad_()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
y = ops.mlp_chain(x, layers)
gy = torch.randn_like(y) * (y.detach() > 0)
print("fwd %.1f us" % t(lambda: ops.mlp_chain(x, layers)))
def fb():
    y = ops.mlp_chain(x, layers); y.backward(gy)
print("fwd+bwd %.1f us (incl. host)" % t(fb))
