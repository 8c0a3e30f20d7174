import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch_scae_amd import ops

def case(B, G, Kin, dims, ident=False):
    g = torch.Generator().manual_seed(1)
    layers, K = [], Kin
    for N in dims:
        w = torch.randn(G, N, K, generator=g) / K ** 0.5
        if ident:
            w = torch.eye(N, K).expand(G, N, K).contiguous()
        layers.append((w.cuda(), None, False))
        K = N
    x = torch.randn(B, G, Kin, generator=g).abs().cuda()
    with ops.mfma_bf16(False):
        y32 = ops.mlp_chain(x, layers)
    with ops.mfma_bf16(True):
        y16 = ops.mlp_chain(x, layers)
    torch.cuda.synchronize()
    d = (y16 - y32).abs()
    print(B, G, Kin, dims, "ident" if ident else "", "max y32", float(y32.abs().max()), "max diff", float(d.max()),
          "bad frac", float((d > 0.05 * y32.abs().max()).float().mean()))
    if ident and float(d.max()) > 0.1:
        i = (d > 0.1).nonzero()[:5]
        print(i.tolist(), [(float(y16[tuple(j)]), float(y32[tuple(j)])) for j in i])

case(16, 1, 16, [16], True)
case(16, 1, 16, [16])
case(16, 1, 64, [16])
case(16, 1, 64, [64])
case(16, 1, 128, [64])
case(16, 1, 256, [128])
case(64, 2, 256, [128])
case(64, 2, 256, [128, 32])
case(64, 2, 256, [128, 32, 128])
case(64, 2, 256, [128, 32, 128, 391])
