"""The 1x1 attention conv's backward at cfg-2 (B*HW = 3200 pixels, C = 128, A*P = 576): its
data- and weight-gradient GEMMs as one launch (what the step runs), and each alone -- what
moving the weight-gradient tiles into another launch could save."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch_scae_amd import ops, _lib
B, HW, C, AP = 128, 25, 128, 576
dev = torch.device("cuda", 0)
f = lambda *s: torch.randn(*s, device=dev)   # noqa: E731
x, w, dy, gate = f(B, HW, C), f(AP, C), f(B, HW, AP), f(B, HW, C)
dx, raw = f(B, HW, C), f(B, HW, C)
p = ops._p
dgrad = ops._gemm_desc(p(dy), p(w), p(dx), 1, B * HW, C, AP, True, AP, 0, False, C, 0, C, 0)
dgrad.mask, dgrad.ldmask, dgrad.c_nomask = gate.data_ptr(), C, raw.data_ptr()
st = ops._stream(x)


def wgrad_of(gsz):
    S, kper, slab = B // gsz, HW * gsz, AP * C + AP
    part = f(S, slab)
    d = ops._gemm_desc(p(dy), p(x), p(part), S, AP, C, kper, False, AP, kper * AP, False, C, kper * C, C, slab,
                       asum=ops._off(part, AP * C), asum_b=slab)
    d._keep = part
    return d


def timed(descs, reps=100):
    arr = (_lib.GemmDesc * len(descs))(*descs)
    for _ in range(5):
        _lib.call("scae_gemm_multi_f32", arr, len(descs), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.call("scae_gemm_multi_f32", arr, len(descs), st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("dgrad alone %.1f us" % timed([dgrad]))
for gsz in (32, 16, 8, 4, 2):
    wgrad = wgrad_of(gsz)
    print("images per weight-gradient group %2d (%2d groups): pair %.1f us   wgrad alone %.1f"
          % (gsz, B // gsz, timed([wgrad, dgrad]), timed([wgrad])))
