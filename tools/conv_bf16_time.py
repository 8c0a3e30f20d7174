"""Stand-alone timing of the bf16-resident K8 kernels (csrc/conv_bf16.hip) at cfg-3's layer shapes (B = 1024): forward, data gradient, weight gradient; algorithmic TFLOP/s."""
import ctypes, sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from torch_scae_amd import _lib
lib = _lib.load(); P = ctypes.c_void_p
def p(t): return None if t is None else P(t.data_ptr())
st = lambda: P(torch.cuda.current_stream().cuda_stream)
def bf(t): return t.to(torch.bfloat16)
# timing at the cfg-3 shapes
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, C = 1024, 128
for IH, s in [(19, 2), (9, 1), (7, 1)]:
    OH = (IH - 3) // s + 1
    x = torch.relu(torch.randn(B, IH, IH, C, device="cuda")).to(torch.bfloat16)
    w = (torch.randn(C, 9, C, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(C, device="cuda")
    out_h = torch.empty(B, OH, OH, C, device="cuda", dtype=torch.bfloat16)
    dpre = torch.randn(B, OH, OH, C, device="cuda").to(torch.bfloat16)
    din_h = torch.empty_like(x)
    splits = lib.scae_conv3x3_wgrad_bf16r_splits(B, OH, OH, C, C)
    part = torch.empty(splits * (9 * C * C + C), device="cuda")
    fl = 2.0 * B * OH * OH * C * 9 * C
    t = timeit(lambda: _lib.call("scae_conv3x3_fwd_bf16r", p(x), p(w), p(bias), p(out_h), None, None, None, B, IH, IH, C, C, s, st()))
    t2 = timeit(lambda: _lib.call("scae_conv3x3_dgrad_bf16r", p(dpre), p(w), p(x), p(din_h), None, B, IH, IH, C, C, s, st()))
    t3 = timeit(lambda: _lib.call("scae_conv3x3_wgrad_bf16r", p(dpre), p(x), p(part), B, IH, IH, C, C, s, st()))
    print(f"B=1024 {IH}->{OH}: fwd {t:.1f} us ({fl/t/1e6:.0f} TF)  dgrad {t2:.1f} us ({fl/t2/1e6:.0f} TF)  wgrad {t3:.1f} us ({fl/t3/1e6:.0f} TF, {splits} splits)")
