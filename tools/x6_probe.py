"""The image-resident forward (K8r) at cfg-2's layers 3 / 4: time and error against conv2d in fp64.
Its products are the exact three-way bf16 split of csrc/bf16x6.h; the fp32 MFMA chain it replaced
measured, same inputs (profiles/r06/x6_probe.txt): 9->7 25.0 us, max err 1.17e-6 / rms 1.16e-7;
7->5 14.7 us, 1.05e-6 / 1.14e-7."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from torch_scae_amd import _lib
lib = _lib.load(); P = ctypes.c_void_p
p = lambda t: None if t is None else P(t.data_ptr())
st = lambda: P(torch.cuda.current_stream().cuda_stream)


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


B, C = int(os.environ.get("B", 128)), 128
g = torch.Generator().manual_seed(0)
for IH, s in [(9, 1), (7, 1)]:
    OH = (IH - 3) // s + 1
    x = torch.relu(torch.randn(B, IH, IH, C, generator=g)).cuda()
    w = (torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5).cuda()
    bias = torch.randn(C, generator=g).cuda()
    wf = torch.empty(3, C, 9, C, device="cuda"); wd = torch.empty(C, 9, C, device="cuda")
    _lib.call("scae_conv3x3_relayout_f32", p(w), p(wf), p(wd), C, C, st())
    out = torch.empty(B, OH, OH, C, device="cuda")
    ref = torch.relu(F.conv2d(x.double().permute(0, 3, 1, 2).cpu(), w.double().cpu(),
                              bias.double().cpu(), stride=s)).permute(0, 2, 3, 1)
    for x6 in ("1",):
        call = lambda: _lib.call("scae_conv3x3_fwd_res_f32", p(x), p(wf[1]), p(bias), p(out), None,
                                 None, B, IH, IH, C, C, s, 0, st())
        call(); torch.cuda.synchronize()
        err = (out.double().cpu() - ref).abs()
        t = timeit(call)
        print(f"{IH}->{OH} B={B} {'bf16 x 6' if x6 == '1' else 'fp32 MFMA'}: {t:6.2f} us   max err "
              f"{float(err.max()):.2e}  rms err {float(err.pow(2).mean().sqrt()):.2e}  (max |ref| "
              f"{float(ref.abs().max()):.2f})")
