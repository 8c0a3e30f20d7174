"""Run-to-run bit equality of the wave-per-tile trunk kernels (forward, backward) at cfg-2's shape."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_scae_amd import _lib
lib = _lib.load() if hasattr(_lib, "load") else None
lib = ctypes.CDLL(os.path.abspath(os.environ.get("SCAE_HIP_LIB", "torch_scae_amd/lib/libscae_hip.so")))
P, I, I64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
B, N, D, L = 128, 24, 16, 3
widths = [6, 1, 16, 121]
torch.manual_seed(0)
segs = [torch.randn(B, N, w, device="cuda") for w in widths]
Din = sum(widths)
npar = lib.scae_set_encoder_param_count(D, Din, 0, L, 1)
grid = lib.scae_set_encoder_grid(B)
params = torch.randn(npar, device="cuda") * 0.1
pres = torch.rand(B, N, device="cuda")
gz = torch.randn(B, N, D, device="cuda")
ptrs = (P * 4)(*[s.data_ptr() for s in segs])
ws = (I * 4)(*widths); rs = (I * 4)(*widths); bs = (I64 * 4)(*[N * w for w in widths])
PP, PI, PL = ctypes.POINTER(P), ctypes.POINTER(I), ctypes.POINTER(I64)
lib.scae_set_encoder_fwd_f32.argtypes = [I, PP, PI, PI, PL, P, P, P, P] + [I] * 7 + [P]
lib.scae_set_encoder_bwd_f32.argtypes = [I, PP, PI, PI, PL, PP, P, P, P, P, P] + [I] * 7 + [P]
res = []
for it in range(4):
    z = torch.full((B, N, D), float("nan"), device="cuda"); hs = torch.full((B, L + 1, N, D), float("nan"), device="cuda")
    grads = [torch.full((B, N, w), float("nan"), device="cuda") for w in widths]
    gptrs = (P * 4)(*[g.data_ptr() for g in grads]); pg = torch.full((grid, npar), float("nan"), device="cuda")
    assert lib.scae_set_encoder_fwd_f32(4, ptrs, ws, rs, bs, P(pres.data_ptr()), P(params.data_ptr()), P(z.data_ptr()), P(hs.data_ptr()), B, N, D, Din, 0, L, 1, None) == 0
    assert lib.scae_set_encoder_bwd_f32(4, ptrs, ws, rs, bs, gptrs, P(pres.data_ptr()), P(params.data_ptr()), P(hs.data_ptr()), P(gz.data_ptr()), P(pg.data_ptr()), B, N, D, Din, 0, L, 1, None) == 0
    torch.cuda.synchronize()
    res.append([z, hs, pg] + grads)
names = ["z", "hsave", "pg_partial"] + [f"grad{w}" for w in widths]
for it in range(1, 4):
    for n, a, b in zip(names, res[0], res[it]):
        same = torch.equal(a, b) or bool(((a == b) | (a.isnan() & b.isnan())).all())
        print(it, n, "same" if same else "DIFFERENT max |d| %.3e nan %d" % (float((a - b).abs().nan_to_num().max()), int(a.isnan().sum())))
