"""Phase timings (s_memtime ticks, 100 MHz) of the wave-per-set trunk forward,
from a -DSCAE_STW_PROF build (scratch/libstw_prof.so)."""
import ctypes, os, sys
import torch
P, I = ctypes.c_void_p, ctypes.c_int
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
B, N, D, L = 128, 24, 16, 3
widths = [6, 1, 16, 121]
segs = [torch.randn(B, N, w, device="cuda") for w in widths]
Din = sum(widths)
npar = lib.scae_set_encoder_param_count(D, Din, 0, L, 1)
params = torch.randn(npar, device="cuda") * 0.1
pres = torch.rand(B, N, device="cuda")
z = torch.zeros(B, N, D, device="cuda"); hs = torch.zeros(B, L + 1, N, D, device="cuda")
ptrs = (P * 4)(*[s.data_ptr() for s in segs]); ws = (I * 4)(*widths); rs = (I * 4)(*widths)
bs = (ctypes.c_int64 * 4)(*[N * w for w in widths])
lib.scae_set_encoder_fwd_f32.argtypes = [I, ctypes.POINTER(P), ctypes.POINTER(I), ctypes.POINTER(I), ctypes.POINTER(ctypes.c_int64), P, P, P, P] + [I] * 7 + [P]
for _ in range(3):
    rc = lib.scae_set_encoder_fwd_f32(4, ptrs, ws, rs, bs, P(pres.data_ptr()), P(params.data_ptr()), P(z.data_ptr()), P(hs.data_ptr()), B, N, D, Din, 0, L, 1, None)
    assert rc == 0, rc
torch.cuda.synchronize()
print("ticks (x10 ns): stage, fc1, layer0..2+save, tail:", z[5, 0, :8].tolist())
print("layer phases (weights+qkv, S, softmax, PV, oproj+LN0, fc+LN1):", z[5, 1, :6].tolist())
