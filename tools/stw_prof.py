"""Stage timings (s_memtime ticks) of the wave-per-tile trunk BACKWARD (stw_bwd_kernel) at cfg-2's
shape, from a -DSCAE_STW_PROF build:
  bash tools/variant_lib.sh tools/ablibs/libstw_prof.so set_encoder_wave.hip -DSCAE_STW_PROF
  python tools/stw_prof.py tools/ablibs/libstw_prof.so"""
import ctypes, os, sys
import torch
P, I, I64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
B, N, D, L = 128, 24, 16, 3
widths = [6, 1, 16, 121]
segs = [torch.randn(B, N, w, device="cuda") for w in widths]
grads = [torch.zeros(B, N, w, device="cuda") for w in widths]
Din = sum(widths)
npar = lib.scae_set_encoder_param_count(D, Din, 0, L, 1)
grid = lib.scae_set_encoder_grid(B)
params = torch.randn(npar, device="cuda") * 0.1
pres = torch.rand(B, N, device="cuda")
z = torch.zeros(B, N, D, device="cuda"); hs = torch.zeros(B, L + 1, N, D, device="cuda")
gz = torch.randn(B, N, D, device="cuda"); pg = torch.zeros(grid, npar, device="cuda")
ptrs = (P * 4)(*[s.data_ptr() for s in segs]); gptrs = (P * 4)(*[g.data_ptr() for g in grads])
ws = (I * 4)(*widths); rs = (I * 4)(*widths); bs = (I64 * 4)(*[N * w for w in widths])
PP, PI, PL = ctypes.POINTER(P), ctypes.POINTER(I), ctypes.POINTER(I64)
lib.scae_set_encoder_fwd_f32.argtypes = [I, PP, PI, PI, PL, P, P, P, P] + [I] * 7 + [P]
lib.scae_set_encoder_bwd_f32.argtypes = [I, PP, PI, PI, PL, PP, P, P, P, P, P] + [I] * 7 + [P]
rc = lib.scae_set_encoder_fwd_f32(4, ptrs, ws, rs, bs, P(pres.data_ptr()), P(params.data_ptr()), P(z.data_ptr()), P(hs.data_ptr()), B, N, D, Din, 0, L, 1, None)
assert rc == 0, rc
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(5):
    if it == 4: ev[0].record()
    rc = lib.scae_set_encoder_bwd_f32(4, ptrs, ws, rs, bs, gptrs, P(pres.data_ptr()), P(params.data_ptr()), P(hs.data_ptr()), P(gz.data_ptr()), P(pg.data_ptr()), B, N, D, Din, 0, L, 1, None)
    assert rc == 0, rc
ev[1].record(); torch.cuda.synchronize()
print("launch (events, incl. probes): %.1f us" % (ev[0].elapsed_time(ev[1]) * 1e3))
buf = (ctypes.c_ulonglong * 160)()
lib.scae_debug_stw_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), I]
n = lib.scae_debug_stw_prof(buf, 160)
t = [buf[i] for i in range(n)]
names = ["start", "lds zero + W1"]
for l in (2, 1, 0):
    names += [f"L{l} fwd: " + x for x in ("weights+h->LDS", "qkv+barrier", "S", "softmax", "PV", "oproj+LN0", "ff+LN1")]
    names += [f"L{l} bwd: " + x for x in ("LN1+ff", "LN0+oproj", "dP+softmax", "dS writes+barrier", "dq dk dv", "projections")]
    names += [f"L{l} flush"]
names += ["fc1: operand loads issued, G -> LDS, barrier", "fc1: products", "fc1: stores"]
print(n, "stamps; total ticks", t[-1] - t[0])
for i in range(1, n):
    print("%6d  %s" % (t[i] - t[i - 1], names[i] if i < len(names) else "?"))
