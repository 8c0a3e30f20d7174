"""Stand-alone time of the wave-per-tile trunk kernels (forward, backward) at cfg-2's / cfg-3's set shapes.
usage: python tools/stw_time.py [lib.so]   (default: the in-tree library, or SCAE_HIP_LIB)"""
import ctypes, os, sys
import torch
path = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("SCAE_HIP_LIB", "torch_scae_amd/lib/libscae_hip.so")
lib = ctypes.CDLL(os.path.abspath(path))
P, I, I64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
PP, PI, PL = ctypes.POINTER(P), ctypes.POINTER(I), ctypes.POINTER(I64)
lib.scae_set_encoder_fwd_f32.argtypes = [I, PP, PI, PI, PL, P, P, P, P] + [I] * 7 + [P]
lib.scae_set_encoder_bwd_f32.argtypes = [I, PP, PI, PI, PL, PP, P, P, P, P, P] + [I] * 7 + [P]
for B, N in ((128, 24), (1024, 48)):
    D, L = 16, 3
    widths = [6, 1, 16, 121]
    torch.manual_seed(0)
    segs = [torch.randn(B, N, w, device="cuda") for w in widths]
    Din = sum(widths)
    npar = lib.scae_set_encoder_param_count(D, Din, 0, L, 1)
    grid = lib.scae_set_encoder_grid(B)
    params = torch.randn(npar, device="cuda") * 0.1
    pres = torch.rand(B, N, device="cuda"); gz = torch.randn(B, N, D, device="cuda")
    z = torch.zeros(B, N, D, device="cuda"); hs = torch.zeros(B, L + 1, N, D, device="cuda")
    grads = [None, None, torch.zeros(B, N, 16, device="cuda"), None]     # (the model's: features only)
    gptrs = (P * 4)(*[g.data_ptr() if g is not None else None for g in grads]); pg = torch.zeros(grid, npar, device="cuda")
    ptrs = (P * 4)(*[s.data_ptr() for s in segs])
    ws = (I * 4)(*widths); rs = (I * 4)(*widths); bs = (I64 * 4)(*[N * w for w in widths])
    fwd = lambda: lib.scae_set_encoder_fwd_f32(4, ptrs, ws, rs, bs, P(pres.data_ptr()), P(params.data_ptr()), P(z.data_ptr()), P(hs.data_ptr()), B, N, D, Din, 0, L, 1, None)
    bwd = lambda: lib.scae_set_encoder_bwd_f32(4, ptrs, ws, rs, bs, gptrs, P(pres.data_ptr()), P(params.data_ptr()), P(hs.data_ptr()), P(gz.data_ptr()), P(pg.data_ptr()), B, N, D, Din, 0, L, 1, None)
    out = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(10): assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(200): fn()
        e1.record(); torch.cuda.synchronize()
        out[name] = round(e0.elapsed_time(e1) * 5, 2)     # us per launch, back to back
    print(f"B={B} N={N}", out)
