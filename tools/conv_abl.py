"""Time scae_conv3x3_fwd_f32 of the cfg-2 layers from a given libconv variant
(built by tools/conv_abl.sh from conv_mfma.hip with ablation macros)."""
import ctypes, os, sys
import torch
I, P = ctypes.c_int, ctypes.c_void_p
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.scae_conv3x3_fwd_f32.argtypes = [P] * 6 + [I] * 6 + [P]
    out = []
    for cfg in (0, 2):
        os.environ["SCAE_K8_FWD"] = str(cfg)
        for (IH, s) in ((19, 2), (9, 1), (7, 1)):
            B, C = 128, 128
            OH = (IH - 3) // s + 1
            x = torch.randn(B, IH, IH, C, device="cuda"); wf = torch.randn(C, 9, C, device="cuda")
            if os.environ.get("ZERO"): x.zero_(); wf.zero_()
            bias = torch.randn(C, device="cuda"); y = torch.empty(B, OH, OH, C, device="cuda")
            st = P(torch.cuda.current_stream().cuda_stream)
            f = lambda: lib.scae_conv3x3_fwd_f32(P(x.data_ptr()), P(wf.data_ptr()), P(bias.data_ptr()), P(y.data_ptr()), None, None, B, IH, IH, C, C, s, st)
            for _ in range(3): assert f() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): f()
            e1.record(); torch.cuda.synchronize()
            out.append(round(e0.elapsed_time(e1) / 30 * 1e3, 1))
    print(os.path.basename(path), "cfg0", out[:3], "cfg2", out[3:], flush=True)
