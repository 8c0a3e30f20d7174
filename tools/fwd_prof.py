"""Phase stamps of every workgroup of the three K8 forward launches (B = 128), from a -DSCAE_FWD_PROF build:
  bash tools/variant_lib.sh tools/ablibs/libfwd_prof.so conv_mfma.hip -DSCAE_FWD_PROF
  python tools/fwd_prof.py tools/ablibs/libfwd_prof.so
entry -> set-up done (descriptors, row offsets) -> main loop done -> end (k-split meeting, bias / ReLU, stores)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
P = ctypes.c_void_p
f = lambda *s: torch.randn(*s, device=dev)
st = P(torch.cuda.current_stream().cuda_stream)
buf = (ctypes.c_ulonglong * (3 * 2048 * 4))()
lib.scae_debug_fwd_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
for slot, (IH, s) in enumerate(((19, 2), (9, 1), (7, 1))):
    OH = (IH - 3) // s + 1
    x, wf, bias, y = f(B, IH, IH, 128), f(128, 9, 128), f(128), f(B, OH, OH, 128)
    junk = f(64 << 20)      # evict: the step's launches find their inputs written by another kernel
    for rep in range(4):
        junk.add_(1.0)
        x.mul_(1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert lib.scae_conv3x3_fwd_f32(P(x.data_ptr()), P(wf.data_ptr()), P(bias.data_ptr()), P(y.data_ptr()),
                                        None, None, B, IH, IH, 128, 128, s, st) == 0
        e1.record()
        torch.cuda.synchronize()
    assert lib.scae_debug_fwd_prof(buf) == 0
    t = torch.tensor(list(buf), dtype=torch.float64).view(3, 2048, 4)[slot]
    n = int((t[:, 0] > 0).sum())
    t = t[:n]
    t0 = t[:, 0].min()
    t = (t - t0) / 100.0
    d = lambda a: f"{float(a.median()):6.2f} / {float(a.max()):6.2f}"
    print(f"layer IH={IH}: {n} workgroups, event time {e0.elapsed_time(e1)*1e3:.1f} us, span {float(t[:,3].max()):.2f} us")
    print(f"   entry  median/max {d(t[:,0])}   set-up {d(t[:,1]-t[:,0])}   main loop {d(t[:,2]-t[:,1])}   "
          f"epilogue {d(t[:,3]-t[:,2])}   end {d(t[:,3])}")
    k = max(1, int(float(t[:, 3].max()) // 2) + 1)
    print("   starts per 2 us:", torch.histc(t[:, 0], bins=k, min=0, max=2 * k).int().tolist())
    print("   ends   per 2 us:", torch.histc(t[:, 3], bins=k, min=0, max=2 * k).int().tolist())
