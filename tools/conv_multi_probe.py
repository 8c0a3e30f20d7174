"""Upper bound for a multi-layer K8 launch: the forward tiles /
the backward pairs of the three 128-channel encoder layers as ONE launch without
dependencies between the layers, against one launch per layer.

  tools/variant_lib.sh tools/ablibs/libscae_multi.so conv_mfma.hip -DSCAE_CONV_MULTI_PROBE
  python tools/conv_multi_probe.py tools/ablibs/libscae_multi.so
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
P, I = ctypes.c_void_p, ctypes.c_int
layers = [(19, 128, 128, 2), (9, 128, 128, 1), (7, 128, 128, 1)]   # IH, Cin, Cout, stride
f = lambda *s: torch.randn(*s, device=dev)   # noqa: E731
T = []
for IH, Ci, Co, s in layers:
    OH = (IH - 3) // s + 1
    t = dict(x=f(B, IH, IH, Ci), wf=f(Co, 9, Ci), wd=f(Ci, 9, Co), bias=f(Co),
             wf2=f(3, Co, 9, Ci), w=f(Co, Ci, 3, 3), y=f(B, OH, OH, Co), dy=f(B, OH, OH, Co), dx=f(B, IH, IH, Ci),
             IH=IH, Ci=Ci, Co=Co, s=s, OH=OH)
    lib.scae_conv3x3_wgrad_splits.restype = I
    sp = lib.scae_conv3x3_wgrad_splits(B, OH, OH, Ci, Co)
    t["part"] = f(sp * (9 * Co * Ci + Co))
    T.append(t)
st = P(torch.cuda.current_stream().cuda_stream)
p = lambda t: P(t.data_ptr())   # noqa: E731


def arr(vals, ty=P):
    return (ty * len(vals))(*vals)


def timed(fn, reps=50):
    for _ in range(5):
        assert fn() == 0
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


def fwd_sep(order=(0, 1, 2)):
    rc = 0
    for i in order:
        t = T[i]
        rc |= lib.scae_conv3x3_fwd_f32(p(t["x"]), p(t["wf"]), p(t["bias"]), p(t["y"]), None, None,
                                       B, t["IH"], t["IH"], t["Ci"], t["Co"], t["s"], st)
    return rc


def fwd_multi(order=(0, 1, 2)):
    ts = [T[i] for i in order]
    return lib.scae_debug_conv_fwd_multi(
        len(ts), arr([t["x"].data_ptr() for t in ts]), arr([t["wf"].data_ptr() for t in ts]),
        arr([t["bias"].data_ptr() for t in ts]), arr([t["y"].data_ptr() for t in ts]),
        arr([B] * len(ts), I), arr([t["IH"] for t in ts], I), arr([t["Ci"] for t in ts], I),
        arr([t["Co"] for t in ts], I), arr([t["s"] for t in ts], I), st)


def bwd_sep(order=(2, 1, 0)):
    rc = 0
    for i in order:
        t = T[i]
        rc |= lib.scae_conv3x3_bwd_pair_f32(p(t["dy"]), p(t["wd"]), p(t["x"]), p(t["dx"]),
                                            p(t["part"]), B, t["IH"], t["IH"], t["Ci"], t["Co"],
                                            t["s"], st)
    return rc


scratch = torch.empty(1 << 16, device=dev, dtype=torch.uint8)


def bwd_multi(order=(2, 1, 0)):
    ts = [T[i] for i in order]
    return lib.scae_debug_conv_bwd_multi(
        len(ts), arr([t["dy"].data_ptr() for t in ts]), arr([t["wd"].data_ptr() for t in ts]),
        arr([t["x"].data_ptr() for t in ts]), arr([t["dx"].data_ptr() for t in ts]),
        arr([t["part"].data_ptr() for t in ts]), arr([B] * len(ts), I),
        arr([t["IH"] for t in ts], I), arr([t["Ci"] for t in ts], I),
        arr([t["Co"] for t in ts], I), arr([t["s"] for t in ts], I), p(scratch), st)


def fwd_res(i, group=0):
    t = T[i]
    return lib.scae_conv3x3_fwd_res_f32(p(t["x"]), P(t["wf2"].data_ptr() + t["wf2"][0].numel() * 4),
                                        p(t["bias"]), p(t["y"]), None, None, B, t["IH"], t["IH"],
                                        t["Ci"], t["Co"], t["s"], group, st)


for t in T:
    lib.scae_conv3x3_relayout_f32(p(t["w"]), p(t["wf2"]), p(t["wd"].clone()), t["Co"], t["Ci"], st)
for i in (1, 2):
    t = T[i]
    lib.scae_conv3x3_fwd_f32(p(t["x"]), p(t["wf2"]), p(t["bias"]), p(t["y"]), None, None, B,
                             t["IH"], t["IH"], t["Ci"], t["Co"], t["s"], st)
    ref = t["y"].clone()
    t["y"].zero_()
    assert fwd_res(i) == 0
    torch.cuda.synchronize()
    print(f"resident layer {i + 2}: max |diff| vs tiles {float((ref - t['y']).abs().max()):.2e} of "
          f"{float(ref.abs().max()):.2f}")
    print(f"   tiles {timed(lambda: fwd_sep((i,))):.1f} us;  resident by group size:",
          {G: round(timed(lambda G=G: fwd_res(i, G)), 1) for G in (1, 2, 3)
           if fwd_res(i, G) == 0})
assert lib.scae_debug_conv_bwd_multi_bytes() <= scratch.numel()
# same results?
fwd_sep()
ref = [t["y"].clone() for t in T]
for t in T:
    t["y"].zero_()
fwd_multi()
torch.cuda.synchronize()
print("fwd multi == separate:", all(torch.equal(a, t["y"]) for a, t in zip(ref, T)))
bwd_sep()
refd = [t["dx"].clone() for t in T]
refp = [t["part"].clone() for t in T]
for t in T:
    t["dx"].zero_()
    t["part"].zero_()
bwd_multi()
torch.cuda.synchronize()
print("bwd multi == separate:", all(torch.equal(a, t["dx"]) for a, t in zip(refd, T)),
      all(torch.equal(a, t["part"]) for a, t in zip(refp, T)))
print(f"B={B}")
print("fwd  per layer:", [round(timed(lambda i=i: fwd_sep((i,))), 1) for i in range(3)])
print("fwd  3 launches: %.1f us   one launch (2,3,4): %.1f   one launch (4,3,2): %.1f"
      % (timed(fwd_sep), timed(fwd_multi), timed(lambda: fwd_multi((2, 1, 0)))))
print("bwd  per layer (4,3,2):", [round(timed(lambda i=i: bwd_sep((i,))), 1) for i in (2, 1, 0)])
for env in ("2", "3"):
    os.environ["SCAE_K8_PAIR"] = env
    print(f"bwd  per layer (4,3,2) with second-generation data-gradient tiles (SCAE_K8_PAIR={env}):",
          [round(timed(lambda i=i: bwd_sep((i,))), 1) for i in (2, 1, 0)])
os.environ.pop("SCAE_K8_PAIR")
print("bwd  3 launches: %.1f us   one launch (4,3,2): %.1f   one launch (2,3,4): %.1f"
      % (timed(bwd_sep), timed(bwd_multi), timed(lambda: bwd_multi((0, 1, 2)))))
