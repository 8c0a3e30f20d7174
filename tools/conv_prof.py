"""Start / end stamps of every workgroup of the three convolution backward launches of a cfg-2 step, from a
-DSCAE_CONV_PROF build:  bash tools/variant_lib.sh tools/ablibs/libconv_prof.so conv_mfma.hip -DSCAE_CONV_PROF
SCAE_HIP_LIB=$PWD/tools/ablibs/libconv_prof.so python tools/conv_prof.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch_scae_amd import _lib
cfg = bench.CONFIGS["mnist_24_24_bs128"]
dev = torch.device("cuda", 0)
step = bench.make_step(cfg, dev)
images, labels = bench.synthetic_batches(cfg, dev, 8)
for i in range(6): step(images[i], labels[i])
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (3 * 4096 * 2))()
lib.scae_debug_conv_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert lib.scae_debug_conv_prof(buf) == 0
t = torch.tensor(list(buf), dtype=torch.float64).view(3, 4096, 2)
for mode, layer in ((1, "layer 4 (mixed<1>)"), (2, "layer 3 (rider<2>)"), (0, "layer 2 (rider<0>)")):
    meta = (int(buf[(mode * 4096 + 4095) * 2]), int(buf[(mode * 4096 + 4095) * 2 + 1]))
    st, en = t[mode, :4095, 0], t[mode, :4095, 1]
    ok = st > 0
    n = int(ok.sum())
    if n == 0: print(layer, "no stamps"); continue
    st, en = st[:n], en[:n]
    t0 = st.min(); st = (st - t0) / 100; en = (en - t0) / 100
    nr, nd = meta[0] >> 32, meta[0] & 0xffffffff
    print(f"{layer}: {n} workgroups (grid {meta[1]}, riders {nr}, dgrad {nd}), span {float(en.max()):.1f} us")
    def part(name, lo, hi):
        if hi <= lo: return
        s, e = st[lo:hi], en[lo:hi]
        print(f"   {name:6s} n {hi-lo:5d}  start median/max {s.median():6.2f} {s.max():6.2f}   end median/max {e.median():6.2f} {e.max():6.2f}   duration median/max {(e-s).median():6.2f} {(e-s).max():6.2f}")
    part("all", 0, n)
    for d in range(10):   # by block index: riders, then data-gradient tiles, then weight-gradient tiles
        part(f"{10*d}%", n * d // 10, n * (d + 1) // 10)
    print("   starts per 5 us:", torch.histc(st, bins=int(float(en.max()) // 5) + 1, min=0, max=5 * (int(float(en.max()) // 5) + 1)).int().tolist())
    print("   ends   per 5 us:", torch.histc(en, bins=int(float(en.max()) // 5) + 1, min=0, max=5 * (int(float(en.max()) // 5) + 1)).int().tolist())
