"""Start / end stamps of every workgroup of the shared trunk + likelihood launch (cfg-2), from a
-DSCAE_TL_PROF build:  bash tools/variant_lib.sh tools/ablibs/libtl_prof.so trunk_logprob.hip -DSCAE_TL_PROF
SCAE_HIP_LIB=$PWD/tools/ablibs/libtl_prof.so python tools/tl_prof.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch_scae_amd import _lib
cfg = bench.CONFIGS["mnist_24_24_bs128"]
dev = torch.device("cuda", 0)
step = bench.make_step(cfg, dev)
images, labels = bench.synthetic_batches(cfg, dev, 8)
for i in range(5): step(images[i], labels[i])       # eager or replayed: the last launch's stamps stay
torch.cuda.synchronize()
lib = _lib.load()
n = 128 + 7 * 128
buf = (ctypes.c_ulonglong * (2 * n))()
lib.scae_debug_tl_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
assert lib.scae_debug_tl_prof(buf, n) == 0
st = torch.tensor([buf[2 * i] for i in range(n)], dtype=torch.float64)
en = torch.tensor([buf[2 * i + 1] for i in range(n)], dtype=torch.float64)
ok = st > 0
print("workgroups with stamps:", int(ok.sum()), "of", n)
t0 = st[ok].min()
st, en = (st - t0) / 100.0, (en - t0) / 100.0     # us
st[~ok] = 0; en[~ok] = 0
def desc(name, s, e):
    print(f"{name}: n {len(s)}  start min/median/max {s.min():.2f} {s.median():.2f} {s.max():.2f} us   "
          f"end min/median/max {e.min():.2f} {e.median():.2f} {e.max():.2f}   duration median/max {(e-s).median():.2f} {(e-s).max():.2f}")
desc("trunk     ", st[:128][ok[:128]], en[:128][ok[:128]]); desc("likelihood", st[128:][ok[128:]], en[128:][ok[128:]])
ls = st[128:][ok[128:]]
print("likelihood start histogram (us):", torch.histc(ls, bins=14, min=0, max=14).int().tolist())
print("kernel span:", float(en.max()), "us")
