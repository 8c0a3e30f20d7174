"""usage: lanes_timeline.py [one|two] -- the recorded cfg-2 step re-issued with a timing event
around every launch (scae_launch_list_timeline): start, duration and lane of every launch as
the two lanes really run (rocprofv3's kernel trace serialises a process's dispatches)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch_scae_amd import _lib
mode = sys.argv[1] if len(sys.argv) > 1 else "two"
cfg = bench.CONFIGS[os.environ.get("WL", "mnist_24_24_bs128")]
dev = torch.device("cuda", 0)
images, labels = bench.synthetic_batches(cfg, dev, 8)
# SKIP=k: k pool streams are taken (and dropped) first, so that the step's own streams are
# other members of torch's pool -- streams share hardware queues (GPU_MAX_HW_QUEUES)
_burn = [torch.cuda.Stream() for _ in range(int(os.environ.get("SKIP", "0")))]
step = bench.make_step(cfg, dev, replay="launches", two_lanes=mode == "two")
if os.environ.get("RESIDENT"):
    step.plan.side_resident = int(os.environ["RESIDENT"])
step.prepare(images[0], labels[0])
main_ctx = torch.cuda.stream(torch.cuda.Stream()) if os.environ.get("MAIN") == "new" \
    else torch.cuda.stream(torch.cuda.current_stream())
main_ctx.__enter__()
names = []
for fn, _, _ in step._launches:
    names.append(getattr(fn, "__name__", "?").replace("scae_", ""))
for i in range(20):
    step(images[i % 8], labels[i % 8])
torch.cuda.synchronize()
lib = _lib.load()
n = lib.scae_launch_list_size(step._klist)
out = (ctypes.c_float * (2 * n))()
P = ctypes.c_void_p
side = step.plan.side_stream
best = None
for rep in range(5):
    step._stage(images[0], labels[0])
    _lib.call("scae_launch_list_timeline", step._klist,
              P(torch.cuda.current_stream().cuda_stream),
              None if side is None else P(side.cuda_stream), out, 2 * n)
    t = list(out)
    end = max(t[1::2])
    if best is None or end < best[0]:
        best = (end, t)
end, t = best
print(f"{mode}: {n} launches, last end {end:.1f} us (events around every launch add to it)")
if side is not None:
    lanes = [lib.scae_launch_list_lane(step._klist, i) for i in range(n)]
    last_side = max(i for i in range(n) if lanes[i] == 1)
    first_side = min(i for i in range(n) if lanes[i] == 1)
    print(f"side lane {t[2 * first_side]:.1f} .. {t[2 * last_side + 1]:.1f} us; the main lane's next "
          f"launch starts at {t[2 * (last_side + 1)]:.1f}; streams main {torch.cuda.current_stream().cuda_stream:#x} "
          f"side {side.cuda_stream:#x}")
if os.environ.get("BRIEF"):
    sys.exit(0)
# (C-ABI calls and kernel launches are one to one here except for multi-kernel launchers)
for i in range(n):
    lane = lib.scae_launch_list_lane(step._klist, i)
    nm = names[i] if len(names) == n else "?"
    print(f"{t[2 * i]:8.1f} +{t[2 * i + 1] - t[2 * i]:7.2f}  lane {lane}  {nm}")
