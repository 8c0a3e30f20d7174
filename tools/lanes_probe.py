"""The captured cfg-2 step three ways: HIP-graph replay; the library's launch list on one
stream; the launch list on two streams (the object path's backward beside the part
decoder's / encoder's, step_plan.SIDE_NODES).  Same state before every run; losses must agree
bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch_scae_amd import _lib

wl = sys.argv[1] if len(sys.argv) > 1 else "mnist_24_24_bs128"
cfg = bench.CONFIGS[wl]
dev = torch.device("cuda", 0)
images, labels = bench.synthetic_batches(cfg, dev, 8)


def build(resident=None, **kw):
    torch.manual_seed(1234)
    from torch_scae_amd import ops
    ops.reset_noise()
    step = bench.make_step(cfg, dev, **kw)
    if resident is not None:
        step.plan.side_resident = resident
    step.prepare(images[0], labels[0])
    torch.cuda.synchronize()
    return step


def run(step, n=200, warm=20):
    snap = step.snapshot()
    out = []
    for rep in range(3):
        step.restore(snap)
        for i in range(warm):
            step(images[i % 8], labels[i % 8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            step(images[i % 8], labels[i % 8])
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / n * 1e3)
    step.restore(snap)
    return min(out), out


variants = [("graph", dict(replay="graph")),
            ("launches, one lane", dict(replay="launches", two_lanes=False)),
            ("launches, two lanes", dict(replay="launches", two_lanes=True))]
for r in os.environ.get("RESIDENT", "").split(","):
    if r:
        variants.append((f"two lanes, {r} resident", dict(replay="launches", two_lanes=True,
                                                          resident=int(r))))
for spec in os.environ.get("SIDE_CUS", "").split(","):
    if spec:
        variants.append((f"two lanes, side on {spec} CUs", dict(replay="launches", two_lanes=spec)))
lib = _lib.load()
for name, kw in variants:
    step = build(**kw)
    best, all_ = run(step)
    info = ""
    if step._klist:
        info = "list %d launches, %d on the side lane; graph nodes %s" % (
            lib.scae_launch_list_size(step._klist),
            lib.scae_launch_list_side_size(step._klist), step.graph_nodes)
    elif kw.get("replay") == "launches":
        info = "NO LIST (graph nodes %s): replayed as a graph" % (step.graph_nodes,)
    # a short trajectory from the restored state: the loss after 5 steps
    for i in range(5):
        loss = step(images[i % 8], labels[i % 8])
    torch.cuda.synchronize()
    print("%-22s %.4f ms/step  (%s)  loss after 5 steps %.6f  %s" % (
        name, best, ", ".join("%.4f" % v for v in all_), float(loss), info), flush=True)
    if step._klist:
        t0 = time.perf_counter()
        for i in range(100):
            step.replay_launches()
        host = (time.perf_counter() - t0) / 100 * 1e6
        torch.cuda.synchronize()
        print("   host time per list replay: %.1f us" % host, flush=True)
    del step
    torch.cuda.empty_cache()
