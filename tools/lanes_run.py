"""usage: lanes_run.py <graph|one|two> [steps] -- the captured cfg-2 step, replayed (for a
kernel trace: rocprofv3 --kernel-trace -- python3 tools/lanes_run.py two 30)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
mode = sys.argv[1] if len(sys.argv) > 1 else "two"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = bench.CONFIGS[os.environ.get("WL", "mnist_24_24_bs128")]
dev = torch.device("cuda", 0)
images, labels = bench.synthetic_batches(cfg, dev, 8)
kw = dict(graph=dict(replay="graph"), one=dict(replay="launches", two_lanes=False),
          two=dict(replay="launches", two_lanes=True))[mode]
step = bench.make_step(cfg, dev, **kw)
step.prepare(images[0], labels[0])
for i in range(n):
    step(images[i % 8], labels[i % 8])
torch.cuda.synchronize()
print("done", float(step.loss))
