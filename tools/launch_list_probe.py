"""Graph replay against the same step re-issued launch by launch from the recorded C-ABI calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
cfg = bench.CONFIGS["mnist_24_24_bs128"]
dev = torch.device("cuda", 0)
step = bench.make_step(cfg, dev)
images, labels = bench.synthetic_batches(cfg, dev, 8)
for i in range(10):
    step(images[i % 8], labels[i % 8])
torch.cuda.synchronize()
print("launches recorded:", len(step._launches))
snap = step.snapshot()
def run(mode, n=200):
    step.restore(snap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step._stage(images[i % 8], labels[i % 8])
        if mode == "graph":
            step.graph.replay()
        else:
            step.replay_launches()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, float(step.loss)
for rep in range(3):
    print("graph    %.4f ms  loss %.6f" % run("graph"))
    print("launches %.4f ms  loss %.6f" % run("launches"))
t0 = time.perf_counter()
for i in range(200):
    step.replay_launches()
host = (time.perf_counter() - t0) / 200 * 1e6
torch.cuda.synchronize()
print("host time per launch list: %.1f us" % host)
