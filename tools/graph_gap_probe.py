"""Where does the ~8.5 us between two steps go?  Replays the captured step graph (a) as the
bench does -- an eager prologue launch, then the graph -- and (b) graph after graph without
the prologue, under `rocprofv3 --kernel-trace`; tools/graph_gap_report.py reads the trace.
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o gap -- python3 tools/graph_gap_probe.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
cfg = bench.CONFIGS["mnist_24_24_bs128"]
dev = torch.device("cuda", 0)
step = bench.make_step(cfg, dev)
images, labels = bench.synthetic_batches(cfg, dev, 8)
for i in range(10):
    step(images[i % 8], labels[i % 8])
torch.cuda.synchronize()
for i in range(40):          # (a) prologue + graph
    step(images[i % 8], labels[i % 8])
torch.cuda.synchronize()
for i in range(40):          # (b) graph after graph
    step.graph.replay()
torch.cuda.synchronize()
