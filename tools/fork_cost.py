"""How much does a fork/join inside a replayed HIP graph cost?  Chain of small kernels,
with and without a side-stream branch of the same total length."""
import time, torch
dev = torch.device("cuda")
a = torch.zeros(1 << 16, device=dev); b = torch.zeros(1 << 16, device=dev)
big = torch.zeros(1 << 24, device=dev); big2 = torch.zeros(1 << 24, device=dev)

def chain(t, n):
    for _ in range(n):
        t.add_(1.0)

def run(build, reps=200):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        build(s)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        build(s)
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

side = torch.cuda.Stream()
def serial_small(s): chain(a, 40); chain(b, 10)
def forked_small(s):
    chain(a, 20)
    side.wait_stream(s)
    with torch.cuda.stream(side): chain(b, 10)
    chain(a, 20)
    s.wait_stream(side)
def serial_big(s): chain(big, 20); chain(big2, 10)      # ~16M-element adds: tens of us each
def forked_big(s):
    chain(big, 5)
    side.wait_stream(s)
    with torch.cuda.stream(side): chain(big2, 10)
    chain(big, 15)
    s.wait_stream(side)
def two_forks(s):
    chain(a, 10)
    for _ in range(2):
        side.wait_stream(s)
        with torch.cuda.stream(side): chain(b, 5)
        chain(a, 15)
        s.wait_stream(side)
for name, f in (("serial 50 small", serial_small), ("forked 40+10 small", forked_small),
                ("two forks 40+10 small", two_forks),
                ("serial 30 big", serial_big), ("forked 20+10 big", forked_big)):
    print(f"{name:28s} {run(f):9.1f} us/replay")
