"""Can a chain of tiny latency-bound kernels run beside a few chip-filling compute kernels?
(a) one graph, serial; (b) one graph with a forked branch; (c) two graphs replayed on two
streams with event joins; (d) eager on two streams."""
import time, torch
dev = torch.device("cuda")
a = torch.zeros(1 << 14, device=dev)
X = torch.randn(4096, 2048, device=dev); Y = torch.randn(2048, 2048, device=dev); Z = torch.empty(4096, 2048, device=dev)
NB, NS = 4, 40

def big():
    for _ in range(NB): torch.mm(X, Y, out=Z)
def small():
    for _ in range(NS): a.add_(1.0)

def timeit(fn, reps=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

def capture(build, s):
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        build()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        build()
    return g

s1, s2, side = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
g_big = capture(big, s1); g_small = capture(small, s2)
g_serial = capture(lambda: (big(), small()), s1)
def forked():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side): small()
    big()
    cur.wait_stream(side)
g_fork = capture(forked, s1)

print(f"big alone (graph)        {timeit(g_big.replay):8.1f} us")
print(f"small alone (graph)      {timeit(g_small.replay):8.1f} us")
print(f"serial one graph         {timeit(g_serial.replay):8.1f} us")
print(f"forked one graph         {timeit(g_fork.replay):8.1f} us")

def two_graphs():
    # both replays ordered after the previous iteration through events
    with torch.cuda.stream(s1):
        g_big.replay()
    with torch.cuda.stream(s2):
        g_small.replay()
    s1.wait_stream(s2)
    s2.wait_stream(s1)
print(f"two graphs, two streams  {timeit(two_graphs):8.1f} us")

def eager2():
    with torch.cuda.stream(s1): big()
    with torch.cuda.stream(s2): small()
    s1.wait_stream(s2); s2.wait_stream(s1)
print(f"eager, two streams       {timeit(eager2):8.1f} us")
def eager1():
    big(); small()
print(f"eager, one stream        {timeit(eager1):8.1f} us")

# three-segment step: G1 (small x10) -> [big || small] -> G3 (small x10), all graphs
g_pre = capture(lambda: [a.add_(1.0) for _ in range(10)], s1)
def seg():
    with torch.cuda.stream(s1):
        g_pre.replay()
    s2.wait_stream(s1)
    with torch.cuda.stream(s1): g_big.replay()
    with torch.cuda.stream(s2): g_small.replay()
    s1.wait_stream(s2)
    with torch.cuda.stream(s1): g_pre.replay()
    s2.wait_stream(s1)
print(f"pre | big||small | post  {timeit(seg):8.1f} us   (pre alone {timeit(g_pre.replay):.1f})")
