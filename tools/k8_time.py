"""K8 timings per layer for tile-shape overrides (SCAE_K8_FWD / SCAE_K8_DG /
SCAE_K8_PAIR / SCAE_K8_WG: -1 = first-generation kernels, 0..3 = shapes)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "mnist_24_24_bs128"
cfg = bench.CONFIGS[wl]
dev = torch.device("cuda", 0)
def run(tag, **env):
    for k in ("SCAE_K8_FWD", "SCAE_K8_DG", "SCAE_K8_PAIR", "SCAE_K8_WG"):
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    k8 = bench.time_k8_kernels(cfg, dev, reps=30)
    out = {}
    for name, ls in k8.items():
        out[name] = [round(l["seconds"] * 1e6, 1) for l in ls] + \
            [round(sum(l["flops"] for l in ls) / sum(l["seconds"] for l in ls) / 1e12, 1)]
    print(tag, json.dumps(out), flush=True)
run("auto")
for a in sys.argv[2:]:
    kind, c = a.split("=")
    run(a, **{dict(fwd="SCAE_K8_FWD", dg="SCAE_K8_DG", pair="SCAE_K8_PAIR", wg="SCAE_K8_WG")[kind]: int(c)})
