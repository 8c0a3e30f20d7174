"""Worst gradient entry (relative to its tensor's max) of the replayed cfg-2 step against the
oracle, over a few batch seeds x 3 steps -- how close to the 1e-4 bar the timed-path test runs.
usage: [SCAE_HIP_LIB=...] python tools/replay_sweep.py [seeds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import test_timed_path as T
from tests.test_hip_model import full_size_params
name = os.environ.get("SWEEP_CFG", "cfg2")
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    cfg, B, sd, g = full_size_params(name)
    g.manual_seed(100 + seed)
    model, step = T.build_step(cfg, B, sd)
    step.capture()
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    worst = []
    for it in range(3):
        try:
            w, l2 = T.check_replayed_steps(model, step, cfg, B, P, g, 1, entry_bar=1.0,
                                           loss_rtol=1.0, update_l2=None, what=f"seed {seed}")
        except AssertionError as e:
            w = ("assert", str(e)[:80])
        worst.append(w)
    print(os.environ.get("SCAE_HIP_LIB", "in-tree"), "seed", seed,
          [(f"{a:.2e}", k.split(".")[-2] + "." + k.split(".")[-1]) if isinstance(a, float) else (a, k) for a, k in worst], flush=True)
