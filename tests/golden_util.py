"""Load the committed golden fixtures (tests/golden/*.npz)."""
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name, device="cpu"):
    """-> (dict key -> torch tensor, meta dict or None)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = None
    out = {}
    for k in z.files:
        if k == "__meta__":
            meta = json.loads(bytes(z[k]).decode())
            continue
        out[k] = torch.from_numpy(z[k]).to(device)
    return out, meta


def sub(blob, prefix):
    """Entries under ``prefix`` with the prefix stripped."""
    n = len(prefix)
    return {k[n:]: v for k, v in blob.items() if k.startswith(prefix)}


def model_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR)
                  if f.startswith("scae_") and f.endswith(".npz"))


def close(a, b, atol, rtol):
    a = torch.as_tensor(a).detach().cpu()
    b = torch.as_tensor(b).detach().cpu()
    if a.dtype != b.dtype:
        a, b = a.double(), b.double()
    if a.shape != b.shape:
        return False, f"shape {tuple(a.shape)} vs {tuple(b.shape)}"
    if a.numel() == 0:
        return True, ""
    if not a.dtype.is_floating_point:
        ok = bool((a == b).all())
        return ok, "" if ok else "integer mismatch"
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    ok = bool((err <= tol).all())
    return ok, f"max abs err {float(err.max()):.3e} (max |ref| {float(b.abs().max()):.3e})"


def assert_close(a, b, atol=1e-6, rtol=1e-5, what=""):
    ok, msg = close(a, b, atol, rtol)
    assert ok, f"{what}: {msg}"
