"""Test infrastructure: batches without borderline ReLU gates.

A parameter gradient at full size is a batch sum over ~10 M ReLU units.  A unit
whose pre-activation lies within fp32 round-off of a kink has two equally
valid fp32 gradients (gate open / closed), which would move single entries of
a weight gradient by one sample's share -- far above the 1e-4 bar -- without
either side being wrong.  Instead of loosening the bar, the full-size parity
tests draw their batch from candidates screened with the oracle in fp64: a
sample is kept only if EVERY piecewise-linear unit it drives (conv / MLP ReLUs,
relu1 colours) sits at least `margin` x (largest pre-activation of that layer)
away from its kinks.  On such a batch the HIP path and the fp32 oracle
evaluate the same smooth function, and every gradient entry is held to 1e-4.
"""
import contextlib

import torch
import torch.nn.functional as F


class _Shim:
    """Stands in for ``torch.nn.functional`` inside the oracle module while
    screening: records, per sample, the smallest relative distance of any
    pre-activation to a kink of relu / relu6."""

    def __init__(self, B):
        self.B = B
        self.margin = torch.full((B,), float("inf"), dtype=torch.float64)
        self.shared = float("inf")     # units driven by parameters alone

    def _record(self, x, kinks):
        z = x.detach().double()
        scale = float(z.abs().max().clamp_min(1e-30))
        d = torch.stack([(z - k).abs() for k in kinks]).amin(0) / scale
        if z.dim() > 0 and z.shape[0] >= self.B and z.shape[0] % self.B == 0:
            self.margin = torch.minimum(self.margin,
                                        d.reshape(self.B, -1).amin(1))
        else:
            self.shared = min(self.shared, float(d.min()))

    def relu(self, x):
        self._record(x, (0.0,))
        return F.relu(x)

    def relu6(self, x):
        self._record(x, (0.0, 6.0))
        return F.relu6(x)

    def __getattr__(self, name):
        return getattr(F, name)


@contextlib.contextmanager
def record_gates(oracle_module, B):
    shim = _Shim(B)
    saved = oracle_module.F
    oracle_module.F = shim
    try:
        yield shim
    finally:
        oracle_module.F = saved


def screened_scae_batch(O, cfg, sd, B, gen, margin=4e-6, n_classes=10,
                        max_rounds=40):
    """(image, label, noise) of B samples for the SCAE config ``cfg`` with
    parameters ``sd``, every sample clean by the criterion above (oracle
    forward in fp64 on candidate chunks of B samples)."""
    M, Oc = cfg["n_part_caps"], cfg["n_obj_caps"]
    P64 = {k: v.double() for k, v in sd.items()}
    ocfg = O.prepare_model_params(**cfg)
    keep = [[], [], [], [], []]
    have = 0
    for _ in range(max_rounds):
        image = torch.rand(B, *cfg["image_shape"], generator=gen)
        label = torch.randint(0, n_classes, (B,), generator=gen)
        noise = [torch.rand(B, M, generator=gen),
                 torch.rand(B, Oc, 1, generator=gen),
                 torch.rand(B, Oc, M, generator=gen)]
        with torch.no_grad(), record_gates(O, B) as rec:
            O.scae_forward(P64, ocfg, image.double(),
                           [n.double() for n in noise], training=True)
        ok = rec.margin >= margin
        for dst, src in zip(keep, [image, label] + noise):
            dst.append(src[ok])
        have += int(ok.sum())
        if have >= B:
            break
    else:
        raise RuntimeError(f"only {have} of {B} clean samples found")
    image, label, n0, n1, n2 = [torch.cat(k)[:B] for k in keep]
    return image, label, [n0, n1, n2]


def conv_margins(image, ws, bs, strides):
    """Per-sample smallest relative distance of any conv pre-activation of
    the stack (fp64) to zero."""
    y = image.double()
    m = torch.full((image.shape[0],), float("inf"), dtype=torch.float64)
    for w, b, s in zip(ws, bs, strides):
        z = F.conv2d(y, w.double(), b.double(), stride=s)
        m = torch.minimum(m, (z.abs() / z.abs().max()).flatten(1).amin(1))
        y = F.relu(z)
    return m
