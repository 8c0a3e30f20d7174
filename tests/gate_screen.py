"""Test infrastructure: batches without borderline ReLU gates.

A parameter gradient at full size is a batch sum over ~10 M ReLU units.  A unit
whose pre-activation lies within fp32 round-off of a kink has two equally
valid fp32 gradients (gate open / closed), which would move single entries of
a weight gradient by one sample's share -- far above the 1e-4 bar -- without
either side being wrong.  Instead of loosening the bar, the full-size parity
tests draw their batch from candidates screened with the oracle: a sample is
kept only if EVERY piecewise-linear unit it drives (conv / MLP ReLUs, relu1
colours) sits clear of its kinks.  On such a batch the HIP path and the fp32
oracle evaluate the same smooth function, and every gradient entry is held to
1e-4.

"Clear" is calibrated per layer, not a global constant: the oracle runs twice
on the candidates, in fp32 and in fp64, and a layer's margin is ``SAFETY`` x
the largest fp32-vs-fp64 deviation any of its pre-activations shows (relative
to the layer's largest pre-activation), at least ``FLOOR``.  A 9-tap image
layer (round-off ~1e-7) then rejects a tenth of what a 1152-term layer does,
and the screen rejects what round-off could flip and nothing else.  The kept /
drawn ratio of every screening call is printed, recorded in ``LAST_STATS`` and
asserted >= ``MIN_KEPT_RATIO``: a screen that rejected most of what it drew
would be choosing the batch, not cleaning it.

Round 5: the CONVOLUTION gates -- 8 of the ~10 M units, and the ones whose sums
(1152 products in MFMA order against the oracle's blocked CPU sums) disagree
most -- are no longer screened at all.  The HIP side's own gate pattern is read
off its activations (``hip_conv_gates``) and IMPOSED on the oracle
(``imposed_gates``: the oracle's k-th ReLU returns x * mask_k), so both
sides evaluate the same piecewise-linear branch on ARBITRARY images; every unit
where the imposed gate differs from the oracle's own is counted and must have a
pre-activation within ``DISAGREE_BAR`` of its layer's largest -- the
attribution of each difference to a gate at round-off distance from its kink.
The per-capsule MLPs (1.5 M units at cfg-2, the chain kernel keeps their
activations) are imposed the same way.  What is left to the screen (``skip`` /
``skip_caps``) are the 0.25 M units inside fused kernels that keep no
activations (colour MLP, the set transformer's feed-forward layers, relu1).
"""
import contextlib

import torch
import torch.nn.functional as F

# margin = SAFETY x observed fp32 round-off of the layer.  The round-off observed is the
# ORACLE's (blocked CPU sums); the HIP convolutions add their 1152 products in MFMA
# order, whose error is a few times larger: with SAFETY 4 a seed sweep of the replayed
# step (tools/replay_sweep.py, 15 step samples, either kernel generation) showed the
# worst gradient entry at 5e-6 .. 3e-5 of its tensor's largest for most steps and at
# 1e-4 .. 2e-4 for one in eight -- single gate flips the screen had let through.
SAFETY = 8.0
FLOOR = 2e-7          # ... but at least this (relative to the layer's max)
MIN_KEPT_RATIO = 0.6       # full screens (the medium configurations of test_step_plan.py:
                           # 0.645 and 0.844 observed, profiles/r05/gate_screen.txt)
MIN_KEPT_IMPOSED = 0.9     # screens beside imposed gates: what is observed is >= 0.97
LAST_STATS = []       # [(what, kept, drawn)] of the most recent calls
ALL_STATS = []        # [(what, kept, drawn, skip)] of the whole session (conftest writes them out)
# a unit whose gate the HIP side and the oracle decide differently has a pre-activation
# this close to zero (relative to its layer's largest): fp32 round-off of a <= 1152-term sum
DISAGREE_BAR = 2e-6       # (worst ever recorded: 2.1e-7, profiles/r05/gate_screen.txt)
LAST_DISAGREEMENTS = []   # [(what, layer, units that differ, all units, worst |pre| / max)]


class _Shim:
    """Stands in for ``torch.nn.functional`` inside the oracle module.

    ``reference`` None: the fp32 pass -- keeps every relu / relu6 input.
    ``reference`` = that list: the fp64 pass -- per call, the layer's margin
    from its fp32 round-off, and per sample the smallest (distance to a kink /
    margin) over all units; a sample is clean iff that ratio is >= 1."""

    def __init__(self, B, reference=None, margin=None, skip=0, skip_caps=False):
        self.B = B
        self.reference = reference
        self.fixed_margin = margin
        self.calls = []
        self.k = 0
        self.ratio = torch.full((B,), float("inf"), dtype=torch.float64)
        self.margins = []
        # units that are not screened because their gates are imposed on the oracle instead:
        # the first ``skip`` ReLU calls (the convolution layers) and, with ``skip_caps``, the
        # ReLUs of ``capsule_layer`` (the per-capsule MLPs)
        self.skip, self.skip_caps = skip, skip_caps
        self.n_relu = 0
        self.caps_base = None     # index of capsule_layer's first ReLU call (_shimmed sets it)

    def _record(self, x, kinks):
        if self.reference is None and self.fixed_margin is None:
            self.calls.append(x.detach().clone())
            return
        z = x.detach().double()
        scale = float(z.abs().max().clamp_min(1e-30))
        if self.fixed_margin is not None:
            margin = self.fixed_margin
        else:
            ref = self.reference[self.k]
            self.k += 1
            assert ref.shape == z.shape, "fp32 / fp64 passes out of step"
            dev = float((z - ref.double()).abs().max()) / scale
            margin = max(SAFETY * dev, FLOOR)
        self.margins.append(margin)
        d = torch.stack([(z - k).abs() for k in kinks]).amin(0) / scale
        if z.dim() > 0 and z.shape[0] >= self.B and z.shape[0] % self.B == 0:
            self.ratio = torch.minimum(self.ratio,
                                       d.reshape(self.B, -1).amin(1) / margin)
        # (units driven by parameters alone are the same for every candidate:
        # nothing to screen)

    def relu(self, x):
        idx = self.n_relu
        self.n_relu += 1
        exempt = idx < self.skip or (self.skip_caps and self.caps_base is not None
                                     and idx >= self.caps_base)
        if not exempt:
            self._record(x, (0.0,))
        return F.relu(x)

    def relu6(self, x):
        self._record(x, (0.0, 6.0))
        return F.relu6(x)

    def __getattr__(self, name):
        return getattr(F, name)


@contextlib.contextmanager
def _shimmed(oracle_module, shim):
    """``shim`` replaces ``torch.nn.functional`` inside the oracle; the entry of
    ``capsule_layer`` tells it at which ReLU call the per-capsule MLPs start."""
    saved, saved_caps = oracle_module.F, oracle_module.capsule_layer

    def capsule_layer(*a, **k):
        shim.caps_base = shim.n_relu
        return saved_caps(*a, **k)
    oracle_module.F = shim
    oracle_module.capsule_layer = capsule_layer
    try:
        yield shim
    finally:
        oracle_module.F = saved
        oracle_module.capsule_layer = saved_caps


def clean_mask(O, ocfg, P32, P64, image, noise, skip=0, skip_caps=False):
    """bool (B,): samples of (image, noise) whose every gate is clear of its
    kinks by the calibrated margin (two oracle forwards: fp32, fp64).
    ``skip`` / ``skip_caps``: units left out because their gates are imposed
    (``imposed_gates``): the first ``skip`` ReLU calls (convolution layers) /
    the per-capsule MLPs."""
    B = image.shape[0]
    with torch.no_grad():
        with _shimmed(O, _Shim(B, skip=skip, skip_caps=skip_caps)) as rec32:
            O.scae_forward(P32, ocfg, image, noise, training=True)
        with _shimmed(O, _Shim(B, reference=rec32.calls, skip=skip,
                               skip_caps=skip_caps)) as rec64:
            O.scae_forward(P64, ocfg, image.double(),
                           [n.double() for n in noise], training=True)
    assert rec64.k == len(rec32.calls)
    return rec64.ratio >= 1.0, rec64.margins


def _note(what, kept, drawn, margins, skip=0):
    LAST_STATS.append((what, kept, drawn))
    ALL_STATS.append((what, kept, drawn, skip))
    del LAST_STATS[:-64]
    ratio = kept / max(1, drawn)
    print(f"[gate screen] {what}: kept {kept} of {drawn} drawn ({ratio:.3f}); "
          f"layer margins {min(margins):.1e} .. {max(margins):.1e}")
    # (with the convolution and capsule-MLP gates imposed the screen covers 0.25 M short-sum
    # units and keeps 0.97 - 1.0 of what it draws, profiles/r05/gate_screen.txt)
    floor = MIN_KEPT_IMPOSED if str(skip).endswith("+caps") else MIN_KEPT_RATIO
    assert ratio >= floor, \
        f"gate screen {what}: kept only {kept} of {drawn} candidates (floor {floor})"
    return ratio


def _params(sd):
    P32 = {k: v.detach().float() for k, v in sd.items()}
    return P32, {k: v.double() for k, v in P32.items()}


class _Impose:
    """Stands in for ``torch.nn.functional`` inside the oracle: a ReLU call
    whose gates are known from the HIP side returns x * mask -- the
    piecewise-linear branch the HIP kernels took -- and records where that
    differs from x > 0.  ``conv``: masks of the first ReLU calls (the CNN
    encoder, which runs first in ``scae_forward``); ``caps``: masks of
    ``capsule_layer``'s ReLU calls in the oracle's order (capsule 0 layer 0,
    capsule 0 layer 1, .. of ``mlps``, then of ``caps_mlps``)."""

    def __init__(self, conv, caps=None):
        self.conv, self.caps = conv, caps or []
        self.n_relu, self.caps_base, self.stats = 0, None, []

    def relu(self, x):
        k = self.n_relu
        self.n_relu += 1
        if k < len(self.conv):
            m, tag = self.conv[k], f"conv layer {k}"
        elif self.caps_base is not None and 0 <= k - self.caps_base < len(self.caps):
            m, tag = self.caps[k - self.caps_base], "capsule MLPs"
        else:
            return F.relu(x)
        m = m.to(x.device)
        assert m.shape == x.shape, (tag, tuple(m.shape), tuple(x.shape))
        z = x.detach()
        differ = m != (z > 0)
        scale = float(z.abs().max().clamp_min(1e-30))
        worst = float(z[differ].abs().max()) / scale if bool(differ.any()) else 0.0
        self.stats.append((tag, int(differ.sum()), differ.numel(), worst))
        return x * m.to(x.dtype)

    def __getattr__(self, name):
        return getattr(F, name)


@contextlib.contextmanager
def imposed_gates(O, conv, caps=None, what=""):
    """Inside the block the oracle's convolution ReLUs (and, with ``caps``, the
    per-capsule MLPs' -- together 97 % of the model's piecewise-linear units)
    take the gate pattern the HIP kernels decided (``hip_gates``).  On exit
    every unit where that differs from the oracle's own gate must have been
    within ``DISAGREE_BAR`` of its kink: the two sides differ by round-off
    there, and only there."""
    shim = _Impose(conv, caps)
    with _shimmed(O, shim):
        yield shim
    assert shim.n_relu >= len(conv) and (not caps or shim.caps_base is not None), \
        "the oracle ran fewer ReLUs than gates were imposed"
    groups = {}
    for tag, n_diff, n, worst in shim.stats:
        g = groups.setdefault(tag, [0, 0, 0.0])
        g[0] += n_diff
        g[1] += n
        g[2] = max(g[2], worst)
    for tag, (n_diff, n, worst) in groups.items():
        LAST_DISAGREEMENTS.append((what, tag, n_diff, n, worst))
        assert worst <= DISAGREE_BAR, \
            f"{what}: {tag}: a gate differs at |pre| = {worst:.2e} of the layer's max"
    del LAST_DISAGREEMENTS[:-4096]
    print(f"[imposed gates] {what}: " + "; ".join(
        f"{tag}: {n_diff} of {n} units decided differently (worst |pre| / max {worst:.1e})"
        for tag, (n_diff, n, worst) in groups.items()))


def imposed_conv_gates(O, masks, what=""):
    return imposed_gates(O, masks, None, what)


def hip_conv_gates(model, image):
    """The gate pattern of the HIP convolution stack for ``image`` (device
    tensor) under the model's CURRENT parameters: bool NCHW masks, one per
    layer, read off the activations the K8 kernels produce (deterministic: the
    same bits as inside a step)."""
    from torch_scae_amd import ops
    enc = model.part_encoder.encoder
    convs = [m for m in enc.network if isinstance(m, torch.nn.Conv2d)]
    assert enc._hip_stack
    with torch.no_grad(), ops.StepPlan("gates").active():
        acts, _, _ = ops._conv_stack_fwd(image, enc.strides,
                                         [c.weight.detach() for c in convs],
                                         [c.bias.detach() for c in convs])
        torch.cuda.synchronize()
        return [(a > 0).permute(0, 3, 1, 2).contiguous().cpu() for a in acts]


def hip_gates(model, image, noise):
    """(conv masks, capsule-MLP masks) of the HIP model for (image, noise)
    under its CURRENT parameters, in the form ``imposed_gates`` takes.  The
    capsule MLPs' gates are read off the activations the chain kernel (K7b)
    stores for its backward, during one eager forward with ``noise`` replayed
    (the same bits as inside a fused / replayed step: the bitwise rider tests);
    None where the model does not take that kernel."""
    from torch_scae_amd import nn_utils, ops
    captured, real = [], ops._chain_forward_desc

    def spy(*a, **k):
        d, acts = real(*a, **k)
        captured.append(acts)
        return d, acts
    ops._chain_forward_desc = spy
    try:
        with torch.no_grad(), nn_utils.fixed_noise([n.clone() for n in noise]):
            model(image)
        torch.cuda.synchronize()
    finally:
        ops._chain_forward_desc = real
    caps = None
    if len(captured) == 1 and len(captured[0]) == 4:
        a0, a1, a2, a3 = captured[0]     # (G,B,N) x 3, (B,G,N)
        caps = []
        for i in range(a0.shape[0]):
            caps += [a0[i] > 0, a1[i] > 0]
        for i in range(a0.shape[0]):
            caps += [a2[i] > 0, a3[:, i] > 0]
        caps = [m.cpu() for m in caps]
    return hip_conv_gates(model, image), caps


def screened_scae_batch(O, cfg, sd, B, gen, n_classes=10, max_rounds=40, skip=0,
                        skip_caps=False):
    """(image, label, noise) of B samples for the SCAE config ``cfg`` with
    parameters ``sd``, every sample clean by the criterion above (candidate
    chunks of B samples).  ``skip``: leading ReLU calls not screened."""
    M, Oc = cfg["n_part_caps"], cfg["n_obj_caps"]
    P32, P64 = _params(sd)
    ocfg = O.prepare_model_params(**cfg)
    keep = [[], [], [], [], []]
    have = drawn = kept = 0
    for _ in range(max_rounds):
        image = torch.rand(B, *cfg["image_shape"], generator=gen)
        label = torch.randint(0, n_classes, (B,), generator=gen)
        noise = [torch.rand(B, M, generator=gen),
                 torch.rand(B, Oc, 1, generator=gen),
                 torch.rand(B, Oc, M, generator=gen)]
        ok, margins = clean_mask(O, ocfg, P32, P64, image, noise, skip=skip,
                                 skip_caps=skip_caps)
        for dst, src in zip(keep, [image, label] + noise):
            dst.append(src[ok])
        drawn += B
        kept += int(ok.sum())
        have += int(ok.sum())
        if have >= B:
            break
    else:
        raise RuntimeError(f"only {have} of {B} clean samples found")
    _note(f"batch of {B} ({M}/{Oc} capsules)", kept, drawn, margins,
          f"{skip}{'+caps' if skip_caps else ''}")
    image, label, n0, n1, n2 = [torch.cat(k)[:B] for k in keep]
    return image, label, [n0, n1, n2]


def screened_batch_for_noise(O, cfg, sd, noise, gen, n_classes=10,
                             max_rounds=40, skip=0, skip_caps=False):
    """(image, label) for GIVEN noise draws (the device generator's: a
    replayed training step draws its own): sample slot b keeps noise[.][b] and
    gets candidate images until one is clean with it.  Gates of a sample
    depend on that sample alone (no batch statistics anywhere in the forward),
    so slots are screened independently."""
    B = noise[0].shape[0]
    P32, P64 = _params(sd)
    ocfg = O.prepare_model_params(**cfg)
    noise = [n.detach().float().cpu() for n in noise]
    image = torch.rand(B, *cfg["image_shape"], generator=gen)
    label = torch.randint(0, n_classes, (B,), generator=gen)
    drawn = B
    for _ in range(max_rounds):
        ok, margins = clean_mask(O, ocfg, P32, P64, image, noise, skip=skip,
                                 skip_caps=skip_caps)
        n_bad = int((~ok).sum())
        if n_bad == 0:
            break
        image[~ok] = torch.rand(n_bad, *cfg["image_shape"], generator=gen)
        drawn += n_bad
    else:
        raise RuntimeError(f"{n_bad} of {B} slots still not clean")
    _note(f"images for {B} fixed noise rows", B, drawn, margins,
          f"{skip}{'+caps' if skip_caps else ''}")
    return image, label


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
