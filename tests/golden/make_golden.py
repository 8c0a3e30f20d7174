#!/usr/bin/env python3
"""Generate golden input/output vectors from the REAL reference (bdsaglam/torch-scae).

Runs ONLY in the build container, where the reference is mounted read-only at
/root/reference.  Nothing of the reference travels to the GPU box: this script
imports it, feeds it seeded inputs, and stores inputs / parameters / noise /
outputs / gradients as small ``.npz`` fixtures next to this file.  Those
fixtures pin the oracle (``oracle/scae_oracle.py``), which in turn is the
checker for the HIP path (tests/, ``__graft_entry__.smoke``, ``bench.py``'s
cpu_baseline leg).

Recipe:
  * ``monty`` is not installed -> register an in-memory ``monty.collections``
    whose ``AttrDict`` is a dict with ``self.__dict__ = self``.
  * ``cv_ops.geometric_transform`` does ``theta *= 2*pi`` on a split view,
    which modern autograd rejects; the reference function is therefore called
    through a wrapper that makes ``torch.split`` return clones for the
    duration of the call (bit-identical values, no reference code restated).
  * the three ``torch.rand_like`` draws of one forward are captured so that
    the oracle / HIP path can be fed the identical noise.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import json
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


# ----------------------------------------------------------------------------
# reference import
# ----------------------------------------------------------------------------
def import_reference():
    class AttrDict(dict):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.__dict__ = self

    monty = types.ModuleType("monty")
    coll = types.ModuleType("monty.collections")
    coll.AttrDict = AttrDict
    monty.collections = coll
    sys.modules["monty"] = monty
    sys.modules["monty.collections"] = coll
    sys.path.insert(0, REF)
    import torch_scae  # noqa
    from torch_scae import cv_ops

    ref_gt = cv_ops.geometric_transform

    def gt_out_of_place(pose_tensor, *args, **kwargs):
        # Run the UNTOUCHED reference function, but hand it cloned split
        # outputs: its in-place ``theta *= 2*pi`` then mutates a private copy
        # instead of a view, which autograd accepts; values are bit-identical.
        orig_split = torch.split

        def split_clone(t, *a, **k):
            return tuple(x.clone() for x in orig_split(t, *a, **k))

        torch.split = split_clone
        try:
            return ref_gt(pose_tensor, *args, **kwargs)
        finally:
            torch.split = orig_split

    return ref_gt, gt_out_of_place


class NoiseTap:
    """Capture (or replay) torch.rand_like draws."""

    def __init__(self):
        self.draws = []
        self._orig = torch.rand_like

    def __enter__(self):
        def tapped(t, *a, **k):
            r = self._orig(t, *a, **k)
            self.draws.append(r.detach().clone())
            return r
        torch.rand_like = tapped
        return self

    def __exit__(self, *exc):
        torch.rand_like = self._orig


def npy(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def save(name, blob, meta=None):
    out = {k: npy(v) for k, v in blob.items()}
    if meta is not None:
        out["__meta__"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name:40s} {os.path.getsize(path) / 1024:8.1f} KiB  ({len(out)} arrays)")


# ----------------------------------------------------------------------------
# full-model goldens
# ----------------------------------------------------------------------------
TINY_BASE = dict(
    image_shape=(1, 16, 16),
    n_classes=4,
    n_part_caps=4,
    n_obj_caps=3,
    pcae_cnn_encoder_params=dict(out_channels=[8, 8], kernel_sizes=[3, 3],
                                 strides=[2, 1]),
    pcae_encoder_params=dict(n_special_features=5),
    pcae_template_generator_params=dict(template_size=(5, 5)),
    ocae_encoder_set_transformer_params=dict(dim_hidden=8, dim_out=16,
                                             n_layers=2),
    ocae_decoder_capsule_params=dict(dim_caps=4, hidden_sizes=(8,)),
    scae_params=dict(reconstruct_alternatives=False),
)


def merged(base, **over):
    import copy
    cfg = copy.deepcopy(base)
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(cfg.get(k), dict):
            cfg[k].update(v)
        else:
            cfg[k] = v
    return cfg


MODEL_CASES = {
    # name: (config, batch, train_mode)
    "scae_base": (TINY_BASE, 3, True),
    "scae_eval": (TINY_BASE, 2, False),
    "scae_rgb_noalpha_scale": (merged(
        TINY_BASE, image_shape=(3, 12, 12),
        pcae_decoder_params=dict(use_alpha_channel=False,
                                 learn_output_scale=True)), 2, True),
    "scae_soft": (merged(
        TINY_BASE, scae_params=dict(reconstruct_alternatives=False,
                                    vote_type="soft", presence_type="soft")),
        2, True),
    "scae_hard": (merged(
        TINY_BASE, scae_params=dict(reconstruct_alternatives=False,
                                    vote_type="hard", presence_type="hard")),
        2, True),
    "scae_heads3": (merged(
        TINY_BASE, ocae_encoder_set_transformer_params=dict(
            dim_hidden=8, dim_out=16, n_layers=2, n_heads=3)), 2, True),
    "scae_isab": (merged(
        TINY_BASE, ocae_encoder_set_transformer_params=dict(
            dim_hidden=8, dim_out=16, n_layers=1, n_inducing_points=4)),
        2, True),
    "scae_similarity": (merged(
        TINY_BASE,
        pcae_encoder_params=dict(n_special_features=5,
                                 similarity_transform=True),
        ocae_decoder_capsule_params=dict(dim_caps=4, hidden_sizes=(8,),
                                         similarity_transform=True,
                                         learn_vote_scale=False)), 2, True),
    "scae_alternatives": (merged(
        TINY_BASE, scae_params=dict(reconstruct_alternatives=True)), 2, False),
    # round 3: more capsules than the matrix-core kernels' 64-element tiles
    # (70 part / 66 object capsules: the general attention kernels, the
    # two-pass capsule likelihood, the unfused object encoder)
    "scae_big": (merged(
        TINY_BASE, n_part_caps=70, n_obj_caps=66,
        ocae_decoder_capsule_params=dict(dim_caps=4, hidden_sizes=(4,))),
        2, True),
    # round 6: a reference-captured model on the TIMED path's kernel instantiations --
    # 64-channel 3x3 layers (the implicit-GEMM encoder K8 / K8r instead of the vendor
    # convolution the 8-channel fixtures fall back to), the defaults everywhere else:
    # dim_hidden = 16 (the one-wave-per-tile trunk), 11 x 11 templates, capsule MLPs of
    # hidden size 128 / 32 capsule parameters (the one-launch chain), a 64-wide object
    # encoding (the folded output attention on the matrix cores; 256 would put the
    # fixture past 3 MB)
    "scae_kernels": (dict(
        image_shape=(1, 24, 24), n_classes=4, n_part_caps=8, n_obj_caps=6,
        pcae_cnn_encoder_params=dict(out_channels=[64, 64, 64],
                                     kernel_sizes=[3, 3, 3], strides=[2, 1, 1]),
        ocae_encoder_set_transformer_params=dict(dim_out=64),
        scae_params=dict(reconstruct_alternatives=False)), 4, True),
}


def jsonable(cfg):
    return json.loads(json.dumps(cfg))


def flatten_res(res, prefix, blob):
    for k, v in res.items():
        if isinstance(v, torch.Tensor):
            blob[f"{prefix}{k}"] = v
        elif isinstance(v, dict):
            for kk, vv in v.items():
                if isinstance(vv, torch.Tensor):
                    blob[f"{prefix}{k}.{kk}"] = vv


def model_golden(name, cfg, batch, train, seed):
    from torch_scae import factory
    np.random.seed(seed)
    torch.manual_seed(seed)
    model = factory.make_scae(cfg)
    # break the symmetric zero inits so every gradient path is exercised
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for pname, p in model.named_parameters():
            if float(p.abs().sum()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    model.train(train)
    image = torch.rand(batch, *cfg["image_shape"], generator=g)
    label = torch.randint(0, cfg["n_classes"], (batch,), generator=g)

    blob = {}
    for k, v in model.state_dict().items():
        blob[f"param/{k}"] = v.clone()
    blob["in/image"] = image
    blob["in/label"] = label

    with NoiseTap() as tap:
        res = model(image)
    for i, d in enumerate(tap.draws):
        blob[f"noise/{i}"] = d
    loss, log = model.loss(res, image, label)
    acc = model.calculate_accuracy(res, label)
    rec_lp = res.rec.pdf.log_prob(image)

    flatten_res(res, "res/", blob)
    for alt in ("bottom_up_rec", "top_down_rec", "top_down_per_caps_rec"):
        if alt in res:
            blob[f"res/{alt}.mode"] = res[alt].pdf.mode()
    blob["res/rec.log_prob"] = rec_lp
    blob["res/rec.mode"] = res.rec.pdf.mode()
    try:
        blob["res/rec.mode_max"] = res.rec.pdf.mode(maximum=True)
    except RuntimeError:
        pass
    blob["res/rec.mean"] = res.rec.pdf.mean()
    blob["res/rec.mixing_log_prob"] = res.rec.pdf.mixing_log_prob()
    blob["out/loss"] = loss
    blob["out/accuracy"] = acc
    for k, v in log.items():
        blob[f"log/{k}"] = v

    loss.backward()
    for pname, p in model.named_parameters():
        if p.grad is not None:
            blob[f"grad/{pname}"] = p.grad
    meta = dict(config=jsonable(cfg), batch=batch, train=train, seed=seed,
                n_noise=len(tap.draws),
                no_grad_params=[n for n, p in model.named_parameters()
                                if p.grad is None])
    save(name, blob, meta)


# ----------------------------------------------------------------------------
# per-op goldens
# ----------------------------------------------------------------------------
def op_geometric_transform(ref_gt):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(5, 7, 6, generator=g)
    x[0, 0] = 0.0
    x[0, 1, 2] = 3.7          # theta wrap (> 1 turn)
    x[0, 2, 2] = -2.25
    blob = {"in/pose": x}
    for sim in (False, True):
        for nonlin in (False, True):
            for mat in (False, True):
                y = ref_gt(x.clone(), similarity=sim, nonlinear=nonlin,
                           as_matrix=mat)
                blob[f"out/sim{int(sim)}_nl{int(nonlin)}_mat{int(mat)}"] = y
    save("op_geometric_transform", blob)


def op_geometric_transform_grad(gt_oop):
    g = torch.Generator().manual_seed(12)
    blob = {}
    for sim in (False, True):
        x = torch.randn(4, 3, 6, generator=g, requires_grad=True)
        w = torch.randn(4, 3, 6, generator=g)
        y = gt_oop(x, similarity=sim, nonlinear=True, as_matrix=False)
        (y * w).sum().backward()
        blob[f"in/pose_sim{int(sim)}"] = x
        blob[f"in/w_sim{int(sim)}"] = w
        blob[f"out/y_sim{int(sim)}"] = y
        blob[f"grad/pose_sim{int(sim)}"] = x.grad
    save("op_geometric_transform_grad", blob)


def op_qkv_attention():
    from torch_scae.set_transformer import qkv_attention
    g = torch.Generator().manual_seed(21)
    blob = {}
    cases = {
        "plain": (4, 10, 10, 16, 32, "rand"),
        "nopresence": (3, 5, 7, 6, 6, None),
        "saturated": (3, 6, 6, 8, 8, "saturated"),
        "ties": (2, 4, 6, 8, 4, "ties"),
        "wide": (2, 24, 24, 256, 256, "saturated"),
    }
    for name, (B, N, M, dk, dv, pres) in cases.items():
        q = torch.randn(B, N, dk, generator=g, requires_grad=True)
        k = torch.randn(B, M, dk, generator=g, requires_grad=True)
        v = torch.randn(B, M, dv, generator=g, requires_grad=True)
        w = torch.randn(B, N, dv, generator=g)
        p = None
        if pres == "rand":
            p = torch.rand(B, M, generator=g)
        elif pres == "saturated":
            # presence exactly 1.0 for most keys, <1 for a few -> ordinary
            # softmax among the saturated keys
            p = torch.ones(B, M)
            p[:, 1] = 0.3
            p[0, :] = 1.0
        elif pres == "ties":
            p = torch.full((B, M), 0.25)
            p[:, 0] = 0.75
            p[:, 3] = 0.75          # two equal maxima -> 0.5/0.5 routing
            p[1, 5] = 1e-20
        out = qkv_attention(q, k, v, p)
        (out * w).sum().backward()
        blob[f"{name}/q"], blob[f"{name}/k"], blob[f"{name}/v"] = q, k, v
        blob[f"{name}/w"] = w
        if p is not None:
            blob[f"{name}/presence"] = p
        blob[f"{name}/out"] = out
        blob[f"{name}/gq"], blob[f"{name}/gk"], blob[f"{name}/gv"] = \
            q.grad, k.grad, v.grad
    save("op_qkv_attention", blob)


def grab_params(mod, blob, prefix="param/"):
    for k, v in mod.state_dict().items():
        blob[prefix + k] = v.clone()


def op_set_transformer_blocks():
    from torch_scae import set_transformer as st
    g = torch.Generator().manual_seed(31)
    torch.manual_seed(31)

    def run(name, mod, inputs, blob):
        for k, v in inputs.items():
            if v is not None:
                blob[f"{name}/in/{k}"] = v
        grab_params(mod, blob, f"{name}/param/")
        out = mod(*[v for v in inputs.values()])
        w = torch.randn(out.shape, generator=g)
        (out * w).sum().backward()
        blob[f"{name}/w"] = w
        blob[f"{name}/out"] = out
        for k, p in mod.named_parameters():
            if p.grad is not None:
                blob[f"{name}/grad/{k}"] = p.grad
        for k, v in inputs.items():
            if v is not None and v.grad is not None:
                blob[f"{name}/gin/{k}"] = v.grad

    blob = {}
    B, N, M, d = 3, 5, 5, 16
    q = torch.randn(B, N, d, generator=g, requires_grad=True)
    k = torch.randn(B, M, d, generator=g, requires_grad=True)
    v = torch.randn(B, M, 32, generator=g, requires_grad=True)
    p = torch.rand(B, M, generator=g)
    p[:, 0] = 1.0
    p[:, 2] = 1.0
    run("mha_h3", st.MultiHeadQKVAttention(d_k=d, d_v=32, n_heads=3),
        dict(q=q, k=k, v=v, presence=p), blob)

    q2 = torch.randn(B, N, d, generator=g, requires_grad=True)
    k2 = torch.randn(B, M, d, generator=g, requires_grad=True)
    run("mab_h3", st.MAB(d=d, n_heads=3, layer_norm=False),
        dict(q=q2, k=k2, presence=p), blob)
    q3 = torch.randn(B, N, d, generator=g, requires_grad=True)
    k3 = torch.randn(B, M, d, generator=g, requires_grad=True)
    run("mab_ln", st.MAB(d=d, n_heads=2, layer_norm=True),
        dict(q=q3, k=k3, presence=p), blob)
    x4 = torch.randn(B, M, d, generator=g, requires_grad=True)
    run("sab", st.SAB(d=d, n_heads=1, layer_norm=True),
        dict(x=x4, presence=p), blob)
    x5 = torch.randn(B, M, d, generator=g, requires_grad=True)
    run("isab", st.ISAB(d=d, n_heads=2, n_inducing_points=M, layer_norm=True),
        dict(x=x5, presence=p), blob)
    x5b = torch.randn(B, M, d, generator=g, requires_grad=True)
    run("isab_nopres", st.ISAB(d=d, n_heads=1, n_inducing_points=3,
                               layer_norm=False),
        dict(x=x5b, presence=None), blob)
    x6 = torch.randn(B, M, d, generator=g, requires_grad=True)
    run("pma", st.PMA(d=d, n_heads=1, n_seeds=M, layer_norm=True),
        dict(x=x6, presence=p), blob)
    x7 = torch.randn(B, M, 11, generator=g, requires_grad=True)
    run("st_sab", st.SetTransformer(dim_in=11, dim_hidden=d, dim_out=24,
                                    n_outputs=4, n_layers=2, n_heads=1,
                                    layer_norm=True),
        dict(x=x7, presence=p), blob)
    x8 = torch.randn(B, M, 11, generator=g, requires_grad=True)
    run("st_isab", st.SetTransformer(dim_in=11, dim_hidden=d, dim_out=24,
                                     n_outputs=4, n_layers=2, n_heads=3,
                                     layer_norm=True, n_inducing_points=M),
        dict(x=x8, presence=p), blob)
    save("op_set_transformer_blocks", blob)


def op_capsule_likelihood():
    from torch_scae.object_decoder import CapsuleLikelihood
    g = torch.Generator().manual_seed(41)
    blob = {}
    for name, (B, O, V) in {"a": (3, 4, 5), "b": (2, 7, 3)}.items():
        P = 6
        vote = torch.randn(B, O, V, P, generator=g, requires_grad=True)
        scale = (torch.rand(B, O, V, generator=g) + 0.2).requires_grad_(True)
        vp = torch.rand(B, O, V, generator=g)
        vp[0, 0, 0] = 0.0                 # log_safe branch
        vp[0, 1, 1] = 1e-20
        vp[0, :, 2] = 0.005               # below the dummy's 0.01 -> binary 0
        vp.requires_grad_(True)
        dummy = torch.randn(1, 1, V, P, generator=g, requires_grad=True)
        x = torch.randn(B, V, P, generator=g, requires_grad=True)
        pres = torch.rand(B, V, generator=g, requires_grad=True)
        res = CapsuleLikelihood(vote, scale, vp, dummy)(x, pres)
        ws = {}
        tot = res.log_prob * 1.7
        for kk in ("winner", "winner_presence", "soft_winner",
                   "soft_winner_presence", "posterior_mixing_prob",
                   "mixing_log_prob", "mixing_logit"):
            ws[kk] = torch.randn(res[kk].shape, generator=g)
            tot = tot + (res[kk] * ws[kk]).sum()
        tot.backward()
        for kk, t in dict(vote=vote, scale=scale, vote_presence=vp,
                          dummy_vote=dummy, x=x, presence=pres).items():
            blob[f"{name}/in/{kk}"] = t
            blob[f"{name}/grad/{kk}"] = t.grad
        for kk, t in res.items():
            blob[f"{name}/out/{kk}"] = t
        for kk, t in ws.items():
            blob[f"{name}/w/{kk}"] = t
    # no-presence variant (forward only)
    vote = torch.randn(2, 3, 4, 6, generator=g)
    scale = torch.rand(2, 3, 4, generator=g) + 0.3
    vp = torch.rand(2, 3, 4, generator=g)
    dummy = torch.randn(1, 1, 4, 6, generator=g)
    x = torch.randn(2, 4, 6, generator=g)
    res = CapsuleLikelihood(vote, scale, vp, dummy)(x, None)
    for kk, t in dict(vote=vote, scale=scale, vote_presence=vp,
                      dummy_vote=dummy, x=x).items():
        blob[f"nopres/in/{kk}"] = t
    for kk, t in res.items():
        blob[f"nopres/out/{kk}"] = t
    save("op_capsule_likelihood", blob)


def op_capsule_layer():
    from torch_scae.object_decoder import CapsuleLayer, CapsuleObjectDecoder
    g = torch.Generator().manual_seed(51)
    blob = {}
    variants = {
        "default": dict(learn_vote_scale=True, allow_deformations=True,
                        noise_type="uniform", noise_scale=4.,
                        similarity_transform=False),
        "sim_nonoise": dict(learn_vote_scale=False, allow_deformations=False,
                            noise_type=None, noise_scale=0.,
                            similarity_transform=True),
    }
    for name, kw in variants.items():
        torch.manual_seed(52)
        B, O, F, V, D = 3, 4, 10, 5, 6
        layer = CapsuleLayer(n_caps=O, dim_feature=F, n_votes=V, dim_caps=D,
                             hidden_sizes=(7,), **kw)
        dec = CapsuleObjectDecoder(layer)
        with torch.no_grad():
            for p in dec.parameters():
                if float(p.abs().sum()) == 0.0:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.2)
        feat = torch.randn(B, O, F, generator=g, requires_grad=True)
        grab_params(layer, blob, f"{name}/layer_param/")
        with NoiseTap() as tap:
            res = layer(feat)
        for i, d in enumerate(tap.draws):
            blob[f"{name}/noise/{i}"] = d
        tot = res.cpr_dynamic_reg_loss * 0.9
        for kk in ("vote", "scale", "vote_presence", "presence_logit_per_caps",
                   "presence_logit_per_vote"):
            w = torch.randn(res[kk].shape, generator=g)
            blob[f"{name}/w/{kk}"] = w
            tot = tot + (res[kk] * w).sum()
            blob[f"{name}/out/{kk}"] = res[kk]
        blob[f"{name}/out/cpr_dynamic_reg_loss"] = res.cpr_dynamic_reg_loss
        tot.backward()
        blob[f"{name}/in/feature"] = feat
        blob[f"{name}/grad/feature"] = feat.grad
        for k, p in layer.named_parameters():
            if p.grad is not None:
                blob[f"{name}/grad/{k}"] = p.grad

        # full object decoder, forward only, same params
        grab_params(dec, blob, f"{name}/dec_param/")
        x = torch.randn(B, V, 6, generator=g)
        pres = torch.rand(B, V, generator=g)
        with torch.no_grad(), NoiseTap() as tap:
            r2 = dec(feat.detach(), x, pres)
        for i, d in enumerate(tap.draws):
            blob[f"{name}/dec_noise/{i}"] = d
        blob[f"{name}/dec_in/x"] = x
        blob[f"{name}/dec_in/presence"] = pres
        for kk, t in r2.items():
            blob[f"{name}/dec_out/{kk}"] = t
    save("op_capsule_layer", blob)


def op_image_decoder():
    from torch_scae.part_decoder import (TemplateBasedImageDecoder,
                                         TemplateGenerator)
    g = torch.Generator().manual_seed(61)
    blob = {}
    meta = {}
    combos = {
        # name: (C, HW, M, tsize, scale, alpha, bgval, presence, bgimage)
        "default": (1, (12, 12), 3, (5, 5), False, True, True, True, False),
        "rgb": (3, (10, 14), 3, (5, 5), False, True, True, True, False),
        "scale": (1, (12, 12), 3, (5, 5), True, True, True, True, False),
        "noalpha": (1, (12, 12), 3, (5, 5), False, False, True, True, False),
        "noalpha_rgb_scale": (3, (9, 9), 2, (4, 6), True, False, True, True,
                              False),
        "nopresence": (1, (12, 12), 3, (5, 5), False, True, True, False,
                       False),
        "bgimage": (1, (12, 12), 3, (5, 5), False, True, True, True, True),
        "bgimage_nobgval": (3, (8, 8), 2, (5, 5), False, True, False, True,
                            True),
        "big_template": (1, (8, 8), 2, (13, 13), False, True, True, True,
                         False),
    }
    for name, (C, HW, M, ts, sc, al, bgv, pr, bgi) in combos.items():
        torch.manual_seed(62)
        dec = TemplateBasedImageDecoder(
            n_templates=M, template_size=ts, output_size=HW,
            learn_output_scale=sc, use_alpha_channel=al, background_value=bgv)
        with torch.no_grad():
            for p in dec.parameters():
                if float(p.abs().sum()) == 0.0:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.5)
        B = 3
        templates = torch.rand(B, M, C, *ts, generator=g).requires_grad_(True)
        # poses: a near-identity one, a random one, one mapping fully outside.
        # (irrational-looking entries on purpose: "nice" values such as 1.1 /
        # 0.05 put output pixels EXACTLY on template texel boundaries, where
        # the bilinear interpolant has a kink and its one-sided derivatives --
        # both legitimate -- differ between fp32 evaluation orders.)
        pose = torch.randn(B, M, 6, generator=g) * 0.7
        pose[:, 0] = torch.tensor([1.1037, 0.0971, 0.0513, -0.1043, 0.9029,
                                   -0.0487])
        pose[0, 1] = torch.tensor([0.1, 0.0, 5.0, 0.0, 0.1, 5.0])   # outside
        pose.requires_grad_(True)
        presence = None
        if pr:
            presence = torch.rand(B, M, generator=g)
            presence[0, 0] = 1.0
            presence[1, 1] = 0.0          # log_safe branch
            presence.requires_grad_(True)
        bg_image = None
        if bgi:
            bg_image = torch.rand(B, C, *HW, generator=g).requires_grad_(True)
        x = torch.rand(B, C, *HW, generator=g)
        res = dec(templates, pose, presence, bg_image)
        lp = res.pdf.log_prob(x)
        w = torch.randn(lp.shape, generator=g)
        (lp * w).sum().backward()
        grab_params(dec, blob, f"{name}/param/")
        blob[f"{name}/in/templates"] = templates
        blob[f"{name}/in/pose"] = pose
        blob[f"{name}/in/x"] = x
        blob[f"{name}/in/w"] = w
        blob[f"{name}/grad/templates"] = templates.grad
        blob[f"{name}/grad/pose"] = pose.grad
        if presence is not None:
            blob[f"{name}/in/presence"] = presence
            blob[f"{name}/grad/presence"] = presence.grad
        if bg_image is not None:
            blob[f"{name}/in/bg_image"] = bg_image
            blob[f"{name}/grad/bg_image"] = bg_image.grad
        for k, p in dec.named_parameters():
            if p.grad is not None:
                blob[f"{name}/pgrad/{k}"] = p.grad
        blob[f"{name}/out/transformed_templates"] = res.transformed_templates
        blob[f"{name}/out/mixing_logits"] = res.mixing_logits
        blob[f"{name}/out/log_prob"] = lp
        with torch.no_grad():
            blob[f"{name}/out/mean"] = res.pdf.mean()
            blob[f"{name}/out/mode"] = res.pdf.mode()
            try:
                blob[f"{name}/out/mode_max"] = res.pdf.mode(maximum=True)
            except RuntimeError:
                # reference quirk: the in-place `+=` in mode(maximum=True)
                # cannot broadcast (B,K,1,H,W) logits against C>1 channels
                pass
            blob[f"{name}/out/mixing_log_prob"] = res.pdf.mixing_log_prob()

        # second backward: gradients THROUGH the materialised tensors
        templates2 = templates.detach().clone().requires_grad_(True)
        pose2 = pose.detach().clone().requires_grad_(True)
        pres2 = None if presence is None else \
            presence.detach().clone().requires_grad_(True)
        for p in dec.parameters():
            p.grad = None
        res2 = dec(templates2, pose2, pres2,
                   None if bg_image is None else bg_image.detach())
        wt = torch.randn(res2.transformed_templates.shape, generator=g)
        wm = torch.randn(res2.mixing_logits.shape, generator=g)
        ((res2.transformed_templates * wt).sum()
         + (res2.mixing_logits * wm).sum()).backward()
        blob[f"{name}/mat/wt"] = wt
        blob[f"{name}/mat/wm"] = wm
        blob[f"{name}/mat/grad/templates"] = templates2.grad
        blob[f"{name}/mat/grad/pose"] = pose2.grad
        if pres2 is not None:
            blob[f"{name}/mat/grad/presence"] = pres2.grad
        for k, p in dec.named_parameters():
            if p.grad is not None:
                blob[f"{name}/mat/pgrad/{k}"] = p.grad
        meta[name] = dict(C=C, HW=list(HW), M=M, template_size=list(ts),
                          learn_output_scale=sc, use_alpha_channel=al,
                          background_value=bgv)

    # template generator variants
    for name, kw in {
        "tg_default": dict(n_templates=3, n_channels=1, template_size=(5, 5),
                           template_nonlin="sigmoid", dim_feature=4,
                           colorize_templates=True, color_nonlin="sigmoid"),
        "tg_relu1_rgb": dict(n_templates=2, n_channels=3, template_size=(4, 6),
                             template_nonlin="relu1", dim_feature=4,
                             colorize_templates=True, color_nonlin="relu1"),
        "tg_nocolor": dict(n_templates=3, n_channels=1, template_size=(5, 5),
                           template_nonlin="relu1", dim_feature=None,
                           colorize_templates=False),
    }.items():
        np.random.seed(63)
        torch.manual_seed(63)
        tg = TemplateGenerator(**kw)
        grab_params(tg, blob, f"{name}/param/")
        feat = None
        if kw["colorize_templates"]:
            feat = torch.randn(2, kw["n_templates"], kw["dim_feature"],
                               generator=g, requires_grad=True)
        r = tg(feature=feat, batch_size=2)
        w = torch.randn(r.templates.shape, generator=g)
        if name != "tg_relu1_rgb":
            # reference quirk: colour relu1 does an in-place `+= .99` on a
            # ReLU output, so its backward raises; forward-only for that one
            (r.templates * w).sum().backward()
        blob[f"{name}/w"] = w
        if feat is not None:
            blob[f"{name}/in/feature"] = feat
            if feat.grad is not None:
                blob[f"{name}/grad/feature"] = feat.grad
        blob[f"{name}/out/templates"] = r.templates
        blob[f"{name}/out/raw_templates"] = r.raw_templates
        for k, p in tg.named_parameters():
            if p.grad is not None:
                blob[f"{name}/pgrad/{k}"] = p.grad
        meta[name] = {k: (list(v) if isinstance(v, tuple) else v)
                      for k, v in kw.items()}
    save("op_image_decoder", blob, meta)


def op_part_encoder():
    from torch_scae.part_encoder import CNNEncoder, CapsuleImageEncoder
    g = torch.Generator().manual_seed(71)
    blob = {}
    for name, (shape, sim, train) in {
        "affine_train": ((1, 16, 16), False, True),
        "similarity_eval": ((3, 14, 14), True, False),
    }.items():
        torch.manual_seed(72)
        cnn = CNNEncoder(input_shape=shape, out_channels=[6, 6],
                         kernel_sizes=[3, 3], strides=[2, 1])
        enc = CapsuleImageEncoder(input_shape=shape, encoder=cnn, n_caps=3,
                                  n_poses=6, n_special_features=4,
                                  similarity_transform=sim)
        with torch.no_grad():
            enc.img_embedding_bias.copy_(
                torch.randn(enc.img_embedding_bias.shape, generator=g) * 0.1)
        enc.train(train)
        img = torch.rand(2, *shape, generator=g)
        grab_params(enc, blob, f"{name}/param/")
        with torch.no_grad(), NoiseTap() as tap:
            r = enc(img)
        for i, d in enumerate(tap.draws):
            blob[f"{name}/noise/{i}"] = d
        blob[f"{name}/in/image"] = img
        for kk in ("pose", "presence", "feature"):
            blob[f"{name}/out/{kk}"] = r[kk]
    save("op_part_encoder", blob)


def op_sparsity():
    from torch_scae.object_decoder import sparsity_loss
    from torch_scae import math_ops
    g = torch.Generator().manual_seed(81)
    blob = {}
    cp = torch.rand(5, 7, generator=g)
    cp[0, 0] = 0.0
    blob["in/caps_presence"] = cp
    for lt in ("l2", "entropy", "kl"):
        x = cp.clone().requires_grad_(True)
        a, b = sparsity_loss(lt, x, n_classes=3, within_example_constant=None)
        (a * 1.3 + b * 0.7).backward()
        blob[f"out/{lt}_within"], blob[f"out/{lt}_between"] = a, b
        blob[f"grad/{lt}"] = x.grad
    a, b = sparsity_loss("l2", cp, n_classes=3, within_example_constant=1.5)
    blob["out/l2c_within"], blob["out/l2c_between"] = a, b
    t = torch.tensor([0.0, 1e-17, 1e-16, 2e-16, 0.5, 1.0, 3.0])
    blob["in/log_safe"] = t
    blob["out/log_safe"] = math_ops.log_safe(t)
    save("op_sparsity", blob)


def op_capsule_layer_hier():
    """The hierarchical form of CapsuleLayer.forward: parent_transform replaces
    the capsule's own OVR (object_decoder.py:184-187), parent_presence its own
    presence (:214-215).  A file of its own (added in round 2; the other
    fixtures are unchanged)."""
    from torch_scae.object_decoder import CapsuleLayer
    g = torch.Generator().manual_seed(53)
    blob = {}
    torch.manual_seed(54)
    B, O, F, V, D = 3, 4, 10, 5, 6
    layer = CapsuleLayer(n_caps=O, dim_feature=F, n_votes=V, dim_caps=D,
                         hidden_sizes=(7,), learn_vote_scale=True,
                         allow_deformations=True, noise_type="uniform",
                         noise_scale=4., similarity_transform=False)
    with torch.no_grad():
        for p in layer.parameters():
            if float(p.abs().sum()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
    feat = torch.randn(B, O, F, generator=g, requires_grad=True)
    pt = torch.randn(B, O, 1, 3, 3, generator=g, requires_grad=True)
    pp = torch.rand(B, O, 1, generator=g, requires_grad=True)
    grab_params(layer, blob, "layer_param/")
    with NoiseTap() as tap:
        res = layer(feat, parent_transform=pt, parent_presence=pp)
    for i, d in enumerate(tap.draws):
        blob[f"noise/{i}"] = d
    tot = res.cpr_dynamic_reg_loss * 0.9
    for kk in ("vote", "scale", "vote_presence", "presence_logit_per_caps",
               "presence_logit_per_vote"):
        w = torch.randn(res[kk].shape, generator=g)
        blob[f"w/{kk}"] = w
        tot = tot + (res[kk] * w).sum()
        blob[f"out/{kk}"] = res[kk]
    blob["out/cpr_dynamic_reg_loss"] = res.cpr_dynamic_reg_loss
    tot.backward()
    for kk, t in (("feature", feat), ("parent_transform", pt),
                  ("parent_presence", pp)):
        blob[f"in/{kk}"] = t
        blob[f"grad/{kk}"] = t.grad
    for k, p in layer.named_parameters():
        if p.grad is not None:
            blob[f"grad/{k}"] = p.grad
    save("op_capsule_layer_hier", blob)


# seeds that differ from 100 + position: with seed 109 one of scae_big's 36 k (pixel,
# template) samples lies within round-off of a texel boundary, where the bilinear
# derivative has two valid one-sided values (the op fixtures avoid that with
# "irrational" poses; a whole model's poses come out of its encoder)
SEEDS = {"scae_big": 1234}


def main():
    ref_gt, gt_oop = import_reference()
    from torch_scae import cv_ops
    if sys.argv[1:] == ["hier"]:             # only the round-2 fixture
        cv_ops.geometric_transform = gt_oop
        op_capsule_layer_hier()
        return
    if len(sys.argv) == 2 and sys.argv[1] in MODEL_CASES:   # one model fixture
        cv_ops.geometric_transform = gt_oop
        name = sys.argv[1]
        cfg, batch, train = MODEL_CASES[name]
        model_golden(name, cfg, batch, train, seed=SEEDS.get(
            name, 100 + list(MODEL_CASES).index(name)))
        return
    op_geometric_transform(ref_gt)          # untouched reference function
    cv_ops.geometric_transform = gt_oop      # needed for every backward below
    op_geometric_transform_grad(gt_oop)
    op_qkv_attention()
    op_set_transformer_blocks()
    op_capsule_likelihood()
    op_capsule_layer()
    op_image_decoder()
    op_part_encoder()
    op_sparsity()
    for i, (name, (cfg, batch, train)) in enumerate(MODEL_CASES.items()):
        model_golden(name, cfg, batch, train, seed=SEEDS.get(name, 100 + i))
    op_capsule_layer_hier()


if __name__ == "__main__":
    main()
