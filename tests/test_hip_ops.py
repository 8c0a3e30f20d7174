"""GPU parity: every HIP op (through the C ABI) against the golden vectors
captured from the reference and against the CPU oracle on random inputs.
Tolerances are the north-star's 1e-4 (fp32), written per assertion."""
import numpy as np
import pytest
import torch

from oracle import scae_oracle as O
from tests.golden_util import assert_close, load, sub

pytestmark = pytest.mark.gpu

ATOL, RTOL = 1e-5, 1e-4      # outputs
GATOL, GRTOL = 2e-5, 1e-4    # gradients


def dev(t):
    return None if t is None else t.cuda()


def leaf(t):
    return t.detach().clone().cuda().requires_grad_(True)


# ------------------------------------------------------------------ K5 -----
def test_geometric_transform_flags_vs_golden():
    from torch_scae_amd.cv_ops import geometric_transform
    blob, _ = load("op_geometric_transform")
    x = blob["in/pose"].cuda()
    for sim in (0, 1):
        for nl in (0, 1):
            for mat in (0, 1):
                y = geometric_transform(x, bool(sim), bool(nl), bool(mat))
                assert_close(y, blob[f"out/sim{sim}_nl{nl}_mat{mat}"], ATOL,
                             RTOL, f"sim{sim} nl{nl} mat{mat}")


def test_geometric_transform_grad_vs_golden():
    from torch_scae_amd.cv_ops import geometric_transform
    blob, _ = load("op_geometric_transform_grad")
    for sim in (0, 1):
        x = leaf(blob[f"in/pose_sim{sim}"])
        y = geometric_transform(x, bool(sim))
        (y * blob[f"in/w_sim{sim}"].cuda()).sum().backward()
        assert_close(y, blob[f"out/y_sim{sim}"], ATOL, RTOL, "y")
        assert_close(x.grad, blob[f"grad/pose_sim{sim}"], GATOL, GRTOL, "g")


@pytest.mark.parametrize("sim,nl,mat", [(0, 1, 1), (1, 1, 0), (0, 0, 0),
                                        (1, 0, 1)])
def test_geometric_transform_vs_oracle_large(sim, nl, mat):
    from torch_scae_amd.cv_ops import geometric_transform
    g = torch.Generator().manual_seed(3)
    x = torch.randn(128, 24, 24, 6, generator=g)
    w = torch.randn(128, 24, 24, 3, 3, generator=g) if mat else \
        torch.randn(128, 24, 24, 6, generator=g)
    xc = x.clone().requires_grad_(True)
    yo = O.geometric_transform(xc, bool(sim), bool(nl), bool(mat))
    (yo * w).sum().backward()
    xg = leaf(x)
    yg = geometric_transform(xg, bool(sim), bool(nl), bool(mat))
    (yg * w.cuda()).sum().backward()
    assert_close(yg, yo, ATOL, RTOL, "y")
    assert_close(xg.grad, xc.grad, 1e-4, 1e-4, "grad")


# ------------------------------------------------------------------ K2 -----
@pytest.mark.parametrize("case", ["plain", "nopresence", "saturated", "ties",
                                  "wide"])
def test_qkv_attention_vs_golden(case):
    from torch_scae_amd.set_transformer import qkv_attention
    blob, _ = load("op_qkv_attention")
    c = sub(blob, case + "/")
    q, k, v = leaf(c["q"]), leaf(c["k"]), leaf(c["v"])
    out = qkv_attention(q, k, v, dev(c.get("presence")))
    (out * c["w"].cuda()).sum().backward()
    assert_close(out, c["out"], 2e-5, 1e-4, "out")
    assert_close(q.grad, c["gq"], 5e-5, 1e-4, "gq")
    assert_close(k.grad, c["gk"], 5e-5, 1e-4, "gk")
    assert_close(v.grad, c["gv"], 5e-5, 1e-4, "gv")


@pytest.mark.parametrize("HB,N,M,dk,dv,pres", [
    (128, 24, 24, 16, 16, "mixed"),      # SAB of cfg-2
    (128, 24, 24, 256, 256, "mixed"),    # output attention of cfg-2
    (16, 64, 48, 256, 256, "ones"),      # cfg-3 shape
    (7, 1, 1, 3, 5, None),               # degenerate
    (5, 33, 17, 70, 130, "mixed"),       # ragged, multi-chunk
    (6, 10, 20, 6, 6, "rand"),           # heads-of-3 shape, one-hot regime
    # beyond the 64-element tiles of the matrix-core kernels: the general
    # kernels of set_attention_big.hip (round 3)
    (9, 80, 80, 16, 16, "mixed"),        # SAB over 80 part capsules
    (5, 72, 80, 256, 256, "mixed"),      # output attention, 72 seeds x 80 keys
    (3, 65, 200, 7, 33, "rand"),         # ragged
])
def test_qkv_attention_vs_oracle(HB, N, M, dk, dv, pres):
    from torch_scae_amd.set_transformer import qkv_attention
    g = torch.Generator().manual_seed(HB * 1000 + N)
    q = torch.randn(HB, N, dk, generator=g)
    k = torch.randn(HB, M, dk, generator=g)
    v = torch.randn(HB, M, dv, generator=g)
    w = torch.randn(HB, N, dv, generator=g)
    p = None
    if pres == "rand":
        p = torch.rand(HB, M, generator=g)
    elif pres == "ones":
        p = torch.ones(HB, M)
    elif pres == "mixed":
        p = torch.ones(HB, M)
        p[:, ::3] = torch.rand(HB, len(range(0, M, 3)), generator=g)
        p[0] = 1.0
    qc, kc, vc = (t.clone().requires_grad_(True) for t in (q, k, v))
    oo = O.qkv_attention(qc, kc, vc, p)
    (oo * w).sum().backward()
    qg, kg, vg = leaf(q), leaf(k), leaf(v)
    og = qkv_attention(qg, kg, vg, dev(p))
    (og * w.cuda()).sum().backward()
    scale = max(1.0, float(oo.abs().max()))
    assert_close(og, oo, 1e-4 * scale, 1e-4, "out")
    for name, a, b in (("gq", qg.grad, qc.grad), ("gk", kg.grad, kc.grad),
                       ("gv", vg.grad, vc.grad)):
        assert_close(a, b, 1e-4 * max(1.0, float(b.abs().max())), 1e-4, name)


def test_qkv_attention_presence_gradient():
    """d/d presence = +1e32 * sum_n dS (set_transformer.py:42): compare in
    units of 1e32 against autograd on the oracle."""
    from torch_scae_amd.set_transformer import qkv_attention
    g = torch.Generator().manual_seed(9)
    q, k, v = (torch.randn(3, 5, 8, generator=g) for _ in range(3))
    p = torch.ones(3, 5)
    pc = p.clone().requires_grad_(True)
    O.qkv_attention(q, k, v, pc).square().sum().backward()
    pg = leaf(p)
    qkv_attention(q.cuda(), k.cuda(), v.cuda(), pg).square().sum().backward()
    assert_close(pg.grad / 1e32, pc.grad / 1e32, 1e-4, 1e-3, "gpresence")


def test_qkv_attention_size_limits():
    """Sets beyond the 64-element tiles run on the general kernels (round 3);
    what no kernel takes -- the backward parks one problem's dS, N x M floats,
    in LDS -- is rejected loudly, not silently wrong."""
    from torch_scae_amd.ops import ScaeHipError
    from torch_scae_amd.set_transformer import qkv_attention
    x = torch.zeros(1, 65, 4, device="cuda")
    assert qkv_attention(x, x, x).shape == (1, 65, 4)
    big = torch.zeros(1, 300, 4, device="cuda", requires_grad=True)
    out = qkv_attention(big, big, big)
    with pytest.raises(ScaeHipError):
        out.sum().backward()


BLOCKS = {
    "mha_h3": ("MultiHeadQKVAttention", dict(d_k=16, d_v=32, n_heads=3),
               ("q", "k", "v", "presence")),
    "mab_h3": ("MAB", dict(d=16, n_heads=3, layer_norm=False),
               ("q", "k", "presence")),
    "mab_ln": ("MAB", dict(d=16, n_heads=2, layer_norm=True),
               ("q", "k", "presence")),
    "sab": ("SAB", dict(d=16, n_heads=1, layer_norm=True), ("x", "presence")),
    "isab": ("ISAB", dict(d=16, n_heads=2, n_inducing_points=5,
                          layer_norm=True), ("x", "presence")),
    "isab_nopres": ("ISAB", dict(d=16, n_heads=1, n_inducing_points=3,
                                 layer_norm=False), ("x",)),
    "pma": ("PMA", dict(d=16, n_heads=1, n_seeds=5, layer_norm=True),
            ("x", "presence")),
    "st_sab": ("SetTransformer", dict(dim_in=11, dim_hidden=16, dim_out=24,
                                      n_outputs=4, n_layers=2, n_heads=1,
                                      layer_norm=True), ("x", "presence")),
    "st_isab": ("SetTransformer", dict(dim_in=11, dim_hidden=16, dim_out=24,
                                       n_outputs=4, n_layers=2, n_heads=3,
                                       layer_norm=True, n_inducing_points=5),
                ("x", "presence")),
}


@pytest.mark.parametrize("name", sorted(BLOCKS))
def test_set_transformer_modules_vs_golden(name):
    from torch_scae_amd import set_transformer as st
    blob, _ = load("op_set_transformer_blocks")
    c = sub(blob, name + "/")
    cls, kw, argn = BLOCKS[name]
    mod = getattr(st, cls)(**kw)
    mod.load_state_dict(sub(c, "param/"))
    mod = mod.cuda()
    ins = {}
    for a in argn:
        t = c["in/" + a]
        ins[a] = t.cuda() if a == "presence" else leaf(t)
    out = mod(*[ins[a] for a in argn])
    (out * c["w"].cuda()).sum().backward()
    assert_close(out, c["out"], 2e-5, 1e-4, "out")
    grads = dict(mod.named_parameters())
    for k, g in sub(c, "grad/").items():
        assert_close(grads[k].grad, g, 5e-5, 2e-4, "grad " + k)
    for k, g in sub(c, "gin/").items():
        assert_close(ins[k].grad, g, 5e-5, 2e-4, "gin " + k)


# --------------------------------------------------------------- K3 / K4 ---
@pytest.mark.parametrize("case", ["a", "b"])
def test_capsule_likelihood_vs_golden(case):
    from torch_scae_amd.object_decoder import CapsuleLikelihood
    blob, _ = load("op_capsule_likelihood")
    c = sub(blob, case + "/")
    i = {k: leaf(v) for k, v in sub(c, "in/").items()}
    res = CapsuleLikelihood(i["vote"], i["scale"], i["vote_presence"],
                            i["dummy_vote"])(i["x"], i["presence"])
    tot = res.log_prob * 1.7
    for k, w in sub(c, "w/").items():
        tot = tot + (res[k] * w.cuda()).sum()
    tot.backward()
    for k, ref in sub(c, "out/").items():
        assert_close(res[k], ref, ATOL, RTOL, "out " + k)
    for k, g in sub(c, "grad/").items():
        assert_close(i[k].grad, g, 5e-5, 2e-4, "grad " + k)


def test_capsule_likelihood_no_presence_vs_golden():
    from torch_scae_amd.object_decoder import CapsuleLikelihood
    blob, _ = load("op_capsule_likelihood")
    c = sub(blob, "nopres/")
    i = {k: v.cuda() for k, v in sub(c, "in/").items()}
    res = CapsuleLikelihood(i["vote"], i["scale"], i["vote_presence"],
                            i["dummy_vote"])(i["x"], None)
    for k, ref in sub(c, "out/").items():
        assert_close(res[k], ref, ATOL, RTOL, "out " + k)


@pytest.mark.parametrize("B,Oc,M", [
    (128, 24, 24),    # cfg-2
    (6, 72, 80),      # more than 64 object capsules: the two-pass statistics
    (3, 130, 100),    # the largest capsule product the factory admits
])
def test_capsule_likelihood_vs_oracle_cfg2(B, Oc, M):
    from torch_scae_amd.object_decoder import CapsuleLikelihood
    g = torch.Generator().manual_seed(77)
    ins = dict(vote=torch.randn(B, Oc, M, 6, generator=g),
               scale=torch.rand(B, Oc, M, generator=g) + 0.1,
               vote_presence=torch.rand(B, Oc, M, generator=g),
               dummy_vote=torch.randn(1, 1, M, 6, generator=g) * 0.1,
               x=torch.randn(B, M, 6, generator=g),
               presence=torch.rand(B, M, generator=g))
    ws = dict(posterior_mixing_prob=torch.randn(B, Oc, M, generator=g),
              soft_winner=torch.randn(B, M, 6, generator=g))
    ic = {k: v.clone().requires_grad_(True) for k, v in ins.items()}
    ro = O.capsule_likelihood(ic["vote"], ic["scale"], ic["vote_presence"],
                              ic["dummy_vote"], ic["x"], ic["presence"])
    (ro.log_prob + sum((ro[k] * w).sum() for k, w in ws.items())).backward()
    ig = {k: leaf(v) for k, v in ins.items()}
    rg = CapsuleLikelihood(ig["vote"], ig["scale"], ig["vote_presence"],
                           ig["dummy_vote"])(ig["x"], ig["presence"])
    (rg.log_prob + sum((rg[k] * w.cuda()).sum()
                       for k, w in ws.items())).backward()
    for k in ro:
        assert_close(rg[k], ro[k], 1e-4, 1e-4, "out " + k)
    for k in ins:
        ref = ic[k].grad
        assert_close(ig[k].grad, ref, 1e-4 * max(1, float(ref.abs().max())),
                     2e-4, "grad " + k)


CAPS_VARIANTS = {
    "default": dict(learn_vote_scale=True, allow_deformations=True,
                    noise_type="uniform", noise_scale=4.,
                    similarity_transform=False),
    "sim_nonoise": dict(learn_vote_scale=False, allow_deformations=False,
                        noise_type=None, noise_scale=0.,
                        similarity_transform=True),
}


@pytest.mark.parametrize("name", sorted(CAPS_VARIANTS))
def test_capsule_layer_and_decoder_vs_golden(name):
    from torch_scae_amd import nn_ext, nn_utils
    from torch_scae_amd.object_decoder import CapsuleLayer, CapsuleObjectDecoder
    blob, _ = load("op_capsule_layer")
    c = sub(blob, name + "/")
    layer = CapsuleLayer(n_caps=4, dim_feature=10, n_votes=5, dim_caps=6,
                         hidden_sizes=(7,), **CAPS_VARIANTS[name])
    layer.load_state_dict(sub(c, "layer_param/"))
    layer = layer.cuda()
    feat = leaf(c["in/feature"])
    noise = sub(c, "noise/")
    with nn_utils.fixed_noise([noise[k] for k in sorted(noise)]):
        res = layer(feat)
    tot = res.cpr_dynamic_reg_loss * 0.9
    for k, w in sub(c, "w/").items():
        tot = tot + (res[k] * w.cuda()).sum()
    tot.backward()
    for k, ref in sub(c, "out/").items():
        assert_close(res[k], ref, ATOL, RTOL, "out " + k)
    assert_close(feat.grad, c["grad/feature"], GATOL, 2e-4, "grad feature")
    grads = nn_ext.named_reference_grads(layer)
    for k, g in sub(c, "grad/").items():
        if k != "feature":
            assert_close(grads[k], g, GATOL, 2e-4, "grad " + k)

    dec = CapsuleObjectDecoder(CapsuleLayer(
        n_caps=4, dim_feature=10, n_votes=5, dim_caps=6, hidden_sizes=(7,),
        **CAPS_VARIANTS[name]))
    dec.load_state_dict(sub(c, "dec_param/"))
    dec = dec.cuda()
    dn = sub(c, "dec_noise/")
    with torch.no_grad(), nn_utils.fixed_noise([dn[k] for k in sorted(dn)]):
        r2 = dec(feat.detach(), c["dec_in/x"].cuda(),
                 c["dec_in/presence"].cuda())
    for k, ref in sub(c, "dec_out/").items():
        assert_close(r2[k], ref, ATOL, RTOL, "dec_out " + k)


@pytest.mark.parametrize("B,Oc,V", [(3, 5, 7), (16, 24, 24), (2, 66, 70)])
def test_mat3_mul_vs_matmul(B, Oc, V):
    """The 3 x 3 products of the hierarchical CapsuleLayer.forward
    (object_decoder.py:184-191) against torch.matmul in fp64, gradients of
    both factors."""
    from torch_scae_amd import ops
    g = torch.Generator().manual_seed(B + V)
    left = torch.randn(B, Oc, 1, 3, 3, generator=g)
    right = torch.randn(B, Oc, V, 3, 3, generator=g)
    w = torch.randn(B, Oc, V, 3, 3, generator=g)
    ld, rd = left.double().requires_grad_(), right.double().requires_grad_()
    ref = torch.matmul(ld.repeat(1, 1, V, 1, 1), rd)
    (ref * w.double()).sum().backward()
    lg, rg = leaf(left), leaf(right)
    out = ops.mat3_mul(lg, rg)
    (out * w.cuda()).sum().backward()
    assert_close(out, ref.float(), 1e-5, 1e-5, "out")
    assert_close(lg.grad, ld.grad.float(), 1e-4, 1e-5, "g_left")
    assert_close(rg.grad, rd.grad.float(), 1e-5, 1e-5, "g_right")
    # a constant parent transform: no gradient buffer is asked for
    out2 = ops.mat3_mul(left.cuda(), leaf(right))
    assert torch.equal(out2, out)


def test_capsule_layer_hierarchical_vs_golden():
    """CapsuleLayer.forward(feature, parent_transform, parent_presence)
    (object_decoder.py:184-187, :214-215) against reference-captured vectors."""
    from torch_scae_amd import nn_ext, nn_utils
    from torch_scae_amd.object_decoder import CapsuleLayer
    blob, _ = load("op_capsule_layer_hier")
    layer = CapsuleLayer(n_caps=4, dim_feature=10, n_votes=5, dim_caps=6,
                         hidden_sizes=(7,), **CAPS_VARIANTS["default"])
    layer.load_state_dict(sub(blob, "layer_param/"))
    layer = layer.cuda()
    ins = {k: leaf(blob["in/" + k])
           for k in ("feature", "parent_transform", "parent_presence")}
    noise = sub(blob, "noise/")
    with nn_utils.fixed_noise([noise[k] for k in sorted(noise)]):
        res = layer(ins["feature"], parent_transform=ins["parent_transform"],
                    parent_presence=ins["parent_presence"])
    tot = res.cpr_dynamic_reg_loss * 0.9
    for k, w in sub(blob, "w/").items():
        tot = tot + (res[k] * w.cuda()).sum()
    tot.backward()
    for k, ref in sub(blob, "out/").items():
        assert_close(res[k], ref, ATOL, RTOL, "out " + k)
    grads = nn_ext.named_reference_grads(layer)
    for k, g in sub(blob, "grad/").items():
        got = ins[k].grad if k in ins else grads[k]
        assert_close(got, g, GATOL, 2e-4, "grad " + k)
    # each argument on its own works too
    with nn_utils.fixed_noise([noise[k] for k in sorted(noise)]):
        r1 = layer(ins["feature"], parent_presence=ins["parent_presence"])
    assert_close(r1.vote_presence, blob["out/vote_presence"], ATOL, RTOL,
                 "presence only")


def test_capsule_layer_error_behaviour():
    from torch_scae_amd.object_decoder import CapsuleLayer, sparsity_loss
    x = torch.zeros(2, 3, 4, device="cuda")
    with pytest.raises(ValueError):
        CapsuleLayer(3, 4, 5, 6, noise_type="bogus").cuda()(x)
    with pytest.raises(NameError):      # reference crashes the same way
        CapsuleLayer(3, 4, 5, 6, caps_dropout_rate=0.5).cuda()(x)
    with pytest.raises(RuntimeError):   # so does its LogisticNormal noise
        CapsuleLayer(3, 4, 5, 6, noise_type="logistic", noise_scale=4.).cuda()(x)
    with pytest.raises(ValueError):
        sparsity_loss("nope", x[0])


# ------------------------------------------------------------------ K1 -----
def decoder_cases():
    _, meta = load("op_image_decoder")
    return sorted(k for k in meta if not k.startswith("tg_"))


def make_decoder(m, params):
    from torch_scae_amd.part_decoder import TemplateBasedImageDecoder
    dec = TemplateBasedImageDecoder(
        n_templates=m["M"], template_size=tuple(m["template_size"]),
        output_size=tuple(m["HW"]), learn_output_scale=m["learn_output_scale"],
        use_alpha_channel=m["use_alpha_channel"],
        background_value=m["background_value"])
    dec.load_state_dict(params)
    return dec.cuda()


@pytest.mark.parametrize("name", decoder_cases())
def test_image_decoder_vs_golden(name):
    blob, meta = load("op_image_decoder")
    c = sub(blob, name + "/")
    dec = make_decoder(meta[name], sub(c, "param/"))
    i = {k: leaf(v) for k, v in sub(c, "in/").items() if k not in ("x", "w")}
    x, w = c["in/x"].cuda(), c["in/w"].cuda()
    r = dec(i["templates"], i["pose"], i.get("presence"), i.get("bg_image"))
    assert_close(r.transformed_templates, c["out/transformed_templates"],
                 ATOL, RTOL, "tt")
    assert_close(r.mixing_logits, c["out/mixing_logits"], 2e-5, RTOL, "ml")

    # fused likelihood path
    lp = r.pdf.log_prob(x)
    (lp * w).sum().backward()
    assert_close(lp, c["out/log_prob"], 5e-5, 1e-4, "log_prob")
    for k, g in sub(c, "grad/").items():
        assert_close(i[k].grad, g, 5e-5, 2e-4, "grad " + k)
    pg = dict(dec.named_parameters())
    for k, g in sub(c, "pgrad/").items():
        assert_close(pg[k].grad, g, 1e-4, 2e-4, "pgrad " + k)

    # the generic (materialised) mixture gives the same numbers
    from torch_scae_amd.distributions import GaussianMixture
    generic = GaussianMixture.make_from_stats(
        r.transformed_templates.detach(), r.pdf.dist.scale.detach(),
        r.mixing_logits.detach())
    assert_close(generic.log_prob(x), c["out/log_prob"], 5e-5, 1e-4,
                 "generic log_prob")
    assert_close(r.pdf.mean(), c["out/mean"], 2e-5, 1e-4, "mean")
    assert_close(r.pdf.mode(), c["out/mode"], 2e-5, 1e-4, "mode")
    assert_close(r.pdf.mixing_log_prob(), c["out/mixing_log_prob"], 2e-5,
                 1e-4, "mixing_log_prob")
    if "out/mode_max" in c:
        assert_close(r.pdf.mode(maximum=True), c["out/mode_max"], 2e-5, 1e-4,
                     "mode_max")
    else:
        with pytest.raises(RuntimeError):
            r.pdf.mode(maximum=True)

    # gradients THROUGH the materialised tensors (autograd of the renderer)
    for p in dec.parameters():
        p.grad = None
    i2 = {k: leaf(v) for k, v in sub(c, "in/").items() if k not in ("x", "w")}
    bg = i2.get("bg_image")
    r2 = dec(i2["templates"], i2["pose"], i2.get("presence"),
             None if bg is None else bg.detach())
    ((r2.transformed_templates * c["mat/wt"].cuda()).sum()
     + (r2.mixing_logits * c["mat/wm"].cuda()).sum()).backward()
    for k, g in sub(c, "mat/grad/").items():
        assert_close(i2[k].grad, g, 1e-4, 2e-4, "mat grad " + k)
    for k, g in sub(c, "mat/pgrad/").items():
        assert_close(pg[k].grad, g, 2e-4, 2e-4, "mat pgrad " + k)

    # generic mixture backward == fused backward
    loc = r.transformed_templates.detach().clone().requires_grad_(True)
    ml = r.mixing_logits.detach().clone().requires_grad_(True)
    sig = r.pdf.dist.scale.detach().clone().requires_grad_(True)
    (GaussianMixture.make_from_stats(loc, sig, ml).log_prob(x) * w) \
        .sum().backward()
    cfg = dict(output_size=tuple(meta[name]["HW"]),
               learn_output_scale=meta[name]["learn_output_scale"],
               use_alpha_channel=meta[name]["use_alpha_channel"])
    locc = r.transformed_templates.detach().cpu().requires_grad_(True)
    mlc = r.mixing_logits.detach().cpu().requires_grad_(True)
    sigc = r.pdf.dist.scale.detach().cpu().requires_grad_(True)
    (O.gmm_log_prob(locc, sigc, mlc, x.cpu()) * w.cpu()).sum().backward()
    assert_close(loc.grad, locc.grad, 5e-5, 2e-4, "generic g_loc")
    assert_close(ml.grad, mlc.grad, 5e-5, 2e-4, "generic g_ml")
    assert_close(sig.grad, sigc.grad, 2e-4, 2e-4, "generic g_sigma")


@pytest.mark.parametrize("B,M,C,HW,ts,alpha,scale", [
    (128, 24, 1, (40, 40), (11, 11), True, False),    # cfg-2
    (16, 32, 3, (32, 32), (11, 11), True, False),     # cfg-5 shape
    (8, 48, 1, (40, 40), (11, 11), True, False),      # cfg-3 shape
    (4, 5, 3, (17, 23), (7, 9), False, True),         # ragged, temperature
    # the wave-form kernels (alpha mode) off the beaten sizes: odd image and
    # template sizes, 2 / 3 / 4 channels, learned output scale, H W not a
    # multiple of 4 (render falls back), a set of 70 components
    (4, 5, 3, (17, 23), (7, 9), True, True),
    (3, 7, 2, (20, 12), (5, 6), True, False),
    (2, 3, 4, (9, 16), (4, 3), True, True),
    (2, 70, 1, (24, 24), (11, 11), True, False),
])
def test_image_decoder_vs_oracle_full_size(B, M, C, HW, ts, alpha, scale):
    _decoder_vs_oracle(B, M, C, HW, ts, alpha, scale,
                       torch.Generator().manual_seed(5))


def _unit_scale_inputs(B, M, g):
    pose = torch.randn(B, M, 6, generator=g) * 0.5
    pose[:, :, 0] += 1.0
    pose[:, :, 4] += 1.0
    return pose, torch.rand(B, M, generator=g)


def _clear_of_cell_boundaries(pose, HW, ts, g, margin=8e-6):
    """d log_prob / d pose jumps where a pixel's sample position crosses a
    texel-cell boundary: a pixel within fp32 round-off of one has two valid
    one-sided derivatives (the golden poses are
    'irrational' for the same reason).  Round-off of a position is a few ulp
    of ~10 texels, ~1e-6; capsules with a pixel within 8e-6 get their
    translation nudged until none is left -- the regimes below draw thousands
    of (capsule, pixel) pairs, the unit-scale test above keeps its round-1
    draws."""
    H, W = HW
    th, tw = ts
    xs = (2 * torch.arange(W, dtype=torch.float64) + 1) / W - 1
    ys = (2 * torch.arange(H, dtype=torch.float64) + 1) / H - 1
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    gx, gy = gx.reshape(-1), gy.reshape(-1)
    pose = pose.clone()
    for _ in range(60):
        a = pose.double()[..., None]                       # (B, M, 6, 1)
        ix = ((a[:, :, 0] * gx + a[:, :, 1] * gy + a[:, :, 2] + 1) * tw - 1) / 2
        iy = ((a[:, :, 3] * gx + a[:, :, 4] * gy + a[:, :, 5] + 1) * th - 1) / 2
        near = lambda v, n: ((v - v.round()).abs() < margin) \
            & (v > -1.5) & (v < n + 0.5)                   # noqa: E731
        bad = (near(ix, tw) | near(iy, th)).any(-1)        # (B, M)
        if not bool(bad.any()):
            return pose
        nudge = torch.randn(pose.shape[0], pose.shape[1], 2, generator=g) * 2e-2
        pose[..., 2] += torch.where(bad, nudge[..., 0], torch.zeros(()))
        pose[..., 5] += torch.where(bad, nudge[..., 1], torch.zeros(()))
    raise AssertionError("poses could not be cleared of cell boundaries")


def _decoder_vs_oracle(B, M, C, HW, ts, alpha, scale, g, inputs=None,
                       tile_sums=False):
    """TemplateBasedImageDecoder + mixture log-likelihood and all gradients
    against the oracle.  ``inputs``: g -> (pose, presence), default unit-scale
    poses.  ``tile_sums``: through ``log_prob_tile_sums`` -- the training
    step's form, whose backward is the cell-gather kernel -- instead of the
    per-pixel map."""
    from torch_scae_amd.part_decoder import TemplateBasedImageDecoder
    torch.manual_seed(0)
    dec = TemplateBasedImageDecoder(M, ts, HW, learn_output_scale=scale,
                                    use_alpha_channel=alpha)
    with torch.no_grad():
        for p in dec.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    P = {"d." + k: v.clone().requires_grad_(True)
         for k, v in dec.state_dict().items()}
    templates = torch.rand(B, M, C, *ts, generator=g)
    if inputs is None:
        pose, presence = _unit_scale_inputs(B, M, g)
    else:
        pose, presence = inputs(g)
        pose = _clear_of_cell_boundaries(pose, HW, ts, g)
    x = torch.rand(B, C, *HW, generator=g)
    cfg = dict(output_size=HW, learn_output_scale=scale,
               use_alpha_channel=alpha)
    tc, pc, prc = (t.clone().requires_grad_(True)
                   for t in (templates, pose, presence))
    ro = O.image_decoder(P, "d", tc, pc, prc, None, cfg)
    lpo = O.gmm_log_prob(ro.transformed_templates, ro.scale, ro.mixing_logits,
                         x)
    rec_ll = lpo.flatten(1).sum(-1).mean()
    rec_ll.backward()

    dec = dec.cuda()
    tg, pg_, prg = leaf(templates), leaf(pose), leaf(presence)
    rg = dec(tg, pg_, prg)
    if tile_sums:
        sums = rg.pdf.log_prob_tile_sums(x.cuda())
        (sums.sum() / B).backward()
        assert_close(sums.sum(1), lpo.flatten(1).sum(1), 1e-4 * float(
            lpo.flatten(1).sum(1).abs().max()), 1e-4, "image totals")
    else:
        lpg = rg.pdf.log_prob(x.cuda())
        lpg.flatten(1).sum(-1).mean().backward()
        assert_close(lpg, lpo, 1e-4, 1e-4, "log_prob")
    assert_close(rg.transformed_templates, ro.transformed_templates, 1e-5,
                 1e-4, "tt")
    assert_close(rg.mixing_logits, ro.mixing_logits, 2e-5, 1e-4, "ml")
    for name, a, b in (("templates", tg.grad, tc.grad),
                       ("pose", pg_.grad, pc.grad),
                       ("presence", prg.grad, prc.grad)):
        assert_close(a, b, 1e-4 * max(1.0, float(b.abs().max())), 2e-4,
                     "grad " + name)
    for k, p in dec.named_parameters():
        ref = P["d." + k].grad
        if ref is None:          # parameter unused in this mode
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, k
            continue
        assert_close(p.grad, ref, 1e-4 * max(1.0, float(ref.abs().max())),
                     5e-4, "pgrad " + k)


def _regime_inputs(regime, B, M, C, HW, g):
    """(pose, presence) of a pose regime a training run passes through.

    init:      what a freshly initialised part encoder emits for U[0,1)
               images (the factory's model of that shape, eval mode);
    collapsed: the state ``bench.py``'s U[0,1) batches train into within
               ~200 steps -- pose scale ~0.01,
               so that the whole image falls into one or two texel cells of a
               template (the K1 backward then splits a cell over row slices x
               row segments), presences 1e-18 and exactly 0;
    mixed:     half of the capsules collapsed, half at unit scale, a few with
               a single collapsed axis (one texel column / row)."""
    if regime == "init":
        from torch_scae_amd import factory
        torch.manual_seed(3)
        np.random.seed(3)
        model = factory.make_scae(dict(
            image_shape=(C, *HW), n_classes=10, n_part_caps=M, n_obj_caps=M,
            scae_params=dict(reconstruct_alternatives=False))).cuda().eval()
        with torch.no_grad():
            parts = model.part_encoder(torch.rand(B, C, *HW, generator=g).cuda())
        return parts.pose.cpu().clone(), parts.presence.cpu().clone()
    unit, _ = _unit_scale_inputs(B, M, g)
    small = torch.randn(B, M, 6, generator=g) * 0.003
    small[:, :, 0] += 0.01
    small[:, :, 4] += 0.01
    # translations anywhere in (and a little beyond) the template
    small[:, :, 2] = torch.rand(B, M, generator=g) * 2.4 - 1.2
    small[:, :, 5] = torch.rand(B, M, generator=g) * 2.4 - 1.2
    presence = torch.rand(B, M, generator=g)
    if regime == "collapsed":
        pose = small
        presence = torch.where(presence < 0.5, torch.full_like(presence, 1e-18),
                               torch.zeros_like(presence))
        presence[:, 0] = torch.rand(B, generator=g)      # one live capsule
        return pose, presence
    assert regime == "mixed"
    pick = torch.rand(B, M, generator=g)
    pose = torch.where((pick < 0.5)[..., None], small, unit)
    one_axis = (pick >= 0.5) & (pick < 0.65)
    pose[..., 0] = torch.where(one_axis, small[..., 0], pose[..., 0])
    pose[..., 1] = torch.where(one_axis, small[..., 1], pose[..., 1])
    presence = torch.where(pick < 0.25, torch.full_like(presence, 1e-18),
                           presence)
    return pose, presence


@pytest.mark.parametrize("tile_sums", [False, True], ids=["pixels", "sums"])
@pytest.mark.parametrize("regime", ["init", "collapsed", "mixed"])
@pytest.mark.parametrize("B,M,C,HW", [(128, 24, 1, (40, 40)),     # cfg-2
                                      (32, 32, 3, (32, 32))])     # cfg-5 shape
def test_image_decoder_vs_oracle_pose_regimes(B, M, C, HW, regime, tile_sums):
    """The pose regimes of a training run (round-3 review: every full-size
    decoder test drew unit-scale poses, while the timed step runs on
    collapsed ones)."""
    _decoder_vs_oracle(B, M, C, HW, (11, 11), True, False,
                       torch.Generator().manual_seed(17),
                       inputs=lambda g: _regime_inputs(regime, B, M, C, HW, g),
                       tile_sums=tile_sums)


@pytest.mark.parametrize("B,M,C,HW,ts,alpha", [
    (128, 24, 1, (40, 40), (11, 11), True),
    (4, 5, 3, (17, 23), (7, 9), False),
])
def test_log_prob_tile_sums_match_per_pixel_path(B, M, C, HW, ts, alpha):
    """K1's tile-sum variant (training loss) against its per-pixel variant:
    same total, and with non-uniform tile weights the same gradients."""
    from torch_scae_amd.part_decoder import TemplateBasedImageDecoder
    torch.manual_seed(1)
    g = torch.Generator().manual_seed(9)
    dec = TemplateBasedImageDecoder(M, ts, HW, use_alpha_channel=alpha).cuda()
    templates = torch.rand(B, M, C, *ts, generator=g)
    pose = torch.randn(B, M, 6, generator=g) * 0.5
    pose[:, :, 0] += 1.0
    pose[:, :, 4] += 1.0
    presence = torch.rand(B, M, generator=g)
    x = torch.rand(B, C, *HW, generator=g).cuda()

    def run(sums):
        ins = [leaf(t) for t in (templates, pose, presence)]
        for p in dec.parameters():
            p.grad = None
        pdf = dec(*ins).pdf
        if sums:
            out = pdf.log_prob_tile_sums(x)               # (B, tiles)
            tiles = out.shape[1]
            w = torch.linspace(0.5, 1.5, B * tiles, device="cuda").view(B, tiles)
            (out * w).sum().backward()
            return out, w, [t.grad for t in ins] + [p.grad for p in dec.parameters()]
        return pdf.log_prob(x), None, ins

    sums, w, g_sums = run(True)
    lp, _, ins = run(False)
    HWn = HW[0] * HW[1]
    tiles = sums.shape[1]
    assert_close(sums.sum(1), lp.flatten(1).sum(1), 1e-3, 1e-5, "image totals")
    # recover the tiling from the ABI and replay the same weights per pixel
    per_tile = None
    for cand in range(1, HWn + 1):
        if -(-HWn // cand) == tiles:
            t0 = lp.flatten(2)[:, :, :cand].sum((1, 2))
            if torch.allclose(t0, sums[:, 0], rtol=1e-4, atol=1e-3):
                per_tile = cand
                break
    assert per_tile is not None, "tile size not recoverable"
    wpix = w.repeat_interleave(per_tile, 1)[:, :HWn].view(B, 1, *HW)
    for p in dec.parameters():
        p.grad = None
    (lp * wpix).sum().backward()
    g_pix = [t.grad for t in ins] + [p.grad for p in dec.parameters()]
    for a, b in zip(g_sums, g_pix):
        if b is None:
            assert a is None
            continue
        assert_close(a, b, 1e-5 * max(1.0, float(b.abs().max())), 1e-4,
                     "grad via tile sums")


def test_image_decoder_error_behaviour():
    from torch_scae_amd.part_decoder import TemplateBasedImageDecoder
    dec = TemplateBasedImageDecoder(2, (5, 5), (8, 8), use_alpha_channel=True,
                                    background_value=False).cuda()
    t = torch.rand(1, 2, 1, 5, 5, device="cuda")
    p = torch.rand(1, 2, 6, device="cuda")
    with pytest.raises(AttributeError):      # like the reference, :192
        dec(t, p)


@pytest.mark.parametrize("name", ["tg_default", "tg_relu1_rgb", "tg_nocolor"])
def test_template_generator_vs_golden(name):
    from torch_scae_amd.part_decoder import TemplateGenerator
    blob, meta = load("op_image_decoder")
    c = sub(blob, name + "/")
    kw = dict(meta[name])
    kw["template_size"] = tuple(kw["template_size"])
    tg = TemplateGenerator(**kw)
    tg.load_state_dict(sub(c, "param/"))
    tg = tg.cuda()
    feat = c.get("in/feature")
    if feat is not None:
        feat = leaf(feat)
    r = tg(feature=feat, batch_size=2)
    assert_close(r.templates, c["out/templates"], ATOL, RTOL, "templates")
    assert_close(r.raw_templates, c["out/raw_templates"], ATOL, RTOL, "raw")


@pytest.mark.parametrize("name,shape,sim,train", [
    ("affine_train", (1, 16, 16), False, True),
    ("similarity_eval", (3, 14, 14), True, False)])
def test_part_encoder_vs_golden(name, shape, sim, train):
    from torch_scae_amd import nn_utils
    from torch_scae_amd.part_encoder import CNNEncoder, PCAE
    blob, _ = load("op_part_encoder")
    c = sub(blob, name + "/")
    cnn = CNNEncoder(input_shape=shape, out_channels=[6, 6],
                     kernel_sizes=[3, 3], strides=[2, 1])
    enc = PCAE(input_shape=shape, encoder=cnn, n_caps=3, n_poses=6,
               n_special_features=4, similarity_transform=sim)
    enc.load_state_dict(sub(c, "param/"))
    enc = enc.cuda().train(train)
    noise = sub(c, "noise/")
    with torch.no_grad(), nn_utils.fixed_noise([noise[k]
                                                for k in sorted(noise)]):
        r = enc(c["in/image"].cuda())
    for k in ("pose", "presence", "feature"):
        assert_close(r[k], c["out/" + k], 2e-5, 1e-4, k)


# ----------------------------------------------------------- K2b fused trunk
@pytest.mark.parametrize("B,N,widths,D,Dout,L,ln,pres", [
    (128, 24, (7, 16, 121), 16, 256, 3, True, "mixed"),   # cfg-2
    (9, 32, (7, 16, 363), 16, 256, 3, True, "rand"),      # cfg-5 shape
    (5, 64, (11,), 16, 70, 2, False, None),               # max set, no LN
    (4, 64, (11,), 32, 70, 2, False, None),               # over the LDS budget -> unfused path
    (3, 1, (5, 3), 8, 9, 1, True, "ones"),                # degenerate
    (300, 10, (13,), 16, 24, 1, True, "mixed"),           # B > grid: block loop
])
def test_fused_set_encoder_vs_oracle(B, N, widths, D, Dout, L, ln, pres):
    from torch_scae_amd.set_transformer import SetTransformer
    torch.manual_seed(B + N)
    Din = sum(widths)
    st = SetTransformer(dim_in=Din, dim_hidden=D, dim_out=Dout, n_outputs=4,
                        n_layers=L, n_heads=1, layer_norm=ln)
    g = torch.Generator().manual_seed(17)
    with torch.no_grad():
        for p in st.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 0.1)
    P = {"m." + k: v.clone().requires_grad_(True)
         for k, v in st.state_dict().items()}
    # segments as strided views of one wider buffer (like the part encoder's
    # split outputs)
    wide = torch.randn(B, N, Din + 5, generator=g)
    p = None
    if pres == "rand":
        p = torch.rand(B, N, generator=g)
    elif pres == "ones":
        p = torch.ones(B, N)
    elif pres == "mixed":
        p = torch.ones(B, N)
        p[:, ::3] = torch.rand(B, len(range(0, N, 3)), generator=g)
    w = torch.randn(B, N, Dout, generator=g)

    xc = wide[..., 2:2 + Din].clone().requires_grad_(True)
    h = O._linear(P, "m.fc1", xc)
    for l in range(L):
        h = O.sab(P, f"m.sabs.{l}", h, p, 1, ln)
    zo = O._linear(P, "m.fc2", h)
    (zo * w).sum().backward()

    st = st.cuda()
    wide_g = wide.cuda()
    segs, col = [], 2
    for wd in widths:
        segs.append(wide_g[..., col:col + wd].detach().requires_grad_(True))
        col += wd
    zg = st.encode_segments(segs, dev(p))
    (zg * w.cuda()).sum().backward()
    assert_close(zg, zo, 1e-4, 1e-4, "z")
    gx = torch.cat([s.grad for s in segs], -1)
    assert_close(gx, xc.grad, 1e-4 * max(1.0, float(xc.grad.abs().max())),
                 2e-4, "gx")
    sd_grads = {k: q.grad for k, q in st.named_parameters()}
    for k, q in P.items():
        name = k[2:]
        if q.grad is None:
            continue
        ref = q.grad
        assert_close(sd_grads[name], ref,
                     2e-4 * max(1.0, float(ref.abs().max())), 5e-4,
                     "grad " + name)


@pytest.mark.parametrize("B,N,widths,L,ln,pres", [
    (128, 24, (6, 1, 16, 121), 3, True, "mixed"),   # cfg-2: two waves per set
    (40, 32, (7, 16, 363), 3, True, "rand"),        # cfg-5 shape (Din = 386)
    (700, 24, (6, 1, 16, 40), 2, True, "rand"),     # B > grid: partial rows accumulate
    (33, 10, (13,), 1, True, "mixed"),              # one wave per set
    (17, 17, (5, 30), 2, False, None),              # ragged second tile, no LayerNorm
    (6, 16, (16,), 3, True, "ones"),                # exactly one tile
    (3, 1, (5, 3), 1, True, "rand"),                # a single element
    (70, 48, (6, 1, 16, 121), 3, True, "mixed"),    # configs[2]: three tiles / waves
    (9, 40, (6, 1, 16, 121), 3, True, "rand"),      # the 40 / 32 default (ragged third tile)
    (5, 64, (11,), 2, False, None),                 # four full tiles
    (4, 57, (23, 9), 1, True, "rand"),              # ragged fourth tile
])
def test_trunk_on_matrix_cores_vs_oracle(B, N, widths, L, ln, pres):
    """K2b without fc2 (the form SetTransformer.forward_segments uses): for
    D = 16 and N <= 64 this is the wave-per-tile MFMA implementation
    (set_encoder_wave.hip) -- trunk output, segment gradients and every
    parameter gradient against the oracle."""
    from torch_scae_amd import ops
    from torch_scae_amd.set_transformer import SetTransformer
    D = 16
    torch.manual_seed(B + N)
    Din = sum(widths)
    st = SetTransformer(dim_in=Din, dim_hidden=D, dim_out=32, n_outputs=4,
                        n_layers=L, n_heads=1, layer_norm=ln)
    g = torch.Generator().manual_seed(23)
    with torch.no_grad():
        for q in st.parameters():
            q.add_(torch.randn(q.shape, generator=g) * 0.1)
    P = {"m." + k: v.clone().requires_grad_(True)
         for k, v in st.state_dict().items()}
    wide = torch.randn(B, N, Din + 3, generator=g)
    p = None
    if pres == "rand":
        p = torch.rand(B, N, generator=g)
    elif pres == "ones":
        p = torch.ones(B, N)
    elif pres == "mixed":
        p = torch.ones(B, N)
        p[:, ::3] = torch.rand(B, len(range(0, N, 3)), generator=g)
    w = torch.randn(B, N, D, generator=g)
    xc = wide[..., 1:1 + Din].clone().requires_grad_(True)
    h = O._linear(P, "m.fc1", xc)
    for l in range(L):
        h = O.sab(P, f"m.sabs.{l}", h, p, 1, ln)
    (h * w).sum().backward()

    st = st.cuda()
    wide_g = wide.cuda()
    segs, col = [], 1
    for wd in widths:
        segs.append(wide_g[..., col:col + wd].detach().requires_grad_(True))
        col += wd
    hg = ops.set_encoder(segs, dev(p), st._packed_trunk(with_fc2=False), D, 0,
                         L, ln)
    (hg * w.cuda()).sum().backward()
    assert_close(hg, h, 1e-4, 1e-4, "trunk output")
    gx = torch.cat([s_.grad for s_ in segs], -1)
    assert_close(gx, xc.grad, 1e-4 * max(1.0, float(xc.grad.abs().max())),
                 1e-4, "segment gradients")
    grads = {k: q.grad for k, q in st.named_parameters()}
    n = 0
    for k, q in P.items():
        if q.grad is None:
            continue
        ref = q.grad
        assert_close(grads[k[2:]], ref, 1e-4 * max(1.0, float(ref.abs().max())),
                     1e-4, "grad " + k[2:])
        n += 1
    assert n == 2 + L * (10 + (4 if ln else 0))


@pytest.mark.parametrize("B,N,widths,L,ln,pres", [
    (128, 24, (6, 1, 16, 121), 3, True, "mixed"),   # cfg-2
    (70, 48, (6, 1, 16, 121), 3, True, "mixed"),    # configs[2]: 48 part capsules
    (9, 40, (6, 1, 16, 121), 3, True, "rand"),      # ragged third tile
    (5, 64, (11,), 2, False, None),                 # four full tiles, no LayerNorm
])
def test_trunk_bf16_attention_vs_fp32_oracle(B, N, widths, L, ln, pres):
    """BASELINE.json configs[2] ("bf16 ... MFMA attention path") inside the fused
    trunk: under ``ops.mfma_bf16()`` the attention products of every block --
    Q K^T, P V and the four products of their backward -- take bf16 operands on
    v_mfma_f32_16x16x16_bf16 (fp32 accumulate); everything else stays fp32.
    Against the fp32 oracle at bf16's bar, written here: 2^-7 of the largest
    entry on the output; 5e-2 relative L2 on the segment gradients, 1e-1 on
    every parameter gradient (an operand carries 8 significant bits; the
    errors of L chained blocks and of the softmax backward's cancellation add
    up, and a batch of 9 sets averages little of it away: 7e-2 measured on
    one feed-forward weight), plus 2e-3 of the largest parameter-gradient norm for the gradients
    that are zero by symmetry (the key bias: the softmax ignores a shift of
    all keys).  The fp32 entry on the same inputs stays at 1e-4, so the
    difference is the operand rounding and nothing else."""
    from torch_scae_amd import ops
    from torch_scae_amd.set_transformer import SetTransformer
    D = 16
    torch.manual_seed(B + N)
    Din = sum(widths)
    st = SetTransformer(dim_in=Din, dim_hidden=D, dim_out=32, n_outputs=4,
                        n_layers=L, n_heads=1, layer_norm=ln)
    g = torch.Generator().manual_seed(23)
    with torch.no_grad():
        for q in st.parameters():
            q.add_(torch.randn(q.shape, generator=g) * 0.1)
    P = {"m." + k: v.clone().requires_grad_(True)
         for k, v in st.state_dict().items()}
    wide = torch.randn(B, N, Din + 3, generator=g)
    p = None
    if pres == "rand":
        p = torch.rand(B, N, generator=g)
    elif pres == "mixed":
        p = torch.ones(B, N)
        p[:, ::3] = torch.rand(B, len(range(0, N, 3)), generator=g)
    w = torch.randn(B, N, D, generator=g)
    xc = wide[..., 1:1 + Din].clone().requires_grad_(True)
    h = O._linear(P, "m.fc1", xc)
    for l in range(L):
        h = O.sab(P, f"m.sabs.{l}", h, p, 1, ln)
    (h * w).sum().backward()

    st = st.cuda()
    wide_g = wide.cuda()
    calls = []
    real = ops._lib.call

    def spy(name, *a):
        calls.append(name)
        return real(name, *a)

    def run(bf16):
        segs, col = [], 1
        for wd in widths:
            segs.append(wide_g[..., col:col + wd].detach().requires_grad_(True))
            col += wd
        st.zero_grad(set_to_none=True)
        with ops.mfma_bf16(bf16):
            hg = ops.set_encoder(segs, dev(p), st._packed_trunk(with_fc2=False),
                                 D, 0, L, ln)
            (hg * w.cuda()).sum().backward()
        return hg, torch.cat([s_.grad for s_ in segs], -1), \
            {k: q.grad.clone() for k, q in st.named_parameters()
             if q.grad is not None}

    ops._lib.call = spy
    try:
        hb, gxb, gb = run(True)
    finally:
        ops._lib.call = real
    assert calls[:2] == ["scae_set_encoder_fwd_bf16",
                         "scae_set_encoder_bwd_bf16"], calls
    hf, gxf, gf = run(False)
    assert_close(hf, h, 1e-4, 1e-4, "fp32 trunk output")
    tol = 2.0 ** -7
    assert not torch.equal(hb, hf)          # the bf16 kernels really ran
    assert_close(hb, h, tol * float(h.abs().max()), 0, "bf16 trunk output")
    rel = float((gxb.cpu() - xc.grad).norm() / xc.grad.norm())
    assert rel <= 5e-2, ("bf16 segment gradients", rel)
    refs = {k: q.grad for k, q in P.items()
            if q.grad is not None and k[2:] in gb}
    floor = 2e-3 * max(float(v.norm()) for v in refs.values())
    worst = sorted(((float((gb[k[2:]].cpu() - v).norm())
                     / (1e-1 * float(v.norm()) + floor), k)
                    for k, v in refs.items()), reverse=True)
    assert len(worst) >= 2 + L * 10 and worst[0][0] <= 1.0, worst[:5]


@pytest.mark.parametrize("B,N,O,C,pres", [
    (128, 24, 24, 256, "mixed"),    # cfg-2
    (600, 24, 24, 64, "rand"),      # B > grid: partial rows accumulate
    (5, 32, 32, 128, "rand"),       # full tiles
    (7, 17, 9, 64, None),           # ragged keys, one query tile
    (3, 1, 1, 64, "ones"),
    (4, 40, 24, 256, "rand"),       # three key tiles (the 40 / 32 default)
    (130, 48, 64, 256, "mixed"),    # configs[2]: 48 keys, 64 queries
    (3, 64, 33, 64, None),
])
def test_seed_attention_vs_fp64(B, N, O, C, pres):
    """K2c, ops.seed_attention(h, q, wk, bk, wv, bv, presence) =
    softmax((q K'^T - (1 - presence) 1e32) / sqrt(C)) V' with K' = h wk^T + bk,
    V' = h wv^T + bv (set_transformer.py:24-47 after the folding of
    seed_attention.hip): output and all gradients against the same formula
    in fp64 (seed_attention_wave.hip for N, O <= 64)."""
    from torch_scae_amd import ops
    D = 16
    g = torch.Generator().manual_seed(B * 7 + N)
    h = torch.randn(B, N, D, generator=g)
    q = torch.randn(O, C, generator=g) * 0.3
    wk, wv = torch.randn(C, D, generator=g) * 0.3, torch.randn(C, D, generator=g) * 0.3
    bk, bv = torch.randn(C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.3
    p = None
    if pres == "rand":
        p = torch.rand(B, N, generator=g)
    elif pres == "ones":
        p = torch.ones(B, N)
    elif pres == "mixed":
        p = torch.ones(B, N)
        p[:, ::3] = torch.rand(B, len(range(0, N, 3)), generator=g)
    w = torch.randn(B, O, C, generator=g)

    ins = [t.double().requires_grad_() for t in (h, q, wk, bk, wv, bv)]
    hd, qd, wkd, bkd, wvd, bvd = ins
    K = hd @ wkd.T + bkd
    V = hd @ wvd.T + bvd
    R = qd @ K.transpose(1, 2)
    if p is not None:
        # the reference's fp32 mask arithmetic absorbs R for every key with
        # presence < 1; in fp64 the same effect needs the mask to dominate
        R = R - (1. - p.double())[:, None, :] * 1e300
    ref = torch.softmax(R / (C ** 0.5), -1) @ V
    (ref * w.double()).sum().backward()

    outs = [leaf(t) for t in (h, q, wk, bk, wv, bv)]
    out = ops.seed_attention(*outs, dev(p))
    (out * w.cuda()).sum().backward()
    assert_close(out, ref.float(), 1e-4, 1e-4, "out")
    for name, got, want in zip(("gh", "gq", "gwk", "gbk", "gwv", "gbv"), outs, ins):
        scale = max(1.0, float(want.grad.abs().max()))
        assert_close(got.grad, want.grad.float(), 1e-4 * scale, 1e-4, name)


@pytest.mark.parametrize("B,N,O,C,pres", [
    (128, 24, 24, 256, "mixed"),    # cfg-2
    (130, 48, 64, 256, "mixed"),    # configs[2]: 48 keys, 64 queries
    (4, 40, 24, 256, "rand"),
])
def test_seed_attention_bf16_vs_fp32(B, N, O, C, pres):
    """configs[2]'s precision in the output attention: inside ``ops.mfma_bf16()``
    the logits, P h and the three products of their backward take bf16
    operands (scae_seed_attention_mfma_fwd/bwd_bf16).  Against the fp32 kernels
    on the same inputs (themselves held to 1e-4 of fp64 above) at bf16's bar:
    2^-7 of the largest entry on the output, 5e-2 relative L2 on every
    gradient (plus 2e-3 of the largest gradient norm for those that are zero
    by the softmax's shift symmetry)."""
    from torch_scae_amd import ops
    D = 16
    g = torch.Generator().manual_seed(B * 7 + N)
    h = torch.randn(B, N, D, generator=g)
    q = torch.randn(O, C, generator=g) * 0.3
    wk, wv = torch.randn(C, D, generator=g) * 0.3, torch.randn(C, D, generator=g) * 0.3
    bk, bv = torch.randn(C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.3
    p = torch.rand(B, N, generator=g)
    if pres == "mixed":
        p = torch.ones(B, N)
        p[:, ::3] = torch.rand(B, len(range(0, N, 3)), generator=g)
    w = torch.randn(B, O, C, generator=g).cuda()
    calls = []
    real = ops._lib.call

    def spy(name, *a):
        calls.append(name)
        return real(name, *a)

    def run(bf16):
        ins = [leaf(t) for t in (h, q, wk, bk, wv, bv)]
        with ops.mfma_bf16(bf16):
            out = ops.seed_attention(*ins, dev(p))
            (out * w).sum().backward()
        return out.detach(), [t.grad for t in ins]

    ops._lib.call = spy
    try:
        ob, gb = run(True)
    finally:
        ops._lib.call = real
    assert calls[:2] == ["scae_seed_attention_mfma_fwd_bf16",
                         "scae_seed_attention_mfma_bwd_bf16"], calls
    of, gf = run(False)
    assert not torch.equal(ob, of)
    assert float((ob - of).abs().max()) <= 2.0 ** -7 * float(of.abs().max())
    floor = 2e-3 * max(float(v.norm()) for v in gf)
    for name, a, b in zip(("gh", "gq", "gwk", "gbk", "gwv", "gbv"), gb, gf):
        assert float((a - b).norm()) <= 5e-2 * float(b.norm()) + floor, \
            (name, float((a - b).norm()), float(b.norm()))


# ------------------------------------------------------------- K6 loss tail
@pytest.mark.parametrize("prior,post,use_label,const", [
    ("l2", "entropy", True, None),       # the default SCAE config
    ("entropy", "l2", True, 1.5),
    ("kl", "kl", False, None),
    ("l2", "l2", True, None),
])
def test_loss_tail_vs_oracle(prior, post, use_label, const):
    from torch_scae_amd import ops
    g = torch.Generator().manual_seed(3)
    B, Oc, M, ncls = 128, 24, 24, 10
    lpp = torch.randn(B, M, generator=g)
    post_full = torch.softmax(torch.randn(B, Oc + 1, M, generator=g), 1)
    cp = torch.rand(B, Oc, generator=g)
    cp[0, 0] = 0.0
    W = torch.randn(ncls, Oc, generator=g) * 0.3
    bb = torch.randn(ncls, generator=g) * 0.1
    label = torch.randint(0, ncls, (B,), generator=g)
    weights = [1.0, 2.0, 0.35, 0.7, 0.2]

    def ref(lpp, post_full, cp, W, bb):
        log_prob = lpp.sum() / B
        pw, pb = O.sparsity_loss(prior, cp, n_classes=ncls,
                                 within_example_constant=const)
        mass = post_full[:, :-1].sum(-1)
        qw, qb = O.sparsity_loss(post, mass / M, n_classes=ncls)
        tot = -weights[0] * log_prob + weights[1] * pw + weights[2] * pb \
            + weights[3] * qw + weights[4] * qb
        xe1 = xe2 = torch.zeros(())
        if use_label:
            import torch.nn.functional as F
            p1 = torch.softmax(F.linear(cp.detach(), W, bb), -1)
            p2 = torch.softmax(F.linear(mass.detach(), W, bb), -1)
            xe1, xe2 = F.cross_entropy(p1, label), F.cross_entropy(p2, label)
            tot = tot + xe1 + xe2
        return torch.stack([tot, log_prob, pw, pb, qw, qb, xe1, xe2])

    ins_c = [t.clone().requires_grad_(True) for t in (lpp, post_full, cp, W, bb)]
    out_c = ref(*ins_c)
    out_c[0].backward()
    ins_g = [leaf(t) for t in (lpp, post_full, cp, W, bb)]
    out_g = ops.loss_tail(ins_g[0], ins_g[1], ins_g[2], ins_g[3], ins_g[4],
                          label.cuda() if use_label else None, ncls, prior,
                          post, True, weights, const)
    out_g[0].backward()
    assert_close(out_g[:8], out_c, 1e-4, 1e-4, "tail outputs")
    assert float(out_g[10]) == -float(out_g[1]) and float(out_g[8]) == 0.0
    names = ("lpp", "posterior", "caps_presence", "cls_w", "cls_b")
    for nme, a, b in zip(names, ins_g, ins_c):
        if b.grad is None:
            assert a.grad is None or float(a.grad.abs().sum()) == 0.0, nme
            continue
        assert_close(a.grad, b.grad,
                     1e-4 * max(1.0, float(b.grad.abs().max())), 2e-4,
                     "grad " + nme)


    # the remaining scalar terms of the training loss folded in: reconstruction
    # tile sums (rec_ll = sum / B) and the dynamic regulariser
    rec = torch.randn(B, 7)
    reg = torch.rand(1)
    ins_e = [leaf(t) for t in (lpp, post_full, cp, W, bb, rec, reg)]
    out_e = ops.loss_tail(*ins_e[:5], label.cuda() if use_label else None, ncls,
                          prior, post, True, weights, const, rec_sums=ins_e[5],
                          reg=ins_e[6], w_reg=0.37)
    out_e[0].backward()
    rec_ll = float(rec.double().sum() / B)
    want = float(out_c[0]) - rec_ll + 0.37 * float(reg)
    assert abs(float(out_e[0]) - want) <= 1e-4 * max(1.0, abs(want))
    assert abs(float(out_e[8]) - rec_ll) <= 1e-5 * max(1.0, abs(rec_ll))
    assert float(out_e[9]) == -float(out_e[8])
    assert_close(ins_e[5].grad, torch.full((B, 7), -1.0 / B), 1e-7, 1e-6, "g_rec")
    assert_close(ins_e[6].grad, torch.tensor([0.37]), 1e-7, 1e-6, "g_reg")
    assert_close(ins_e[0].grad, ins_g[0].grad, 1e-7, 1e-6, "g_lpp with extras")


# --------------------------------------------------------------------------
# K8 CNN encoder convolutions (part_encoder.py:26-44) vs torch's conv2d in fp64
# --------------------------------------------------------------------------
@pytest.mark.parametrize("B,C0,HW,chans,strides", [
    (5, 1, 40, [128, 128, 128, 128], [2, 2, 1, 1]),    # cfg-2 MNIST encoder
    (3, 3, 32, [64, 128, 64], [1, 2, 1]),
    (2, 3, 21, [64, 64], [2, 2]),                      # odd sizes, ragged tiles
    (130, 1, 17, [64, 64, 64], [1, 1, 2]),
    (64, 1, 40, [64, 64], [1, 1]),         # enough pixels for the 64x64 tiles
    (48, 2, 40, [64, 64], [1, 2]),         # 64x64 tiles, strided data gradient
    (3, 1, 12, [384, 384], [1, 1]),        # 64x64 weight-gradient tiles
    (128, 1, 40, [128, 128, 128], [2, 2, 1]),   # cfg-2 sizes: 32x64 split-K tiles
    (4, 4, 11, [64], [1]),                 # image layer only, 4 input channels
    (2, 2, 30, [128, 64, 64, 64, 64], [2, 1, 1, 1, 1]),   # five layers
])
def test_conv_stack_vs_conv2d(B, C0, HW, chans, strides):
    import torch.nn.functional as F
    from torch_scae_amd import ops
    assert ops.conv_stack_supported(C0, chans, [3] * len(chans), strides)
    from tests.gate_screen import conv_margins
    g = torch.Generator().manual_seed(B * 100 + HW)
    ws, bs, cin = [], [], C0
    for c in chans:
        bound = 1.0 / (cin * 9) ** 0.5
        ws.append((torch.rand(c, cin, 3, 3, generator=g) * 2 - 1) * bound)
        bs.append((torch.rand(c, generator=g) * 2 - 1) * bound)
        cin = c
    # a pre-activation within fp32 round-off of zero has two valid ReLU gates;
    # the batch is drawn from candidates whose every pre-activation (fp64) is
    # at least 4e-6 x the layer's largest away from zero (tests/gate_screen.py)
    kept = []
    for _ in range(60):
        cand = torch.rand(B, C0, HW, HW, generator=g)
        kept.append(cand[conv_margins(cand, ws, bs, strides) >= 4e-6])
        if sum(k.shape[0] for k in kept) >= B:
            break
    image = torch.cat(kept)[:B]
    assert image.shape[0] == B

    def run(dev, dt):
        w = [t.clone().to(dev, dt).requires_grad_() for t in ws]
        b = [t.clone().to(dev, dt).requires_grad_() for t in bs]
        x = image.to(dev, dt)
        if dev == "cpu":
            y = x
            for wi, bi, s in zip(w, b, strides):
                y = F.relu(F.conv2d(y, wi, bi, stride=s))
        else:
            y = ops.conv_stack(x, w, b, strides)
        return y, w, b

    y_ref, w_ref, b_ref = run("cpu", torch.float64)
    y, w, b = run("cuda", torch.float32)
    assert y.shape == y_ref.shape
    assert_close(y, y_ref.float(), rtol=1e-4, atol=2e-5, what="conv out")
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy.double())
    y.backward(gy.cuda())
    for l in range(len(chans)):
        for name, got, ref in (("dW", w[l].grad, w_ref[l].grad),
                               ("db", b[l].grad, b_ref[l].grad)):
            # every entry within 1e-4 of the tensor's largest entry (fp64 ref)
            scale = float(ref.abs().max())
            assert_close(got, ref.float(), rtol=0.0, atol=1e-4 * scale,
                         what=f"{name}{l}")


@pytest.mark.parametrize("B,IH,IW,Ci,Co,stride", [
    (2, 9, 9, 128, 128, 1),       # all 25 tap-class pairs of stride 1, every tile ragged
    (3, 9, 11, 128, 64, 2),       # the merged parity classes of stride 2: range-checked taps
    (2, 10, 12, 256, 64, 2),      # a last row / column no tap reaches; two channel tiles
    (40, 19, 19, 128, 128, 2),    # cfg-2's second layer: full tiles + a ragged one per class
    (130, 5, 5, 128, 128, 1),     # the smallest layer: one-pixel classes
])
@pytest.mark.parametrize("form", ["tile", "ksplit"])
def test_conv_data_gradient_dma_tile_vs_fp64(B, IH, IW, Ci, Co, stride, form, monkeypatch):
    """The DMA-fed data-gradient tiles of the pair launch (conv_mfma.hip; fp32 rows by LDS-DMA,
    six exact bf16 products per fragment pair) forced onto shapes of every tap-class structure
    -- "tile": DMODE 4, 64 x 128, SCAE_K8_DGX=1 (any layer takes it); "ksplit": DMODE 5, 32 x 64
    with the K loop split over the waves, SCAE_K8_DGK=1 --: the gated input gradient against
    conv2d in fp64 at 2e-6 of its largest entry, and against the first-generation tiles of the
    same launch; the weight-gradient partials of the launch are the same bits either way
    ("ksplit": to round-off -- its launch always takes the 32-pixel ring)."""
    import ctypes
    import torch.nn.functional as F
    from torch_scae_amd import _lib
    lib = _lib.load()
    P = ctypes.c_void_p
    s = stride
    g = torch.Generator().manual_seed(B * 100 + IH)
    OH, OW = (IH - 3) // s + 1, (IW - 3) // s + 1
    st = P(torch.cuda.current_stream().cuda_stream)
    x = torch.relu(torch.randn(B, IH, IW, Ci, generator=g)).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).cuda()
    wd = w.permute(1, 2, 3, 0).contiguous()       # (ci, 9, co)
    dpre = torch.randn(B, OH, OW, Co, generator=g).cuda()
    splits = lib.scae_conv3x3_wgrad_splits(B, OH, OW, Ci, Co)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SCAE_K8_DGX", mode if form == "tile" else "0")
        monkeypatch.setenv("SCAE_K8_DGK", mode if form == "ksplit" else "0")
        din = torch.full((B, IH, IW, Ci), 7.0, device="cuda")
        part = torch.full((splits * (9 * Co * Ci + Co),), 3.0, device="cuda")
        _lib.call("scae_conv3x3_bwd_pair_f32", P(dpre.data_ptr()), P(wd.data_ptr()),
                  P(x.data_ptr()), P(din.data_ptr()), P(part.data_ptr()), B, IH, IW, Ci, Co, s, st)
        torch.cuda.synchronize()
        outs[mode] = (din, part)
    xr = x.double().permute(0, 3, 1, 2).cpu().requires_grad_(True)
    F.conv2d(xr, w.double().cpu(), None, stride=s).backward(dpre.double().permute(0, 3, 1, 2).cpu())
    dref = (xr.grad * (xr.detach() > 0)).permute(0, 2, 3, 1)
    top = float(dref.abs().max())
    assert float((outs["1"][0].double().cpu() - dref).abs().max()) <= 2e-6 * top
    assert float((outs["1"][0] - outs["0"][0]).abs().max()) <= 4e-6 * top
    if form == "tile":
        assert torch.equal(outs["1"][1], outs["0"][1])
    else:   # (the K-split form's launch takes the 32-pixel weight-gradient ring whatever the
        # layer's size: the splits cut the pixels elsewhere, the sums over the splits agree)
        n = 9 * Co * Ci
        for lo, hi, width in ((0, splits * n, n), (splits * n, splits * (n + Co), Co)):
            a, b = (outs[m][1][lo:hi].view(splits, width).double().sum(0) for m in ("1", "0"))
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


@pytest.mark.parametrize("B,IH,IW,Ci,Co,stride,post", [
    (3, 9, 9, 128, 64, 1, True),      # one ragged tile, the embedding bias output
    (2, 21, 19, 64, 128, 2, False),   # two channel tiles, two chunks per tap, stride 2
    (130, 5, 5, 128, 128, 1, True),   # several tiles, the last one ragged
    (40, 19, 19, 128, 128, 2, False), # cfg-2's second layer
])
def test_conv_forward_ksplit_vs_conv2d(B, IH, IW, Ci, Co, stride, post, monkeypatch):
    """scae_conv3x3_fwd_f32 on the tile form with the K loop dealt to the waves (conv_mfma.hip,
    fwd_x6k_tile: an option, SCAE_K8_FWDK=1) against conv2d in fp64 at 2e-6 of the output's largest entry, and
    against the ring-pipelined tiles (SCAE_K8_FWDK=0); rows past the last pixel are not written."""
    import ctypes
    import torch.nn.functional as F
    from torch_scae_amd import _lib
    P = ctypes.c_void_p
    s = stride
    g = torch.Generator().manual_seed(B * 10 + IH)
    OH, OW = (IH - 3) // s + 1, (IW - 3) // s + 1
    st = P(torch.cuda.current_stream().cuda_stream)
    x = torch.relu(torch.randn(B, IH, IW, Ci, generator=g)).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5).cuda()
    wf = w.permute(0, 2, 3, 1).contiguous()       # (co, 9, ci)
    bias, pb = torch.randn(Co, generator=g).cuda(), torch.randn(Co, OH, OW, generator=g).cuda()
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SCAE_K8_FWDK", mode)
        out = torch.full((B * OH * OW + 5, Co), 7.0, device="cuda")
        outp = torch.full((B * OH * OW + 5, Co), 7.0, device="cuda") if post else None
        _lib.call("scae_conv3x3_fwd_f32", P(x.data_ptr()), P(wf.data_ptr()), P(bias.data_ptr()),
                  P(out.data_ptr()), P(pb.data_ptr()) if post else None,
                  P(outp.data_ptr()) if post else None, B, IH, IW, Ci, Co, s, st)
        torch.cuda.synchronize()
        assert bool((out[-5:] == 7.0).all()) and (outp is None or bool((outp[-5:] == 7.0).all()))
        outs[mode] = (out[:-5].view(B, OH, OW, Co), None if outp is None else outp[:-5].view(B, OH, OW, Co))
    ref = torch.relu(F.conv2d(x.double().permute(0, 3, 1, 2).cpu(), w.double().cpu(),
                              bias.double().cpu(), stride=s)).permute(0, 2, 3, 1)
    top = float(ref.abs().max())
    assert float((outs["1"][0].double().cpu() - ref).abs().max()) <= 2e-6 * top
    assert float((outs["1"][0] - outs["0"][0]).abs().max()) <= 4e-6 * top
    if post:
        refp = ref + pb.double().cpu().permute(1, 2, 0)
        assert float((outs["1"][1].double().cpu() - refp).abs().max()) <= 2e-6 * float(refp.abs().max())


def test_conv_stack_falls_back_for_small_channel_counts():
    from torch_scae_amd import ops
    from torch_scae_amd.part_encoder import CNNEncoder
    assert not ops.conv_stack_supported(1, [8, 8], [3, 3], [2, 1])
    assert not ops.conv_stack_supported(1, [64, 64], [3, 5], [2, 1])
    assert not ops.conv_stack_supported(1, [64, 64], [3, 3], [3, 1])
    assert not ops.conv_stack_supported(5, [64], [3], [1])
    # unsupported shapes still run (library convolutions), same module surface
    torch.manual_seed(0)
    enc = CNNEncoder((1, 20, 20), [8, 8], [3, 3], [3, 1]).cuda()
    x = torch.rand(2, 1, 20, 20, device="cuda")
    y = enc(x)
    assert tuple(y.shape[1:]) == tuple(enc.output_shape) == (8, 4, 4)
    y.sum().backward()
    assert enc.network[0].weight.grad is not None


def test_mnist_transform_on_device():
    from torch_scae_amd.data import pad_and_translate
    g = torch.Generator().manual_seed(3)
    digits = torch.randint(0, 256, (64, 1, 28, 28), generator=g, dtype=torch.uint8)
    shifts = torch.randint(-6, 7, (64, 2), generator=g)
    a = pad_and_translate(digits, (40, 40), shifts=shifts)
    b = pad_and_translate(digits.cuda(), (40, 40), shifts=shifts)
    # (x / 255 may round differently on the device: 1 ulp)
    assert b.is_cuda and torch.allclose(a, b.cpu(), rtol=0, atol=1e-7)
    assert torch.equal(a == 0, b.cpu() == 0)          # same placement


# --------------------------------------------------------------------------
# K2d weight folding (set_transformer.py:218-223 projections) vs fp64 algebra
# --------------------------------------------------------------------------
@pytest.mark.parametrize("B,N,O,C,G,rows", [
    (128, 24, 24, 256, 24, 128),    # the bench workload's shapes
    (37, 20, 17, 128, 5, 37),
])
def test_attention_backward_carrying_weight_gemms_equals_separate_launches(
        B, N, O, C, G, rows):
    """scae_seed_attention_mfma_bwd_gemm_f32 (csrc/seed_bwd_gemm.hip): the
    output attention's backward with weight-gradient GEMM tiles as the tail of
    its grid, against the two launches it replaces -- bit for bit."""
    import ctypes
    from torch_scae_amd import _lib, ops
    g = torch.Generator().manual_seed(B + O)
    rnd = lambda *shape: torch.randn(*shape, generator=g).cuda()
    D = 16
    h, q, wk, wv = rnd(B, N, D), rnd(O, C) / C ** .5, rnd(C, D) * .3, rnd(C, D) * .3
    presence, gout = torch.rand(B, N, generator=g).cuda(), rnd(B, O, C)
    # two weight-gradient problems gW[g] (n x k) = gpre[g]^T x[g], batch = G groups
    probs = []
    for n, k in ((32, 48), (50, 33)):
        gpre, x = rnd(G, rows, n), rnd(G, rows, k)
        probs.append((gpre, x, n, k))
    st = torch.cuda.current_stream().cuda_stream
    nrows = _lib.load().scae_seed_attention_mfma_rows(B)
    results = []
    for merged in (False, True):
        gh = torch.empty(B, N, D, device="cuda")
        partial = torch.empty(nrows, O * D + C * D + C, device="cuda")
        gws = [torch.empty(G, n, k, device="cuda") for _, _, n, k in probs]
        descs = (_lib.GemmDesc * len(probs))()
        for i, ((gpre, x, n, k), gw) in enumerate(zip(probs, gws)):
            descs[i] = ops._gemm_desc(ops._p(gpre), ops._p(x), ops._p(gw), G, n, k,
                                      rows, False, n, rows * n, False, k,
                                      rows * k, k, n * k)
        args = (ops._p(h), ops._p(q), ops._p(wk), ops._p(wv), ops._p(presence),
                ops._p(gout), ops._p(gh), ops._p(partial), B, N, O, C)
        if merged:
            _lib.call("scae_seed_attention_mfma_bwd_gemm_f32", *args, descs,
                      len(probs), st)
        else:
            _lib.call("scae_seed_attention_mfma_bwd_f32", *args, st)
            _lib.call("scae_gemm_multi_f32", descs, len(probs), st)
        torch.cuda.synchronize()
        results.append([gh, partial, *gws])
    for a, b in zip(*results):
        assert torch.equal(a, b)
    ref = torch.einsum("grn,grk->gnk", probs[0][0].double(), probs[0][1].double())
    assert_close(results[1][2], ref.float(), atol=1e-4, rtol=1e-5, what="gW")


@pytest.mark.parametrize("O,C,D", [(24, 256, 16), (5, 64, 8), (32, 512, 32),
                                   (3, 128, 16), (9, 1024, 8)])
def test_seed_fold_vs_fp64(O, C, D):
    from torch_scae_amd import ops
    assert ops.seed_fold_supported(O, C, D)
    g = torch.Generator().manual_seed(O * C + D)
    shapes = [(O, C), (C, C), (C,), (C, C), (C,), (C, C), (C,), (C, C), (C,),
              (C, D), (C,)]
    vals = [torch.randn(*s, generator=g) / (s[-1] ** 0.5) for s in shapes]

    def fold(seeds, wq, bq, wk, bk, wv, bv, wo, bo, w2, b2):
        q = seeds @ wq.T + bq
        wv2, bv2 = wv @ w2, wv @ b2 + bv
        return q, wk @ w2, wk @ b2 + bk, wo @ wv2, wo @ bv2 + bo

    ref_in = [v.double().requires_grad_() for v in vals]
    ref_out = fold(*ref_in)
    hip_in = [v.cuda().requires_grad_() for v in vals]
    hip_out = ops.seed_fold(*hip_in)
    names = ["q", "wkf", "bkf", "wvf", "bvf"]
    for n, a, b in zip(names, hip_out, ref_out):
        assert_close(a, b.float(), rtol=1e-4, atol=1e-5, what=n)
    gouts = [torch.randn(o.shape, generator=g) for o in ref_out]
    torch.autograd.backward(ref_out, [t.double() for t in gouts])
    torch.autograd.backward(hip_out, [t.cuda() for t in gouts])
    pn = ["seeds", "wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "w2", "b2"]
    for n, a, b in zip(pn, hip_in, ref_in):
        scale = float(b.grad.abs().max())
        assert_close(a.grad, b.grad.float(), rtol=1e-4, atol=1e-5 * scale,
                     what="d_" + n)
    # partially-used outputs: missing gradients count as zeros
    hip_in2 = [v.cuda().requires_grad_() for v in vals]
    out2 = ops.seed_fold(*hip_in2)
    out2[0].sum().backward()
    assert float(hip_in2[3].grad.abs().max()) == 0.0
    assert_close(hip_in2[2].grad, torch.full((C,), float(O)), what="d_bq")


def test_head_backward_with_colour_backward_in_front_equals_separate_launches():
    """scae_capsule_head_bwd_tc_f32 (the colour MLP's backward and the head's in
    one grid, workgroup by workgroup) against scae_template_color_bwd_f32 +
    scae_capsule_head_bwd_f32 -- bit for bit."""
    from torch_scae_amd import _lib, ops
    g = torch.Generator().manual_seed(44)
    rnd = lambda *shape: torch.randn(*shape, generator=g).cuda()
    new = lambda *shape: torch.empty(*shape, device="cuda")
    B, HW, A, F, C, hw, H1 = 128, 25, 24, 16, 1, 121, 32
    P = F + 8
    y, pooled = rnd(B, HW, A * P), rnd(B, A, P - 1)
    noise_u = torch.rand(B, A, generator=g).cuda()
    g_pose, g_pres, g_feat = rnd(B, A, 6), rnd(B, A), rnd(B, A, F)
    logits, feature = rnd(A, C, hw), rnd(B, A, F)
    w1, b1, w2, b2 = rnd(H1, F) * .2, rnd(H1) * .1, rnd(C, H1) * .2, rnd(C) * .1
    color, g_templates, g_raw = torch.rand(B, A, C, generator=g).cuda(), \
        rnd(B, A, C, hw), rnd(A, C, hw)
    rows = _lib.load().scae_template_color_partial_rows(B, A)
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for merged in (False, True):
        dy, g_logits, tc_gf = new(B, HW, A * P), new(A, C, hw), new(B, A, F)
        partial = new(rows, H1 * F + H1 + C * H1 + C)
        tc_ptrs = tuple(ops._p(t) for t in (logits, feature, w1, b1, w2, b2, color,
                                            g_templates, g_raw, g_logits, tc_gf,
                                            partial))
        dims = (C, hw, F, H1, 1, 1)
        head = (ops._p(y), ops._p(pooled), ops._p(noise_u), 4.0, 1, ops._p(g_pose),
                ops._p(g_pres), ops._p(g_feat))
        if merged:
            _lib.call("scae_capsule_head_bwd_tc_f32", *head, ops._p(dy), B, HW, A,
                      P, *tc_ptrs, *dims, st)
        else:
            _lib.call("scae_template_color_bwd_f32", *tc_ptrs, B, A, *dims, st)
            _lib.call("scae_capsule_head_bwd_f32", *head, ops._p(tc_gf),
                      ops._p(dy), B, HW, A, P, st)
        torch.cuda.synchronize()
        res.append((dy, g_logits, tc_gf, partial))
    for a, b, what in zip(res[0], res[1], ("dy", "g_logits", "g_feature", "partial")):
        assert torch.equal(a, b), what


def test_conv_backward_carrying_reduce_and_fold_backward_equals_separate_launches():
    """scae_conv3x3_bwd_pair_reduce_f32 / _fold_f32: a conv layer's backward
    launch with the output attention's partial-row reduction, resp. the
    folding products' backward, as 256-thread workgroups at the head of its
    grid -- bit for bit what scae_conv3x3_bwd_pair_f32 +
    scae_seed_attention_mfma_reduce_f32 + scae_seed_fold_bwd_f32 (1024-thread
    workgroups) give."""
    import ctypes
    from torch_scae_amd import _lib, ops
    g = torch.Generator().manual_seed(33)
    rnd = lambda *shape: torch.randn(*shape, generator=g).cuda()
    new = lambda *shape: torch.empty(*shape, device="cuda")
    B, ih, ci, co, s = 128, 9, 128, 128, 1
    oh = ih - 2
    xin, dpre = torch.relu(rnd(B, ih, ih, ci)), rnd(B, oh, oh, co)
    wd = rnd(ci, 9, co) * .03
    O, C, D, rows = 24, 256, 16, 128
    partial_rows = rnd(rows, O * D + C * D + C)
    q, wk = rnd(O, C), rnd(C, D)
    shapes = [(O, C), (C, C), (C,), (C, C), (C,), (C, C), (C,), (C, C), (C,),
              (C, D), (C,)]
    vals = [rnd(*sh) / sh[-1] ** .5 for sh in shapes]
    fold_outs = (new(O, C), new(C, D), new(C), new(C, D), new(C), new(C, D + 1),
                 new(C, C))
    st = torch.cuda.current_stream().cuda_stream
    lib = _lib.load()
    _lib.call("scae_seed_fold_fwd_f32",
              ctypes.byref(ops._fold_desc(vals, fold_outs, O, C, D)), st)
    splits = lib.scae_conv3x3_wgrad_splits(B, oh, oh, ci, co)
    res = []
    for carried in (False, True):
        outs = []
        for which in ("reduce", "fold"):
            din, part = new(B, ih, ih, ci), new(splits * (9 * co * ci + co))
            pair = (ops._p(dpre), ops._p(wd), ops._p(xin), ops._p(din),
                    ops._p(part), B, ih, ih, ci, co, s)
            if which == "reduce":
                red_out = [new(O, C), new(C, D), new(C), new(C, D), new(C)]
                red = (ops._p(partial_rows), rows, ops._p(q), ops._p(wk),
                       *[ops._p(t) for t in red_out], O, C)
                if carried:
                    _lib.call("scae_conv3x3_bwd_pair_reduce_f32", *pair, *red, st)
                else:
                    _lib.call("scae_conv3x3_bwd_pair_f32", *pair, st)
                    _lib.call("scae_seed_attention_mfma_reduce_f32", *red, st)
                outs += [din, part, *red_out]
            else:
                desc = ops._fold_desc(vals, fold_outs, O, C, D)
                gr = _lib.SeedFoldGrads()
                for name, t in zip(("g_q", "g_wkf", "g_bkf", "g_wvf", "g_bvf"),
                                   red_out):
                    setattr(gr, name, t.data_ptr())
                grads = [torch.empty_like(v) for v in vals]
                for name, t in zip(ops._FOLD_INPUTS, grads):
                    setattr(gr, "d_" + name, t.data_ptr())
                if carried:
                    _lib.call("scae_conv3x3_bwd_pair_fold_f32", *pair,
                              ctypes.byref(desc), ctypes.byref(gr), st)
                else:
                    _lib.call("scae_conv3x3_bwd_pair_f32", *pair, st)
                    _lib.call("scae_seed_fold_bwd_f32", ctypes.byref(desc),
                              ctypes.byref(gr), st)
                outs += [din, part, *grads]
            torch.cuda.synchronize()
        res.append(outs)
    assert len(res[0]) == len(res[1]) == 7 + 2 + 11
    for i, (a, b) in enumerate(zip(*res)):
        assert torch.equal(a, b), i


@pytest.mark.parametrize("ksplit", ["0", "1"])
def test_conv_layer_carrying_the_folding_products_equals_separate_launches(ksplit, monkeypatch):
    """scae_conv3x3_fwd_fold_f32: the encoder's second conv layer with the
    output attention's folding products as the tail of its grid, against
    scae_conv3x3_fwd_f32 + scae_seed_fold_fwd_f32 (conv output bit for bit;
    the products to round-off: the carried form keeps 16 instead of 32 loads in
    flight, same sums in the same order).  Both launches on the same tile form
    (they choose it by the same rule): SCAE_K8_FWDK=0 the ring-pipelined tiles,
    =1 the form with the K loop dealt to the waves."""
    import ctypes
    from torch_scae_amd import _lib, ops
    monkeypatch.setenv("SCAE_K8_FWDK", ksplit)
    g = torch.Generator().manual_seed(21)
    B, ih, iw, ci, co, s = 16, 19, 19, 128, 128, 2
    O, C, D = 24, 256, 16
    act = torch.rand(B, ih, iw, ci, generator=g).cuda()
    wf = (torch.randn(co, 9, ci, generator=g) * .03).cuda()
    bias = torch.randn(co, generator=g).cuda()
    shapes = [(O, C), (C, C), (C,), (C, C), (C,), (C, C), (C,), (C, C), (C,),
              (C, D), (C,)]
    vals = [(torch.randn(*sh, generator=g) / sh[-1] ** .5).cuda() for sh in shapes]
    oh = (ih - 3) // s + 1
    st = torch.cuda.current_stream().cuda_stream
    new = lambda *shape: torch.empty(*shape, device="cuda")
    res = []
    for carried in (False, True):
        out = new(B, oh, oh, co)
        fold = (new(O, C), new(C, D), new(C), new(C, D), new(C), new(C, D + 1),
                new(C, C))
        desc = ops._fold_desc(vals, fold, O, C, D)
        conv = (ops._p(act), ops._p(wf), ops._p(bias), ops._p(out), None, None, B,
                ih, iw, ci, co, s)
        if carried:
            _lib.call("scae_conv3x3_fwd_fold_f32", *conv, ctypes.byref(desc), st)
        else:
            _lib.call("scae_conv3x3_fwd_f32", *conv, st)
            _lib.call("scae_seed_fold_fwd_f32", ctypes.byref(desc), st)
        torch.cuda.synchronize()
        res.append((out, *fold))
    assert torch.equal(res[0][0], res[1][0])
    for a, b, what in zip(res[0][1:], res[1][1:],
                          ("q", "wkf", "bkf", "wvf", "bvf", "wv2e", "wowv")):
        assert_close(a, b, rtol=1e-5, atol=1e-6, what=what)


# --------------------------------------------------------------------------
# K7 batched MFMA GEMM: all operand layouts, ragged shapes, epilogues
# --------------------------------------------------------------------------
@pytest.mark.parametrize("G,M,N,K,ak,bk", [
    (3, 70, 45, 37, True, True),       # ragged, scalar loads / stores
    (2, 64, 128, 96, True, False),
    (5, 33, 17, 130, False, True),
    (4, 128, 64, 200, False, False),
    (1, 3200, 576, 128, True, True),   # the 1x1 attention conv (64x64 tiles)
    (40, 256, 128, 64, False, False),  # enough tiles for the 64x64 shape
])
def test_gemm_layouts_and_epilogues(G, M, N, K, ak, bk):
    from torch_scae_amd import ops
    g = torch.Generator().manual_seed(G * M + N * K)
    A = torch.randn(G, M, K, generator=g)
    Bm = torch.randn(G, N, K, generator=g)
    bias = torch.randn(G, N, generator=g)
    mask = torch.randn(G, M, N, generator=g)
    ref = torch.einsum("gmk,gnk->gmn", A.double(), Bm.double())
    a_dev = (A if ak else A.transpose(1, 2).contiguous()).cuda()
    b_dev = (Bm if bk else Bm.transpose(1, 2).contiguous()).cuda()
    lda, ldb = (K if ak else M), (K if bk else N)
    tol = dict(rtol=2e-4, atol=2e-4 * K ** 0.5)

    def run(**kw):
        out = torch.full((G, M, N), float("nan"), device="cuda")
        ops._gemm(ops._p(a_dev), ops._p(b_dev), ops._p(out), G, M, N, K, ak,
                  lda, M * K, bk, ldb, N * K, N, M * N, ref=out, **kw)
        return out

    assert_close(run(), ref.float(), what="plain", **tol)
    bias_d, mask_d = bias.cuda(), mask.cuda()
    want = torch.relu(ref + bias.double()[:, None, :]) * (mask > 0)
    got = run(bias=ops._p(bias_d), bias_ld=1, bias_b=N, relu=True,
              mask=ops._p(mask_d), ldmask=N, mask_b=M * N)
    assert_close(got, want.float(), what="bias+relu+gate", **tol)
    if not ak:
        asum = torch.full((G, M), float("nan"), device="cuda")
        run(asum=ops._p(asum), asum_b=M)
        assert_close(asum, A.double().sum(2).float(), what="asum", **tol)
        # strided destination (a column of a wider matrix)
        wide = torch.full((G, M, 3), float("nan"), device="cuda")
        run(asum=ops._off(wide, 2), asum_b=3 * M, asum_ld=3)
        assert_close(wide[:, :, 2], A.double().sum(2).float(),
                     what="strided asum", **tol)
        assert bool(torch.isnan(wide[:, :, :2]).all())


# --------------------------------------------------------------------------
# K9 1x1 attention conv + attention pooling (part_encoder.py:71-74,
# nn_ext.py:76-101) vs the composed torch ops in fp64
# --------------------------------------------------------------------------
@pytest.mark.parametrize("B,C,H,W,A,P", [(128, 128, 5, 5, 24, 24),
                                         (6, 8, 3, 4, 5, 9),
                                         (7, 64, 6, 6, 32, 24),
                                         (3, 20, 1, 1, 4, 2)])
def test_attention_conv_pool_vs_torch(B, C, H, W, A, P):
    import torch.nn.functional as F
    from torch_scae_amd import ops
    from torch_scae_amd.nn_ext import multiple_attention_pooling_2d
    assert ops.attention_pool_supported(H * W, A, P)
    g = torch.Generator().manual_seed(B + C + A)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(A * P, C, generator=g) / C ** 0.5
    b = torch.randn(A * P, generator=g)
    gout = torch.randn(B, A, P - 1, generator=g)

    xr, wr, br = (t.double().requires_grad_() for t in (x, w, b))
    ref = multiple_attention_pooling_2d(
        F.conv2d(xr, wr.view(A * P, C, 1, 1), br), A).view(B, A, P - 1)
    ref.backward(gout.double())

    xh = x.permute(0, 2, 3, 1).reshape(B, H * W, C).contiguous().cuda() \
        .requires_grad_()
    wh, bh = w.cuda().requires_grad_(), b.cuda().requires_grad_()
    out = ops.attention_conv_pool(xh, wh, bh, A)
    out.backward(gout.cuda())
    assert_close(out, ref.float(), rtol=1e-4, atol=1e-5, what="pooled")
    for name, a, r in (("dx", xh.grad, xr.grad.permute(0, 2, 3, 1)
                        .reshape(B, H * W, C)), ("dw", wh.grad, wr.grad),
                       ("db", bh.grad, br.grad)):
        assert_close(a, r.float(), rtol=2e-4,
                     atol=2e-5 * max(1.0, float(r.abs().max())), what=name)


# --------------------------------------------------------------------------
# fused RMSprop step (base_experiment.py:44-77) vs torch.optim.RMSprop
# --------------------------------------------------------------------------
@pytest.mark.parametrize("momentum", [0.9, 0.0])
def test_rmsprop_flat_vs_torch_optim(momentum):
    import torch.nn as nn
    from torch_scae_amd.data_parallel import FlatParameters, RMSpropFlat
    torch.manual_seed(3)
    net = nn.Sequential(nn.Linear(13, 7), nn.ReLU(), nn.Linear(7, 5)).cuda()
    ref = nn.Sequential(nn.Linear(13, 7), nn.ReLU(), nn.Linear(7, 5)).cuda()
    ref.load_state_dict(net.state_dict())
    flat = FlatParameters(net)
    opt = RMSpropFlat(flat, lr=1e-2, eps=1e-4, momentum=momentum)
    ropt = torch.optim.RMSprop(ref.parameters(), lr=1e-2, eps=1e-4,
                               momentum=momentum)
    for it in range(4):
        x = torch.randn(9, 13, device="cuda")
        for m, o in ((net, None), (ref, ropt)):
            if o is None:
                flat.clear_grads()
            else:
                o.zero_grad()
            m(x).square().sum().backward()
        flat.gather_grads()
        opt.step()
        ropt.step()
    for a, b in zip(net.parameters(), ref.parameters()):
        assert_close(a, b, rtol=1e-5, atol=1e-6, what="param after 4 steps")


@pytest.mark.parametrize("B,HW,C,A,P", [
    (128, 25, 128, 24, 23),    # the bench workload: preferred
    (40, 25, 128, 48, 23),     # many 16-channel tiles per wave
    (2048, 9, 128, 8, 16),     # a batch where the GEMM form is preferred
])
def test_fused_head_conv_equals_gemm_plus_head(B, HW, C, A, P):
    """scae_capsule_head_conv_fwd_f32 (1x1 conv inside the pooling workgroups)
    against the K7 GEMM + scae_capsule_head_fwd_f32 it replaces, called
    directly so that the size preference does not pick for us."""
    from torch_scae_amd import _lib, ops
    g = torch.Generator().manual_seed(B + HW + A)
    x = torch.randn(B, HW, C, generator=g).cuda()
    w = (torch.randn(A * P, C, generator=g) / C ** 0.5).cuda()
    bias = torch.randn(A * P, generator=g).cuda()
    u = torch.rand(B, A, generator=g).cuda()
    assert _lib.load().scae_capsule_head_conv_supported(HW, A, P, C)
    if (B, A) == (128, 24):
        assert _lib.load().scae_capsule_head_conv_preferred(B, HW, A, P, C)
    if B == 2048:
        assert not _lib.load().scae_capsule_head_conv_preferred(B, HW, A, P, C)
    new = lambda *shape: torch.empty(*shape, device="cuda")
    outs = []
    for fused in (True, False):
        pooled, pose, pres = new(B, A, P - 1), new(B, A, 6), new(B, A)
        feat, absence = new(B, A, max(P - 8, 1)), new(B, A, 1)
        fp = feat.data_ptr() if P > 8 else None
        head = (u.data_ptr(), 4.0, 1, pooled.data_ptr(), pose.data_ptr(),
                pres.data_ptr(), fp, absence.data_ptr(), B, HW, A, P,
                torch.cuda.current_stream().cuda_stream)
        if fused:
            y = new(B, HW, A * P)
            _lib.call("scae_capsule_head_conv_fwd_f32", x.data_ptr(),
                      w.data_ptr(), bias.data_ptr(), C, y.data_ptr(), *head)
        else:
            y = ops._conv1x1_fwd(x, w, bias)
            _lib.call("scae_capsule_head_fwd_f32", y.data_ptr(), *head)
        torch.cuda.synchronize()
        outs.append((y, pooled, pose, pres, absence) + ((feat,) if P > 8 else ()))
    ref = (x.double() @ w.double().t() + bias.double()).float()
    assert_close(outs[0][0], ref, rtol=1e-5, atol=1e-5, what="fused y vs fp64")
    for a, b, what in zip(outs[0], outs[1],
                          ("y", "pooled", "pose", "presence", "absence", "feature")):
        assert_close(a, b, rtol=1e-5, atol=1e-5, what=what)


@pytest.mark.parametrize("B,C,H,W,A,F,sim,noisy", [
    (128, 128, 5, 5, 24, 16, False, True),
    (5, 8, 3, 4, 5, 0, True, False),
    (7, 64, 2, 2, 6, 3, False, True),
    (9, 192, 4, 8, 7, 5, True, True),      # fused conv: 32 pixels, ragged tiles
    (3, 256, 3, 3, 16, 16, False, False),  # ... at its largest channel count
    (3, 320, 3, 3, 4, 2, False, False),    # beyond it: GEMM + head
])
def test_capsule_head_vs_oracle(B, C, H, W, A, F, sim, noisy):
    """part_encoder.py:71-92 end to end (conv1x1, pooling, split, noise,
    sigmoid, geometric_transform) against the composed fp64 reference ops."""
    import torch.nn.functional as Fn
    from torch_scae_amd import ops
    from torch_scae_amd.nn_ext import multiple_attention_pooling_2d
    P = 6 + 1 + F + 1
    g = torch.Generator().manual_seed(B * 7 + A)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(A * P, C, generator=g) / C ** 0.5
    b = torch.randn(A * P, generator=g)
    u = torch.rand(B, A, generator=g) if noisy else None
    scale = 4.0
    gouts = [torch.randn(B, A, 6, generator=g), torch.randn(B, A, generator=g),
             torch.randn(B, A, F, generator=g)]

    xr, wr, br = (t.double().requires_grad_() for t in (x, w, b))
    h = multiple_attention_pooling_2d(
        Fn.conv2d(xr, wr.view(A * P, C, 1, 1), br), A).view(B, A, P - 1)
    pose_r = O.geometric_transform(h[..., :6], similarity=sim)
    logit = h[..., 6]
    if noisy:
        logit = logit + (u.double() - .5) * scale
    pres_r, feat_r = torch.sigmoid(logit), h[..., 7:]
    loss = (pose_r * gouts[0].double()).sum() + (pres_r * gouts[1].double()).sum()
    if F:
        loss = loss + (feat_r * gouts[2].double()).sum()
    loss.backward()

    xh = x.permute(0, 2, 3, 1).reshape(B, H * W, C).contiguous().cuda() \
        .requires_grad_()
    wh, bh = w.cuda().requires_grad_(), b.cuda().requires_grad_()
    pose, pres, feat, absence = ops.capsule_head(
        xh, wh, bh, A, u.cuda() if noisy else None, scale, sim)
    assert torch.equal(absence, 1. - pres.detach().unsqueeze(-1))
    assert (feat is None) == (F == 0)
    assert_close(pose, pose_r.float(), rtol=1e-4, atol=1e-5, what="pose")
    assert_close(pres, pres_r.float(), rtol=1e-4, atol=1e-5, what="presence")
    lh = (pose * gouts[0].cuda()).sum() + (pres * gouts[1].cuda()).sum()
    if F:
        assert_close(feat, feat_r.float(), rtol=1e-4, atol=1e-5, what="feat")
        lh = lh + (feat * gouts[2].cuda()).sum()
    lh.backward()
    for name, a, r in (("dx", xh.grad, xr.grad.permute(0, 2, 3, 1)
                        .reshape(B, H * W, C)), ("dw", wh.grad, wr.grad),
                       ("db", bh.grad, br.grad)):
        assert_close(a, r.float(), rtol=5e-4,
                     atol=5e-5 * max(1.0, float(r.abs().max())), what=name)


# --------------------------------------------------------------------------
# K10 coloured templates (part_decoder.py:78-110) vs the oracle, fwd + grads
# --------------------------------------------------------------------------
@pytest.mark.parametrize("tnl,cnl,C,F,ts", [("sigmoid", "sigmoid", 1, 16, 11),
                                            ("relu1", "relu1", 3, 5, 7),
                                            ("sigmoid", "relu1", 2, 16, 4)])
def test_colored_templates_vs_oracle(tnl, cnl, C, F, ts):
    from torch_scae_amd.part_decoder import TemplateGenerator
    torch.manual_seed(11)
    B, M = 9, 6
    tg = TemplateGenerator(M, C, (ts, ts), template_nonlin=tnl, dim_feature=F,
                           colorize_templates=True, color_nonlin=cnl)
    with torch.no_grad():     # spread the logits over both clamp regions
        tg.template_logits.mul_(1.6).sub_(.3)
    feature = torch.randn(B, M, F)
    gt, gr = torch.randn(B, M, C, ts, ts), torch.randn(1, M, C, ts, ts)
    cfg = dict(template_nonlin=tnl, color_nonlin=cnl, colorize_templates=True)

    P = {"tg." + k: v.detach().clone().requires_grad_()
         for k, v in tg.state_dict().items()}
    f_ref = feature.clone().requires_grad_()
    ref = O.template_generator(P, "tg", f_ref, B, cfg)
    ((ref.templates * gt).sum() + (ref.raw_templates * gr).sum()).backward()

    tg = tg.cuda()
    f_hip = feature.cuda().requires_grad_()
    out = tg(feature=f_hip)
    ((out.templates * gt.cuda()).sum()
     + (out.raw_templates * gr.cuda()).sum()).backward()
    assert_close(out.templates, ref.templates, rtol=1e-5, atol=1e-6,
                 what="templates")
    assert_close(out.raw_templates, ref.raw_templates, rtol=1e-5, atol=1e-6,
                 what="raw")
    assert_close(f_hip.grad, f_ref.grad, rtol=1e-4, atol=1e-5, what="d_feature")
    for name, p in tg.named_parameters():
        assert_close(p.grad, P["tg." + name].grad, rtol=1e-4, atol=2e-5,
                     what="d_" + name)


@pytest.mark.parametrize("rows,cols", [(1, 5), (16, 300), (128, 14848),
                                       (1024, 1280), (3, 70000)])
def test_sum_rows_scatter(rows, cols):
    from torch_scae_amd import ops
    g = torch.Generator().manual_seed(rows + cols)
    src = torch.randn(rows, cols, generator=g)
    want = src.double().sum(0)
    cuts = sorted({0, cols // 7, cols // 3, cols // 2, cols - 1, cols})
    shapes = [(b - a,) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    outs = ops._sum_rows(src.cuda(), shapes)
    pos = 0
    for o in outs:
        assert_close(o, want[pos:pos + o.numel()].float(), 1e-5 * rows ** 0.5,
                     1e-5, "segment")
        pos += o.numel()
    # explicit starts: a window in the middle, the rest dropped
    if cols >= 8:
        (mid,) = ops._sum_rows(src.cuda(), [(2, 2)], starts=[cols // 2])
        assert_close(mid.flatten(), want[cols // 2:cols // 2 + 4].float(),
                     1e-5 * rows ** 0.5, 1e-5, "window")


def test_class_probs_vs_torch():
    """stacked_capsule_auto_encoder.py:205-212 in one launch, incl. the
    classifier gradients of the un-fused loss path."""
    import torch.nn.functional as F
    from torch_scae_amd import ops
    g = torch.Generator().manual_seed(2)
    B, Oc, M, ncls = 37, 24, 24, 10
    cp = torch.rand(B, Oc, generator=g)
    post = torch.softmax(torch.randn(B, Oc + 1, M, generator=g), 1)
    W = torch.randn(ncls, Oc, generator=g) * 0.4
    bb = torch.randn(ncls, generator=g) * 0.1
    label = torch.randint(0, ncls, (B,), generator=g)
    Wr, br = W.clone().requires_grad_(), bb.clone().requires_grad_()
    p1 = torch.softmax(F.linear(cp, Wr, br), -1)
    p2 = torch.softmax(F.linear(post[:, :-1].sum(-1), Wr, br), -1)
    (F.cross_entropy(p1, label) + F.cross_entropy(p2, label)).backward()
    Wh, bh = leaf(W), leaf(bb)
    q1, q2 = ops.class_probs(cp.cuda(), post.cuda(), Wh, bh)
    assert_close(q1, p1, 1e-6, 1e-5, "prior_cls_prob")
    assert_close(q2, p2, 1e-6, 1e-5, "posterior_cls_prob")
    (F.cross_entropy(q1, label.cuda()) + F.cross_entropy(q2, label.cuda())) \
        .backward()
    assert_close(Wh.grad, Wr.grad, 1e-6, 1e-4, "d_w")
    assert_close(bh.grad, br.grad, 1e-6, 1e-4, "d_b")


def test_class_probs_riding_in_the_loss_tail_launch_changes_nothing():
    """Inside ``ops.step_fusion`` the class-probability launch of SCAE.forward
    waits for the loss tail's per-image launch and rides there
    (scae_loss_tail_fwd_class_probs_f32): probabilities, the rider sum, the tail's
    12-vector -- bit for bit what the separate launches give; without a tail
    the block's exit launches it."""
    from torch_scae_amd import ops
    g = torch.Generator().manual_seed(5)
    B, Oc, M, ncls = 128, 24, 24, 10
    lpp = torch.randn(B, M, generator=g).cuda()
    post = torch.softmax(torch.randn(B, Oc + 1, M, generator=g), 1).cuda()
    cp = torch.rand(B, Oc, generator=g).cuda()
    W, bb = (torch.randn(ncls, Oc, generator=g) * .3).cuda(), \
        (torch.randn(ncls, generator=g) * .1).cuda()
    label = torch.randint(0, ncls, (B,), generator=g).cuda()
    src = torch.randn(5000, generator=g).cuda()
    target = torch.empty(1, device="cuda")      # any tensor: "a fused step"

    def run(fused, with_tail=True):
        total = torch.zeros((), device="cuda")
        import contextlib
        with (ops.step_fusion(target) if fused else contextlib.nullcontext()):
            q1, q2 = ops.class_probs(cp, post, W, bb,
                                     extra_sums=[(src, 0.5, total)])
            out = ops.loss_tail(lpp, post, cp, W, bb, label, ncls, "l2", "l2",
                                True, [1., 2., .35, .7, .2], None) \
                if with_tail else None
        torch.cuda.synchronize()
        return q1, q2, total, out

    plain, fused = run(False), run(True)
    for a, b in zip(plain, fused):
        assert torch.equal(a, b)
    lone = run(True, with_tail=False)
    for a, b in zip(plain[:3], lone[:3]):
        assert torch.equal(a, b)


def test_loss_tail_combine_inside_the_backward_launch_changes_nothing():
    """Inside ``ops.step_fusion`` the whole-scalar loss tail leaves its batch
    combine to the backward launch (scae_loss_extras.defer_combine): scalars
    and gradients bit for bit those of the three-launch form; a forward no
    backward follows is completed when the block exits."""
    import contextlib
    from torch_scae_amd import ops
    g = torch.Generator().manual_seed(9)
    B, Oc, M, ncls = 128, 24, 24, 10
    base = [torch.randn(B, M, generator=g),
            torch.softmax(torch.randn(B, Oc + 1, M, generator=g), 1),
            torch.rand(B, Oc, generator=g),
            torch.randn(ncls, Oc, generator=g) * .3,
            torch.randn(ncls, generator=g) * .1,
            torch.randn(B, 25, generator=g), torch.rand(1, generator=g)]
    label = torch.randint(0, ncls, (B,), generator=g).cuda()
    target = torch.empty(1, device="cuda")

    def run(fused, backward=True):
        ins = [leaf(t) for t in base]
        with (ops.step_fusion(target) if fused else contextlib.nullcontext()):
            loss, out = ops.loss_tail_scalar(
                *ins[:5], label, ncls, "l2", "kl", True, [1., 2., .35, .7, .2],
                None, rec_sums=ins[5], reg=ins[6], w_reg=0.3)
            if backward:
                loss.backward()
        torch.cuda.synchronize()
        return [loss.detach(), out.detach()] + \
            ([t.grad for t in ins] if backward else [])

    plain, fused = run(False), run(True)
    assert len(plain) == len(fused) == 9
    for a, b in zip(plain, fused):
        assert torch.equal(a, b)
    lone = run(True, backward=False)
    for a, b in zip(plain[:2], lone):
        assert torch.equal(a, b)


def test_gemm_pair_equals_two_launches():
    """scae_gemm_pair_f32: two differently shaped / laid out problems at once."""
    from torch_scae_amd import ops
    g = torch.Generator().manual_seed(4)
    A0, B0 = torch.randn(3, 70, 37, generator=g), torch.randn(3, 45, 37, generator=g)
    A1, B1 = torch.randn(5, 129, 33, generator=g), torch.randn(5, 64, 129, generator=g)
    a0, b0, a1 = A0.cuda(), B0.cuda(), A1.cuda()         # A1 is (G, K, M)
    b1 = B1.transpose(1, 2).contiguous().cuda()          # both k-strided
    c0 = torch.full((3, 70, 45), float("nan"), device="cuda")
    c1 = torch.full((5, 33, 64), float("nan"), device="cuda")
    asum = torch.full((5, 33), float("nan"), device="cuda")
    ops._gemm_pair(
        ops._gemm_desc(ops._p(a0), ops._p(b0), ops._p(c0), 3, 70, 45, 37, True,
                       37, 70 * 37, True, 37, 45 * 37, 45, 70 * 45, relu=True),
        ops._gemm_desc(ops._p(a1), ops._p(b1), ops._p(c1), 5, 33, 64, 129,
                       False, 33, 129 * 33, False, 64, 129 * 64, 64, 33 * 64,
                       asum=ops._p(asum), asum_b=33), c0)
    want0 = torch.relu(torch.einsum("gmk,gnk->gmn", A0.double(), B0.double()))
    want1 = torch.einsum("gkm,gnk->gmn", A1.double(), B1.double())
    assert_close(c0, want0.float(), 2e-4, 2e-3, "first")
    assert_close(c1, want1.float(), 2e-4, 3e-3, "second")
    assert_close(asum, A1.double().sum(1).float(), 2e-4, 2e-3, "asum")


@pytest.mark.parametrize("maximum", [False, True])
def test_gmm_mode_straight_through_vs_oracle(maximum):
    """distributions.py:50-77 with straight_through_gradient=True."""
    from torch_scae_amd.distributions import GaussianMixture
    g = torch.Generator().manual_seed(12)
    B, K, C, H, W = 3, 5, 1, 6, 7
    loc = torch.rand(B, K, C, H, W, generator=g)
    ml = torch.randn(B, K, 1, H, W, generator=g)
    scale = torch.tensor([0.7])
    lr, mr = loc.clone().requires_grad_(), ml.clone().requires_grad_()
    ref = O.gmm_mode(lr, scale, mr, straight_through_gradient=True,
                     maximum=maximum)
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    lh, mh = leaf(loc), leaf(ml)
    pdf = GaussianMixture.make_from_stats(lh, scale.cuda(), mh)
    out = pdf.mode(straight_through_gradient=True, maximum=maximum)
    out.backward(gout.cuda())
    assert_close(out, ref, 1e-6, 1e-5, "mode")
    assert_close(lh.grad, lr.grad, 1e-6, 1e-5, "d_loc")
    assert_close(mh.grad, mr.grad, 1e-6, 1e-5, "d_logits")


@pytest.mark.parametrize("C,Cm,maximum", [(1, 1, False), (1, 1, True),
                                          (3, 1, False), (3, 3, True)])
def test_gmm_mode_and_mean_gradients_vs_oracle(C, Cm, maximum):
    """distributions.py:37-39 / :50-77 without the straight-through estimator:
    mode() is differentiable w.r.t. loc (the gradient goes to the winning
    component), mean() w.r.t. loc and the mixing logits."""
    from torch_scae_amd.distributions import GaussianMixture
    g = torch.Generator().manual_seed(21 + C + Cm)
    B, K, H, W = 3, 6, 5, 7
    loc = torch.rand(B, K, C, H, W, generator=g)
    ml = torch.randn(B, K, Cm, H, W, generator=g)
    scale = torch.tensor([0.6])
    gout = torch.randn(B, C, H, W, generator=g)
    lr, mr = loc.clone().requires_grad_(), ml.clone().requires_grad_()
    ref = O.gmm_mode(lr, scale, mr, maximum=maximum)
    ref.backward(gout)
    lh, mh = leaf(loc), leaf(ml)
    pdf = GaussianMixture.make_from_stats(lh, scale.cuda(), mh)
    out = pdf.mode(maximum=maximum)
    assert out.requires_grad
    out.backward(gout.cuda())
    assert_close(out, ref, 1e-6, 1e-5, "mode")
    assert_close(lh.grad, lr.grad, 1e-6, 1e-5, "mode d_loc")
    assert mh.grad is None and mr.grad is None      # argmax: no gradient
    lr, mr = loc.clone().requires_grad_(), ml.clone().requires_grad_()
    ref = O.gmm_mean(lr, mr)
    ref.backward(gout)
    lh, mh = leaf(loc), leaf(ml)
    out = GaussianMixture.make_from_stats(lh, scale.cuda(), mh).mean()
    out.backward(gout.cuda())
    assert_close(out, ref, 1e-6, 1e-5, "mean")
    assert_close(lh.grad, lr.grad, 1e-6, 1e-5, "mean d_loc")
    assert_close(mh.grad, mr.grad, 1e-6, 1e-5, "mean d_logits")


# --------------------------------------------------------------------------
# device-resident noise generator (replaces torch.rand_like on the hot path)
# --------------------------------------------------------------------------
def test_uniform_generator_statistics_and_state():
    from torch_scae_amd import ops
    ref = torch.zeros(1, device="cuda")
    torch.manual_seed(1234)
    ops.reset_noise()
    a = ops.uniform(79872, ref)
    b = ops.uniform(79872, ref)
    assert a.shape == (79872,) and a.dtype == torch.float32
    assert float(a.min()) >= 0.0 and float(a.max()) < 1.0
    assert abs(float(a.mean()) - 0.5) < 5e-3
    assert abs(float(a.var()) - 1.0 / 12.0) < 2e-3
    assert not torch.equal(a, b)                 # the state advanced
    # neighbouring draws are uncorrelated
    assert abs(float(((a[:-1] - .5) * (a[1:] - .5)).mean())) < 2e-3
    ops.reset_noise()                            # restart from the torch seed
    assert torch.equal(ops.uniform(79872, ref), a)
    torch.manual_seed(4321)                      # a new seed: a new stream
    assert not torch.equal(ops.uniform(79872, ref), b)
    odd = ops.uniform(5, ref)                    # ragged tail of a group of 4
    assert odd.shape == (5,) and float(odd.max()) < 1.0


def test_uniform_generator_advances_under_graph_replay():
    from torch_scae_amd import ops
    ref = torch.zeros(1, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.uniform(1000, ref)                   # creates the stream's state
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        out = ops.uniform(1000, ref)
    g.replay()
    first = out.clone()
    g.replay()
    torch.cuda.synchronize()
    assert not torch.equal(first, out)
    assert 0.0 <= float(out.min()) and float(out.max()) < 1.0


# --------------------------------------------------------------------------
# the whole part encoder as one autograd node (part_encoder.py:86-113)
# --------------------------------------------------------------------------
@pytest.mark.parametrize("B,C0,HW,chans,strides,A,F,noisy", [
    (128, 1, 40, [128, 128, 128, 128], [2, 2, 1, 1], 24, 16, True),   # cfg-2
    (6, 3, 18, [64, 128], [2, 1], 5, 0, False),
    (9, 1, 14, [64, 64, 64], [1, 2, 1], 7, 3, True),
])
def test_part_encoder_node_vs_fp64_composition(B, C0, HW, chans, strides, A, F,
                                               noisy):
    import torch.nn.functional as Fn
    from torch_scae_amd import ops
    from torch_scae_amd.nn_ext import multiple_attention_pooling_2d
    g = torch.Generator().manual_seed(B + 13 * A)
    P = 6 + 1 + F + 1
    image = torch.rand(B, C0, HW, HW, generator=g)
    ws, bs, cin, size = [], [], C0, HW
    for c, s in zip(chans, strides):
        ws.append(torch.randn(c, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** .5)
        bs.append(torch.randn(c, generator=g) * 0.1)
        cin, size = c, (size - 3) // s + 1
    pb = torch.randn(cin, size, size, generator=g) * 0.5
    aw = torch.randn(A * P, cin, 1, 1, generator=g) / cin ** 0.5
    ab = torch.randn(A * P, generator=g)
    u = torch.rand(B, A, generator=g) if noisy else None
    gouts = [torch.randn(B, A, 6, generator=g), torch.randn(B, A, generator=g),
             torch.randn(B, A, F, generator=g), torch.randn(B, A, F, generator=g)]
    scale = 4.0

    leaves = [t.double().requires_grad_() for t in (pb, aw, ab, *ws, *bs)]
    rpb, raw, rab = leaves[:3]
    rws, rbs = leaves[3:3 + len(ws)], leaves[3 + len(ws):]
    h = image.double()
    for w, b, s in zip(rws, rbs, strides):
        h = torch.relu(Fn.conv2d(h, w, b, stride=s))
    h = multiple_attention_pooling_2d(Fn.conv2d(h + rpb, raw, rab), A) \
        .view(B, A, P - 1)
    pose_r = O.geometric_transform(h[..., :6], similarity=False)
    logit = h[..., 6] + ((u.double() - .5) * scale if noisy else 0.)
    pres_r, feat_r = torch.sigmoid(logit), h[..., 7:]
    loss = (pose_r * gouts[0].double()).sum() + (pres_r * gouts[1].double()).sum()
    if F:   # two consumers of the feature
        loss = loss + (feat_r * (gouts[2] + gouts[3]).double()).sum()
    loss.backward()

    dev = [t.cuda().requires_grad_() for t in (pb, aw, ab, *ws, *bs)]
    pose, pres, feat, twin, absence = ops.part_encoder(
        image.cuda(), dev[3:3 + len(ws)], dev[3 + len(ws):], strides, dev[0],
        dev[1], dev[2], A, u.cuda() if noisy else None, scale, False)
    assert torch.equal(absence, 1. - pres.detach().unsqueeze(-1))
    assert_close(pose, pose_r.float(), rtol=1e-4, atol=1e-4, what="pose")
    assert_close(pres, pres_r.float(), rtol=1e-4, atol=1e-4, what="presence")
    lh = (pose * gouts[0].cuda()).sum() + (pres * gouts[1].cuda()).sum()
    if F:
        assert twin is not feat and torch.equal(twin, feat)
        assert_close(feat, feat_r.float(), rtol=1e-4, atol=1e-4, what="feature")
        lh = lh + (feat * gouts[2].cuda()).sum() + (twin * gouts[3].cuda()).sum()
    else:
        assert feat is None and twin is None
    lh.backward()
    names = ["embedding bias", "att weight", "att bias"] + \
        [f"w{i}" for i in range(len(ws))] + [f"b{i}" for i in range(len(ws))]
    for name, a, r in zip(names, dev, leaves):
        # sums over up to ~1e5 ReLU-gated terms: a pre-activation within
        # round-off of 0 may gate differently in fp32 and fp64
        scale_r = max(1.0, float(r.grad.abs().max()))
        bad = ((a.grad.cpu() - r.grad.float()).abs() > 1e-4 * scale_r)
        assert float(bad.float().mean()) <= 0.02, (name, float(bad.float().mean()))
        assert_close(a.grad, r.grad.float(), rtol=1e-2, atol=1e-2 * scale_r,
                     what="grad " + name)


# --------------------------------------------------------------------------
# K2 bf16 forward (BASELINE.json configs[2]: bf16, MFMA attention path)
# --------------------------------------------------------------------------
@pytest.mark.parametrize("HB,N,M,dk,dv,pres", [
    (64, 64, 48, 256, 256, "mixed"),     # output attention of cfg-3 (O=64, M=48)
    (64, 48, 48, 16, 16, "mixed"),       # SAB of cfg-3
    (5, 33, 17, 70, 130, "mixed"),       # ragged, multi-chunk
    (7, 1, 1, 3, 5, None),               # degenerate
    (3, 72, 80, 32, 48, "mixed"),        # beyond the bf16 kernel's 64-element
                                         # tiles: general fp32 kernel, upcast
])
def test_qkv_attention_bf16_vs_oracle(HB, N, M, dk, dv, pres):
    """bf16 operands on the bf16 matrix cores, fp32 accumulate / softmax.  The
    oracle runs in fp32 on the SAME bf16-rounded inputs, so the differences
    left are the bf16 rounding of P (as the A operand of P V) and of the
    output: tolerance 2^-7 relative to the output scale, written below; the
    attention probabilities themselves (fp32) must agree to 1e-4."""
    from torch_scae_amd import ops
    g = torch.Generator().manual_seed(HB * 1000 + N)
    bf = torch.bfloat16
    q = torch.randn(HB, N, dk, generator=g).to(bf)
    k = torch.randn(HB, M, dk, generator=g).to(bf)
    v = torch.randn(HB, M, dv, generator=g).to(bf)
    w = torch.randn(HB, N, dv, generator=g)
    p = None
    if pres == "mixed":
        p = torch.ones(HB, M)
        p[:, ::3] = torch.rand(HB, len(range(0, M, 3)), generator=g)
        p[0] = 1.0
    qc, kc, vc = (t.float().requires_grad_(True) for t in (q, k, v))
    oo = O.qkv_attention(qc, kc, vc, p)
    (oo * w).sum().backward()

    qg, kg, vg = (t.cuda().requires_grad_(True) for t in (q, k, v))
    import numpy as np
    og = ops.qkv_attention(qg, kg, vg, dev(p))
    assert og.dtype == bf
    # probabilities (saved for the backward pass): fp32 softmax of exact bf16
    # products accumulated in fp32
    ref_p = torch.softmax(
        (qc @ kc.transpose(1, 2) - ((1. - p[:, None, :]) * 1e32 if p is not None
                                    else 0.)) / np.float32(np.sqrt(dk)), -1)
    got_p = og.grad_fn.saved_tensors[3]
    assert got_p.dtype == torch.float32
    assert_close(got_p, ref_p.detach(), 1e-4, 1e-4, "probs (fp32)")
    (og.float() * w.cuda()).sum().backward()
    scale = max(1.0, float(oo.abs().max()))
    tol = 2.0 ** -7
    assert_close(og.float(), oo, tol * scale, tol, "out (bf16)")
    for name, a, b in (("gq", qg.grad, qc.grad), ("gk", kg.grad, kc.grad),
                       ("gv", vg.grad, vc.grad)):
        assert a.dtype == bf
        assert_close(a.float(), b, tol * max(1.0, float(b.abs().max())), tol,
                     name + " (bf16)")


# --------------------------------------------------------------------------
# launch-merging helpers: multi-matrix / transposed / periodic column sums,
# scaled scalar sums (alone and riding in the class-probability launch), the
# one-launch batch hand-over
# --------------------------------------------------------------------------
def test_sum_rows_multi_transpose_period_and_tall_skinny():
    from torch_scae_amd import ops
    g = torch.Generator().manual_seed(3)
    a = torch.randn(128, 25 * 128, generator=g)      # (B, HW*C) -> (C, HW) transposed
    b = torch.randn(3200, 4, generator=g)            # tall and skinny (K1 scalars)
    c = torch.randn(128, 24 * 199, generator=g)      # periodic windows (K3 biases)
    d = torch.randn(7, 33, generator=g)              # few rows
    given = torch.full((128, 5, 5), float("nan"), device="cuda")
    (ta,), (b0, b3), (p0, p1), (dd,) = ops._sum_rows_multi([
        dict(partial=a.cuda(), shapes=[(128, 5, 5)], transpose=128, outs=[given]),
        dict(partial=b.cuda(), shapes=[(1,), (1,)], starts=[0, 3]),
        dict(partial=c.cuda(), shapes=[(1, 24, 1, 6), (1, 24, 24)],
             starts=[144, 151], period=199),
        dict(partial=d.cuda(), shapes=[(33,)])])
    assert ta.data_ptr() == given.data_ptr()
    want_a = a.double().sum(0).view(25, 128).t().reshape(128, 5, 5)
    assert_close(ta, want_a.float(), 2e-4, 1e-5, "transposed scatter")
    sb = b.double().sum(0)
    assert_close(b0, sb[0:1].float(), 1e-3, 1e-5, "tall-skinny col 0")
    assert_close(b3, sb[3:4].float(), 1e-3, 1e-5, "tall-skinny col 3")
    sc = c.double().sum(0).view(24, 199)
    assert_close(p0, sc[:, 144:150].reshape(1, 24, 1, 6).float(), 2e-4, 1e-5,
                 "periodic window 0")
    assert_close(p1, sc[:, 151:175].reshape(1, 24, 24).float(), 2e-4, 1e-5,
                 "periodic window 1")
    assert_close(dd, d.double().sum(0).float(), 1e-5, 1e-5, "few rows")


def test_scaled_sums_alone_and_riding_in_class_probs():
    from torch_scae_amd import ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn(128, 24, generator=g)
    y = torch.randn(3072, generator=g)
    jobs = lambda: [(x.cuda(), 1.0 / 128, torch.empty((), device="cuda")),   # noqa: E731
                    (y.cuda(), 0.5 / 128, torch.empty((), device="cuda"))]
    alone = jobs()
    ops.scaled_sums(alone)
    assert_close(alone[0][2], (x.double().sum() / 128).float(), 1e-5, 1e-5, "x")
    assert_close(alone[1][2], (y.double().sum() * 0.5 / 128).float(), 1e-5, 1e-5, "y")
    # the same jobs as riders of the class-probability kernel
    cp = torch.rand(128, 24, generator=g)
    post = torch.rand(128, 25, 24, generator=g)
    w, b = torch.randn(10, 24, generator=g), torch.randn(10, generator=g)
    riders = jobs()
    prior, posterior = ops.class_probs(cp.cuda(), post.cuda(), w.cuda(), b.cuda(),
                                       extra_sums=riders)
    assert torch.equal(riders[0][2].cpu().reshape(()), alone[0][2].cpu().reshape(())) \
        or abs(float(riders[0][2]) - float(alone[0][2])) <= 1e-6
    assert_close(riders[1][2], alone[1][2], 1e-6, 1e-5, "rider y")
    assert_close(prior, torch.softmax(cp @ w.t() + b, -1), 1e-5, 1e-4, "prior")
    assert_close(posterior, torch.softmax(post[:, :-1].sum(-1) @ w.t() + b, -1),
                 1e-5, 1e-4, "posterior")


def test_launch_list_replays_recorded_launches():
    """scae_launch_list_*: the launches two entry points make on the recording's stream are
    re-issued by scae_launch_list_run on another stream with the recorded arguments -- the
    outputs reappear after being wiped -- and nothing is recorded after end.  A second
    recording open at the same time on ANOTHER stream sees only that stream's launches (two
    steps capturing in one process do not pollute each other's list)."""
    import ctypes
    from torch_scae_amd import _lib
    lib = _lib.load()
    P = ctypes.c_void_p
    g = torch.Generator().manual_seed(7)
    pose = torch.randn(300, 6, generator=g).cuda()
    out = torch.empty(300, 6, device="cuda")
    out_b = torch.empty(300, 6, device="cuda")
    src = torch.rand(1000, generator=g).cuda()
    lab = torch.randint(0, 10, (16,), generator=g).cuda()
    u = torch.empty(1000, device="cuda")
    lab2 = torch.empty(16, dtype=torch.long, device="cuda")
    torch.cuda.synchronize()
    other = torch.cuda.Stream()
    st = P(torch.cuda.current_stream().cuda_stream)
    st_b = P(other.cuda_stream)
    lst = lib.scae_launch_list_begin(st)
    lst_b = lib.scae_launch_list_begin(st_b)
    assert lst and lst_b
    _lib.call("scae_geometric_transform_fwd_f32", P(pose.data_ptr()), P(out.data_ptr()), 300,
              0, 1, 0, st)
    _lib.call("scae_geometric_transform_fwd_f32", P(pose.data_ptr()), P(out_b.data_ptr()),
              300, 0, 1, 0, st_b)                # (the other recording's stream)
    _lib.call("scae_stage_batch", P(u.data_ptr()), P(src.data_ptr()), 1000,
              P(lab2.data_ptr()), P(lab.data_ptr()), 16, st)
    assert lib.scae_launch_list_end(P(lst)) == 0
    assert lib.scae_launch_list_size(P(lst)) == 2
    assert lib.scae_launch_list_size(P(lst_b)) == 1
    _lib.call("scae_geometric_transform_fwd_f32", P(pose.data_ptr()), P(out.data_ptr()), 300,
              0, 1, 0, st)                       # (not recorded)
    assert lib.scae_launch_list_size(P(lst)) == 2
    assert lib.scae_launch_list_end(P(lst_b)) == 0
    torch.cuda.synchronize()
    ref_out = out.clone()
    assert torch.equal(u, src) and torch.equal(lab2, lab) and torch.equal(out_b, ref_out)
    out.zero_()
    out_b.zero_()
    u.zero_()
    lab2.zero_()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        assert lib.scae_launch_list_run(P(lst), P(side.cuda_stream)) == 0
    side.synchronize()
    lib.scae_launch_list_free(P(lst))
    assert torch.equal(out, ref_out) and torch.equal(u, src) and torch.equal(lab2, lab)
    assert float(out_b.abs().max()) == 0         # (list A holds nothing of stream B)
    assert lib.scae_launch_list_run(P(lst_b), st) == 0
    torch.cuda.synchronize()
    lib.scae_launch_list_free(P(lst_b))
    assert torch.equal(out_b, ref_out)
    assert float(ref_out.abs().max()) > 0


def test_stage_batch_one_launch():
    import ctypes
    from torch_scae_amd import _lib
    g = torch.Generator().manual_seed(5)
    for shape in ((128, 1, 40, 40), (5, 3, 7, 9)):       # vector and scalar-tail paths
        image = torch.rand(*shape, generator=g).cuda()
        label = torch.randint(0, 10, (shape[0],), generator=g).cuda()
        di = torch.full(shape, float("nan"), device="cuda")
        dl = torch.full((shape[0],), -1, dtype=torch.long, device="cuda")
        P = ctypes.c_void_p
        _lib.call("scae_stage_batch", P(di.data_ptr()), P(image.data_ptr()),
                  image.numel(), P(dl.data_ptr()), P(label.data_ptr()), label.numel(),
                  P(torch.cuda.current_stream().cuda_stream))
        assert torch.equal(di, image) and torch.equal(dl, label)


@pytest.mark.parametrize("B,G,Kin,dims,ones_at", [
    (128, 24, 256, [128, 32, 128, 199], 2),    # cfg-2's capsule MLPs
    (37, 3, 50, [37, 12, 29, 45], 2),          # ragged everything, scalar weight loads
    (16, 2, 64, [8, 4, 8, 31], 2),             # the smoke model's widths
    (70, 5, 96, [64, 40], None),               # a two-layer chain, no ones column
    (40, 2, 300, [200, 70, 33, 391], 2),       # 3 k-chunks, several tiles per wave, cfg-3's width
    (20, 3, 704, [130, 7], None),              # the widest input the kernel takes
])
@pytest.mark.parametrize("row_tile", [16, 32])
def test_mlp_chain_vs_fp64(B, G, Kin, dims, ones_at, row_tile, monkeypatch):
    """K7b (object_decoder.py:86-107, :137-158, the cat with caps_exist of
    :149): all layers of the per-capsule MLP chain in one launch, and its
    backward (data-gradient chain + one weight-gradient launch), against fp64
    algebra; bar 1e-4 relative to each tensor's largest entry.  ``row_tile``:
    batch rows per workgroup -- 32 is what large batches take (the activation
    buffers then are as wide as what each holds), forced here at every size."""
    from torch_scae_amd import ops
    monkeypatch.setattr(ops, "_CHAIN_ROW_TILE", row_tile)
    g = torch.Generator().manual_seed(B + G + Kin)
    layers, K = [], Kin
    for l, N in enumerate(dims):
        ones = ones_at is not None and l == ones_at
        w = torch.randn(G, N, K + (1 if ones else 0), generator=g) / K ** 0.5
        b = None if (ones_at is not None and l >= ones_at) \
            else torch.randn(G, N, generator=g) * 0.1
        layers.append((w, b, ones))
        K = N
    x = torch.randn(B, G, Kin, generator=g)
    R = torch.randn(B, G, dims[-1], generator=g)

    def ref(x, layers):
        h, near = x.transpose(0, 1), None                  # (G, B, K)
        for w, b, ones in layers:
            if ones:
                h = torch.cat([h, torch.ones_like(h[..., :1])], -1)
            h = torch.bmm(h, w.transpose(1, 2))
            if b is not None:
                h = h + b.unsqueeze(1)
            # ReLU gates within fp32 round-off of zero: their (b, g) rows are
            # taken out of the gradient comparison (either gate is a valid fp32
            # evaluation)
            n = (h.detach().abs() < 1e-5 * float(h.detach().abs().max())).any(-1)
            near = n if near is None else near | n
            h = torch.relu(h)
        return h.transpose(0, 1), near.transpose(0, 1)

    xr = x.double().requires_grad_()
    lr = [(w.double().requires_grad_(),
           None if b is None else b.double().requires_grad_(), o)
          for w, b, o in layers]
    yr, near = ref(xr, lr)
    assert float(near.float().mean()) < 0.2
    R = R * (~near).unsqueeze(-1)
    (yr * R.double()).sum().backward()

    xh = x.cuda().requires_grad_()
    lh = [(w.cuda().requires_grad_(),
           None if b is None else b.cuda().requires_grad_(), o)
          for w, b, o in layers]
    assert ops.mlp_chain_supported(xh, lh)
    yh = ops.mlp_chain(xh, lh)
    assert yh.shape == (B, G, dims[-1])
    assert_close(yh, yr.float(), rtol=1e-4,
                 atol=1e-4 * float(yr.abs().max()), what="y")
    # the consumer's contract: gradient w.r.t. the last PRE-activation
    yh.backward(R.cuda() * (yh.detach() > 0))
    assert_close(xh.grad, xr.grad.float(), rtol=1e-4,
                 atol=1e-4 * float(xr.grad.abs().max()), what="gx")
    for l, ((wh, bh, _), (wr, br, _)) in enumerate(zip(lh, lr)):
        assert_close(wh.grad, wr.grad.float(), rtol=1e-4,
                     atol=1e-4 * float(wr.grad.abs().max()), what=f"gw{l}")
        if bh is not None:
            assert_close(bh.grad, br.grad.float(), rtol=1e-4,
                         atol=1e-4 * float(br.grad.abs().max()), what=f"gb{l}")
    # and it is the same function as the layer-by-layer GEMM path
    if ones_at == 2 and len(dims) == 4:
        x2 = x.cuda()
        h = ops.grouped_mlp(x2, [w.cuda() for w, _, _ in layers[:2]],
                            [b.cuda() for _, b, _ in layers[:2]])
        y2 = ops.grouped_mlp(h, [w.cuda() for w, _, _ in layers[2:]], None,
                             ones_input=True)
        assert_close(yh, y2, rtol=1e-5, atol=1e-5 * float(yr.abs().max()),
                     what="vs K7")


@pytest.mark.parametrize("B,O,V,noise,sim", [(128, 24, 24, True, True),
                                              (37, 5, 7, False, False),
                                              (16, 4, 5, True, True),
                                              (70, 6, 48, True, False)])
@pytest.mark.parametrize("row_tile", [16, 32])
def test_chain_votes_matches_chain_then_votes(B, O, V, noise, sim, row_tile,
                                              monkeypatch):
    """K7b + K3 in one launch (forward) / K3-backward + data-gradient chain in one
    launch (backward) against the same two ops launched separately: every
    output and every gradient, 1e-5 relative to the tensor's largest entry
    (the only difference is the order of two small sums).  ``row_tile`` as in
    test_mlp_chain_vs_fp64: with 32 rows the vote blocks run twice per
    workgroup, their scratch in the weight tiles."""
    from torch_scae_amd import ops
    monkeypatch.setattr(ops, "_CHAIN_ROW_TILE", row_tile)
    g = torch.Generator().manual_seed(B * O + V)
    Kin, H, Dc, A = 64, 48, 12, 8 * V + 7
    dims = [(H, Kin, True, False), (Dc, H, True, False), (H, Dc + 1, False, True),
            (A, H, False, False)]
    vals = dict(
        x=torch.randn(B, O, Kin, generator=g),
        cpr=torch.randn(1, O, V, 6, generator=g) * 0.3,
        b_cvr=torch.randn(1, O, 1, 6, generator=g) * 0.3,
        b_caps=torch.randn(1, O, 1, generator=g),
        b_vote=torch.randn(1, O, V, generator=g),
        b_scale=torch.randn(1, O, V, generator=g))
    ws = [torch.randn(O, n, k, generator=g) / k ** 0.5 for n, k, _, _ in dims]
    bs = [torch.randn(O, n, generator=g) * 0.1 if hb else None
          for n, _, hb, _ in dims]
    nz = [torch.rand(B, O, 1, generator=g).cuda(),
          torch.rand(B, O, V, generator=g).cuda()] if noise else [None, None]
    gouts = [torch.randn(B, O, V, 6, generator=g), torch.randn(B, O, V, generator=g),
             torch.randn(B, O, V, generator=g), torch.randn(B, O, 1, generator=g),
             torch.randn(B, O, V, generator=g), torch.randn((), generator=g),
             torch.randn(B, O, generator=g)]

    def run(fused):
        t = {k: v.cuda().requires_grad_() for k, v in vals.items()}
        w = [x.cuda().requires_grad_() for x in ws]
        b = [None if x is None else x.cuda().requires_grad_() for x in bs]
        layers = [(w[i], b[i], dims[i][3]) for i in range(4)]
        kw = dict(noise_caps=nz[0], noise_vote=nz[1], noise_scale=4.0,
                  similarity=sim, learn_vote_scale=True, allow_deformations=True)
        if fused:
            outs = ops.chain_votes(t["x"], layers, t["cpr"], t["b_cvr"],
                                   t["b_caps"], t["b_vote"], t["b_scale"], **kw)
        else:
            ap = ops.mlp_chain(t["x"], layers)
            outs = ops.capsule_votes(ap, t["cpr"], t["b_cvr"], t["b_caps"],
                                     t["b_vote"], t["b_scale"],
                                     param_is_relu=True, **kw)
        torch.autograd.backward(list(outs[:7]), [go.cuda() for go in gouts])
        grads = [t[k].grad for k in vals] + [x.grad for x in w] + \
            [x.grad for x in b if x is not None]
        return [o.detach() for o in outs], grads

    o1, g1 = run(True)
    o0, g0 = run(False)
    for i, (a, b_) in enumerate(zip(o1, o0)):
        assert_close(a, b_, rtol=1e-5, atol=1e-5 * float(b_.abs().max()) + 1e-7,
                     what=f"out{i}")
    for i, (a, b_) in enumerate(zip(g1, g0)):
        assert_close(a, b_, rtol=1e-5, atol=1e-5 * float(b_.abs().max()) + 1e-7,
                     what=f"grad{i}")


@pytest.mark.parametrize("shape,N,bias", [((7, 24, 16), 48, True), ((128, 24, 256), 256, True),
                                          ((5, 3, 50), 17, False)])
def test_hip_linear_vs_fp64(shape, N, bias):
    """ops.HipLinear (the projections / feed-forward layers of the module-by-module
    set-transformer blocks, set_transformer.py:56-133) on K7 against fp64."""
    from torch_scae_amd import ops
    g = torch.Generator().manual_seed(sum(shape) + N)
    K = shape[-1]
    lin = ops.HipLinear(K, N, bias=bias)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(N, K, generator=g) / K ** 0.5)
        if bias:
            lin.bias.copy_(torch.randn(N, generator=g))
    x = torch.randn(*shape, generator=g)
    gy = torch.randn(*shape[:-1], N, generator=g)
    xr = x.double().requires_grad_()
    wr = lin.weight.detach().double().requires_grad_()
    br = lin.bias.detach().double().requires_grad_() if bias else None
    yr = torch.nn.functional.linear(xr, wr, br)
    yr.backward(gy.double())
    lin = lin.cuda()
    xh = x.cuda().requires_grad_()
    yh = lin(xh)
    yh.backward(gy.cuda())
    assert_close(yh, yr.float(), 1e-4, 1e-4 * float(yr.abs().max()), "y")
    assert_close(xh.grad, xr.grad.float(), 1e-4, 1e-4 * float(xr.grad.abs().max()), "gx")
    assert_close(lin.weight.grad, wr.grad.float(), 1e-4, 1e-4 * float(wr.grad.abs().max()), "gw")
    if bias:
        assert_close(lin.bias.grad, br.grad.float(), 1e-4, 1e-4 * float(br.grad.abs().max()), "gb")


@pytest.mark.parametrize("shape,affine", [((7, 24, 16), True), ((128, 24, 256), True),
                                          ((3, 5, 70), True), ((2, 1000), False)])
def test_hip_layer_norm_vs_fp64(shape, affine):
    """ops.HipLayerNorm (nn.LayerNorm(d) of MAB, set_transformer.py:114-131)
    against fp64, outputs and all three gradients."""
    from torch_scae_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    d = shape[-1]
    ln = ops.HipLayerNorm(d, elementwise_affine=affine)
    if affine:
        with torch.no_grad():
            ln.weight.copy_(torch.randn(d, generator=g))
            ln.bias.copy_(torch.randn(d, generator=g))
    x = torch.randn(*shape, generator=g) * 3 + 1
    gy = torch.randn(*shape, generator=g)
    xr = x.double().requires_grad_()
    wr = ln.weight.detach().double().requires_grad_() if affine else None
    br = ln.bias.detach().double().requires_grad_() if affine else None
    yr = torch.nn.functional.layer_norm(xr, (d,), wr, br, ln.eps)
    yr.backward(gy.double())
    ln = ln.cuda()
    xh = x.cuda().requires_grad_()
    yh = ln(xh)
    yh.backward(gy.cuda())
    assert_close(yh, yr.float(), 1e-4, 1e-4 * float(yr.abs().max()), "y")
    assert_close(xh.grad, xr.grad.float(), 1e-4, 1e-4 * float(xr.grad.abs().max()), "gx")
    if affine:
        assert_close(ln.weight.grad, wr.grad.float(), 1e-4,
                     1e-4 * float(wr.grad.abs().max()), "gw")
        assert_close(ln.bias.grad, br.grad.float(), 1e-4,
                     1e-4 * float(br.grad.abs().max()), "gb")


def test_step_prologue_matches_the_three_launches():
    """scae_step_prologue_f32 = scae_stage_batch + scae_uniform_f32 +
    scae_seed_fold_fwd_f32 in one launch: bit-identical outputs, and the noise
    generator advances exactly as a scae_uniform_f32 launch advances it."""
    from torch_scae_amd import ops
    g = torch.Generator().manual_seed(11)
    O, C, D = 24, 256, 16
    shapes = [(O, C), (C, C), (C,), (C, C), (C,), (C, C), (C,), (C, C), (C,),
              (C, D), (C,)]
    vals = [(torch.randn(*s, generator=g) / (s[-1] ** 0.5)).cuda() for s in shapes]
    image = torch.rand(128, 1, 40, 40, generator=g).cuda()
    label = torch.randint(0, 10, (128,), generator=g).cuda()
    n = 128 * 24 + 128 * 24 + 128 * 24 * 24
    torch.manual_seed(1234)
    ops.reset_noise()
    ref_noise = [ops.uniform(n, image).clone() for _ in range(3)]
    ref_fold = ops.seed_fold(*vals)

    torch.manual_seed(1234)
    ops.reset_noise()
    pro = ops.StepPrologue()
    di, dl = torch.zeros_like(image), torch.zeros_like(label)
    with ops.step_prologue(pro):
        first = ops.uniform(n, image)            # establishes the buffer, launches itself
        assert torch.equal(first, ref_noise[0])
        fold0 = ops.seed_fold(*vals)             # likewise
        for a, b in zip(fold0, ref_fold):
            assert torch.equal(a, b)
        for a in pro.fold_outs:
            a.fill_(float("nan"))
        pro.launch(di, image, dl, label)
        assert torch.equal(di, image) and torch.equal(dl, label)
        second = ops.uniform(n, image)           # no launch: the prologue's draw
        assert second.data_ptr() == pro.noise.data_ptr()
        assert torch.equal(second, ref_noise[1])
        fold1 = ops.seed_fold(*vals)
        for a, b in zip(fold1, ref_fold):
            assert torch.equal(a, b)
        third = ops.uniform(n, image)            # consumed: launches again
        assert torch.equal(third, ref_noise[2])
        # gradients flow through the prologue-filled outputs as usual
        vin = [v.clone().requires_grad_() for v in vals]
        pro2 = ops.StepPrologue()
        with ops.step_prologue(pro2):
            ops.seed_fold(*vin)
            pro2.launch()
            outs = ops.seed_fold(*vin)
        ref_in = [v.clone().requires_grad_() for v in vals]
        ref_outs = ops.seed_fold(*ref_in)
        gouts = [torch.randn(o.shape, generator=g).cuda() for o in outs]
        torch.autograd.backward(outs, gouts)
        torch.autograd.backward(ref_outs, gouts)
        for a, b in zip(vin, ref_in):
            assert torch.equal(a.grad, b.grad)
    # parts can be left out
    pro3 = ops.StepPrologue()
    pro3.launch()                                # nothing to do: no launch, no error
    pro3.launch(di.zero_(), image, dl.zero_(), label)
    assert torch.equal(di, image) and torch.equal(dl, label)


@pytest.mark.gpu
@pytest.mark.parametrize("C0", [1, 3])
def test_step_prologue_runs_the_image_layer(C0):
    """The encoder's image layer + filter re-layouts as a fourth job of the
    prologue launch (scae_step_prologue_first_f32): the conv stack then finds its
    first activation and the re-laid filters already there -- bit-identical to
    the stack's own launch -- reads the batch at the hand-over's SOURCE, and
    gradients are unchanged."""
    from torch_scae_amd import ops
    g = torch.Generator().manual_seed(5)
    B, H = 16, 20
    chans, strides = [C0, 64, 128, 64], (2, 1, 1)
    ws = [(torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)).cuda()
          for ci, co in zip(chans[:-1], chans[1:])]
    bs = [(0.1 * torch.randn(co, generator=g)).cuda() for co in chans[1:]]
    batches = [torch.rand(B, C0, H, H, generator=g).cuda() for _ in range(3)]
    label = torch.zeros(B, dtype=torch.int64).cuda()
    resident, rl = torch.zeros_like(batches[0]), torch.zeros_like(label)

    def run(image, wlist, blist):
        return ops.conv_stack(image, wlist, blist, strides)

    refs = [run(x, ws, bs).clone() for x in batches]
    pro = ops.StepPrologue()
    with ops.step_prologue(pro):
        resident.copy_(batches[0])
        y0 = run(resident, ws, bs)            # registers the layer, launches itself
        assert torch.equal(y0, refs[0]) and pro.first_outs is not None
        for t in (pro.first_outs[0], *pro.first_outs[1], *pro.first_outs[2]):
            t.fill_(float("nan"))
        pro.launch(resident, batches[1], rl, label)   # hand-over + image layer
        assert torch.equal(resident, batches[1]) and pro.first_fresh
        y1 = run(resident, ws, bs)            # no image-layer launch of its own
        assert not pro.first_fresh and torch.equal(y1, refs[1])
        resident.copy_(batches[2])
        y2 = run(resident, ws, bs)            # consumed: launches again
        assert torch.equal(y2, refs[2])
        pro.launch()                          # refresh without a batch: the resident image
        assert pro.first_fresh
        assert torch.equal(run(resident, ws, bs), refs[2])
        # another image tensor: not the registered layer, its own launch
        pro.launch()
        assert torch.equal(run(batches[0], ws, bs), refs[0])
        # gradients through the prologue-filled activation
        win = [w.clone().requires_grad_() for w in ws]
        bin_ = [b.clone().requires_grad_() for b in bs]
        pro2 = ops.StepPrologue()
        with ops.step_prologue(pro2):
            run(resident, win, bin_)
            pro2.launch()
            y = run(resident, win, bin_)
        wref = [w.clone().requires_grad_() for w in ws]
        bref = [b.clone().requires_grad_() for b in bs]
        yr = run(resident, wref, bref)
        gy = torch.randn(y.shape, generator=g).cuda()
        y.backward(gy)
        yr.backward(gy)
        for a, b in zip(win + bin_, wref + bref):
            assert torch.equal(a.grad, b.grad)


# --------------------------------------------------------------------------
# configs[2]'s precision: bf16 operands / fp32 accumulation on K7 and K8
# (ops.mfma_bf16), against fp64 at bf16's bar
# --------------------------------------------------------------------------
def _rel_l2(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).norm() / (ref.norm() + 1e-30))


@pytest.mark.parametrize("resident", [True, False])
@pytest.mark.parametrize("B,HW,chans,strides", [
    (256, 40, [128, 128, 128], [2, 2, 1]),     # 128-channel layers: bf16 tiles
    (64, 24, [128, 128], [1, 1]),
    (24, 29, [128, 128, 128, 128], [2, 2, 1, 1]),   # ragged tiles, every class of the stride-2 gradient
])
def test_conv_stack_bf16_operands_vs_fp64(B, HW, chans, strides, resident, monkeypatch):
    """K8 with bf16 operands (v_mfma_f32_32x32x16_bf16, fp32 accumulate): outputs
    and weight / bias gradients against conv2d in fp64.  Bar: every operand is
    rounded to 8 significant bits (relative 2^-9), a K-long dot product of
    such terms is off by ~2^-9 / sqrt(K) x its term scale: relative L2 error
    <= 1e-2 on the outputs, every entry within 2^-6 of the largest; gradients
    additionally see the ReLU gates that bf16's rounding flips (pre-activations
    within ~2^-9 of zero, a few tenths of a per cent of the units per layer):
    <= 8e-2 relative L2 per tensor, the lower layers' being the worst."""
    import torch.nn.functional as F
    from torch_scae_amd import ops
    # resident: the operands of layers 1.. kept as bf16 in HBM (csrc/conv_bf16.hip); else the
    # first form, fp32 tensors rounded on their way into LDS -- the same rounded operands
    monkeypatch.setattr(ops, "_CONV_BF16R", resident)
    g = torch.Generator().manual_seed(B + HW)
    image = torch.rand(B, 1, HW, HW, generator=g)
    ws, bs, cin = [], [], 1
    for c in chans:
        bound = 1.0 / (cin * 9) ** 0.5
        ws.append((torch.rand(c, cin, 3, 3, generator=g) * 2 - 1) * bound)
        bs.append((torch.rand(c, generator=g) * 2 - 1) * bound)
        cin = c
    w64 = [t.double().requires_grad_() for t in ws]
    b64 = [t.double().requires_grad_() for t in bs]
    y_ref = image.double()
    for wi, bi, s in zip(w64, b64, strides):
        y_ref = F.relu(F.conv2d(y_ref, wi, bi, stride=s))
    wg, bg = [leaf(t) for t in ws], [leaf(t) for t in bs]
    calls = []
    real = ops._lib.call

    def spy(name, *a):
        calls.append(name)
        return real(name, *a)
    ops._lib.call = spy
    try:
        with ops.mfma_bf16():
            y = ops.conv_stack(image.cuda(), wg, bg, strides)
            gy = torch.randn(y_ref.shape, generator=g)
            y.backward(gy.cuda())
    finally:
        ops._lib.call = real
    if resident:
        assert {"scae_conv3x3_fwd_bf16r", "scae_conv3x3_dgrad_bf16r",
                "scae_conv3x3_wgrad_bf16r"} <= set(calls), calls
        assert "scae_conv3x3_fwd_bf16" not in calls
    elif B >= 64:    # (the first form needs a few 128 x 128 tiles per CU)
        assert "scae_conv3x3_fwd_bf16" in calls and "scae_conv3x3_bwd_pair_bf16" in calls
    y_ref.backward(gy.double())
    assert _rel_l2(y, y_ref) <= 1e-2
    assert float((y.cpu().double() - y_ref).abs().max()) <= 2 ** -6 * float(y_ref.abs().max())
    for l in range(len(chans)):
        for name, got, ref in (("dW", wg[l].grad, w64[l].grad),
                               ("db", bg[l].grad, b64[l].grad)):
            assert _rel_l2(got, ref) <= 8e-2, (name, l, _rel_l2(got, ref))


@pytest.mark.parametrize("B,IH,IW,stride", [
    (2, 9, 9, 1),        # one partial tile
    (3, 9, 11, 2),       # every tap-set class of the stride-2 data gradient, ragged
    (2, 10, 12, 2),      # ... with a last row / column that no tap reaches (zero gradient)
    (5, 19, 19, 2),      # cfg-3's second layer
    (130, 5, 5, 1),      # several tiles, the last one ragged; 2 weight-gradient splits
])
def test_conv_bf16_resident_kernels_vs_fp64(B, IH, IW, stride):
    """csrc/conv_bf16.hip, each pass on its own through the C ABI: operands that ARE bf16
    (so nothing is rounded on the way in), products on v_mfma_f32_32x32x16_bf16 with fp32
    accumulation -- the fp32 outputs (forward: out_f / out_post; data gradient: din_f; weight
    gradient: the summed partial slabs and the bias partials) against conv2d in fp64 at fp32
    round-off (1e-5 of the tensor's largest entry), the bf16 outputs at bf16's (2^-8)."""
    import ctypes
    import torch.nn.functional as F
    from torch_scae_amd import _lib
    lib = _lib.load()
    P = ctypes.c_void_p
    C, s = 128, stride
    g = torch.Generator().manual_seed(B * 100 + IH)
    OH, OW = (IH - 3) // s + 1, (IW - 3) // s + 1
    st = P(torch.cuda.current_stream().cuda_stream)
    p = lambda t: None if t is None else P(t.data_ptr())
    x = torch.relu(torch.randn(B, IH, IW, C, generator=g)).to(torch.bfloat16).cuda()
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).to(torch.bfloat16).cuda()   # (co,ci,3,3)
    bias, post = torch.randn(C, generator=g).cuda(), torch.randn(C, OH, OW, generator=g).cuda()
    wf = w.permute(0, 2, 3, 1).contiguous()       # (co, 9, ci)
    wd = w.permute(1, 2, 3, 0).contiguous()       # (ci, 9, co)
    assert lib.scae_conv3x3_bf16r_supported(B, IH, IW, C, C, s) == 1
    assert lib.scae_conv3x3_bf16r_supported(B, IH, IW, 64, C, s) == 0
    out_h = torch.empty(B, OH, OW, C, device="cuda", dtype=torch.bfloat16)
    out_f, out_p = torch.empty(B, OH, OW, C, device="cuda"), torch.empty(B, OH, OW, C, device="cuda")
    _lib.call("scae_conv3x3_fwd_bf16r", p(x), p(wf), p(bias), p(out_h), p(out_f), p(post), p(out_p),
              B, IH, IW, C, C, s, st)
    x64 = x.double().permute(0, 3, 1, 2).cpu()
    ref = torch.relu(F.conv2d(x64, w.double().cpu(), bias.double().cpu(), stride=s)).permute(0, 2, 3, 1)
    top = float(ref.abs().max())
    assert float((out_f.double().cpu() - ref).abs().max()) <= 1e-5 * top
    assert float((out_h.double().cpu() - ref).abs().max()) <= 2 ** -8 * top
    assert float((out_p.double().cpu() - ref - post.double().cpu().permute(1, 2, 0)).abs().max()) \
        <= 1e-5 * (top + float(post.abs().max()))
    # data gradient, gated by x > 0
    dpre = torch.randn(B, OH, OW, C, generator=g).to(torch.bfloat16).cuda()
    din_h = torch.empty(B, IH, IW, C, device="cuda", dtype=torch.bfloat16)
    din_f = torch.full((B, IH, IW, C), 7.0, device="cuda")
    _lib.call("scae_conv3x3_dgrad_bf16r", p(dpre), p(wd), p(x), p(din_h), p(din_f), B, IH, IW, C, C,
              s, st)
    xr = x64.clone().requires_grad_(True)
    F.conv2d(xr, w.double().cpu(), None, stride=s).backward(dpre.double().permute(0, 3, 1, 2).cpu())
    dref = (xr.grad * (xr.detach() > 0)).permute(0, 2, 3, 1)
    top = float(dref.abs().max())
    assert float((din_f.double().cpu() - dref).abs().max()) <= 1e-5 * top
    assert float((din_h.double().cpu() - dref).abs().max()) <= 2 ** -8 * top
    # weight gradient: split partials in the fp32 kernels' layout
    splits = lib.scae_conv3x3_wgrad_bf16r_splits(B, OH, OW, C, C)
    assert splits >= 1
    part = torch.full((splits * (9 * C * C + C),), 3.0, device="cuda")
    _lib.call("scae_conv3x3_wgrad_bf16r", p(dpre), p(x), p(part), B, IH, IW, C, C, s, st)
    torch.cuda.synchronize()
    dw = part[:splits * 9 * C * C].view(splits, 9, C, C).sum(0).double().cpu()     # (tap, co, ci)
    db = part[splits * 9 * C * C:].view(splits, C).sum(0).double().cpu()
    wr = w.double().cpu().requires_grad_(True)
    F.conv2d(x64, wr, None, stride=s).backward(dpre.double().permute(0, 3, 1, 2).cpu())
    wref = wr.grad.permute(2, 3, 0, 1).reshape(9, C, C)
    assert float((dw - wref).abs().max()) <= 1e-5 * float(wref.abs().max())
    bref = dpre.double().cpu().sum((0, 1, 2))
    assert float((db - bref).abs().max()) <= 1e-5 * float(dpre.double().abs().sum((0, 1, 2)).max())
    # the batched fp32 -> bf16 copy: round to nearest even, like torch's
    a, b = torch.randn(1000 * 8, generator=g).cuda(), torch.randn(24, generator=g).cuda()
    ah = torch.empty(a.shape, device="cuda", dtype=torch.bfloat16)
    bh = torch.empty(b.shape, device="cuda", dtype=torch.bfloat16)
    _lib.call("scae_cvt_bf16_batch", 2, (P * 2)(a.data_ptr(), b.data_ptr()),
              (P * 2)(ah.data_ptr(), bh.data_ptr()), (ctypes.c_int64 * 2)(a.numel(), b.numel()), st)
    torch.cuda.synchronize()
    assert torch.equal(ah, a.to(torch.bfloat16)) and torch.equal(bh, b.to(torch.bfloat16))


@pytest.mark.parametrize("B,HW,C,AP,gsz", [
    (128, 25, 128, 576, 32),    # cfg-2's capsule head: K = 576 and 800
    (32, 25, 64, 128, 8),       # a ragged last K chunk of the k-strided pair (K = 200)
    (64, 9, 128, 160, 32),      # five K chunks: waves with one and with two
])
def test_gemm_pair_on_wave_private_pipelines_vs_fp64(B, HW, C, AP, gsz, monkeypatch):
    """csrc/gemm_ksplit.hip through scae_gemm_pair_f32: the backward of the 1 x 1 attention
    convolution -- dx = dy W with the ReLU gate and the ungated copy (A k-contiguous), dW per
    group of images = dy^T x with its bias sums (both operands k-strided, batch > 1) -- against
    fp64 at 2e-6 of each result's largest entry, and against the register-staged tiles
    (SCAE_GEMM_KSPLIT=0).  The fast form must actually take these shapes: the two runs differ
    somewhere in the last bits."""
    import ctypes
    from torch_scae_amd import _lib, ops
    g = torch.Generator().manual_seed(B + HW)
    S, kper, slab = B // gsz, HW * gsz, AP * C + AP
    x = torch.randn(B, HW, C, generator=g).cuda()
    gate = torch.relu(torch.randn(B, HW, C, generator=g)).cuda()
    w = (torch.randn(AP, C, generator=g) / C ** 0.5).cuda()
    dy = torch.randn(B, HW, AP, generator=g).cuda()
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SCAE_GEMM_KSPLIT", mode)
        part = torch.full((S, slab), 7.0, device="cuda")
        dx, raw = torch.full_like(x, 7.0), torch.full_like(x, 7.0)
        dgrad = ops._gemm_desc(ops._p(dy), ops._p(w), ops._p(dx), 1, B * HW, C, AP, True, AP, 0,
                               False, C, 0, C, 0)
        dgrad.mask, dgrad.ldmask, dgrad.c_nomask = gate.data_ptr(), C, raw.data_ptr()
        wgrad = ops._gemm_desc(ops._p(dy), ops._p(x), ops._p(part), S, AP, C, kper, False, AP,
                               kper * AP, False, C, kper * C, C, slab,
                               asum=ops._off(part, AP * C), asum_b=slab)
        ops._gemm_pair(wgrad, dgrad, x)
        torch.cuda.synchronize()
        outs[mode] = (dx, raw, part)
    dy64, x64, w64 = dy.double().cpu(), x.double().cpu(), w.double().cpu()
    raw_ref = dy64 @ w64
    dx_ref = raw_ref * (gate.cpu() > 0)
    dw_ref = torch.einsum("skm,skn->smn", dy64.view(S, kper, AP), x64.view(S, kper, C))
    db_ref = dy64.view(S, kper, AP).sum(1)
    dx, raw, part = outs["1"]
    for got, ref in ((dx, dx_ref), (raw, raw_ref), (part[:, :AP * C].view(S, AP, C), dw_ref),
                     (part[:, AP * C:], db_ref)):
        assert float((got.double().cpu() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    assert not all(torch.equal(a, b) for a, b in zip(outs["1"], outs["0"]))
    for a, b in zip(outs["1"], outs["0"]):
        assert float((a - b).abs().max()) <= 4e-6 * float(b.abs().max())


def test_grouped_mlp_bf16_operands_vs_fp64():
    """K7 (the per-capsule MLPs as batched GEMMs) with bf16 operands: forward,
    input gradient and weight / bias gradients at configs[2]'s layer sizes."""
    from torch_scae_amd import ops
    from torch_scae_amd.nn_ext import GroupedMLP
    G, B = 8, 1024
    torch.manual_seed(3)
    mlp = GroupedMLP(G, [256, 128, 32], bias=True).cuda()
    x = torch.randn(B, G, 256, device="cuda", requires_grad=True)
    w = torch.randn(B, G, 32, device="cuda")
    calls = []
    real = ops._lib.call

    def spy(name, *a):
        calls.append(name)
        return real(name, *a)
    ops._lib.call = spy
    try:
        with ops.mfma_bf16():
            y = mlp(x)
            (y * w).sum().backward()
    finally:
        ops._lib.call = real
    assert any(n.endswith("_bf16") for n in calls), calls
    params = [p for p in mlp.parameters()]
    got = [x.grad.clone()] + [p.grad.clone() for p in params]
    x.grad = None
    for p in params:
        p.grad = None
    y_ref = mlp(x)            # the fp32 kernels as the yardstick (held to fp64 elsewhere)
    (y_ref * w).sum().backward()
    ref = [x.grad] + [p.grad for p in params]
    assert _rel_l2(y, y_ref) <= 1e-2
    # (gradients pass two ReLU gates: the ~0.3 % of units whose pre-activation lies
    # within bf16's rounding of zero flip their gate, which alone moves a gradient
    # tensor by a few per cent in L2)
    for a, b in zip(got, ref):
        assert _rel_l2(a, b) <= 8e-2, _rel_l2(a, b)


@pytest.mark.parametrize("row_tile", [16, 32])
def test_mlp_chain_bf16_operands_vs_fp32(row_tile, monkeypatch):
    """K7b inside ``ops.mfma_bf16()`` (BASELINE.json configs[2]): the one-launch
    chain stays the path -- its layer products take bf16 operands
    (v_mfma_f32_16x16x16_bf16, rounded at the matrix core, fp32 accumulate,
    fp32 tensors in memory), the weight-gradient launch runs the bf16 tiles.
    Bars: outputs 1e-2 in relative L2 against the fp32 chain (held to fp64
    above); gradients 1.2e-1 -- the batched-GEMM test's 8e-2 is for a chain of
    two layers, here a gradient crosses up to three ReLU gates whose
    borderline units bf16's rounding flips (measured: 0.091 on the input
    gradient, below 0.08 on the weights)."""
    from torch_scae_amd import ops
    monkeypatch.setattr(ops, "_CHAIN_ROW_TILE", row_tile)
    G, B, Kin, dims = 6, 256, 256, [128, 32, 128, 391]
    g = torch.Generator().manual_seed(11)
    layers, K = [], Kin
    for l, N in enumerate(dims):
        ones = l == 2
        w = torch.randn(G, N, K + (1 if ones else 0), generator=g) / K ** 0.5
        b = None if l >= 2 else torch.randn(G, N, generator=g) * 0.1
        layers.append((w, b, ones))
        K = N
    x = torch.randn(B, G, Kin, generator=g)
    R = torch.randn(B, G, dims[-1], generator=g)

    def run(bf16):
        xh = x.cuda().requires_grad_()
        lh = [(w.cuda().requires_grad_(),
               None if b is None else b.cuda().requires_grad_(), o)
              for w, b, o in layers]
        calls = []
        real = ops._lib.call

        def spy(name, *a):
            calls.append(name)
            return real(name, *a)
        ops._lib.call = spy
        try:
            with ops.mfma_bf16(bf16):
                assert ops.mlp_chain_supported(xh, lh)
                y = ops.mlp_chain(xh, lh)
                y.backward(R.cuda() * (y.detach() > 0))
        finally:
            ops._lib.call = real
        grads = [xh.grad] + [w.grad for w, _, _ in lh] + \
            [b.grad for _, b, _ in lh if b is not None]
        return y.detach(), grads, calls

    y32, g32, _ = run(False)
    y16, g16, calls = run(True)
    assert "scae_mlp_chain_fwd_f32" in calls and "scae_mlp_chain_bwd_f32" in calls
    assert "scae_gemm_multi_bf16" in calls
    assert not torch.equal(y16, y32)            # (the flag reaches the kernel)
    assert _rel_l2(y16, y32) <= 1e-2, _rel_l2(y16, y32)
    errs = [_rel_l2(a, b) for a, b in zip(g16, g32)]
    assert max(errs) <= 1.2e-1, errs


def test_deferred_sums_with_a_parameter_used_twice():
    """under ``deferred_param_sums`` a column sum may only wait when
    its output is the parameter's own slot AND nobody else contributes to that
    parameter before it has run.  A module applied twice in one forward (shared
    parameters) must therefore get its gradients by autograd's accumulation of
    two launched sums -- compared with a plain backward of the same graph, and
    with the once-used module, whose sums do wait."""
    import copy
    from torch_scae_amd import ops
    from torch_scae_amd.data_parallel import FlatParameters
    from torch_scae_amd.part_decoder import TemplateGenerator
    torch.manual_seed(3)
    B, M, C, F, ts = 8, 6, 1, 16, 11
    tg = TemplateGenerator(M, C, (ts, ts), dim_feature=F,
                           colorize_templates=True).cuda()
    plain = copy.deepcopy(tg)
    f1, f2 = torch.randn(2, B, M, F, device="cuda").unbind(0)
    w1, w2 = torch.randn(2, B, M, C, ts, ts, device="cuda").unbind(0)

    def loss_of(m, twice):
        a = (m(feature=f1).templates * w1).sum()
        return a + (m(feature=f2).templates * w2).sum() if twice else a

    flat = FlatParameters(tg)
    for twice in (True, False, True):
        plain.zero_grad(set_to_none=True)
        loss_of(plain, twice).backward()
        flat.clear_grads()
        launched = []
        real = ops._launch_sum_units

        def spy(units):
            q = ops.step_plan.current().deferred
            launched.append((len(units), q is not None and len(q)))
            return real(units)
        ops._launch_sum_units = spy
        try:
            with ops.deferred_param_sums():
                loss_of(tg, twice).backward()
                waiting = len(ops.step_plan.current().deferred)
        finally:
            ops._launch_sum_units = real
        flat.gather_grads()
        torch.cuda.synchronize()
        # once: every sum waits for the end of the backward; twice: the second
        # use's backward flushed what the first had queued and launched its own
        assert (waiting > 0) == (not twice), (twice, waiting, launched)
        ref = dict(plain.named_parameters())
        names = {id(p): n for n, p in tg.named_parameters()}
        for p, v in zip(flat.params, flat.grad_views()):
            assert torch.equal(v, ref[names[id(p)]].grad), (twice, names[id(p)])


def test_grouped_mlp_equals_per_capsule_loop():
    """Stacked-weight batched evaluation on K7 == the reference's loop of
    per-capsule MLPs (object_decoder.py:137-158; the loop evaluated with stock
    torch modules on the CPU from the same state_dict), with bias and with the
    ``caps_exist`` ones column."""
    from torch_scae_amd.nn_ext import MLP, GroupedMLP
    torch.manual_seed(0)
    G, B = 5, 7
    for bias, ones in ((True, False), (False, True)):
        d_in = 6 + (1 if ones else 0)
        gm = GroupedMLP(G, [d_in, 9, 4], bias=bias, ones_input=ones)
        loop = torch.nn.ModuleList([MLP([d_in, 9, 4], bias=bias)
                                    for _ in range(G)])
        loop.load_state_dict(gm.state_dict())
        x = torch.randn(B, G, 6)
        xin = torch.cat([x, torch.ones(B, G, 1)], -1) if ones else x
        want = torch.stack([loop[g](xin[:, g]) for g in range(G)], 1)
        got = gm.cuda()(x.cuda()).cpu()
        assert torch.allclose(got, want, atol=1e-6), (bias, ones)


@pytest.mark.parametrize("B,IH,Ci,Co,s,group,post", [
    (128, 9, 128, 128, 1, 0, False),    # the encoder's third layer at cfg-2
    (128, 7, 128, 128, 1, 0, True),     # its fourth, with the embedding bias output
    (5, 7, 128, 128, 1, 2, True),       # a last group with one image of two
    (7, 7, 128, 64, 1, 3, False),       # three images per workgroup, 3 row tiles
    (3, 9, 256, 96, 2, 1, False),       # stride 2, two channel blocks per wave and tap
    (6, 5, 128, 32, 1, 4, True),        # 3 x 3 outputs, four images per workgroup
])
def test_conv_resident_forward_vs_conv2d(B, IH, Ci, Co, s, group, post):
    """K8r (csrc/conv_resident.hip): the forward of a small layer with the input
    images resident in LDS and the filter read from its fragment-major copy,
    against fp64 conv2d + ReLU (part_encoder.py:26-44) at 1e-5 of the output's
    largest entry, and against the tile kernel of the same layer."""
    import ctypes
    from torch_scae_amd import _lib, ops
    g = torch.Generator().manual_seed(B * IH + Ci + Co)
    x = torch.randn(B, Ci, IH, IH, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** .5
    bias = torch.randn(Co, generator=g) * .1
    OH = (IH - 3) // s + 1
    pbias = torch.randn(Co, OH, OH, generator=g)
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), bias.double(),
                                                stride=s))
    lib = _lib.load()
    if group == 0:      # (a forced group may exceed what the launcher picks by itself)
        assert lib.scae_conv3x3_fwd_res_supported(B, IH, IH, Ci, Co, s) == 1
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    wd_, wf = torch.empty(Ci, 9, Co, device="cuda"), torch.empty(3, Co, 9, Ci, device="cuda")
    assert wf.numel() >= lib.scae_conv3x3_wf_floats(Co, Ci)
    st = torch.cuda.current_stream().cuda_stream
    P = ops._p
    _lib.call("scae_conv3x3_relayout_f32", P(w.cuda()), P(wf), P(wd_), Co, Ci, st)
    out = torch.full((B, OH, OH, Co), float("nan"), device="cuda")
    outp = torch.full((B, OH, OH, Co), float("nan"), device="cuda") if post else None
    pb = pbias.cuda() if post else None
    _lib.call("scae_conv3x3_fwd_res_f32", P(xn), P(wf[1]), P(bias.cuda()), P(out), P(pb),
              P(outp), B, IH, IH, Ci, Co, s, group, st)
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2).cpu().double()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 1e-5 * scale
    if post:
        gotp = outp.permute(0, 3, 1, 2).cpu().double()
        assert float((gotp - (ref + pbias.double())).abs().max()) <= 1e-5 * scale
    if Co % 64 == 0:        # the ring-pipelined tiles of the same layer
        out2 = torch.empty_like(out)
        _lib.call("scae_conv3x3_fwd_f32", P(xn), P(wf[0]), P(bias.cuda()), P(out2), None,
                  None, B, IH, IH, Ci, Co, s, st)
        assert float((out2 - out).abs().max()) <= 2e-6 * scale


def test_exact_split_products_reproduce_fp32_values_bitwise():
    """csrc/bf16x6.h: the K8 loops multiply fp32 operands as six exact partial products of a
    three-way bf16 split (8 + 8 + 8 significant bits) on the bf16 matrix cores.  What "exact"
    means is checked bit for bit where the result must be an input value itself:
      * forward with a one-hot filter (centre tap, co == ci, weight 1): out = x at the window
        centre -- x's full 24-bit values must come back from hi + mid + lo -- through the
        image-resident kernel (K8r) and the ring-pipelined tiles;
      * weight gradient with a one-hot pre-activation gradient (one pixel, one channel, 1.0):
        dW[co, :, tap] = x at that tap's input pixel, and the products with the other channels
        and pixels are exact zeros.
    And the accuracy claim: on random data the K8r forward is within 1e-6 of the output's
    largest entry of fp64 (the fp32 MFMA chain measures 1.2e-6 max / 1.2e-7 rms there, the split
    form 7.4e-7 / 5.9e-8: tools/x6_probe.py)."""
    import ctypes
    import torch.nn.functional as F
    from torch_scae_amd import _lib, ops
    lib, P = _lib.load(), ops._p
    st = torch.cuda.current_stream().cuda_stream
    B, IH, C = 6, 9, 128
    OH = IH - 2
    g = torch.Generator().manual_seed(5)
    # full-mantissa values of mixed magnitude (and exact zeros, as ReLU outputs have)
    x = torch.relu(torch.randn(B, IH, IH, C, generator=g) * torch.exp(torch.randn(B, IH, IH, C, generator=g) * 3)).cuda()
    w = torch.zeros(C, C, 3, 3)
    w[torch.arange(C), torch.arange(C), 1, 1] = 1.0
    wf, wd = torch.empty(3, C, 9, C, device="cuda"), torch.empty(C, 9, C, device="cuda")
    _lib.call("scae_conv3x3_relayout_f32", P(w.cuda()), P(wf), P(wd), C, C, st)
    zero = torch.zeros(C, device="cuda")
    want = x[:, 1:-1, 1:-1, :].contiguous()
    out = torch.full((B, OH, OH, C), float("nan"), device="cuda")
    _lib.call("scae_conv3x3_fwd_res_f32", P(x), P(wf[1]), P(zero), P(out), None, None, B, IH, IH,
              C, C, 1, 0, st)
    assert torch.equal(out, want)
    out.fill_(float("nan"))
    _lib.call("scae_conv3x3_fwd_f32", P(x), P(wf[0]), P(zero), P(out), None, None, B, IH, IH, C, C,
              1, st)
    assert torch.equal(out, want)
    # weight gradient: dpre = 1 at (image 2, output pixel (3, 4), channel 17), 0 elsewhere
    dpre = torch.zeros(B, OH, OH, C, device="cuda")
    dpre[2, 3, 4, 17] = 1.0
    splits = lib.scae_conv3x3_wgrad_splits(B, OH, OH, C, C)
    part = torch.empty(splits * (9 * C * C + C), device="cuda")
    dw, db = torch.empty(C, C, 3, 3, device="cuda"), torch.empty(C, device="cuda")
    _lib.call("scae_conv3x3_wgrad_f32", P(dpre), P(x), P(part), P(dw), P(db), B, IH, IH, C, C, 1, st)
    torch.cuda.synchronize()
    expect = torch.zeros(C, C, 3, 3, device="cuda")
    for kh in range(3):
        for kw in range(3):
            expect[17, :, kh, kw] = x[2, 3 + kh, 4 + kw, :]
    assert torch.equal(dw, expect)
    assert torch.equal(db, F.one_hot(torch.tensor(17), C).float().cuda())
    # accuracy on random data, against fp64
    xr = torch.relu(torch.randn(B, IH, IH, C, generator=g)).cuda()
    wr = (torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5).cuda()
    br = torch.randn(C, generator=g).cuda()
    _lib.call("scae_conv3x3_relayout_f32", P(wr), P(wf), P(wd), C, C, st)
    _lib.call("scae_conv3x3_fwd_res_f32", P(xr), P(wf[1]), P(br), P(out), None, None, B, IH, IH, C,
              C, 1, 0, st)
    ref = torch.relu(F.conv2d(xr.double().permute(0, 3, 1, 2).cpu(), wr.double().cpu(),
                              br.double().cpu())).permute(0, 2, 3, 1)
    assert float((out.double().cpu() - ref).abs().max()) <= 1e-6 * float(ref.abs().max())


def test_conv_resident_forward_rejects_what_it_cannot_hold():
    from torch_scae_amd import _lib
    lib = _lib.load()
    assert lib.scae_conv3x3_fwd_res_supported(128, 19, 19, 128, 128, 2) == 0   # 185 KB of pixels
    assert lib.scae_conv3x3_fwd_res_supported(128, 7, 7, 64, 128, 1) == 0      # Cin % 128
    assert lib.scae_conv3x3_fwd_res_supported(128, 7, 7, 128, 48, 1) == 0      # Cout % 32


def test_sums_riding_in_the_optimizer_launch_equal_two_launches_bitwise():
    """scae_rmsprop_sums_step_f32: a step's last column sums (plain, periodic and
    transposed windows into slots of the flat gradient buffer) as the head of the
    RMSprop launch -- bit for bit scae_sum_rows_multi_f32 + scae_rmsprop_step_f32, on
    every one of the four flat buffers, with slots that start off 16-byte boundaries."""
    import ctypes
    from torch_scae_amd import _lib, ops
    g = torch.Generator().manual_seed(11)
    n = 20011
    P = ctypes.c_void_p
    st = P(torch.cuda.current_stream().cuda_stream)

    def fresh():
        gg = torch.Generator().manual_seed(5)
        return [torch.randn(n, generator=gg).cuda() for _ in range(2)] + \
            [torch.rand(n, generator=gg).cuda(), torch.randn(n, generator=gg).cuda() * .1]
    partials = [torch.randn(22, 9 * 40, generator=g).cuda(),     # transposed (n x W)
                torch.randn(128, 5 * 12, generator=g).cuda(),    # periodic windows
                torch.randn(7, 333, generator=g).cuda(),         # plain, two outputs
                torch.randn(300, 6, generator=g).cuda()]         # tall and skinny
    # (offset into the flat gradient, segments) per job
    layout = [(1, [(0, 360, -40, 360)]),
              (3001, [(0, 5, 12, 50), (5, 12, 12, 70)]),
              (7002, [(0, 100, 0, 100), (120, 333, 0, 213)]),
              (19990, [(0, 6, 0, 6)])]
    results = []
    for fused in (False, True):
        param, grad, sq, buf = fresh()
        keep, jobs = [], (_lib.SumJob * len(partials))()
        for job, part, (off, segs) in zip(jobs, partials, layout):
            arr = (_lib.SumSegment * len(segs))()
            pos = off
            for a, (b, e, per, length) in zip(arr, segs):
                a.dst, a.begin, a.end, a.period = grad.data_ptr() + 4 * pos, b, e, per
                pos += length
            keep.append(arr)
            job.src, job.rows, job.cols = part.data_ptr(), part.shape[0], part.shape[1]
            job.segments, job.n_segments = arr, len(segs)
        args = (P(param.data_ptr()), P(grad.data_ptr()), P(sq.data_ptr()), P(buf.data_ptr()),
                n, 1e-3, None, 0.99, 1e-4, 0.9)
        if fused:
            _lib.call("scae_rmsprop_sums_step_f32", *args, 1.0, jobs, len(partials), st)
        else:
            _lib.call("scae_sum_rows_multi_f32", jobs, len(partials), st)
            _lib.call("scae_rmsprop_step_f32", *args, 0.0, 1.0, st)
        torch.cuda.synchronize()
        results.append([t.clone() for t in (param, grad, sq, buf)])
    for a, b, what in zip(results[0], results[1], ("param", "grad", "square_avg", "buf")):
        bad = (a != b).nonzero().flatten()
        assert bad.numel() == 0, (what, bad[:8].tolist(), bad.numel())
    assert float((results[0][0] - fresh()[0]).abs().max()) > 0
