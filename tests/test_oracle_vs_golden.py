"""Pin the oracle (oracle/scae_oracle.py) to vectors captured from the real
reference (tests/golden/make_golden.py).  CPU only."""
import pytest
import torch

from oracle import scae_oracle as O
from tests.golden_util import assert_close, load, model_names, sub

ATOL, RTOL = 2e-6, 2e-5


def leafify(d):
    return {k: v.clone().requires_grad_(v.dtype.is_floating_point)
            for k, v in d.items()}


# ---------------------------------------------------------------- per op ----
def test_geometric_transform_all_flags():
    blob, _ = load("op_geometric_transform")
    x = blob["in/pose"]
    for sim in (0, 1):
        for nl in (0, 1):
            for mat in (0, 1):
                y = O.geometric_transform(x, bool(sim), bool(nl), bool(mat))
                assert_close(y, blob[f"out/sim{sim}_nl{nl}_mat{mat}"], ATOL,
                             RTOL, f"gt sim{sim} nl{nl} mat{mat}")


def test_geometric_transform_grad():
    blob, _ = load("op_geometric_transform_grad")
    for sim in (0, 1):
        x = blob[f"in/pose_sim{sim}"].clone().requires_grad_(True)
        y = O.geometric_transform(x, bool(sim))
        (y * blob[f"in/w_sim{sim}"]).sum().backward()
        assert_close(y, blob[f"out/y_sim{sim}"], ATOL, RTOL, "y")
        assert_close(x.grad, blob[f"grad/pose_sim{sim}"], ATOL, RTOL, "grad")


@pytest.mark.parametrize("case", ["plain", "nopresence", "saturated", "ties",
                                  "wide"])
def test_qkv_attention(case):
    blob, _ = load("op_qkv_attention")
    c = sub(blob, case + "/")
    q, k, v = (c[n].clone().requires_grad_(True) for n in "qkv")
    out = O.qkv_attention(q, k, v, c.get("presence"))
    (out * c["w"]).sum().backward()
    assert_close(out, c["out"], 1e-5, 1e-5, "out")
    assert_close(q.grad, c["gq"], 1e-5, 1e-4, "gq")
    assert_close(k.grad, c["gk"], 1e-5, 1e-4, "gk")
    assert_close(v.grad, c["gv"], 1e-5, 1e-4, "gv")


BLOCKS = {
    "mha_h3": lambda P, i: O.multi_head_attention(P, "m", i["q"], i["k"],
                                                  i["v"], i["presence"], 3),
    "mab_h3": lambda P, i: O.mab(P, "m", i["q"], i["k"], i["presence"], 3,
                                 False),
    "mab_ln": lambda P, i: O.mab(P, "m", i["q"], i["k"], i["presence"], 2,
                                 True),
    "sab": lambda P, i: O.sab(P, "m", i["x"], i["presence"], 1, True),
    "isab": lambda P, i: O.isab(P, "m", i["x"], i["presence"], 2, True),
    "isab_nopres": lambda P, i: O.isab(P, "m", i["x"], None, 1, False),
    "pma": lambda P, i: O.pma(P, "m", i["x"], i["presence"], 1, True),
    "st_sab": lambda P, i: O.set_transformer(P, "m", i["x"], i["presence"], 2,
                                             1, True, None),
    "st_isab": lambda P, i: O.set_transformer(P, "m", i["x"], i["presence"],
                                              2, 3, True, 5),
}


@pytest.mark.parametrize("name", sorted(BLOCKS))
def test_set_transformer_blocks(name):
    blob, _ = load("op_set_transformer_blocks")
    c = sub(blob, name + "/")
    P = leafify({"m." + k: v for k, v in sub(c, "param/").items()})
    ins = leafify(sub(c, "in/"))
    ins.setdefault("presence", None)
    if ins["presence"] is not None:
        ins["presence"] = ins["presence"].detach()
    out = BLOCKS[name](P, ins)
    (out * c["w"]).sum().backward()
    assert_close(out, c["out"], 1e-5, 1e-5, "out")
    for k, g in sub(c, "grad/").items():
        assert_close(P["m." + k].grad, g, 1e-5, 1e-4, "grad " + k)
    for k, g in sub(c, "gin/").items():
        assert_close(ins[k].grad, g, 1e-5, 1e-4, "gin " + k)


@pytest.mark.parametrize("case", ["a", "b"])
def test_capsule_likelihood(case):
    blob, _ = load("op_capsule_likelihood")
    c = sub(blob, case + "/")
    i = leafify(sub(c, "in/"))
    res = O.capsule_likelihood(i["vote"], i["scale"], i["vote_presence"],
                               i["dummy_vote"], i["x"], i["presence"])
    tot = res.log_prob * 1.7
    for k, w in sub(c, "w/").items():
        tot = tot + (res[k] * w).sum()
    tot.backward()
    for k, ref in sub(c, "out/").items():
        assert_close(res[k], ref, 1e-5, 1e-5, "out " + k)
    for k, g in sub(c, "grad/").items():
        assert_close(i[k].grad, g, 2e-5, 1e-4, "grad " + k)


def test_capsule_likelihood_no_presence():
    blob, _ = load("op_capsule_likelihood")
    c = sub(blob, "nopres/")
    i = sub(c, "in/")
    res = O.capsule_likelihood(i["vote"], i["scale"], i["vote_presence"],
                               i["dummy_vote"], i["x"], None)
    for k, ref in sub(c, "out/").items():
        assert_close(res[k], ref, 1e-5, 1e-5, "out " + k)


CAPS_VARIANTS = {
    "default": dict(learn_vote_scale=True, allow_deformations=True,
                    noise_type="uniform", noise_scale=4.,
                    similarity_transform=False),
    "sim_nonoise": dict(learn_vote_scale=False, allow_deformations=False,
                        noise_type=None, noise_scale=0.,
                        similarity_transform=True),
}


@pytest.mark.parametrize("name", sorted(CAPS_VARIANTS))
def test_capsule_layer(name):
    blob, _ = load("op_capsule_layer")
    c = sub(blob, name + "/")
    cfg = dict(n_caps=4, dim_feature=10, n_votes=5, dim_caps=6,
               hidden_sizes=(7,), **CAPS_VARIANTS[name])
    P = leafify({"L." + k: v for k, v in sub(c, "layer_param/").items()})
    feat = c["in/feature"].clone().requires_grad_(True)
    noise = sub(c, "noise/")
    res = O.capsule_layer(P, "L", feat, cfg, noise.get("0"), noise.get("1"))
    tot = res.cpr_dynamic_reg_loss * 0.9
    for k, w in sub(c, "w/").items():
        tot = tot + (res[k] * w).sum()
    tot.backward()
    for k, ref in sub(c, "out/").items():
        assert_close(res[k], ref, 1e-5, 1e-5, "out " + k)
    assert_close(feat.grad, c["grad/feature"], 1e-5, 1e-4, "grad feature")
    for k, g in sub(c, "grad/").items():
        if k != "feature":
            assert_close(P["L." + k].grad, g, 1e-5, 1e-4, "grad " + k)

    # whole object decoder, forward
    P2 = {"D." + k: v for k, v in sub(c, "dec_param/").items()}
    dn = sub(c, "dec_noise/")
    with torch.no_grad():
        r2 = O.capsule_object_decoder(P2, "D", feat.detach(), c["dec_in/x"],
                                      c["dec_in/presence"], cfg, dn.get("0"),
                                      dn.get("1"))
    for k, ref in sub(c, "dec_out/").items():
        assert_close(r2[k], ref, 1e-5, 1e-5, "dec_out " + k)


def test_capsule_layer_hierarchical():
    """parent_transform / parent_presence (object_decoder.py:184-187, :214-215)."""
    blob, _ = load("op_capsule_layer_hier")
    cfg = dict(n_caps=4, dim_feature=10, n_votes=5, dim_caps=6,
               hidden_sizes=(7,), **CAPS_VARIANTS["default"])
    P = leafify({"L." + k: v for k, v in sub(blob, "layer_param/").items()})
    ins = {k: blob["in/" + k].clone().requires_grad_(True)
           for k in ("feature", "parent_transform", "parent_presence")}
    noise = sub(blob, "noise/")
    res = O.capsule_layer(P, "L", ins["feature"], cfg, noise.get("0"),
                          noise.get("1"),
                          parent_transform=ins["parent_transform"],
                          parent_presence=ins["parent_presence"])
    tot = res.cpr_dynamic_reg_loss * 0.9
    for k, w in sub(blob, "w/").items():
        tot = tot + (res[k] * w).sum()
    tot.backward()
    for k, ref in sub(blob, "out/").items():
        assert_close(res[k], ref, 1e-5, 1e-5, "out " + k)
    for k, g in sub(blob, "grad/").items():
        got = ins[k].grad if k in ins else P["L." + k].grad
        assert_close(got, g, 1e-5, 1e-4, "grad " + k)


def decoder_cases():
    _, meta = load("op_image_decoder")
    return sorted(k for k in meta if not k.startswith("tg_"))


@pytest.mark.parametrize("name", decoder_cases())
def test_image_decoder(name):
    blob, meta = load("op_image_decoder")
    m = meta[name]
    c = sub(blob, name + "/")
    cfg = dict(output_size=tuple(m["HW"]),
               learn_output_scale=m["learn_output_scale"],
               use_alpha_channel=m["use_alpha_channel"],
               background_value=m["background_value"])
    P = leafify({"d." + k: v for k, v in sub(c, "param/").items()})
    i = leafify({k: v for k, v in sub(c, "in/").items() if k not in ("x", "w")})
    r = O.image_decoder(P, "d", i["templates"], i["pose"], i.get("presence"),
                        i.get("bg_image"), cfg)
    lp = O.gmm_log_prob(r.transformed_templates, r.scale, r.mixing_logits,
                        c["in/x"])
    (lp * c["in/w"]).sum().backward()
    assert_close(r.transformed_templates, c["out/transformed_templates"],
                 ATOL, RTOL, "tt")
    assert_close(r.mixing_logits, c["out/mixing_logits"], 1e-5, 1e-5, "ml")
    assert_close(lp, c["out/log_prob"], 1e-5, 1e-5, "log_prob")
    for k, g in sub(c, "grad/").items():
        assert_close(i[k].grad, g, 1e-5, 1e-4, "grad " + k)
    for k, g in sub(c, "pgrad/").items():
        assert_close(P["d." + k].grad, g, 2e-5, 1e-4, "pgrad " + k)
    with torch.no_grad():
        assert_close(O.gmm_mean(r.transformed_templates, r.mixing_logits),
                     c["out/mean"], 1e-5, 1e-5, "mean")
        assert_close(O.gmm_mode(r.transformed_templates, r.scale,
                                r.mixing_logits), c["out/mode"], 1e-5, 1e-5,
                     "mode")
        assert_close(O.gmm_mixing_log_prob(r.mixing_logits),
                     c["out/mixing_log_prob"], 1e-5, 1e-5, "mixing_log_prob")
        if "out/mode_max" in c:
            assert_close(O.gmm_mode(r.transformed_templates, r.scale,
                                    r.mixing_logits, maximum=True),
                         c["out/mode_max"], 1e-5, 1e-5, "mode_max")
        else:
            with pytest.raises(RuntimeError):
                O.gmm_mode(r.transformed_templates, r.scale, r.mixing_logits,
                           maximum=True)

    # gradients through the materialised tensors
    P = leafify({"d." + k: v for k, v in sub(c, "param/").items()})
    i = leafify({k: v for k, v in sub(c, "in/").items() if k not in ("x", "w")})
    bg = i.get("bg_image")
    r = O.image_decoder(P, "d", i["templates"], i["pose"], i.get("presence"),
                        None if bg is None else bg.detach(), cfg)
    ((r.transformed_templates * c["mat/wt"]).sum()
     + (r.mixing_logits * c["mat/wm"]).sum()).backward()
    for k, g in sub(c, "mat/grad/").items():
        assert_close(i[k].grad, g, 2e-5, 1e-4, "mat grad " + k)
    for k, g in sub(c, "mat/pgrad/").items():
        assert_close(P["d." + k].grad, g, 2e-5, 1e-4, "mat pgrad " + k)


def test_bilinear_warp_matches_torch_ops():
    """The explicit sampling formulas (what the HIP kernel implements) equal
    F.affine_grid + F.grid_sample, incl. poses mapping outside the template."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    src = torch.rand(6, 2, 5, 7, generator=g, dtype=torch.float64)
    theta = torch.randn(6, 2, 3, generator=g, dtype=torch.float64)
    theta[0] = torch.tensor([[0.1, 0, 4.], [0, 0.1, 4.]])
    grid = F.affine_grid(theta, [6, 2, 9, 8], align_corners=False)
    ref = F.grid_sample(src, grid, align_corners=False)
    got = O.bilinear_warp(src, theta, (9, 8))
    assert float((ref - got).abs().max()) < 1e-13


@pytest.mark.parametrize("name", ["tg_default", "tg_relu1_rgb", "tg_nocolor"])
def test_template_generator(name):
    blob, meta = load("op_image_decoder")
    c = sub(blob, name + "/")
    P = leafify({"t." + k: v for k, v in sub(c, "param/").items()})
    feat = c.get("in/feature")
    if feat is not None:
        feat = feat.clone().requires_grad_(True)
    r = O.template_generator(P, "t", feat, 2, meta[name])
    assert_close(r.templates, c["out/templates"], ATOL, RTOL, "templates")
    assert_close(r.raw_templates, c["out/raw_templates"], ATOL, RTOL, "raw")
    if sub(c, "pgrad/"):
        (r.templates * c["w"]).sum().backward()
        for k, g in sub(c, "pgrad/").items():
            assert_close(P["t." + k].grad, g, 1e-5, 1e-4, "pgrad " + k)
        if "grad/feature" in c:
            assert_close(feat.grad, c["grad/feature"], 1e-5, 1e-4, "gfeat")


@pytest.mark.parametrize("name,shape,sim,train", [
    ("affine_train", (1, 16, 16), False, True),
    ("similarity_eval", (3, 14, 14), True, False)])
def test_part_encoder(name, shape, sim, train):
    blob, _ = load("op_part_encoder")
    c = sub(blob, name + "/")
    P = {"e." + k: v for k, v in sub(c, "param/").items()}
    cnn = dict(strides=[2, 1])
    enc = dict(n_caps=3, n_poses=6, n_special_features=4,
               similarity_transform=sim)
    with torch.no_grad():
        r = O.capsule_image_encoder(P, "e", c["in/image"], cnn, enc, train,
                                    c.get("noise/0"))
    for k in ("pose", "presence", "feature"):
        assert_close(r[k], c["out/" + k], 1e-5, 1e-5, k)


def test_sparsity_and_log_safe():
    blob, _ = load("op_sparsity")
    cp = blob["in/caps_presence"]
    for lt in ("l2", "entropy", "kl"):
        x = cp.clone().requires_grad_(True)
        a, b = O.sparsity_loss(lt, x, n_classes=3)
        (a * 1.3 + b * 0.7).backward()
        assert_close(a, blob[f"out/{lt}_within"], ATOL, RTOL, lt + " within")
        assert_close(b, blob[f"out/{lt}_between"], ATOL, RTOL, lt + " between")
        assert_close(x.grad, blob[f"grad/{lt}"], ATOL, 1e-4, lt + " grad")
    a, b = O.sparsity_loss("l2", cp, n_classes=3, within_example_constant=1.5)
    assert_close(a, blob["out/l2c_within"], ATOL, RTOL, "l2c")
    with pytest.raises(ValueError):
        O.sparsity_loss("nope", cp)
    assert torch.equal(O.log_safe(blob["in/log_safe"]), blob["out/log_safe"])


# ------------------------------------------------------------ full model ----
@pytest.mark.parametrize("name", model_names())
def test_full_model(name):
    blob, meta = load(name)
    cfg = O.prepare_model_params(**meta["config"])
    P = leafify(sub(blob, "param/"))
    noise = sub(blob, "noise/")
    if meta["train"]:
        n = (noise["0"], noise["1"], noise["2"])
    else:
        n = (None, noise["0"], noise["1"])
    image, label = blob["in/image"], blob["in/label"]
    res = O.scae_forward(P, cfg, image, n, training=meta["train"])
    loss, log = O.scae_loss(cfg, res, image, label)
    loss.backward()

    assert_close(loss, blob["out/loss"], 1e-4, 2e-6, "loss")
    for k, ref in sub(blob, "log/").items():
        assert_close(log[k], ref, 1e-4, 1e-5, "log " + k)
    assert_close(O.calculate_accuracy(res, label), blob["out/accuracy"],
                 0, 0, "accuracy")
    checked = 0
    for k, ref in sub(blob, "res/").items():
        if "." in k:
            head, tail = k.split(".", 1)
            rec = res[head]
            if tail in rec:
                got = rec[tail]
            elif tail == "log_prob":
                got = O.gmm_log_prob(rec.transformed_templates, rec.scale,
                                     rec.mixing_logits, image)
            elif tail == "mode":
                got = O.gmm_mode(rec.transformed_templates, rec.scale,
                                 rec.mixing_logits)
            elif tail == "mode_max":
                got = O.gmm_mode(rec.transformed_templates, rec.scale,
                                 rec.mixing_logits, maximum=True)
            elif tail == "mean":
                got = O.gmm_mean(rec.transformed_templates, rec.mixing_logits)
            elif tail == "mixing_log_prob":
                got = O.gmm_mixing_log_prob(rec.mixing_logits)
            else:
                raise KeyError(k)
        else:
            got = res[k]
        assert_close(got, ref, 1e-5, 1e-5, "res " + k)
        checked += 1
    assert checked >= 27
    ngrad = 0
    for k, g in sub(blob, "grad/").items():
        assert P[k].grad is not None, k
        assert_close(P[k].grad, g, 2e-5, 2e-4, "grad " + k)
        ngrad += 1
    assert ngrad > 20
    for k in meta["no_grad_params"]:
        assert P[k].grad is None or float(P[k].grad.abs().sum()) == 0.0, k
