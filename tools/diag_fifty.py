"""Diagnostic: replay tests/test_timed_path.py's 50-step run up to step N, then compare, on the
model's own state and batch, (a) the eager model's gradients and (b) the part decoder alone
(K1 forward + backward on identical inputs) with the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import scae_oracle as O
from tests import test_timed_path as T
from torch_scae_amd import factory, nn_utils, nn_ext
from torch_scae_amd.data import stroke_batches
N = int(sys.argv[1]) if len(sys.argv) > 1 else 17
cfg = dict(image_shape=(1, 16, 16), n_classes=4, n_part_caps=5, n_obj_caps=4,
           pcae_cnn_encoder_params=dict(out_channels=[64, 64], kernel_sizes=[3, 3], strides=[2, 1]),
           pcae_template_generator_params=dict(template_size=(5, 5)),
           ocae_encoder_set_transformer_params=dict(dim_hidden=8, dim_out=64, n_layers=2),
           ocae_decoder_capsule_params=dict(dim_caps=4, hidden_sizes=(8,)),
           scae_params=dict(reconstruct_alternatives=False))
np.random.seed(3); torch.manual_seed(3)
proto = factory.make_scae(cfg)
g = torch.Generator().manual_seed(4)
with torch.no_grad():
    for p in proto.parameters():
        if float(p.abs().sum()) == 0.0:
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
sd = {k: v.clone() for k, v in proto.state_dict().items()}
B, lr = 8, 1e-3
model, step = T.build_step(cfg, B, sd, lr=lr)
step.capture()
warm = stroke_batches(1, B, cfg["image_shape"], seed=8, n_classes=4)
step(warm[0][0].cuda(), warm[1][0].cuda())
pool = stroke_batches(50, B, cfg["image_shape"], seed=9, n_classes=4)
ocfg = O.prepare_model_params(**cfg)
for it in range(N + 1):
    noise = T.split_noise(T.predict_noise(step), cfg, B)
    image, label = pool[0][it], pool[1][it]
    if it == N:
        break
    step(image.cuda(), label.cuda())
torch.cuda.synchronize()
state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
P = {k: v.clone().requires_grad_(True) for k, v in state.items()}
ref_loss, _, ref_grads = O.train_step(P, ocfg, image, label, noise)
# (a) eager model, same noise
m2 = factory.make_scae(cfg); m2.load_state_dict(state); m2 = m2.cuda().train()
with nn_utils.fixed_noise([n.clone() for n in noise]):
    res = m2(image.cuda())
loss, _ = m2.loss(res, image.cuda(), label.cuda())
loss.backward()
got = nn_ext.named_reference_grads(m2)
rows = []
for k, r in ref_grads.items():
    if r is None or float(r.abs().max()) < 1e-12: continue
    rows.append((float((got[k].cpu() - r).abs().max()) / float(r.abs().max()), k, float(r.abs().max())))
rows.sort(reverse=True)
print("step", N, "loss", float(loss), float(ref_loss))
print("eager model vs oracle, worst:", [(f"{a:.2e}", k, f"{s:.2e}") for a, k, s in rows[:6]])
# (b) the part decoder alone on the oracle's own decoder inputs
with torch.no_grad():
    enc = O.capsule_image_encoder(state, "part_encoder", image, ocfg["pcae_cnn_encoder"], ocfg["pcae_encoder"], True, noise[0])
    tmpl = O.template_generator(state, "template_generator", enc.feature, B, ocfg["pcae_template_generator"]).templates
print("pose abs max", float(enc.pose.abs().max()), "lin median", float(enc.pose[..., [0, 1, 3, 4]].abs().amax(-1).median()),
      "presence min/max", float(enc.presence.min()), float(enc.presence.max()))
leafc = lambda t: t.clone().requires_grad_(True)
tc, pc, prc = leafc(tmpl), leafc(enc.pose), leafc(enc.presence)
Pd = {k: v.clone().requires_grad_(True) for k, v in state.items() if k.startswith("part_decoder.")}
ro = O.image_decoder(Pd, "part_decoder", tc, pc, prc, None, ocfg["pcae_decoder"])
lpo = O.gmm_log_prob(ro.transformed_templates, ro.scale, ro.mixing_logits, image)
(lpo.flatten(1).sum(-1).mean()).backward()
dec = m2.part_decoder
for p_ in dec.parameters(): p_.grad = None
tg, pg, prg = (t.detach().clone().cuda().requires_grad_(True) for t in (tmpl, enc.pose, enc.presence))
rg = dec(tg, pg, prg)
sums = rg.pdf.log_prob_tile_sums(image.cuda())
(sums.sum() / B).backward()
for name, a, b in (("templates", tg.grad, tc.grad), ("pose", pg.grad, pc.grad), ("presence", prg.grad, prc.grad)):
    d = (a.cpu() - b).abs()
    idx = int(d.reshape(-1).argmax())
    print(f"K1 grad {name}: worst {float(d.max()):.3e} of max {float(b.abs().max()):.3e}; at flat index {idx}")
d = (pg.grad.cpu() - pc.grad).abs().amax(-1)
bi, ki = divmod(int(d.reshape(-1).argmax()), d.shape[1])
print("worst pose grad at image", bi, "capsule", ki, "hip", pg.grad[bi, ki].cpu().tolist(), "oracle", pc.grad[bi, ki].tolist(), "pose", enc.pose[bi, ki].tolist(), "presence", float(enc.presence[bi, ki]))
# (c) gradients arriving at the part encoder's outputs, both sides
cap_o = {}
orig = O.capsule_image_encoder
def spy(*a, **k):
    e = orig(*a, **k)
    for n in ("pose", "presence", "feature"):
        t = getattr(e, n)
        if t is not None:
            t.retain_grad(); cap_o[n] = t
    return e
O.capsule_image_encoder = spy
P2 = {k: v.clone().requires_grad_(True) for k, v in state.items()}
O.train_step(P2, ocfg, image, label, noise)
O.capsule_image_encoder = orig
m3 = factory.make_scae(cfg); m3.load_state_dict(state); m3 = m3.cuda().train()
cap_h = {}
fwd = m3.part_encoder.forward
def spy_h(img):
    parts = fwd(img)
    for n in ("pose", "presence", "feature", "_feature_twin"):
        t = parts.get(n)
        if t is not None and t.requires_grad:
            t.retain_grad(); cap_h[n] = t
    return parts
m3.part_encoder.forward = spy_h
with nn_utils.fixed_noise([n.clone() for n in noise]):
    res = m3(image.cuda())
loss, _ = m3.loss(res, image.cuda(), label.cuda())
loss.backward()
for n in ("pose", "presence", "feature"):
    gh = cap_h[n].grad.cpu()
    if n == "feature" and "_feature_twin" in cap_h and cap_h["_feature_twin"].grad is not None:
        gh = gh + cap_h["_feature_twin"].grad.cpu()
    go = cap_o[n].grad
    d = (gh - go).abs()
    print(f"grad at encoder output {n}: worst {float(d.max()):.3e} of max {float(go.abs().max()):.3e}; values {float((cap_h[n].detach().cpu() - cap_o[n].detach()).abs().max()):.2e} apart")
