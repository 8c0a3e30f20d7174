import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch_scae_amd import ops, _lib
cfg = bench.CONFIGS["mnist_24_24_bs128"]
dev = torch.device("cuda", 0)
model = bench.build_model(cfg, 0).to(dev).train()
cap = {}
orig = model.part_decoder.forward
def fwd(templates, pose, presence=None, bg_image=None):
    cap.update(templates=templates.detach(), pose=pose.detach(), presence=presence.detach())
    return orig(templates=templates, pose=pose, presence=presence, bg_image=bg_image)
model.part_decoder.forward = fwd
from torch_scae_amd.train_step import TrainStep
NSTEPS = int(os.environ.get("NSTEPS", "0"))
if NSTEPS:
    step = TrainStep(model, 128, (1, 40, 40))
    images, labels = bench.synthetic_batches(cfg, dev, 1000)
    for i in range(NSTEPS):
        step(images[i % 8], labels[i % 8])
    torch.cuda.synchronize()
    print("trained", NSTEPS, "loss", float(step.loss))
img = torch.rand(128, 1, 40, 40, device=dev)
with torch.no_grad():
    model(img)
if os.environ.get("SAVE_CAP"):
    os.makedirs("gpurun_out", exist_ok=True)
    torch.save({k: v.cpu() for k, v in cap.items()} | dict(alpha=model.part_decoder.templates_alpha.detach().cpu(), bgv=model.part_decoder.bg_value.detach().cpu(), bgm=model.part_decoder.bg_mixing_logit.detach().cpu(), img=img.cpu()), os.environ["SAVE_CAP"])
if os.environ.get("LOAD_CAP"):
    blob = torch.load(os.environ["LOAD_CAP"])
    cap = {k: blob[k].to(dev) for k in ("templates", "pose", "presence")}
    with torch.no_grad():
        model.part_decoder.templates_alpha.copy_(blob["alpha"]); model.part_decoder.bg_value.copy_(blob["bgv"]); model.part_decoder.bg_mixing_logit.copy_(blob["bgm"])
    img = blob["img"].to(dev)
pose = cap["pose"]
print("pose mean", pose.mean((0, 1)).cpu(), "std", pose.std((0, 1)).cpu())
print("presence min/max", float(cap["presence"].min()), float(cap["presence"].max()))
a = pose.reshape(-1, 6)
det = a[:, 0] * a[:, 4] - a[:, 1] * a[:, 3]
print("det min abs", float(det.abs().min()), "frac |a3|*0.275<1e-3", float(((a[:, 3] * 0.275).abs() < 1e-3).float().mean()))
dec = model.part_decoder
B, M, C, H, W = 128, 24, 1, 40, 40
tensors = [cap["templates"].contiguous(), dec.templates_alpha.detach().reshape(M, 11, 11).contiguous(), pose.contiguous(),
           cap["presence"].contiguous(), None, dec.bg_value.detach(), dec.bg_mixing_logit.detach(), None, None]
desc, _ = ops._make_desc(tensors, (H, W))
dref = ctypes.byref(desc)
f = lambda *s: torch.empty(*s, device=dev)
lp, lse_post, lse_prior = f(B, C, H, W), f(B, C, H, W), f(B, 1, H, W)
glp = torch.ones(B, C, H, W, device=dev)
g_t, g_a, g_pose, g_pres, g_scal = f(B, M, C, 11, 11), f(B, M, 11, 11), f(B, M, 6), f(B, M), f(B, M + 1, 4)
p, st = ops._p, ops._stream(img)
lib = _lib.load()
assert lib.scae_render_gmm_logprob_fwd_f32(dref, p(img), p(lp), p(lse_post), p(lse_prior), st) == 0
def bwd():
    return lib.scae_render_gmm_bwd_f32(dref, p(img), p(lse_post), p(lse_prior), p(glp), None, None, p(g_t), p(g_a), p(g_pose), p(g_pres), None, p(g_scal), st)
for _ in range(5): assert bwd() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): bwd()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("SCAE_HIP_LIB", "in-tree"), "bwd us", e0.elapsed_time(e1) / 50 * 1e3)
tiles = lib.scae_render_gmm_logprob_tiles(dref)
ts = f(B, tiles)
assert lib.scae_render_gmm_logprob_sums_fwd_f32(dref, p(img), p(ts), p(lse_post), p(lse_prior), st) == 0
gts = torch.full((B, tiles), -1.0 / B, device=dev)
def bwd2():
    return lib.scae_render_gmm_sums_bwd_f32(dref, p(img), p(lse_post), p(lse_prior), p(gts), p(g_t), p(g_a), p(g_pose), p(g_pres), None, p(g_scal), st)
for _ in range(5): assert bwd2() == 0
torch.cuda.synchronize()
e0.record()
for _ in range(50): bwd2()
e1.record(); torch.cuda.synchronize()
print("tiles", tiles, "sums bwd us", e0.elapsed_time(e1) / 50 * 1e3)

