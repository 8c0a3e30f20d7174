"""Does the HIP path TRAIN?  cfg-2 on structured synthetic images (data.stroke_batches: ten
stroke glyphs under random affine warps) for N steps through TrainStep (replayed graph, fused
RMSprop, the reference's optimiser settings but for the learning rate given): loss trajectory,
the capsules' state, and the unsupervised / linear-head accuracies SCAE reports
(stacked_capsule_auto_encoder.py:289-293) on held-out batches.
usage: python tools/train_strokes.py <out.json> [steps] [lr]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch_scae_amd.data import stroke_batches
from torch_scae_amd.train_step import TrainStep

out, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20000
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 3e-5
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["mnist_24_24_bs128"]
B, shape = cfg["batch"], cfg["model"]["image_shape"]
model = bench.build_model(cfg, seed=0).to(dev).train()
step = TrainStep(model, B, shape, lr=lr)
train = stroke_batches(256, B, shape, seed=1, device=dev, glyph_seed=7)
held = stroke_batches(16, B, shape, seed=2, device=dev, glyph_seed=7)   # same glyphs, fresh warps


def evaluate():
    model.eval()
    accs, losses = [], []
    with torch.no_grad():
        for i in range(held[0].shape[0]):
            res = model(held[0][i])
            loss, _ = model.loss(res, held[0][i], held[1][i])
            losses.append(float(loss))
            prior = (res.prior_cls_prob.argmax(-1) == held[1][i]).float().mean()
            post = (res.posterior_cls_prob.argmax(-1) == held[1][i]).float().mean()
            accs.append((float(prior), float(post)))
    model.train()
    return dict(loss=sum(losses) / len(losses),
                prior_acc=sum(a for a, _ in accs) / len(accs),
                posterior_acc=sum(b for _, b in accs) / len(accs))


log = [dict(step=0, **evaluate(), capsules=bench.capsule_state(model, held[0][0]))]
t0 = time.perf_counter()
every = max(1, steps // 10)
for s in range(1, steps + 1):
    i = s % train[0].shape[0]
    loss = step(train[0][i], train[1][i])
    if s % every == 0:
        torch.cuda.synchronize()
        log.append(dict(step=s, train_loss=float(loss), **evaluate(),
                        capsules=bench.capsule_state(model, held[0][0]),
                        wall_s=round(time.perf_counter() - t0, 2)))
        print(log[-1], flush=True)
json.dump(dict(workload="mnist_24_24_bs128 on data.stroke_batches (10 glyph classes)", lr=lr,
               steps=steps, batch=B, log=log), open(out, "w"), indent=1)
