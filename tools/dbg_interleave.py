import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib, torch
from tests import test_step_plan as T
cfgs = [T._medium_cfg(vote_type="enc", presence_type="enc"),
        dict(T._medium_cfg(vote_type="enc", presence_type="enc"), n_part_caps=6, n_obj_caps=5)]
B = 16
g = torch.Generator().manual_seed(21)
batches = [(torch.rand(B, 1, 32, 32, generator=g).cuda(), torch.randint(0, 10, (B,), generator=g).cuda()) for _ in cfgs]
states = [T._filled_state(c, seed=7 + i)[0] for i, c in enumerate(cfgs)]
def build():
    from torch_scae_amd import ops
    steps = []
    for cfg, sd in zip(cfgs, states):
        torch.manual_seed(99); ops.reset_noise()
        steps.append(T._eager_step(cfg, sd, B)[1])
    return steps
def forward(step, batch, stack):
    step._stage(*batch); step.flat.clear_grads()
    plan = step.plan
    with plan.active():
        stack.enter_context(step._lazy()); stack.enter_context(plan.fusing(step.image))
        res = step.model(step.image)
        loss, _ = step.model.loss(res, step.image, step.label)
    return loss
def backward(step, loss, stack):
    plan = step.plan
    with plan.active(), plan.deferring():
        loss.backward()
    stack.close(); step.flat.gather_grads(); torch.cuda.synchronize()
    return float(loss), step.flat.flat_grad.clone()
def names(step):
    nm = {id(p): n for n, p in step.model.named_parameters()}
    return [(nm[id(p)], off, p.numel()) for p, off in zip(step.flat.params, step.flat.offsets)]
def diff(step, a, b, tag):
    bad = [(n, float((a[o:o+k] - b[o:o+k]).abs().max())) for n, o, k in names(step) if not torch.equal(a[o:o+k], b[o:o+k])]
    print(tag, "differs in", len(bad), "tensors", bad[:8])
runs = {}
for rep in range(2):
    out = []
    for step, batch in zip(build(), batches):
        with contextlib.ExitStack() as stack:
            out.append(backward(step, forward(step, batch, stack), stack))
    runs[rep] = out
for i in range(2):
    print("apart run0 vs run1 model", i, "loss", runs[0][i][0], runs[1][i][0], "equal grads", torch.equal(runs[0][i][1], runs[1][i][1]))
sa, sb = build()
with contextlib.ExitStack() as ka, contextlib.ExitStack() as kb:
    la = forward(sa, batches[0], ka); lb = forward(sb, batches[1], kb)
    ra = backward(sa, la, ka); rb = backward(sb, lb, kb)
for i, (st, r) in enumerate(((sa, ra), (sb, rb))):
    print("interleaved model", i, "loss", r[0], "equal", torch.equal(r[1], runs[0][i][1]))
    if not torch.equal(r[1], runs[0][i][1]): diff(st, r[1], runs[0][i][1], "  ")
for i, (step, batch) in enumerate(zip(build(), batches)):
    loss = step(*batch); torch.cuda.synchronize()
    print("plain model", i, float(loss), "equal", torch.equal(step.flat.flat_grad, runs[0][i][1]))
    if not torch.equal(step.flat.flat_grad, runs[0][i][1]): diff(step, step.flat.flat_grad, runs[0][i][1], "  ")
