"""Time the parts of the step prologue launch separately (C ABI, HIP events)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch_scae_amd import _lib, ops
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
O, C, D = 24, 256, 16
shapes = [(O, C), (C, C), (C,), (C, C), (C,), (C, C), (C,), (C, C), (C,), (C, D), (C,)]
vals = [(torch.randn(*s, generator=g) / (s[-1] ** 0.5)).cuda() for s in shapes]
image = torch.rand(128, 1, 40, 40, generator=g).cuda(); label = torch.randint(0, 10, (128,), generator=g).cuda()
di, dl = torch.zeros_like(image), torch.zeros_like(label)
pro = ops.StepPrologue()
with ops.step_prologue(pro):
    ops.uniform(128 * 24 * 26, image); ops.seed_fold(*vals)
noise, state, fin, fout, fdims = pro.noise, pro.noise_state, pro.fold_inputs, pro.fold_outs, pro.fold_dims

def t(fn, reps=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

def variant(stage, nz, fold):
    p = ops.StepPrologue()
    if nz: p.noise, p.noise_state = noise, state
    if fold: p.fold_inputs, p.fold_outs, p.fold_dims = fin, fout, fdims
    return (lambda: p.launch(di, image, dl, label)) if stage else (lambda: p.launch())
for name, v in (("stage", (1, 0, 0)), ("noise", (0, 1, 0)), ("fold", (0, 0, 1)), ("all", (1, 1, 1))):
    print(f"{name:6s} {t(variant(*v)):7.2f} us (back-to-back launches incl. host enqueue)")
desc = ops._fold_desc(fin, fout, *fdims)
st = ops._stream(image)
print(f"seed_fold_fwd alone {t(lambda: _lib.call('scae_seed_fold_fwd_f32', ctypes.byref(desc), st)):7.2f} us")
