"""Dump K1 backward outputs for a fixed input (run under two SCAE_HIP_LIB builds, then compare)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch_scae_amd.part_decoder import TemplateBasedImageDecoder
B, M, C, HW, ts = 128, 24, 1, (40, 40), (11, 11)
torch.manual_seed(0)
dec = TemplateBasedImageDecoder(M, ts, HW, learn_output_scale=False, use_alpha_channel=True)
g = torch.Generator().manual_seed(5)
with torch.no_grad():
    for p in dec.parameters():
        p.copy_(torch.randn(p.shape, generator=g) * 0.5)
templates = torch.rand(B, M, C, *ts, generator=g)
pose = torch.randn(B, M, 6, generator=g) * 0.5
pose[:, :, 0] += 1.0
pose[:, :, 4] += 1.0
presence = torch.rand(B, M, generator=g)
x = torch.rand(B, C, *HW, generator=g)
dec = dec.cuda()
leaf = lambda t: t.clone().cuda().requires_grad_(True)
tg, pg, prg = leaf(templates), leaf(pose), leaf(presence)
rg = dec(tg, pg, prg)
lp = rg.pdf.log_prob(x.cuda())
lp.flatten(1).sum(-1).mean().backward()
out = dict(lp=lp.detach().cpu(), gt=tg.grad.cpu(), gp=pg.grad.cpu(), gpr=prg.grad.cpu(),
           **{"p_" + k: v.grad.cpu() for k, v in dec.named_parameters() if v.grad is not None})
torch.save(out, sys.argv[1])
