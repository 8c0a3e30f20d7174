"""Input formats either side of the hot path.

``pad_and_translate`` is the MNIST training transform of the reference
(torch_scae_experiments/mnist/experiment.py:23-40): 28x28 digits are
zero-padded to the model's 40x40 input and shifted by a random whole number of
pixels of at most the padding in each direction
(``Pad(6)`` + ``RandomAffine(degrees=0, translate=(6/40, 6/40))`` + ``ToTensor``)
-- done here for a whole batch on the device instead of per sample on the
host.

``stroke_batches`` is a structured synthetic stand-in for MNIST where no
dataset can be fetched: ten glyph classes, each a fixed set of pen strokes,
rendered under a random affine warp per sample -- images with parts that recur
under pose changes, which is what the part capsules model.  (U[0,1) noise
images have no such structure: trained on them the capsules switch off within
a few hundred steps)"""
import math

import torch


def pad_and_translate(images, out_size=(40, 40), generator=None, shifts=None):
    """images (B, C, h, w) uint8 or float -> (B, C, H, W) float32 in [0, 1].

    ``shifts`` (B, 2) integer (dy, dx) overrides the random draw; otherwise
    each is round(U(-pad, pad)) like torchvision's RandomAffine.get_params
    with the padding of that axis as the maximal shift."""
    B, C, h, w = images.shape
    H, W = out_size
    if H < h or W < w:
        raise ValueError("output smaller than the images")
    x = images.to(torch.float32)
    if not images.dtype.is_floating_point:
        x = x / 255.0                         # ToTensor
    ph, pw = (H - h) // 2, (W - w) // 2
    if shifts is None:
        u = torch.rand(B, 2, generator=generator, device="cpu")
        lim = torch.tensor([ph, pw], dtype=torch.float32)
        shifts = torch.round((u * 2 - 1) * lim).to(torch.int64)
    shifts = shifts.to(images.device)
    # out[b, :, i, j] = x[b, :, i - top_b, j - left_b] where that exists
    top = (ph + shifts[:, 0]).view(B, 1, 1)
    left = (pw + shifts[:, 1]).view(B, 1, 1)
    ii = torch.arange(H, device=images.device).view(1, H, 1) - top   # (B,H,1)
    jj = torch.arange(W, device=images.device).view(1, 1, W) - left  # (B,1,W)
    valid = (ii >= 0) & (ii < h) & (jj >= 0) & (jj < w)              # (B,H,W)
    flat = ii.clamp(0, h - 1) * w + jj.clamp(0, w - 1)                # (B,H,W)
    out = torch.gather(x.reshape(B, C, h * w), 2,
                       flat.view(B, 1, H * W).expand(B, C, H * W))
    return (out.view(B, C, H, W) * valid.unsqueeze(1)).contiguous()


def stroke_batches(n_batches, batch, image_shape, seed=0, device="cpu",
                   n_classes=10, strokes=4, glyph_seed=None):
    """-> (images (n_batches, B, C, H, W) float32 in [0, 1], labels
    (n_batches, B) int64).  Class c is a fixed glyph of ``strokes`` line
    segments (drawn once from ``seed``, or from ``glyph_seed`` when given: a
    held-out set is the same ``glyph_seed`` under another ``seed``); a sample is its glyph under a random
    rotation (+-25 degrees), scale (0.75 .. 1.1), shear and translation
    (+-0.2), drawn with a soft pen (Gaussian profile, sigma 0.07 of the half
    image) -- evaluated analytically per pixel on ``device``."""
    C, H, W = image_shape
    g = torch.Generator(device="cpu").manual_seed(seed)
    # glyphs: endpoints in [-0.75, 0.75]^2, consecutive strokes share an endpoint
    gg = g if glyph_seed is None else torch.Generator(device="cpu").manual_seed(glyph_seed)
    pts = torch.rand(n_classes, strokes + 1, 2, generator=gg) * 1.5 - 0.75
    N = n_batches * batch
    labels = torch.randint(0, n_classes, (N,), generator=g)
    ang = (torch.rand(N, generator=g) * 2 - 1) * math.radians(25.0)
    scale = 0.75 + 0.35 * torch.rand(N, generator=g)
    shear = (torch.rand(N, generator=g) * 2 - 1) * 0.2
    shift = (torch.rand(N, 2, generator=g) * 2 - 1) * 0.2
    colour = 0.6 + 0.4 * torch.rand(N, C, generator=g)
    cos, sin = torch.cos(ang) * scale, torch.sin(ang) * scale
    A = torch.stack([torch.stack([cos, -sin + shear * cos], -1),
                     torch.stack([sin, cos + shear * sin], -1)], -2)   # (N,2,2)
    P = torch.einsum("nij,nkj->nki", A, pts[labels]) + shift[:, None, :]
    P, colour = P.to(device), colour.to(device)
    ys = (2 * torch.arange(H, device=device, dtype=torch.float32) + 1) / H - 1
    xs = (2 * torch.arange(W, device=device, dtype=torch.float32) + 1) / W - 1
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    q = torch.stack([gx, gy], -1).view(1, 1, H * W, 2)         # pixel centres
    a, b = P[:, :-1, None, :], P[:, 1:, None, :]               # (N,S,1,2)
    ab = b - a
    t = ((q - a) * ab).sum(-1) / (ab * ab).sum(-1).clamp_min(1e-8)
    d2 = ((q - (a + t.clamp(0, 1).unsqueeze(-1) * ab)) ** 2).sum(-1)  # (N,S,HW)
    ink = torch.exp(-d2.amin(1) / (2 * 0.07 ** 2)).view(N, 1, H, W)
    images = (ink * colour.view(N, C, 1, 1)).clamp(0, 1)
    return (images.view(n_batches, batch, C, H, W).contiguous(),
            labels.view(n_batches, batch).to(device))
