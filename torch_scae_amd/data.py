"""Input formats either side of the hot path.

``pad_and_translate`` is the MNIST training transform of the reference
(torch_scae_experiments/mnist/experiment.py:23-40): 28x28 digits are
zero-padded to the model's 40x40 input and shifted by a random whole number of
pixels of at most the padding in each direction
(``Pad(6)`` + ``RandomAffine(degrees=0, translate=(6/40, 6/40))`` + ``ToTensor``)
-- done here for a whole batch on the device instead of per sample on the
host."""
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
