"""Scalar-ish loss helpers of the host side (reference: torch_scae/math_ops.py).
They act on tiny (B,O) tensors in SCAE.loss; the per-pixel / per-vote
``log_safe`` work of the hot path lives inside the HIP kernels."""
import torch


def log_safe(tensor, eps=1e-16):
    """log(x) with exactly -1e8 below eps (math_ops.py:18-22)."""
    tiny = tensor < eps
    return torch.where(tiny, torch.full_like(tensor, -1e8),
                       torch.log(torch.where(tiny, torch.ones_like(tensor),
                                             tensor)))


def cross_entropy_safe(true_probs, probs, dim=-1):
    """math_ops.py:25-26."""
    return torch.mean(-torch.sum(true_probs * log_safe(probs), dim=dim))


def normalize(tensor, dim):
    """math_ops.py:29-30."""
    return tensor / (torch.sum(tensor, dim, keepdim=True) + 1e-8)


def l2_loss(tensor):
    """math_ops.py:33-34."""
    return torch.sum(tensor ** 2) / 2
