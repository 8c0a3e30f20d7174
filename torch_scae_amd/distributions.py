"""Gaussian mixture over dim 1 on the HIP kernels (reference:
torch_scae/distributions.py:20-89).

Two flavours share the reference's class name and methods:
  * built with ``make_from_stats`` from arbitrary (materialised) tensors ->
    the generic one-lane-per-pixel mixture kernels;
  * built by ``TemplateBasedImageDecoder`` -> additionally carries the compact
    decoder inputs, and ``log_prob`` takes the fused render+mixture kernel
    (K1) that never reads the (B,K,C,H,W) tensors back from HBM.
"""
import math

import torch

from . import ops


class _NormalView:
    """The slice of torch.distributions.Normal the reference touches
    (``pdf.dist.loc`` / ``.mean`` / ``.scale``)."""

    def __init__(self, loc, scale):
        self._loc = loc            # a tensor, or a thunk that renders it
        self.scale = scale

    @property
    def loc(self):
        if callable(self._loc):
            self._loc = self._loc()
        return self._loc

    @property
    def mean(self):
        return self.loc


class GaussianMixture:
    def __init__(self, normal_dist, mixing_logits, _decoder_inputs=None):
        """
        Args:
          normal_dist: object with ``loc`` [B, K, ...] and ``scale`` (one
            element), e.g. torch.distributions.Normal.
          mixing_logits: tensor [B, K, ...] with K the number of components.
        """
        self.dist = normal_dist
        self._mixing_logits = mixing_logits    # a tensor, or a thunk
        self._decoder_inputs = _decoder_inputs

    @property
    def mixing_logits(self):
        if callable(self._mixing_logits):
            self._mixing_logits = self._mixing_logits()
        return self._mixing_logits

    @property
    def n_components(self):
        if callable(self._mixing_logits) and self._decoder_inputs is not None:
            return self._decoder_inputs.templates.shape[1] + 1
        return self.mixing_logits.shape[1]

    def _sigma(self):
        scale = self.dist.scale
        if not torch.is_tensor(scale):
            scale = torch.tensor([float(scale)], device=self.dist.loc.device)
        if scale.numel() != 1:
            raise ops.ScaeHipError("GaussianMixture kernels take one scalar "
                                   "scale shared by all components")
        return scale

    def mixing_log_prob(self):
        """distributions.py:34-35."""
        return torch.log_softmax(self.mixing_logits, 1)

    def mean(self):
        """distributions.py:37-39."""
        return ops.gmm_mean(self._loc5(), self._ml5()).view(self._out_shape())

    def log_prob(self, x):
        """distributions.py:41-44: logsumexp_K(N(x; loc_k, scale) + log pi_k),
        per pixel and per channel."""
        if self._decoder_inputs is not None and not x.requires_grad:
            return ops.render_gmm_log_prob(self._decoder_inputs, x)
        lp = ops.gmm_log_prob(self._loc5(), self._ml5(), self._sigma(),
                              x.reshape(self._out_shape5()))
        return lp.view(self._out_shape())

    def log_prob_tile_sums(self, x):
        """Partial sums (B, tiles) of ``log_prob(x)`` whose total is
        sum_{b,c,h,w} log_prob -- all the training loss needs -- or None when
        the mixture was not built from compact decoder inputs."""
        if self._decoder_inputs is None or x.requires_grad:
            return None
        return ops.render_gmm_log_prob_sums(self._decoder_inputs, x)

    def mode(self, straight_through_gradient=False, maximum=False):
        """distributions.py:50-77: value of the component with the largest
        mixing log-prob (``maximum``: plus its density at its own mean)."""
        if straight_through_gradient:
            # distributions.py:62-75 with the straight-through estimator: the
            # hard one-hot in the forward value, softmax gradients w.r.t. the
            # mixing coefficients.  A rarely used inspection path: composed
            # from device tensor ops rather than a dedicated kernel.
            mlp = self.mixing_log_prob()
            loc = self.dist.loc
            if maximum:
                sigma = self._sigma().to(loc.dtype)
                extra = (-torch.log(sigma) - 0.5 * math.log(2 * math.pi)) \
                    .expand(loc.shape)       # N(loc; loc, sigma)
                if torch.broadcast_shapes(mlp.shape, extra.shape) != mlp.shape:
                    raise RuntimeError(
                        f"output with shape {list(mlp.shape)} doesn't match "
                        f"the broadcast shape {list(loc.shape)}")
                mlp = mlp + extra
            hard = torch.nn.functional.one_hot(mlp.argmax(1), mlp.shape[1]) \
                .movedim(-1, 1).to(mlp.dtype)
            soft = torch.softmax(mlp, 1)
            mask = (hard - soft).detach() + soft
            return torch.sum(mask * loc, 1)
        return ops.gmm_mode(self._loc5(), self._ml5(), self._sigma(),
                            maximum).view(self._out_shape())

    # (B, K, C, P) views for the kernels ------------------------------------
    def _loc5(self):
        loc = self.dist.loc
        if loc.dim() < 3:
            raise ops.ScaeHipError("mixture kernels need loc of rank >= 3 "
                                   "(B, K, C, ...)")
        return loc.reshape(loc.shape[0], loc.shape[1], loc.shape[2], -1)

    def _ml5(self):
        ml = self.mixing_logits
        return ml.reshape(ml.shape[0], ml.shape[1], ml.shape[2], -1)

    def _out_shape(self):
        loc = self.dist.loc
        return (loc.shape[0], *loc.shape[2:])

    def _out_shape5(self):
        loc = self.dist.loc
        return (loc.shape[0], loc.shape[2], -1)

    @classmethod
    def make_from_stats(cls, loc, scale, mixing_logits):
        """distributions.py:79-89."""
        return cls(_NormalView(loc, scale), mixing_logits)
