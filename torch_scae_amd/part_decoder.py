"""Template-based part decoder on the HIP kernel K1 (reference:
torch_scae/part_decoder.py)."""
from typing import Tuple

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .distributions import GaussianMixture, _NormalView
from .general_utils import AttrDict, LazyAttrDict, prod
from .nn_ext import MLP, relu1
from .nn_utils import choose_activation


class TemplateGenerator(nn.Module):
    """Learned templates, optionally coloured per capsule from its features
    (part_decoder.py:31-110)."""

    def __init__(self, n_templates, n_channels, template_size,
                 template_nonlin='relu1', dim_feature=None,
                 colorize_templates=False, color_nonlin='relu1'):
        super().__init__()
        self.n_templates = n_templates
        self.template_size = template_size
        self.n_channels = n_channels
        self._nonlin_names = (template_nonlin, color_nonlin)
        self.template_nonlin = choose_activation(template_nonlin)
        self.dim_feature = dim_feature
        self.colorize_templates = colorize_templates
        self.color_nonlin = choose_activation(color_nonlin)

        # mutually orthogonal templates at init (part_decoder.py:60-69)
        shape = (1, n_templates, n_channels, *template_size)
        n_elems = prod(shape[2:])
        n = max(n_templates, n_elems)
        basis = np.linalg.qr(np.random.uniform(size=[n, n]))[0]
        basis = basis[:n_templates, :n_elems].reshape(shape).astype(np.float32)
        basis = (basis - basis.min()) / (basis.max() - basis.min())
        self.template_logits = nn.Parameter(torch.from_numpy(basis))
        if colorize_templates:
            self.templates_color_mlp = MLP(sizes=[dim_feature, 32, n_channels])

    def forward(self, feature=None, batch_size=None):
        """feature [B, M, F] or None (+ batch_size) ->
        AttrDict(raw_templates (1,M,C,h,w), templates (B,M,C,h,w))."""
        if feature is not None:
            batch_size = feature.shape[0]
        if self.colorize_templates and feature is not None and \
                feature.is_cuda and feature.dtype == torch.float32 and \
                len(self.templates_color_mlp) == 4 and \
                ops.template_color_supported(
                    feature.shape[1], self.n_channels, feature.shape[2],
                    self.templates_color_mlp[0].out_features,
                    *self._nonlin_names):
            # non-linearities, colour MLP and the product in one kernel
            mlp = self.templates_color_mlp
            raw_templates, templates = ops.colored_templates(
                self.template_logits, feature, mlp[0].weight, mlp[0].bias,
                mlp[2].weight, mlp[2].bias, *self._nonlin_names)
            return AttrDict(raw_templates=raw_templates, templates=templates)
        raw_templates = self.template_nonlin(self.template_logits)
        if self.colorize_templates and feature is not None:
            n_templates = feature.shape[1]
            color = self.templates_color_mlp(
                feature.reshape(batch_size * n_templates, -1))
            if self.color_nonlin == relu1:
                color = color + .99      # out of place (part_decoder.py:97-98)
            color = self.color_nonlin(color).view(batch_size, n_templates, -1)
            templates = raw_templates * color[:, :, :, None, None]
        else:
            templates = raw_templates.repeat(batch_size, 1, 1, 1, 1)
        return AttrDict(raw_templates=raw_templates, templates=templates)


class _Rendered:
    """The two rendered outputs of a decoder call, made on first use."""

    def __init__(self, inputs):
        self.inputs, self.out = inputs, None

    def get(self, i):
        if self.out is None:
            self.out = ops.render_templates(self.inputs)
        return self.out[i]


class TemplateBasedImageDecoder(nn.Module):
    """Affine-warps every template (and its alpha map) into the image frame
    and returns the per-pixel Gaussian mixture over the M templates plus a
    background component (part_decoder.py:113-243)."""

    def __init__(self, n_templates: int, template_size: Tuple[int, int],
                 output_size: Tuple[int, int], learn_output_scale=False,
                 use_alpha_channel=False, background_value=True):
        super().__init__()
        self.n_templates = n_templates
        self.template_size = template_size
        self.output_size = output_size
        self.learn_output_scale = learn_output_scale
        self.use_alpha_channel = use_alpha_channel
        self.background_value = background_value
        # True: the two (B, M+1, ., H, W) outputs are rendered on first access
        # instead of in forward() -- a training step only needs
        # pdf.log_prob(...) of the fused kernel, which never reads them
        # (SURVEY.md section 7's loss-only path; train_step.TrainStep sets it)
        self.lazy_render = False

        if use_alpha_channel:
            self.templates_alpha = nn.Parameter(
                torch.zeros(1, n_templates, 1, *template_size))
        else:
            self.temperature_logit = nn.Parameter(torch.rand(1))
        if learn_output_scale:
            self.scale = nn.Parameter(torch.rand(1))
        self.bg_mixing_logit = nn.Parameter(torch.tensor([0.0]))
        # the fixed output scale of the not-learned case (part_decoder.py:201)
        self.register_buffer("_unit_scale", torch.ones(1), persistent=False)
        if background_value:
            self.bg_value = nn.Parameter(torch.tensor([0.0]))

    def decoder_inputs(self, templates, pose, presence=None, bg_image=None):
        """The compact description of one decoder call (what the fused
        likelihood / render kernels read)."""
        return ops.DecoderInputs(
            tuple(self.output_size),
            templates=templates,
            templates_alpha=self.templates_alpha if self.use_alpha_channel
            else None,
            pose=pose, presence=presence, bg_image=bg_image,
            bg_value=self.bg_value if self.background_value else None,
            bg_mixing_logit=self.bg_mixing_logit,
            temperature_logit=None if self.use_alpha_channel
            else self.temperature_logit,
            out_scale=self.scale if self.learn_output_scale else None)

    def forward(self, templates, pose, presence=None, bg_image=None):
        """templates (B,M,C,h,w), pose [B,M,6], presence [B,M] or None,
        bg_image [B,C,H,W] or None -> AttrDict(transformed_templates
        (B,M+1,C,H,W), mixing_logits (B,M+1,1|C,H,W), pdf)."""
        if bg_image is None and not self.background_value:
            # the reference reads self.bg_value here (part_decoder.py:192)
            raise AttributeError("'TemplateBasedImageDecoder' object has no "
                                 "attribute 'bg_value'")
        if pose.shape[-1] != 6 or pose.shape[1] != templates.shape[1] or \
                pose.shape[0] % templates.shape[0]:
            raise ValueError("pose must be [B, n_templates, 6]")
        if pose.shape[0] != templates.shape[0] and torch.is_grad_enabled() \
                and templates.requires_grad:
            # consecutive groups of images sharing a template set is a
            # forward-only feature of the kernels
            templates = templates.repeat_interleave(
                pose.shape[0] // templates.shape[0], dim=0)
        inputs = self.decoder_inputs(templates, pose, presence, bg_image)
        if self.learn_output_scale:
            scale = nn.functional.softplus(self.scale) + 1e-4
        else:
            scale = self._unit_scale
        if self.lazy_render:
            rendered = _Rendered(inputs)

            def fill(d):
                dict.__setitem__(d, "transformed_templates", rendered.get(0))
                dict.__setitem__(d, "mixing_logits", rendered.get(1))
                d["_lazy"].clear()

            out = LazyAttrDict(pdf=GaussianMixture(
                _NormalView(lambda: rendered.get(0), scale),
                lambda: rendered.get(1), _decoder_inputs=inputs))
            out["_lazy"] = dict(transformed_templates=fill, mixing_logits=fill)
            return out
        transformed_templates, mixing_logits = ops.render_templates(inputs)
        pdf = GaussianMixture(_NormalView(transformed_templates, scale),
                              mixing_logits, _decoder_inputs=inputs)
        return AttrDict(transformed_templates=transformed_templates,
                        mixing_logits=mixing_logits, pdf=pdf)
