"""Part-capsule image encoder (reference: torch_scae/part_encoder.py).

The 3x3 conv + ReLU stack runs on the implicit-GEMM MFMA kernels (K8) for the
reference's channel counts, the 1x1 attention conv on the library, the pose
non-linearity on HIP kernel K5."""
from typing import Tuple

import torch
import torch.nn as nn

from . import cv_ops, ops
from .general_utils import AttrDict
from .nn_ext import Conv2dStack, multiple_attention_pooling_2d
from .nn_utils import measure_shape, rand_like


class CNNEncoder(nn.Module):
    def __init__(self, input_shape, out_channels, kernel_sizes, strides,
                 activation=nn.ReLU, activate_final=True):
        super().__init__()
        self.network = Conv2dStack(in_channels=input_shape[0],
                                   out_channels=out_channels,
                                   kernel_sizes=kernel_sizes, strides=strides,
                                   activation=activation,
                                   activate_final=activate_final)
        self.output_shape = measure_shape(self.network,
                                          input_shape=input_shape)
        self.strides = [int(s) for s in strides]
        # the 3x3 / ReLU / 64-multiple-channel stacks of the reference configs
        # run on the implicit-GEMM HIP kernels; other shapes stay on MIOpen
        self._hip_stack = (activation is nn.ReLU and activate_final
                           and ops.conv_stack_supported(
                               input_shape[0], out_channels, kernel_sizes,
                               strides))

    def forward(self, image):
        if self._hip_stack and image.is_cuda and image.dtype == torch.float32:
            convs = [m for m in self.network if isinstance(m, nn.Conv2d)]
            return ops.conv_stack(image, [c.weight for c in convs],
                                  [c.bias for c in convs], self.strides)
        return self.network(image)


class CapsuleImageEncoder(nn.Module):
    """image -> per part capsule: pose (6), presence, special features
    (part_encoder.py:47-113)."""

    def __init__(self, input_shape: Tuple[int, int, int], encoder: CNNEncoder,
                 n_caps: int, n_poses: int, n_special_features: int = 0,
                 noise_scale: float = 4., similarity_transform: bool = False):
        super().__init__()
        self.input_shape = input_shape
        self.encoder = encoder
        self.n_caps = n_caps
        self.n_poses = n_poses
        self.n_special_features = n_special_features
        self.noise_scale = noise_scale
        self.similarity_transform = similarity_transform

        self.img_embedding_bias = nn.Parameter(
            torch.zeros(tuple(encoder.output_shape), dtype=torch.float32))
        self.caps_dim_splits = [n_poses, 1, n_special_features]
        self.n_total_caps_dims = sum(self.caps_dim_splits)
        self.att_conv = nn.Conv2d(encoder.output_shape[0],
                                  n_caps * (self.n_total_caps_dims + 1),
                                  kernel_size=1, stride=1)
        self.output_shapes = AttrDict(pose=(n_caps, n_poses),
                                      presence=(n_caps,),
                                      feature=(n_caps, n_special_features))

    def forward(self, image):
        batch_size = image.shape[0]
        noisy = self.training and self.noise_scale > 0.
        C, H, W = self.encoder.output_shape
        if image.is_cuda and image.dtype == torch.float32 and \
                self.n_poses == 6 and \
                getattr(self.encoder, "_hip_stack", False) and \
                len(self.encoder.strides) >= 2 and \
                ops.attention_pool_supported(H * W, self.n_caps,
                                             self.n_total_caps_dims + 1):
            # the whole encoder as one autograd node over the HIP kernels
            convs = [m for m in self.encoder.network
                     if isinstance(m, nn.Conv2d)]
            noise = rand_like(image.new_empty(batch_size, self.n_caps)) \
                if noisy else None
            pose, presence, feature, twin, absence = ops.part_encoder(
                image, [c.weight for c in convs], [c.bias for c in convs],
                self.encoder.strides, self.img_embedding_bias,
                self.att_conv.weight, self.att_conv.bias, self.n_caps, noise,
                self.noise_scale, self.similarity_transform)
            return AttrDict(pose=pose, presence=presence, feature=feature,
                            _feature_twin=twin, _absence=absence)
        h = self.encoder(image)
        C, H, W = h.shape[1:]
        if h.is_cuda and h.dtype == torch.float32 and \
                ops.attention_pool_supported(H * W, self.n_caps,
                                             self.n_total_caps_dims + 1):
            # NHWC pixels x (1x1 conv as a GEMM) -> attention pooling kernel
            x = h.permute(0, 2, 3, 1) + self.img_embedding_bias.permute(1, 2, 0)
            x = x.reshape(batch_size, H * W, C)
            weight = self.att_conv.weight.view(-1, C)
            if self.n_poses == 6:
                # ... with the split / noise / sigmoid / pose non-linearity
                # fused behind the pooling
                noise = rand_like(x.new_empty(batch_size, self.n_caps)) \
                    if noisy else None
                pose, presence, feature, absence = ops.capsule_head(
                    x, weight, self.att_conv.bias, self.n_caps, noise,
                    self.noise_scale, self.similarity_transform)
                return AttrDict(pose=pose, presence=presence, feature=feature,
                                _absence=absence)
            h = ops.attention_conv_pool(x, weight, self.att_conv.bias,
                                        self.n_caps)
        else:
            h = self.att_conv(h + self.img_embedding_bias.unsqueeze(0))
            h = multiple_attention_pooling_2d(h, self.n_caps)
            h = h.view(batch_size, self.n_caps, self.n_total_caps_dims)
        pose, presence_logit, special_feature = torch.split(
            h, self.caps_dim_splits, -1)
        if self.n_special_features == 0:
            special_feature = None
        presence_logit = presence_logit.squeeze(-1)
        if noisy:
            noise = (rand_like(presence_logit) - .5) * self.noise_scale
            presence_logit = presence_logit + noise
        presence = torch.sigmoid(presence_logit)
        pose = cv_ops.geometric_transform(pose, self.similarity_transform)
        return AttrDict(pose=pose, presence=presence, feature=special_feature)


# BASELINE.json's north star names this class `part_encoder.PCAE`.
PCAE = CapsuleImageEncoder
