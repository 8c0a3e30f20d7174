"""Shape helpers, activation lookup and the injectable noise source
(reference: torch_scae/nn_utils.py)."""
import contextlib
from typing import Tuple

import torch
import torch.nn.functional as F


def conv_output_size(in_size: int, kernel_size: int, stride: int = 1,
                     padding: int = 0) -> int:
    return (in_size - kernel_size + 2 * padding) // stride + 1


def conv_output_shape(input_shape: Tuple[int, int, int], out_channels: int,
                      kernel_size: int, stride: int = 1,
                      padding: int = 0) -> Tuple[int, int, int]:
    return (out_channels,
            conv_output_size(input_shape[1], kernel_size, stride, padding),
            conv_output_size(input_shape[2], kernel_size, stride, padding))


def measure_shape(network, input_shape, input_dtype=torch.float32):
    """Output shape (C, H, W) of a conv stack.  The reference pushes a random
    image through the network (nn_utils.py:48-52, consuming one torch RNG
    draw); here the shape is derived from the layer hyper-parameters."""
    shape = tuple(input_shape)
    for layer in network.modules():
        if isinstance(layer, torch.nn.Conv2d):
            k, s, p = layer.kernel_size, layer.stride, layer.padding
            shape = (layer.out_channels,
                     conv_output_size(shape[1], k[0], s[0], p[0]),
                     conv_output_size(shape[2], k[1], s[1], p[1]))
    return torch.Size(shape)


def choose_activation(name):
    """nn_utils.py:55-66."""
    from . import nn_ext
    if name == 'sigmoid':
        return torch.sigmoid
    if name == 'relu1':
        return nn_ext.relu1
    act_fn = getattr(F, name, None)
    if act_fn is None:
        raise ValueError('Invalid activation function: "{}".'.format(name))
    return act_fn


# -- noise ------------------------------------------------------------------
# The reference draws its presence-logit noise with torch.rand_like
# (part_encoder.py:106, object_decoder.py:201).  Device and host RNG streams
# differ, so parity runs replay recorded U[0,1) draws through this hook.
_replay = []


def rand_like(tensor):
    """U[0,1) noise shaped like ``tensor``; replays ``fixed_noise`` draws."""
    if _replay and _replay[-1]:
        draw = _replay[-1].pop(0)
        if draw is None:
            return torch.rand_like(tensor)
        if tuple(draw.shape) != tuple(tensor.shape):
            raise ValueError(f"replayed noise has shape {tuple(draw.shape)}, "
                             f"expected {tuple(tensor.shape)}")
        return draw.to(device=tensor.device, dtype=tensor.dtype)
    return torch.rand_like(tensor)


def replaying():
    """True inside a ``fixed_noise`` block."""
    return bool(_replay)


@contextlib.contextmanager
def fixed_noise(draws):
    """Replay ``draws`` (a list of U[0,1) tensors) for the ``rand_like`` calls
    made inside the block, in call order."""
    _replay.append(list(draws))
    try:
        yield
    finally:
        _replay.pop()
