"""Layer builders shared by the capsule modules (reference: torch_scae/nn_ext.py).

``MLP`` / ``Conv2dStack`` keep the reference's ``nn.Sequential`` layout (and so
its state_dict keys).  ``GroupedMLP`` is new: the O independent per-capsule
MLPs that the reference evaluates in a Python loop of tiny GEMMs
(object_decoder.py:137-158) are held as stacked weights and run as batched
GEMMs, while ``state_dict`` still reads and writes the reference's per-capsule
keys.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def MLP(sizes, activation=nn.ReLU, activate_final=True, bias=True):
    """nn_ext.py:19-31 -- note the activation after the last layer too."""
    assert len(sizes) >= 2, "There must be at least two sizes"
    layers = []
    for fan_in, fan_out in zip(sizes[:-1], sizes[1:]):
        layers += [nn.Linear(fan_in, fan_out, bias=bias), activation()]
    if not activate_final:
        layers.pop()
    return nn.Sequential(*layers)


def Conv2dStack(in_channels, out_channels, kernel_sizes, strides,
                activation=nn.ReLU, activate_final=True):
    """nn_ext.py:34-59."""
    assert len(out_channels) == len(kernel_sizes) == len(strides)
    layers = []
    c_in = in_channels
    for c_out, k, s in zip(out_channels, kernel_sizes, strides):
        layers += [nn.Conv2d(c_in, c_out, kernel_size=k, stride=s),
                   activation()]
        c_in = c_out
    if not activate_final:
        layers.pop()
    return nn.Sequential(*layers)


class GroupedMLP(nn.Module):
    """``n_groups`` independent ReLU MLPs evaluated as batched GEMMs.

    forward: (B, n_groups, sizes[0]) -> (B, n_groups, sizes[-1]); ReLU after
    every layer (nn_ext.MLP's activate_final=True default).  With
    ``ones_input`` the input is taken to have an implicit trailing 1.0 column
    (the ``caps_exist`` concat of object_decoder.py:144-151), handled as an
    additive term instead of a concat.

    state_dict layout == the reference's ``ModuleList([MLP(...)] * n_groups)``:
    ``{g}.{2*layer}.weight`` / ``.bias``.
    """

    def __init__(self, n_groups, sizes, bias=True, ones_input=False):
        super().__init__()
        self.n_groups = n_groups
        self.sizes = list(sizes)
        self.ones_input = ones_input
        self.n_layers = len(self.sizes) - 1
        self.has_bias = bias
        # stacked parameters are registered directly on this module (no child
        # containers), so the per-capsule checkpoint mapping below is the only
        # thing load_state_dict / state_dict see
        for j, (fan_in, fan_out) in enumerate(zip(self.sizes[:-1],
                                                  self.sizes[1:])):
            bound = 1.0 / math.sqrt(fan_in)    # nn.Linear's default init
            self.register_parameter(f"stacked_weight_{j}", nn.Parameter(
                torch.empty(n_groups, fan_out, fan_in).uniform_(-bound, bound)))
            if bias:
                self.register_parameter(f"stacked_bias_{j}", nn.Parameter(
                    torch.empty(n_groups, fan_out).uniform_(-bound, bound)))

    @property
    def weights(self):
        return [getattr(self, f"stacked_weight_{j}")
                for j in range(self.n_layers)]

    @property
    def biases(self):
        if not self.has_bias:
            return None
        return [getattr(self, f"stacked_bias_{j}")
                for j in range(self.n_layers)]

    def forward(self, x, grad_pregated=False, x_is_relu=False, pad_out=False):
        """``grad_pregated`` / ``x_is_relu``: private backward contracts of
        a fused chain (ops._GroupedMLP); HIP path only."""
        # batched fp32-MFMA GEMMs with fused bias / ReLU (K7); like every hot
        # op there is no eager form: a CPU tensor raises ops.ScaeHipError
        from . import ops
        return ops.grouped_mlp(x, self.weights, self.biases,
                               ones_input=self.ones_input,
                               grad_pregated=grad_pregated,
                               x_is_relu=x_is_relu, pad_out=pad_out)

    # -- reference-compatible (per-capsule) checkpoint keys ------------------
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for g in range(self.n_groups):
            for j, w in enumerate(self.weights):
                t = w[g]
                destination[f"{prefix}{g}.{2 * j}.weight"] = \
                    t if keep_vars else t.detach()
                if self.biases is not None:
                    b = self.biases[j][g]
                    destination[f"{prefix}{g}.{2 * j}.bias"] = \
                        b if keep_vars else b.detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict,
                              missing_keys, unexpected_keys, error_msgs):
        def gather(param, suffix, j):
            rows = []
            for g in range(self.n_groups):
                key = f"{prefix}{g}.{2 * j}.{suffix}"
                if key not in state_dict:
                    missing_keys.append(key)
                    return
                rows.append(state_dict[key])
            stacked = torch.stack([r.to(param.dtype) for r in rows])
            if stacked.shape != param.shape:
                error_msgs.append(
                    f"size mismatch for {prefix}*.{2 * j}.{suffix}: "
                    f"{tuple(stacked.shape)} vs {tuple(param.shape)}")
                return
            with torch.no_grad():
                param.copy_(stacked)

        for j, w in enumerate(self.weights):
            gather(w, "weight", j)
            if self.biases is not None:
                gather(self.biases[j], "bias", j)
        if strict:
            known = {f"{prefix}{g}.{2 * j}.{s}"
                     for g in range(self.n_groups)
                     for j in range(len(self.weights))
                     for s in (("weight", "bias") if self.biases is not None
                               else ("weight",))}
            for key in state_dict:
                if key.startswith(prefix) and key not in known:
                    unexpected_keys.append(key)


def multiple_soft_attention(feature_map, n_attention_map):
    """nn_ext.py:76-93: per attention map, softmax over pixels of its last
    channel, applied to its other channels."""
    batch_size, n_channels, height, width = feature_map.shape
    assert n_attention_map > 0
    assert n_channels > n_attention_map, \
        "Attention maps cannot be more than feature maps"
    assert n_channels % n_attention_map == 0, "Incompatible attention map count"
    per_map = n_channels // n_attention_map
    fm = feature_map.view(batch_size, n_attention_map, per_map, height * width)
    mask = F.softmax(fm[:, :, -1:, :], dim=-1)
    out = fm[:, :, :-1, :] * mask
    return out.reshape(batch_size, n_channels - n_attention_map, height, width)


def multiple_attention_pooling_2d(feature_map, n_attention_map):
    """nn_ext.py:96-101 -> (B, C - A, 1, 1)."""
    x = multiple_soft_attention(feature_map, n_attention_map)
    return x.flatten(2).sum(-1, keepdim=True).unsqueeze(-1)


def relu1(x):
    """nn_ext.py:139-140."""
    return F.relu6(x * 6.) / 6.


def named_reference_grads(model, grad_of=None):
    """{reference state_dict key: gradient} for every parameter of ``model``,
    un-stacking ``GroupedMLP`` parameters into the per-capsule keys (None for
    parameters that received no gradient).  ``grad_of``: parameter -> its
    gradient (default ``p.grad``; e.g. a lookup into the flat gradient buffer
    of ``data_parallel.FlatParameters`` after a replayed step)."""
    if grad_of is None:
        grad_of = lambda p: p.grad      # noqa: E731
    out = {}
    for mod_name, mod in model.named_modules():
        prefix = mod_name + "." if mod_name else ""
        if isinstance(mod, GroupedMLP):
            for j in range(mod.n_layers):
                for kind, stacked in (("weight", mod.weights[j]),
                                      ("bias", mod.biases[j] if mod.has_bias
                                       else None)):
                    if stacked is None:
                        continue
                    grad = grad_of(stacked)
                    for g in range(mod.n_groups):
                        out[f"{prefix}{g}.{2 * j}.{kind}"] = \
                            None if grad is None else grad[g]
        else:
            for pname, p in mod.named_parameters(recurse=False):
                out[prefix + pname] = grad_of(p)
    return out
