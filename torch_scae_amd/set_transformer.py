"""Set-transformer object encoder (reference: torch_scae/set_transformer.py).
Module by module (multi-head, ISAB, PMA -- the default single-head SAB stack
runs fused, see ``SetTransformer``): projections and the feed-forward layer on
the batched MFMA GEMM K7 (``ops.HipLinear``), the masked softmax(QK^T)V core
on ``ops.qkv_attention`` (fp32 MFMA), LayerNorm on ``ops.HipLayerNorm``."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def qkv_attention(queries, keys, values, presence=None):
    """Transformer-like attention with a soft presence mask on the keys.

    queries [B, N, d_k], keys [B, M, d_k], values [B, M, d_v], presence None or
    [B, M] -> [B, N, d_v]   (set_transformer.py:24-47).

    The mask is the reference's arithmetic, not a -inf shortcut: the kernel
    subtracts (1 - presence) * 1e32 from the fp32 logits, divides by
    sqrt(d_k) and takes a max-subtracted softmax.
    """
    return ops.qkv_attention(queries, keys, values, presence)


class MultiHeadQKVAttention(nn.Module):
    """Multi-head attention with head-padded projections
    (set_transformer.py:50-104)."""

    def __init__(self, d_k, d_v, n_heads):
        super().__init__()
        self.d_k = d_k
        self.d_v = d_v
        self.n_heads = n_heads
        d_k_p = int(math.ceil(d_k / n_heads)) * n_heads
        d_v_p = int(math.ceil(d_v / n_heads)) * n_heads
        self.q_projector = ops.HipLinear(d_k, d_k_p)
        self.k_projector = ops.HipLinear(d_k, d_k_p)
        self.v_projector = ops.HipLinear(d_v, d_v_p)
        self.o_projector = ops.HipLinear(d_v_p, d_v)

    def forward(self, queries, keys, values, presence=None):
        assert queries.shape[2] == keys.shape[2]
        assert keys.shape[1] == values.shape[1]
        if presence is not None:
            assert values.shape[:2] == presence.shape
        B, N = queries.shape[:2]
        M = values.shape[1]
        H = self.n_heads

        q = self.q_projector(queries)
        k = self.k_projector(keys)
        v = self.v_projector(values)
        if H > 1:      # (B, n, H*d) -> (H*B, n, d), head-major like the reference
            q, k, v = (t.view(B, n, H, -1).permute(2, 0, 1, 3)
                       .reshape(H * B, n, -1)
                       for t, n in ((q, N), (k, M), (v, M)))
            if presence is not None:
                presence = presence.repeat(H, 1)
        o = qkv_attention(q, k, v, presence)
        if H > 1:
            o = o.view(H, B, N, -1).permute(1, 2, 0, 3).reshape(B, N, -1)
        return self.o_projector(o)


class MAB(nn.Module):
    """Multi-head attention block (set_transformer.py:107-133)."""

    def __init__(self, d, n_heads, layer_norm=False):
        super().__init__()
        self.layer_norm = layer_norm
        self.mqkv = MultiHeadQKVAttention(d_k=d, d_v=d, n_heads=n_heads)
        if layer_norm:
            self.ln0 = ops.HipLayerNorm(d)
            self.ln1 = ops.HipLayerNorm(d)
        self.fc = ops.HipLinear(d, d)

    def forward(self, queries, keys, presence=None):
        h = self.mqkv(queries, keys, keys, presence) + queries
        if presence is not None:
            assert presence.shape[1] == queries.shape[1] == keys.shape[1]
            h = h * presence.unsqueeze(-1)
        if self.layer_norm:
            h = self.ln0(h)
        h = h + F.relu(self.fc(h))
        if self.layer_norm:
            h = self.ln1(h)
        return h


class SAB(nn.Module):
    def __init__(self, d, n_heads, layer_norm=False):
        super().__init__()
        self.mab = MAB(d=d, n_heads=n_heads, layer_norm=layer_norm)

    def forward(self, x, presence=None):
        return self.mab(x, x, presence)


class ISAB(nn.Module):
    def __init__(self, d, n_heads, n_inducing_points, layer_norm=False):
        super().__init__()
        self.mab0 = MAB(d=d, n_heads=n_heads, layer_norm=layer_norm)
        self.mab1 = MAB(d=d, n_heads=n_heads, layer_norm=layer_norm)
        self.I = nn.Parameter(torch.zeros(1, n_inducing_points, d))
        with torch.no_grad():
            nn.init.xavier_uniform_(self.I)

    def forward(self, x, presence=None):
        h = self.mab0(self.I.expand(x.shape[0], -1, -1), x, presence)
        return self.mab1(x, h)


class PMA(nn.Module):
    def __init__(self, d, n_heads, n_seeds, layer_norm=False):
        super().__init__()
        self.mab = MAB(d=d, n_heads=n_heads, layer_norm=layer_norm)
        self.S = nn.Parameter(torch.zeros(1, n_seeds, d))
        with torch.no_grad():
            nn.init.xavier_uniform_(self.S)

    def forward(self, x, presence=None):
        return self.mab(self.S.expand(x.shape[0], -1, -1), x, presence)


class SetTransformer(nn.Module):
    """Permutation-invariant transformer (set_transformer.py:174-223):
    fc1 -> n_layers x SAB/ISAB -> fc2 -> attention from learned seeds."""

    def __init__(self, dim_in, dim_hidden, dim_out, n_outputs, n_layers,
                 n_heads, layer_norm=False, n_inducing_points: int = None):
        super().__init__()
        self.fc1 = ops.HipLinear(dim_in, dim_hidden)
        if n_inducing_points is None:
            blocks = [SAB(d=dim_hidden, n_heads=n_heads, layer_norm=layer_norm)
                      for _ in range(n_layers)]
        else:
            blocks = [ISAB(d=dim_hidden, n_heads=n_heads,
                           n_inducing_points=n_inducing_points,
                           layer_norm=layer_norm) for _ in range(n_layers)]
        self.sabs = nn.ModuleList(blocks)
        self.fc2 = ops.HipLinear(dim_hidden, dim_out)
        self.seeds = nn.Parameter(torch.zeros(1, n_outputs, dim_out))
        with torch.no_grad():
            nn.init.xavier_uniform_(self.seeds)
        self.multi_head_attention = MultiHeadQKVAttention(
            d_k=dim_out, d_v=dim_out, n_heads=n_heads)

    # -- fused trunk ---------------------------------------------------------
    def _fusable(self, n_items, presence):
        # the fused trunk is fp32 only: under a bf16 autocast the modules run
        # one by one (bf16 linears, bf16 MFMA attention kernel)
        if not getattr(self, "fused", True) or (
                torch.is_autocast_enabled("cuda")
                and torch.get_autocast_dtype("cuda") != torch.float32):
            return False
        if not (all(isinstance(b, SAB) for b in self.sabs)
                and self.multi_head_attention.n_heads == 1
                and (presence is None or not presence.requires_grad)):
            return False
        layer_norm = bool(self.sabs) and self.sabs[0].mab.layer_norm
        from . import _lib
        return bool(_lib.load().scae_set_encoder_supported(
            n_items, self.fc1.out_features, self.fc1.in_features,
            self.fc2.out_features, len(self.sabs), int(layer_norm)))

    def _flat_param_groups(self):
        """Parameters data_parallel.FlatParameters should lay out back to back
        (then ``_packed_trunk`` is a view, not a copy)."""
        return [self._trunk_parts(True)]

    def _packed_trunk(self, with_fc2=True):
        """All trunk parameters as one flat buffer in the layout the fused
        kernel reads (include/scae_hip.h, K2b)."""
        return ops.pack_params(self._trunk_parts(with_fc2))

    def _trunk_parts(self, with_fc2):
        parts = [self.fc1.weight, self.fc1.bias]
        for sab in self.sabs:
            m = sab.mab
            for lin in (m.mqkv.q_projector, m.mqkv.k_projector,
                        m.mqkv.v_projector, m.mqkv.o_projector):
                parts += [lin.weight, lin.bias]
            if m.layer_norm:
                parts += [m.ln0.weight, m.ln0.bias]
            parts += [m.fc.weight, m.fc.bias]
            if m.layer_norm:
                parts += [m.ln1.weight, m.ln1.bias]
        if with_fc2:
            parts += [self.fc2.weight, self.fc2.bias]
        return parts

    def encode_segments(self, segments, presence=None):
        """fc1 -> blocks -> fc2 on the set whose features are the column-wise
        concatenation of ``segments`` (each (B, N, w)); the concat itself is
        only materialised on the unfused path."""
        if segments[0].is_cuda and self._fusable(segments[0].shape[1], presence):
            layer_norm = bool(self.sabs) and self.sabs[0].mab.layer_norm
            return ops.set_encoder(segments, presence, self._packed_trunk(),
                                   self.fc1.out_features,
                                   self.fc2.out_features, len(self.sabs),
                                   layer_norm)
        h = self.fc1(torch.cat(list(segments), -1))
        for sab in self.sabs:
            h = sab(h, presence)
        return self.fc2(h)

    def _folded_attention_ok(self, n_items, presence):
        mha = self.multi_head_attention
        return (self._fusable(n_items, presence)
                and ops.seed_attention_supported(
                    n_items, self.seeds.shape[1], self.fc1.out_features,
                    mha.o_projector.out_features)
                and mha.q_projector.out_features == mha.o_projector.out_features)

    def forward_segments(self, segments, presence=None):
        if segments[0].is_cuda and \
                self._folded_attention_ok(segments[0].shape[1], presence):
            # trunk without fc2, then the output attention with fc2 and the
            # k / v / o projections folded into two (C x D) maps (kernel K2c);
            # the folding products are tiny batch-invariant GEMMs
            layer_norm = bool(self.sabs) and self.sabs[0].mab.layer_norm
            h = ops.set_encoder(segments, presence,
                                self._packed_trunk(with_fc2=False),
                                self.fc1.out_features, 0, len(self.sabs),
                                layer_norm)
            mha, w2, b2 = self.multi_head_attention, self.fc2.weight, \
                self.fc2.bias
            C, D = w2.shape
            if ops.seed_fold_supported(self.seeds.shape[1], C, D):
                q, wk, bk, wv, bv = ops.seed_fold(
                    self.seeds, mha.q_projector.weight,
                    mha.q_projector.bias, mha.k_projector.weight,
                    mha.k_projector.bias, mha.v_projector.weight,
                    mha.v_projector.bias, mha.o_projector.weight,
                    mha.o_projector.bias, w2, b2)
                return ops.seed_attention(h, q, wk, bk, wv, bv, presence)
            q = mha.q_projector(self.seeds[0])                      # (O, C)
            wk = mha.k_projector.weight @ w2                        # (C, D)
            bk = mha.k_projector.weight @ b2 + mha.k_projector.bias
            wv2 = mha.v_projector.weight @ w2
            bv2 = mha.v_projector.weight @ b2 + mha.v_projector.bias
            wv = mha.o_projector.weight @ wv2
            bv = mha.o_projector.weight @ bv2 + mha.o_projector.bias
            return ops.seed_attention(h, q, wk, bk, wv, bv, presence)
        z = self.encode_segments(segments, presence)
        seeds = self.seeds.expand(z.shape[0], -1, -1)
        return self.multi_head_attention(seeds, z, z, presence)

    def forward(self, x, presence=None):
        return self.forward_segments([x], presence)
