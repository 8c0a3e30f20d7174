"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A CPU, fp32, op-for-op restatement (plain PyTorch eager, functional style) of
the SCAE forward / loss hot path of bdsaglam/torch-scae, the reference this
repository is a drop-in for.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this file, and only as the
CHECKER (or the timed CPU baseline) -- never as something the product routes
through.  ``torch_scae_amd`` must not import it.

Parity pinning: every function here is checked against golden vectors captured
from the real reference (imported from /root/reference in the build container
by ``tests/golden/make_golden.py``; fixtures in ``tests/golden/*.npz``) by
``tests/test_oracle_vs_golden.py``.  The reference's own unit tests hold no
numeric vectors (they are shape-only), so those captured vectors are the pin.

All ``file:line`` citations are relative to the reference checkout
(``torch_scae/...``).  Parameters are addressed through a flat ``dict`` that
uses the reference's ``state_dict`` key names, so one parameter set feeds the
reference, this oracle and the HIP modules alike.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LOG_2PI_HALF = 0.5 * math.log(2.0 * math.pi)


class Bag(dict):
    """dict with attribute access (stand-in for monty's AttrDict)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


# ----------------------------------------------------------------------------
# math_ops.py
# ----------------------------------------------------------------------------
def log_safe(x, eps=1e-16):
    """math_ops.py:18-22 -- log(x), but exactly -1e8 where x < eps."""
    small = x < eps
    guarded = torch.where(small, torch.ones_like(x), x)
    return torch.where(small, torch.full_like(x, -1e8), torch.log(guarded))


def normalize(x, dim):
    """math_ops.py:29-30."""
    return x / (x.sum(dim, keepdim=True) + 1e-8)


def cross_entropy_safe(p_true, p, dim=-1):
    """math_ops.py:25-26."""
    return (-(p_true * log_safe(p)).sum(dim)).mean()


def l2_loss(x):
    """math_ops.py:33-34."""
    return (x ** 2).sum() / 2


def relu1(x):
    """nn_ext.py:139-140."""
    return F.relu6(x * 6.) / 6.


def activation_by_name(name):
    """nn_utils.py:55-66."""
    if name == 'sigmoid':
        return torch.sigmoid
    if name == 'relu1':
        return relu1
    fn = getattr(F, name, None)
    if fn is None:
        raise ValueError('Invalid activation function: "{}".'.format(name))
    return fn


# ----------------------------------------------------------------------------
# cv_ops.py
# ----------------------------------------------------------------------------
def geometric_transform(pose, similarity=False, nonlinear=True,
                        as_matrix=False):
    """cv_ops.py:20-76.  (..., 6) -> (..., 6) or (..., 3, 3)."""
    sx, sy, theta, shear, tx, ty = (pose[..., i:i + 1] for i in range(6))
    if nonlinear:
        sx = torch.sigmoid(sx) + 1e-2                       # :41
        sy = torch.sigmoid(sy) + 1e-2
        tx = torch.tanh(tx * 5.)                            # :43-44
        ty = torch.tanh(ty * 5.)
        shear = torch.tanh(shear * 5.)
        theta = theta * (2. * math.pi)                      # :45
    else:
        sx = abs(sx) + 1e-2                                 # :47
        sy = abs(sy) + 1e-2
    c, s = torch.cos(theta), torch.sin(theta)
    if similarity:                                          # :51-54
        rows = [sx * c, -sx * s, tx, sx * s, sx * c, ty]
    else:                                                   # :56-63
        rows = [sx * c + shear * sy * s, -sx * s + shear * sy * c, tx,
                sy * s, sy * c, ty]
    out = torch.cat(rows, -1)
    if as_matrix:                                           # :68-74
        out = out.reshape(*out.shape[:-1], 2, 3)
        last = torch.zeros_like(out[..., :1, :])
        last[..., 0, 2] = 1.0
        out = torch.cat([out, last], -2)
    return out


# ----------------------------------------------------------------------------
# set_transformer.py
# ----------------------------------------------------------------------------
def qkv_attention(q, k, v, presence=None):
    """set_transformer.py:24-47."""
    d_k = q.shape[-1]
    routing = torch.matmul(q, k.transpose(1, 2))            # :40
    if presence is not None:
        routing = routing - (1. - presence.unsqueeze(-2)) * 1e32   # :42
    routing = F.softmax(routing / np.sqrt(d_k), -1)         # :43
    return torch.matmul(routing, v)                         # :47


def _linear(P, prefix, x):
    b = P.get(prefix + '.bias')
    return F.linear(x, P[prefix + '.weight'], b)


def multi_head_attention(P, prefix, q, k, v, presence, n_heads):
    """MultiHeadQKVAttention.forward, set_transformer.py:68-104."""
    assert q.shape[2] == k.shape[2]
    assert k.shape[1] == v.shape[1]
    if presence is not None:
        assert v.shape[:2] == presence.shape
    B, N, _ = q.shape
    M = v.shape[1]
    H = n_heads
    qp = _linear(P, prefix + '.q_projector', q)
    kp = _linear(P, prefix + '.k_projector', k)
    vp = _linear(P, prefix + '.v_projector', v)

    def heads(t, n):                                        # :95-97
        return t.view(B, n, H, -1).permute(2, 0, 1, 3).contiguous() \
                .view(H * B, n, -1)

    if presence is not None:
        presence = presence.repeat(H, 1)                    # :100
    o = qkv_attention(heads(qp, N), heads(kp, M), heads(vp, M), presence)
    o = o.view(H, B, N, -1).permute(1, 2, 0, 3).contiguous().view(B, N, -1)
    return _linear(P, prefix + '.o_projector', o)           # :104


def mab(P, prefix, queries, keys, presence, n_heads, layer_norm):
    """MAB.forward, set_transformer.py:118-133."""
    h = multi_head_attention(P, prefix + '.mqkv', queries, keys, keys,
                             presence, n_heads)
    h = h + queries
    if presence is not None:
        assert presence.shape[1] == queries.shape[1] == keys.shape[1]  # :123
        h = h * presence.unsqueeze(-1)
    d = h.shape[-1]
    if layer_norm:
        h = F.layer_norm(h, (d,), P[prefix + '.ln0.weight'],
                         P[prefix + '.ln0.bias'])
    h = h + F.relu(_linear(P, prefix + '.fc', h))           # :130
    if layer_norm:
        h = F.layer_norm(h, (d,), P[prefix + '.ln1.weight'],
                         P[prefix + '.ln1.bias'])
    return h


def sab(P, prefix, x, presence, n_heads, layer_norm):
    """SAB, set_transformer.py:136-142."""
    return mab(P, prefix + '.mab', x, x, presence, n_heads, layer_norm)


def isab(P, prefix, x, presence, n_heads, layer_norm):
    """ISAB, set_transformer.py:145-158."""
    ind = P[prefix + '.I'].repeat(x.shape[0], 1, 1)
    h = mab(P, prefix + '.mab0', ind, x, presence, n_heads, layer_norm)
    return mab(P, prefix + '.mab1', x, h, None, n_heads, layer_norm)


def pma(P, prefix, x, presence, n_heads, layer_norm):
    """PMA, set_transformer.py:161-171."""
    s = P[prefix + '.S'].repeat(x.shape[0], 1, 1)
    return mab(P, prefix + '.mab', s, x, presence, n_heads, layer_norm)


def set_transformer(P, prefix, x, presence, n_layers, n_heads,
                    layer_norm=False, n_inducing_points=None):
    """SetTransformer.forward, set_transformer.py:212-223."""
    h = _linear(P, prefix + '.fc1', x)
    for l in range(n_layers):
        if n_inducing_points is None:
            h = sab(P, f'{prefix}.sabs.{l}', h, presence, n_heads, layer_norm)
        else:
            h = isab(P, f'{prefix}.sabs.{l}', h, presence, n_heads,
                     layer_norm)
    z = _linear(P, prefix + '.fc2', h)
    s = P[prefix + '.seeds'].repeat(x.shape[0], 1, 1)
    return multi_head_attention(P, prefix + '.multi_head_attention', s, z, z,
                                presence, n_heads)


# ----------------------------------------------------------------------------
# nn_ext.py
# ----------------------------------------------------------------------------
def mlp(P, prefix, x, n_linear):
    """nn_ext.MLP, nn_ext.py:19-31 -- ReLU after EVERY layer (final one too).
    Sequential indices are 0,2,4,.. (ReLUs sit on the odd ones)."""
    for j in range(n_linear):
        x = F.relu(_linear(P, f'{prefix}.{2 * j}', x))
    return x


def multiple_attention_pooling_2d(fmap, n_maps):
    """nn_ext.py:76-101."""
    B, C, H, W = fmap.shape
    assert n_maps > 0
    assert C > n_maps, "Attention maps cannot be more than feature maps"
    assert C % n_maps == 0, "Incompatible attention map count"
    f = fmap.view(B, n_maps, C // n_maps, H * W)
    mask = F.softmax(f[:, :, -1:, :], dim=-1)
    pooled = (f[:, :, :-1, :] * mask).sum(-1)
    return pooled.reshape(B, C - n_maps, 1, 1)


# ----------------------------------------------------------------------------
# part_encoder.py
# ----------------------------------------------------------------------------
def cnn_encoder(P, prefix, image, strides):
    """CNNEncoder / Conv2dStack, part_encoder.py:26-44, nn_ext.py:34-59."""
    h = image
    for i, s in enumerate(strides):
        h = F.relu(F.conv2d(h, P[f'{prefix}.network.{2 * i}.weight'],
                            P[f'{prefix}.network.{2 * i}.bias'], stride=s))
    return h


def capsule_image_encoder(P, prefix, image, cfg_cnn, cfg_enc, training,
                          noise=None):
    """CapsuleImageEncoder.forward, part_encoder.py:86-113.

    ``noise``: the U[0,1) draw of part_encoder.py:106 (shape (B, M)); required
    when ``training`` and noise_scale > 0.
    """
    B = image.shape[0]
    M = cfg_enc['n_caps']
    n_poses = cfg_enc['n_poses']
    n_special = cfg_enc.get('n_special_features', 0)
    noise_scale = cfg_enc.get('noise_scale', 4.)
    emb = cnn_encoder(P, prefix + '.encoder', image, cfg_cnn['strides'])
    h = emb + P[prefix + '.img_embedding_bias'].unsqueeze(0)
    h = F.conv2d(h, P[prefix + '.att_conv.weight'], P[prefix + '.att_conv.bias'])
    h = multiple_attention_pooling_2d(h, M).view(B, M, n_poses + 1 + n_special)
    pose, logit, feature = torch.split(h, [n_poses, 1, n_special], -1)
    if n_special == 0:
        feature = None
    logit = logit.squeeze(-1)
    if training and noise_scale > 0.:
        logit = logit + (noise - .5) * noise_scale          # :106-107
    presence = torch.sigmoid(logit)
    pose = geometric_transform(pose, cfg_enc.get('similarity_transform', False))
    return Bag(pose=pose, presence=presence, feature=feature)


# ----------------------------------------------------------------------------
# part_decoder.py / distributions.py
# ----------------------------------------------------------------------------
def template_generator(P, prefix, feature, batch_size, cfg):
    """TemplateGenerator.forward, part_decoder.py:75-110."""
    if feature is not None:
        batch_size = feature.shape[0]
    raw = activation_by_name(cfg.get('template_nonlin', 'relu1'))(
        P[prefix + '.template_logits'])
    if cfg.get('colorize_templates', False) and feature is not None:
        M = feature.shape[1]
        color = mlp(P, prefix + '.templates_color_mlp',
                    feature.reshape(batch_size * M, -1), 2)
        cname = cfg.get('color_nonlin', 'relu1')
        if cname == 'relu1':
            color = color + .99                             # :97-98
        color = activation_by_name(cname)(color).view(batch_size, M, -1)
        templates = raw * color[:, :, :, None, None]
    else:
        templates = raw.repeat(batch_size, 1, 1, 1, 1)
    return Bag(raw_templates=raw, templates=templates)


def bilinear_warp(src, theta, out_hw):
    """affine_grid + grid_sample(bilinear, zeros, align_corners=False) written
    out explicitly (the semantics part_decoder.py:181-183 gets from torch).

    src (N,C,h,w), theta (N,2,3) -> (N,C,H,W).  Used by the tests to pin the
    formulas the HIP kernel implements; ``image_decoder`` below calls the torch
    ops themselves, exactly like the reference.
    """
    N, C, h, w = src.shape
    H, W = out_hw
    xs = (2 * torch.arange(W, dtype=src.dtype) + 1) / W - 1
    ys = (2 * torch.arange(H, dtype=src.dtype) + 1) / H - 1
    gy, gx = torch.meshgrid(ys, xs, indexing='ij')
    base = torch.stack([gx, gy, torch.ones_like(gx)], -1).view(1, H * W, 3)
    g = torch.matmul(base, theta.transpose(1, 2)).view(N, H, W, 2)
    ix = ((g[..., 0] + 1) * w - 1) / 2
    iy = ((g[..., 1] + 1) * h - 1) / 2
    x0, y0 = torch.floor(ix), torch.floor(iy)
    out = torch.zeros(N, C, H, W, dtype=src.dtype)
    flat = src.reshape(N, C, h * w)
    for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):
        xi, yi = x0 + dx, y0 + dy
        wgt = (1 - (ix - xi).abs()) * (1 - (iy - yi).abs())
        ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
        idx = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1)).long()
        tap = torch.gather(flat, 2, idx.view(N, 1, H * W).expand(N, C, H * W))
        out = out + tap.view(N, C, H, W) * (wgt * ok).unsqueeze(1)
    return out


def image_decoder(P, prefix, templates, pose, presence, bg_image, cfg):
    """TemplateBasedImageDecoder.forward, part_decoder.py:152-243.

    Returns Bag(transformed_templates, mixing_logits, scale); the Gaussian
    mixture of part_decoder.py:233-237 is (loc=transformed_templates,
    scale, mixing_logits) -- see ``gmm_*`` below.
    """
    B, M, C, th, tw = templates.shape
    H, W = cfg['output_size']
    grids = F.affine_grid(pose.reshape(B * M, 2, 3), [B * M, C, H, W],
                          align_corners=False)              # :181
    tt = F.grid_sample(templates.reshape(B * M, C, th, tw), grids,
                       align_corners=False).view(B, M, C, H, W)   # :182-185
    if bg_image is not None:
        bg = bg_image.unsqueeze(1)                          # :190
    else:
        bg = torch.sigmoid(P[prefix + '.bg_value']).repeat(B, 1, C, H, W)
    tt = torch.cat([tt, bg], 1)                             # :195
    if cfg.get('use_alpha_channel', False):
        alpha = P[prefix + '.templates_alpha'].repeat(B, 1, 1, 1, 1) \
            .view(B * M, 1, th, tw)
        ml = F.grid_sample(alpha, grids, align_corners=False) \
            .view(B, M, 1, H, W)                            # :205-208
        bg_ml = F.softplus(P[prefix + '.bg_mixing_logit']).repeat(B, 1, 1, H, W)
        ml = torch.cat([ml, bg_ml], 1)                      # :210-213
    else:
        temperature = F.softplus(P[prefix + '.temperature_logit'] + .5) + 1e-4
        ml = tt / temperature                               # :216-217
    if cfg.get('learn_output_scale', False):
        scale = F.softplus(P[prefix + '.scale']) + 1e-4     # :221
    else:
        scale = torch.tensor([1.0])
    if presence is not None:
        full = torch.cat([presence, presence.new_ones(B, 1)], 1)
        ml = ml + log_safe(full).view(B, M + 1, 1, 1, 1)    # :225-231
    return Bag(transformed_templates=tt, mixing_logits=ml, scale=scale)


def normal_log_prob(x, loc, scale):
    """torch.distributions.Normal.log_prob (used at distributions.py:47,
    object_decoder.py:268)."""
    var = scale ** 2
    return -((x - loc) ** 2) / (2 * var) - torch.log(scale) - LOG_2PI_HALF


def gmm_mixing_log_prob(mixing_logits):
    """distributions.py:34-35."""
    return F.log_softmax(mixing_logits, 1)


def gmm_log_prob(loc, scale, mixing_logits, x):
    """GaussianMixture.log_prob, distributions.py:41-44."""
    lp = normal_log_prob(x.unsqueeze(1), loc, scale)
    return torch.logsumexp(lp + gmm_mixing_log_prob(mixing_logits), 1)


def gmm_mean(loc, mixing_logits):
    """distributions.py:37-39."""
    return (F.softmax(mixing_logits, 1) * loc).sum(1)


def gmm_mode(loc, scale, mixing_logits, straight_through_gradient=False,
             maximum=False):
    """distributions.py:50-77.  With ``maximum`` the reference adds the
    component log-prob IN PLACE into the (B,K,1,..) logits, which only
    broadcasts when loc has one channel (or logits are per-channel); the
    restatement raises the same RuntimeError otherwise."""
    mlp_ = gmm_mixing_log_prob(mixing_logits)
    if maximum:
        extra = normal_log_prob(loc, loc, scale)
        if torch.broadcast_shapes(mlp_.shape, extra.shape) != mlp_.shape:
            raise RuntimeError("output with shape %s doesn't match the "
                               "broadcast shape" % (list(mlp_.shape),))
        mlp_ = mlp_ + extra
    K = mlp_.shape[1]
    mask = F.one_hot(mlp_.argmax(1), K).movedim(-1, 1)
    if straight_through_gradient:
        soft = F.softmax(mlp_, 1)
        mask = (mask - soft).detach() + soft
    return (mask * loc).sum(1)


# ----------------------------------------------------------------------------
# object_decoder.py
# ----------------------------------------------------------------------------
def capsule_layer(P, prefix, feature, cfg, noise_caps=None, noise_vote=None,
                  parent_transform=None, parent_presence=None):
    """CapsuleLayer.forward, object_decoder.py:120-236 (``parent_transform`` /
    ``parent_presence``: the hierarchical form, :184-187, :214-215).

    ``noise_caps`` (B,O,1) / ``noise_vote`` (B,O,V): the two U[0,1) draws of
    object_decoder.py:201 (made at :211 and :212) when noise_type='uniform'.
    """
    B = feature.shape[0]
    O, V = cfg['n_caps'], cfg['n_votes']
    n_hidden = len(cfg.get('hidden_sizes', (128,)))
    if cfg.get('caps_dropout_rate', 0.0) != 0.0:
        # reference deletes caps_exist (:152) and then reads it (:196)
        raise NameError("caps_exist")
    raw = torch.stack([mlp(P, f'{prefix}.mlps.{i}', feature[:, i], n_hidden + 1)
                       for i in range(O)], 1)               # :137-141
    caps_param = torch.cat([raw, torch.ones(B, O, 1)], -1)  # :144-151
    allp = torch.stack([mlp(P, f'{prefix}.caps_mlps.{i}', caps_param[:, i],
                            n_hidden + 1) for i in range(O)], 1)   # :154-158
    splits = [6 * V, 6, 1, V, V]
    shapes = [(V, 6), (1, 6), (1,), (V,), (V,)]
    parts = [t.view(B, O, *s)
             for t, s in zip(torch.split(allp, splits, -1), shapes)]
    cpr_dyn = parts[0]
    if not cfg.get('allow_deformations', True):
        cpr_dyn = torch.zeros_like(cpr_dyn)                 # :167-168
    reg = l2_loss(cpr_dyn) / B                              # :170
    sim = cfg.get('similarity_transform', True)
    cpr = geometric_transform(cpr_dyn + P[prefix + '.cpr_static'], sim,
                              nonlinear=True, as_matrix=True)      # :171
    cvr, logit_caps, logit_vote, scale = [
        t + P[f'{prefix}.caps_bias_list.{j}']
        for j, t in enumerate(parts[1:])]                   # :176-179
    if parent_transform is None:                            # :184-187
        cvr = geometric_transform(cvr, sim, nonlinear=True, as_matrix=True)
    else:
        cvr = parent_transform
    vote = torch.matmul(cvr.repeat(1, 1, V, 1, 1), cpr)     # :189-191

    ntype = cfg.get('noise_type', None)
    nscale = cfg.get('noise_scale', 0.)
    if ntype == 'uniform':                                  # :198-212
        logit_caps = logit_caps + (noise_caps - 0.5) * nscale
        logit_vote = logit_vote + (noise_vote - 0.5) * nscale
    elif ntype == 'logistic':
        raise NotImplementedError("oracle: LogisticNormal noise not restated")
    elif ntype:
        raise ValueError(f'Invalid noise type: {ntype}')
    presence_per_caps = parent_presence if parent_presence is not None \
        else torch.sigmoid(logit_caps)                      # :214-217
    vote_presence = presence_per_caps * torch.sigmoid(logit_vote)
    if cfg.get('learn_vote_scale', False):
        scale = F.softplus(scale + .5) + 1e-2               # :225
    else:
        scale = torch.ones_like(scale)
    return Bag(vote=vote, scale=scale, vote_presence=vote_presence,
               presence_logit_per_caps=logit_caps,
               presence_logit_per_vote=logit_vote, cpr_dynamic_reg_loss=reg)


def capsule_likelihood(vote, scale, vote_presence, dummy_vote, x,
                       presence=None):
    """CapsuleLikelihood.__call__, object_decoder.py:257-372."""
    B, M, Pd = x.shape
    vlp = normal_log_prob(x.unsqueeze(1), vote, scale.unsqueeze(-1)).sum(-1)
    log001 = float(np.log(0.01))
    dummy_lp = torch.zeros(B, 1, M) + log001                # :273-274
    vlp = torch.cat([vlp, dummy_lp], 1)                     # :277
    dummy_logit = torch.full((B, 1, M), log001)             # :281-282
    mixing_logit = torch.cat([log_safe(vote_presence), dummy_logit], 1)
    mixing_log_prob = mixing_logit - mixing_logit.logsumexp(1, keepdim=True)
    binary = (mixing_logit[:, :-1] > mixing_logit[:, -1:]).float()   # :289
    post = mixing_logit + vlp                               # :292
    per_point = post.logsumexp(1)                           # :296
    if presence is not None:
        per_point = per_point * presence.float()
    log_prob = per_point.sum(1).mean()                      # :302-306
    win = torch.argmax(post[:, :-1], 1)                     # :310-311
    bi = torch.arange(B).unsqueeze(1).expand(B, M)
    mi = torch.arange(M).unsqueeze(0).expand(B, M)
    winner = vote[bi, win, mi]                              # :324
    winner_presence = vote_presence[bi, win, mi]            # :328-329
    is_from_capsule = win // M                              # :334
    pprob = F.softmax(post, 1)                              # :338
    votes = torch.cat([vote, dummy_vote.repeat(B, 1, 1, 1)], 1)
    vps = torch.cat([vote_presence, torch.zeros(B, 1, M)], 1)
    soft_winner = (pprob.unsqueeze(-1) * votes).sum(1)      # :350
    soft_winner_presence = (pprob * vps).sum(1)             # :354
    return Bag(log_prob=log_prob, vote_presence_binary=binary, winner=winner,
               winner_presence=winner_presence, soft_winner=soft_winner,
               soft_winner_presence=soft_winner_presence,
               posterior_mixing_prob=pprob[:, :-1],
               mixing_log_prob=mixing_log_prob, mixing_logit=mixing_logit,
               is_from_capsule=is_from_capsule)


def capsule_object_decoder(P, prefix, obj_encoding, part_pose, part_presence,
                           cfg, noise_caps=None, noise_vote=None):
    """CapsuleObjectDecoder.forward, object_decoder.py:393-428."""
    B, O = obj_encoding.shape[:2]
    V = part_pose.shape[1]
    res = capsule_layer(P, prefix + '.capsule_layer', obj_encoding, cfg,
                        noise_caps, noise_vote)
    res.vote = res.vote[..., :-1, :].reshape(B, O, V, -1)   # :413
    res.caps_presence = res.vote_presence.max(-1)[0]        # :415
    res.update(capsule_likelihood(res.vote, res.scale, res.vote_presence,
                                  P[prefix + '.dummy_vote'], part_pose,
                                  part_presence))
    return res


def capsule_l2_loss(caps_presence, n_classes, within_example_constant=None):
    """object_decoder.py:433-452."""
    B, O = caps_presence.shape
    if within_example_constant is None:
        within_example_constant = float(O) / n_classes
    within = ((caps_presence.sum(1) - within_example_constant) ** 2).mean()
    between = ((caps_presence.sum(0) - float(B) / n_classes) ** 2).mean()
    return within, between


def capsule_entropy_loss(caps_presence, k=1):
    """object_decoder.py:456-471."""
    wp = normalize(caps_presence, 1)
    within = cross_entropy_safe(wp, wp * k)
    bp = normalize(caps_presence.sum(0), 0)
    between = cross_entropy_safe(bp, bp * k)
    return within, -between


def sparsity_loss(loss_type, caps_presence, n_classes=None,
                  within_example_constant=None):
    """object_decoder.py:482-493 (+ neg_capsule_kl :475-479)."""
    if loss_type == 'l2':
        return capsule_l2_loss(caps_presence, n_classes,
                               within_example_constant)
    if loss_type == 'entropy':
        return capsule_entropy_loss(caps_presence, k=1)
    if loss_type == 'kl':
        return capsule_entropy_loss(caps_presence,
                                    k=int(caps_presence.shape[-1]))
    raise ValueError(f"Invalid sparsity loss: {loss_type}")


# ----------------------------------------------------------------------------
# factory.py defaults + stacked_capsule_auto_encoder.py
# ----------------------------------------------------------------------------
def prepare_model_params(image_shape, n_classes, n_part_caps, n_obj_caps,
                         pcae_cnn_encoder_params=None,
                         pcae_encoder_params=None,
                         pcae_template_generator_params=None,
                         pcae_decoder_params=None,
                         ocae_encoder_set_transformer_params=None,
                         ocae_decoder_capsule_params=None, scae_params=None):
    """Defaults of factory.prepare_model_params, factory.py:10-149."""
    cnn = dict(input_shape=tuple(image_shape), out_channels=[128] * 4,
               kernel_sizes=[3] * 4, strides=[2, 2, 1, 1], activate_final=True)
    cnn.update(pcae_cnn_encoder_params or {})
    enc = dict(input_shape=tuple(image_shape), n_caps=n_part_caps, n_poses=6,
               n_special_features=16, similarity_transform=False)
    enc.update(pcae_encoder_params or {})
    tg = dict(n_templates=enc['n_caps'], n_channels=image_shape[0],
              template_size=(11, 11), template_nonlin='sigmoid',
              dim_feature=enc['n_special_features'], colorize_templates=True,
              color_nonlin='sigmoid')
    tg.update(pcae_template_generator_params or {})
    tg['template_size'] = tuple(tg['template_size'])
    dec = dict(n_templates=tg['n_templates'], template_size=tg['template_size'],
               output_size=tuple(image_shape[1:]), learn_output_scale=False,
               use_alpha_channel=True, background_value=True)
    dec.update(pcae_decoder_params or {})
    # factory.py:79-86 -- template_size[0] is used twice (reference quirk)
    dim_in = (enc['n_poses'] + tg['dim_feature'] + 1
              + tg['n_channels'] * tg['template_size'][0] * tg['template_size'][0])
    st = dict(n_layers=3, n_heads=1, dim_in=dim_in, dim_hidden=16, dim_out=256,
              n_outputs=n_obj_caps, layer_norm=True)
    st.update(ocae_encoder_set_transformer_params or {})
    caps = dict(n_caps=st['n_outputs'], dim_feature=st['dim_out'],
                n_votes=dec['n_templates'], dim_caps=32, hidden_sizes=(128,),
                caps_dropout_rate=0.0, learn_vote_scale=True,
                allow_deformations=True, noise_type='uniform', noise_scale=4.,
                similarity_transform=False)
    caps.update(ocae_decoder_capsule_params or {})
    scae = dict(n_classes=n_classes, vote_type='enc', presence_type='enc',
                stop_grad_caps_input=True, stop_grad_caps_target=True,
                caps_ll_weight=1., cpr_dynamic_reg_weight=10,
                prior_sparsity_loss_type='l2',
                prior_within_example_sparsity_weight=2.0,
                prior_between_example_sparsity_weight=0.35,
                posterior_sparsity_loss_type='entropy',
                posterior_within_example_sparsity_weight=0.7,
                posterior_between_example_sparsity_weight=0.2)
    scae.update(scae_params or {})
    return dict(image_shape=tuple(image_shape), n_classes=n_classes,
                n_part_caps=n_part_caps, n_obj_caps=n_obj_caps,
                pcae_cnn_encoder=cnn, pcae_encoder=enc,
                pcae_template_generator=tg, pcae_decoder=dec,
                ocae_encoder_set_transformer=st, ocae_decoder_capsule=caps,
                scae=scae)


# SCAE.__init__ defaults, stacked_capsule_auto_encoder.py:25-49
SCAE_CTOR_DEFAULTS = dict(
    n_classes=None, vote_type='soft', presence_type='enc',
    stop_grad_caps_input=True, stop_grad_caps_target=True,
    recon_mse_weight=0, part_caps_sparsity_weight=0.,
    cpr_dynamic_reg_weight=0., caps_ll_weight=0.,
    prior_sparsity_loss_type='l2', prior_within_example_sparsity_weight=0.,
    prior_between_example_sparsity_weight=0.,
    prior_within_example_constant=None,
    posterior_sparsity_loss_type='entropy',
    posterior_within_example_sparsity_weight=0.,
    posterior_between_example_sparsity_weight=0.,
    reconstruct_alternatives=True)


def _scae_cfg(cfg):
    s = dict(SCAE_CTOR_DEFAULTS)
    s.update(cfg['scae'])
    return s


def scae_forward(P, cfg, image, noise=(None, None, None), training=True):
    """SCAE.forward, stacked_capsule_auto_encoder.py:92-215.

    ``cfg``: output of ``prepare_model_params``.  ``noise``: the three U[0,1)
    draws (part-encoder (B,M) [training only], capsule (B,O,1), vote (B,O,V)).
    """
    s = _scae_cfg(cfg)
    B = image.shape[0]
    n_enc, n_caps, n_vote = noise
    enc = capsule_image_encoder(P, 'part_encoder', image,
                                cfg['pcae_cnn_encoder'], cfg['pcae_encoder'],
                                training, n_enc)            # :96
    templates = template_generator(P, 'template_generator', enc.feature, B,
                                   cfg['pcae_template_generator']).templates
    part_param = torch.cat([enc.pose, 1. - enc.presence.unsqueeze(-1)], -1)
    in_presence = enc.presence
    if s['stop_grad_caps_input']:                           # :111-113
        part_param = part_param.detach()
        in_presence = in_presence.detach()
    if enc.feature is not None:
        part_param = torch.cat([part_param, enc.feature], -1)   # :117
    in_templates = templates.detach() if s['stop_grad_caps_input'] \
        else templates
    x = torch.cat([part_param, in_templates.reshape(B, templates.shape[1], -1)],
                  -1)                                       # :124
    st = cfg['ocae_encoder_set_transformer']
    obj_enc = set_transformer(P, 'obj_encoder', x, in_presence, st['n_layers'],
                              st['n_heads'], st.get('layer_norm', False),
                              st.get('n_inducing_points'))  # :126
    t_pose, t_pres = enc.pose, enc.presence
    if s['stop_grad_caps_target']:                          # :131-133
        t_pose, t_pres = t_pose.detach(), t_pres.detach()
    res = capsule_object_decoder(P, 'obj_decoder', obj_enc, t_pose, t_pres,
                                 cfg['ocae_decoder_capsule'], n_caps, n_vote)
    res.part_presence = enc.presence
    vt, pt = s['vote_type'], s['presence_type']
    if vt not in ('enc', 'soft', 'hard'):
        raise ValueError(f'Invalid vote_type: {vt}')
    if pt not in ('enc', 'soft', 'hard'):
        raise ValueError(f'Invalid presence_type: {pt}')
    dec_pose = {'enc': enc.pose, 'soft': res.soft_winner,
                'hard': res.winner}[vt]                     # :141-148
    dec_pres = {'enc': enc.presence, 'soft': res.soft_winner_presence,
                'hard': res.winner_presence}[pt]            # :150-157
    dcfg = cfg['pcae_decoder']
    res.rec = image_decoder(P, 'part_decoder', templates, dec_pose, dec_pres,
                            None, dcfg)                     # :159-162
    if s['reconstruct_alternatives']:                       # :164-195
        with torch.no_grad():
            res.bottom_up_rec = image_decoder(P, 'part_decoder', templates,
                                              enc.pose, enc.presence, None,
                                              dcfg)
            res.top_down_rec = image_decoder(P, 'part_decoder', templates,
                                             res.winner, enc.presence, None,
                                             dcfg)
            O = res.vote.shape[1]
            td_pres = enc.presence.repeat_interleave(O, 0) \
                * res.vote_presence_binary.reshape(B * O, -1)
            res.top_down_per_caps_rec = image_decoder(
                P, 'part_decoder', templates.repeat_interleave(O, 0),
                res.vote.reshape(B * O, *res.vote.shape[2:]), td_pres, None,
                dcfg)
    res.templates = templates
    res.template_presence = enc.presence
    res.transformed_templates = res.rec.transformed_templates
    if s['n_classes'] is not None:                          # :203-213
        def prior_cls(t):
            return F.softmax(_linear(P, 'prior_classifier.0', t), -1)
        res.prior_cls_prob = prior_cls(res.caps_presence.detach())
        # reference quirk: posterior probs also go through prior_classifier
        res.posterior_cls_prob = prior_cls(
            res.posterior_mixing_prob.sum(-1).detach())
    return res


def scae_loss(cfg, res, target, label=None):
    """SCAE.loss, stacked_capsule_auto_encoder.py:217-287."""
    s = _scae_cfg(cfg)
    log = {}
    lp = gmm_log_prob(res.rec.transformed_templates, res.rec.scale,
                      res.rec.mixing_logits, target)        # :220
    rec_ll = lp.reshape(lp.shape[0], -1).sum(-1).mean()
    loss = -rec_ll
    log['rec_ll_loss'] = -rec_ll
    if s['recon_mse_weight'] > 0:                           # :226-230
        mode = gmm_mode(res.rec.transformed_templates, res.rec.scale,
                        res.rec.mixing_logits)
        mse = ((target - mode) ** 2).reshape(target.shape[0], -1).sum(-1).mean()
        loss = loss + s['recon_mse_weight'] * mse
        log['mse'] = mse
    if s['part_caps_sparsity_weight'] > 0:                  # :233-236
        l1 = res.part_presence.sum(-1).mean()
        loss = loss + s['part_caps_sparsity_weight'] * l1
        log['part_caps_loss'] = l1
    loss = loss + -s['caps_ll_weight'] * res.log_prob       # :239
    log['log_prob_loss'] = -res.log_prob
    prior_on = (s['prior_within_example_sparsity_weight'] > 0
                or s['prior_between_example_sparsity_weight'] > 0)
    if prior_on:                                            # :243-255
        w, b = sparsity_loss(s['prior_sparsity_loss_type'], res.caps_presence,
                             n_classes=s['n_classes'],
                             within_example_constant=s[
                                 'prior_within_example_constant'])
        loss = loss + (s['prior_within_example_sparsity_weight'] * w
                       + s['prior_between_example_sparsity_weight'] * b)
        log['prior_within_sparsity_loss'] = w
        log['prior_between_sparsity_loss'] = b
    if prior_on:       # :258-259 -- gated by the PRIOR weights (ref. quirk)
        n_points = res.posterior_mixing_prob.shape[-1]
        mass = res.posterior_mixing_prob.sum(-1)
        w, b = sparsity_loss(s['posterior_sparsity_loss_type'],
                             mass / n_points, n_classes=s['n_classes'])
        loss = loss + (s['posterior_within_example_sparsity_weight'] * w
                       + s['posterior_between_example_sparsity_weight'] * b)
        log['posterior_within_sparsity_loss'] = w
        log['posterior_between_sparsity_loss'] = b
    loss = loss + s['cpr_dynamic_reg_weight'] * res.cpr_dynamic_reg_loss
    log['cpr_dynamic_reg_loss'] = res.cpr_dynamic_reg_loss
    if label is not None:                                   # :278-285
        assert s['n_classes'] is not None
        # cross_entropy applied to probabilities (reference quirk)
        pxe = F.cross_entropy(res.prior_cls_prob, label)
        qxe = F.cross_entropy(res.posterior_cls_prob, label)
        loss = loss + pxe + qxe
        log['prior_cls_xe'] = pxe
        log['posterior_cls_xe'] = qxe
    return loss, log


def calculate_accuracy(res, label):
    """stacked_capsule_auto_encoder.py:289-297."""
    a = (res.prior_cls_prob.argmax(-1) == label).float().mean()
    b = (res.posterior_cls_prob.argmax(-1) == label).float().mean()
    return torch.max(a, b)


def train_step(P, cfg, image, label, noise):
    """forward + loss + backward on CPU; returns (loss, log, grads dict).
    ``P`` values must be leaf tensors with requires_grad=True."""
    for p in P.values():
        p.grad = None
    res = scae_forward(P, cfg, image, noise, training=True)
    loss, log = scae_loss(cfg, res, image, label)
    loss.backward()
    return loss, log, {k: p.grad for k, p in P.items()}
