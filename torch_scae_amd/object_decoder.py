"""Object-capsule decoder on the HIP kernels K3 (votes) and K4 (likelihood)
(reference: torch_scae/object_decoder.py)."""
import torch
import torch.nn as nn

from . import math_ops, ops
from .general_utils import AttrDict, prod
from .nn_ext import GroupedMLP
from .nn_utils import rand_like


class CapsuleLayer(nn.Module):
    """Object capsules -> per-part votes (object_decoder.py:28-240).

    The reference's two Python loops over n_caps tiny MLPs are batched GEMMs
    here (``GroupedMLP``; checkpoints keep the per-capsule keys), and the
    split / geometric_transform / 3x3 product / presence / scale chain is one
    kernel.
    """

    n_transform_params = 6

    def __init__(self, n_caps, dim_feature, n_votes, dim_caps,
                 hidden_sizes=(128,), caps_dropout_rate=0.0,
                 learn_vote_scale=False, allow_deformations=True,
                 noise_type=None, noise_scale=0., similarity_transform=True):
        super().__init__()
        self.n_caps = n_caps
        self.dim_feature = dim_feature
        self.hidden_sizes = list(hidden_sizes)
        self.dim_caps = dim_caps
        self.caps_dropout_rate = caps_dropout_rate
        self.n_votes = n_votes
        self.learn_vote_scale = learn_vote_scale
        self.allow_deformations = allow_deformations
        self.noise_type = noise_type
        self.noise_scale = noise_scale
        self.similarity_transform = similarity_transform

        self.mlps = GroupedMLP(
            n_caps, [dim_feature] + self.hidden_sizes + [dim_caps])
        self.output_shapes = (
            [n_votes, self.n_transform_params],   # OPR-dynamic
            [1, self.n_transform_params],         # OVR
            [1],                                  # per-object presence
            [n_votes],                            # per-vote presence
            [n_votes],                            # per-vote scale
        )
        self.splits = [prod(s) for s in self.output_shapes]
        self.n_outputs = sum(self.splits)
        # bias-free output MLPs: the static part of the OPR is cpr_static
        self.caps_mlps = GroupedMLP(
            n_caps, [dim_caps + 1] + self.hidden_sizes + [self.n_outputs],
            bias=False, ones_input=True)
        self.caps_bias_list = nn.ParameterList([
            nn.Parameter(torch.zeros(1, n_caps, *shape))
            for shape in self.output_shapes[1:]])
        self.cpr_static = nn.Parameter(
            torch.zeros(1, n_caps, n_votes, self.n_transform_params))

    def _check_config(self):
        if self.caps_dropout_rate != 0.0:
            # the reference deletes `caps_exist` (:152) before reading it (:196)
            raise NameError("free variable 'caps_exist' referenced before "
                            "assignment in enclosing scope")
        if self.noise_type not in (None, False, '', 'uniform', 'logistic'):
            raise ValueError(f'Invalid noise type: {self.noise_type}')
        if self.noise_type == 'logistic':
            # the reference cannot run this either: LogisticNormal(0, s)
            # .sample(shape) carries a trailing event dimension of 2 that does
            # not broadcast against the logits (object_decoder.py:201-207)
            raise RuntimeError(
                "The size of tensor a must match the size of tensor b: "
                "LogisticNormal noise samples have a trailing dimension of 2 "
                "(noise_type='logistic' fails the same way in the reference)")

    def _votes(self, feature, defer_reg=False):
        self._check_config()
        B = feature.shape[0]
        # on the HIP path the three stages hand each other gradients w.r.t.
        # PRE-activations (the ReLU gates ride in the GEMM / K3 epilogues)
        fused = feature.is_cuda
        chain = [(w, b, False) for w, b in zip(self.mlps.weights,
                                               self.mlps.biases)] + \
            [(w, None, j == 0) for j, w in enumerate(self.caps_mlps.weights)]
        chained = fused and ops.mlp_chain_supported(feature, chain)
        noise_caps = noise_vote = None
        if self.noise_type == 'uniform':
            proto = feature.new_empty(B, self.n_caps, 1)
            noise_caps = rand_like(proto)
            noise_vote = rand_like(feature.new_empty(B, self.n_caps,
                                                     self.n_votes))
        if chained:
            # both MLPs -- all four layers -- and the vote kernel in one launch
            # (K7b + K3)
            return ops.chain_votes(
                feature, chain, self.cpr_static, *self.caps_bias_list,
                noise_caps=noise_caps, noise_vote=noise_vote,
                noise_scale=self.noise_scale,
                similarity=self.similarity_transform,
                learn_vote_scale=self.learn_vote_scale,
                allow_deformations=self.allow_deformations,
                defer_reg=defer_reg)
        raw_caps_param = self.mlps(feature, grad_pregated=fused)  # (B, O, D)
        all_param = self.caps_mlps(raw_caps_param, grad_pregated=fused,
                                   x_is_relu=fused, pad_out=fused)  # (B, O, A)
        return ops.capsule_votes(
            all_param, self.cpr_static, *self.caps_bias_list,
            noise_caps=noise_caps, noise_vote=noise_vote,
            noise_scale=self.noise_scale,
            similarity=self.similarity_transform,
            learn_vote_scale=self.learn_vote_scale,
            allow_deformations=self.allow_deformations,
            param_is_relu=fused, defer_reg=defer_reg)

    def forward(self, feature, parent_transform=None, parent_presence=None):
        """feature [B, O, F] -> AttrDict(vote (B,O,V,3,3), scale,
        vote_presence, presence_logit_per_caps, presence_logit_per_vote,
        cpr_dynamic_reg_loss)."""
        if parent_transform is not None or parent_presence is not None:
            return self._forward_hierarchical(feature, parent_transform,
                                              parent_presence)
        vote6, scale, vote_presence, logit_caps, logit_vote, reg, _, _ = \
            self._votes(feature)
        last_row = vote6.new_tensor([0., 0., 1.]).expand(*vote6.shape[:-1], 3)
        vote = torch.cat([vote6, last_row], -1).view(*vote6.shape[:-1], 3, 3)
        return AttrDict(vote=vote, scale=scale, vote_presence=vote_presence,
                        presence_logit_per_caps=logit_caps,
                        presence_logit_per_vote=logit_vote,
                        cpr_dynamic_reg_loss=reg)


    def _forward_hierarchical(self, feature, parent_transform,
                              parent_presence):
        """object_decoder.py:184-187, :214-215: a parent capsule's transform
        (B,O,1,3,3) stands in for the capsule's own OVR and / or its presence
        (B,O,1) for the capsule's own.  Not on SCAE's path: the per-capsule
        MLPs on K7, the pose transform on K5, the 3x3 products on
        ``scae_mat3_mul_*`` (geometric_transform.hip)."""
        self._check_config()
        B = feature.shape[0]
        raw_caps_param = self.mlps(feature)                       # (B, O, D)
        all_param = self.caps_mlps(raw_caps_param)                # (B, O, A)
        parts = [t.reshape(B, self.n_caps, *shape) for t, shape in
                 zip(torch.split(all_param, self.splits, -1),
                     self.output_shapes)]
        cpr_dynamic = parts[0]
        if not self.allow_deformations:
            cpr_dynamic = torch.zeros_like(cpr_dynamic)
        reg = math_ops.l2_loss(cpr_dynamic) / B
        cpr = ops.geometric_transform(cpr_dynamic + self.cpr_static,
                                      self.similarity_transform,
                                      nonlinear=True, as_matrix=True)
        cvr, logit_caps, logit_vote, scale = [
            t + bias for t, bias in zip(parts[1:], self.caps_bias_list)]
        if parent_transform is None:
            cvr = ops.geometric_transform(cvr, self.similarity_transform,
                                          nonlinear=True, as_matrix=True)
        else:
            cvr = parent_transform
        # the 3 x 3 object -> part pose products (object_decoder.py:189-191) on
        # the HIP kernel, no ``repeat``: one parent matrix per capsule, or --
        # a parent_transform given per vote -- one per (capsule, vote)
        if cvr.shape[2] == 1:
            vote = ops.mat3_mul(cvr, cpr)
        elif cvr.shape[2] == self.n_votes:
            vote = ops.mat3_mul(cvr.reshape(-1, 1, 3, 3),
                                cpr.reshape(-1, 1, 3, 3)).view(cpr.shape)
        else:
            raise ValueError(
                f"parent transform has {cvr.shape[2]} matrices per capsule, "
                f"expected 1 or n_votes = {self.n_votes}")
        if self.noise_type == 'uniform':
            logit_caps = logit_caps + (rand_like(logit_caps) - 0.5) \
                * self.noise_scale
            logit_vote = logit_vote + (rand_like(logit_vote) - 0.5) \
                * self.noise_scale
        presence_per_caps = parent_presence if parent_presence is not None \
            else torch.sigmoid(logit_caps)
        vote_presence = presence_per_caps * torch.sigmoid(logit_vote)
        if self.learn_vote_scale:
            scale = torch.nn.functional.softplus(scale + .5) + 1e-2
        else:
            scale = torch.ones_like(scale)
        return AttrDict(vote=vote, scale=scale, vote_presence=vote_presence,
                        presence_logit_per_caps=logit_caps,
                        presence_logit_per_vote=logit_vote,
                        cpr_dynamic_reg_loss=reg)


class CapsuleLikelihood:
    """Capsule voting mechanism (object_decoder.py:243-372)."""

    def __init__(self, vote, scale, vote_presence, dummy_vote):
        self.n_caps = vote.shape[1]
        self.vote = vote                      # (B, O, M, P)
        self.scale = scale                    # (B, O, M)
        self.vote_presence = vote_presence    # (B, O, M)
        self.dummy_vote = dummy_vote          # (1, 1, M, P)

    def __call__(self, x, presence=None, _extra_sums=(), _defer_sums=False):
        """x (B, M, P), presence (B, M).  ``_extra_sums``: further
        (src, scale, dst) scalar sums to ride in this call's one
        scalar-sum launch; ``_defer_sums``: do not launch it -- the jobs are
        returned as ``_pending_sums`` for a later kernel to carry."""
        batch_size, n_input_points, dim_in = x.shape
        if dim_in != 6 or self.vote.shape[-1] != 6:
            raise ValueError("the capsule likelihood kernel is built for "
                             "6-dim poses")
        (lpp, binary, winner, winner_presence, _widx, is_from_capsule,
         soft_winner, soft_winner_presence, posterior, mixing_log_prob,
         mixing_logit, log_prob) = ops.capsule_likelihood(
            self.vote, self.scale, self.vote_presence, self.dummy_vote, x,
            presence.float() if presence is not None else None,
            defer_sum=True)
        sums = [(lpp.detach(), 1.0 / batch_size, log_prob.detach()),
                *_extra_sums]
        if not _defer_sums:
            ops.scaled_sums(sums)
        return AttrDict(
            **(dict(_pending_sums=sums) if _defer_sums else {}),
            _log_prob_per_point=lpp,           # inputs of the fused loss tail
            _posterior_full=posterior,
            log_prob=log_prob,                 # = lpp.sum() / batch_size
            vote_presence_binary=binary,
            winner=winner,
            winner_presence=winner_presence,
            soft_winner=soft_winner,
            soft_winner_presence=soft_winner_presence,
            posterior_mixing_prob=posterior[:, :-1],
            mixing_log_prob=mixing_log_prob,
            mixing_logit=mixing_logit,
            is_from_capsule=is_from_capsule,
        )


class CapsuleObjectDecoder(nn.Module):
    def __init__(self, capsule_layer: CapsuleLayer):
        super().__init__()
        self.capsule_layer = capsule_layer
        self.dummy_vote = nn.Parameter(torch.zeros(
            1, 1, capsule_layer.n_votes, capsule_layer.n_transform_params))

    @property
    def n_obj_capsules(self):
        return self.capsule_layer.n_caps

    def forward(self, obj_encoding: torch.Tensor, part_pose: torch.Tensor,
                part_presence: torch.Tensor = None, _defer_sums=False):
        """obj_encoding [B, O, D], part_pose [B, M, P], part_presence [B, M]
        or None -> AttrDict (object_decoder.py:393-428).  ``_defer_sums``
        (SCAE.forward only): the two scalar outputs are filled by a later
        launch the caller makes (``res._pending_sums``)."""
        vote, scale, vote_presence, logit_caps, logit_vote, reg, \
            caps_presence, reg_partial = self.capsule_layer._votes(
                obj_encoding, defer_reg=True)
        res = AttrDict(vote=vote,             # (B, O, V, 6): rows 0..1 only
                       scale=scale, vote_presence=vote_presence,
                       presence_logit_per_caps=logit_caps,
                       presence_logit_per_vote=logit_vote,
                       cpr_dynamic_reg_loss=reg)
        res.caps_presence = caps_presence     # = vote_presence.max(-1)[0]
        likelihood = CapsuleLikelihood(vote=res.vote, scale=res.scale,
                                       vote_presence=res.vote_presence,
                                       dummy_vote=self.dummy_vote)
        # the two scalar outputs (reg loss, log_prob) in one launch
        res.update(likelihood(part_pose, presence=part_presence, _extra_sums=[
            (reg_partial, 0.5 / obj_encoding.shape[0], reg.detach())],
            _defer_sums=_defer_sums))
        return res


# Host-side (op-by-op) forms of the three sparsity penalties; inside a fused
# step the loss-tail kernel (csrc/loss_tail.hip) computes the same terms.  Each
# returns (within-example term, between-example term).
def capsule_l2_loss(caps_presence, n_classes: int,
                    within_example_constant=None, **_ignored):
    """Squared distance of the activation mass per example (row sums) and per
    capsule (column sums) from an even spread over ``n_classes`` classes
    (object_decoder.py:433-452)."""
    n_examples, n_caps = caps_presence.shape
    row_target = n_caps / n_classes if within_example_constant is None \
        else within_example_constant
    col_target = n_examples / n_classes
    row_mass, col_mass = caps_presence.sum(1), caps_presence.sum(0)
    return ((row_mass - row_target).square().mean(),
            (col_mass - col_target).square().mean())


def _scaled_self_entropy(mass, dim, k):
    """cross_entropy_safe(p, k p) of ``mass`` normalised along ``dim``."""
    p = math_ops.normalize(mass, dim)
    return math_ops.cross_entropy_safe(p, k * p)


def capsule_entropy_loss(caps_presence, k=1, **_ignored):
    """Entropy of the activations within an example (to be lowered) and of the
    batch totals between capsules (to be raised, hence the sign)
    (object_decoder.py:456-471)."""
    within = _scaled_self_entropy(caps_presence, 1, k)
    between = _scaled_self_entropy(caps_presence.sum(0), 0, k)
    return within, -between


def neg_capsule_kl(caps_presence, **_ignored):
    """KL to the uniform distribution over capsules = the entropy form with
    k = number of capsules (object_decoder.py:475-479)."""
    return capsule_entropy_loss(caps_presence, k=caps_presence.shape[-1])


def sparsity_loss(loss_type, *args, **kwargs):
    """object_decoder.py:482-493."""
    table = dict(l2=capsule_l2_loss, entropy=capsule_entropy_loss,
                 kl=neg_capsule_kl)
    if loss_type not in table:
        raise ValueError(f"Invalid sparsity loss: {loss_type}")
    return table[loss_type](*args, **kwargs)
