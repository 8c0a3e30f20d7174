"""Stacked Capsule Auto-Encoder: wiring + composite loss over the HIP-backed
blocks (reference: torch_scae/stacked_capsule_auto_encoder.py)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import nn_utils, ops
from .general_utils import LazyAttrDict
from .object_decoder import sparsity_loss


class SCAE(nn.Module):
    """Stacked Capsule Auto-Encoder (constructor arguments and defaults as in
    stacked_capsule_auto_encoder.py:25-49)."""

    def __init__(self, part_encoder, template_generator, part_decoder,
                 obj_encoder, obj_decoder, n_classes=None, vote_type='soft',
                 presence_type='enc', stop_grad_caps_input=True,
                 stop_grad_caps_target=True, recon_mse_weight=0,
                 part_caps_sparsity_weight=0., cpr_dynamic_reg_weight=0.,
                 caps_ll_weight=0., prior_sparsity_loss_type='l2',
                 prior_within_example_sparsity_weight=0.,
                 prior_between_example_sparsity_weight=0.,
                 prior_within_example_constant=None,
                 posterior_sparsity_loss_type='entropy',
                 posterior_within_example_sparsity_weight=0.,
                 posterior_between_example_sparsity_weight=0.,
                 reconstruct_alternatives=True):
        super().__init__()
        self.part_encoder = part_encoder
        self.template_generator = template_generator
        self.part_decoder = part_decoder
        self.obj_encoder = obj_encoder
        self.obj_decoder = obj_decoder
        self.n_classes = n_classes
        self.vote_type = vote_type
        self.presence_type = presence_type
        self.stop_grad_caps_input = stop_grad_caps_input
        self.stop_grad_caps_target = stop_grad_caps_target

        if n_classes:
            n_obj = obj_decoder.n_obj_capsules
            self.prior_classifier = nn.Sequential(nn.Linear(n_obj, n_classes),
                                                  nn.Softmax(-1))
            self.posterior_classifier = nn.Sequential(
                nn.Linear(n_obj, n_classes), nn.Softmax(-1))
        else:
            self.prior_classifier = None
            self.posterior_classifier = None

        self.cpr_dynamic_reg_weight = cpr_dynamic_reg_weight
        self.caps_ll_weight = caps_ll_weight
        self.recon_mse_weight = recon_mse_weight
        self.prior_sparsity_loss_type = prior_sparsity_loss_type
        self.prior_within_example_sparsity_weight = \
            prior_within_example_sparsity_weight
        self.prior_between_example_sparsity_weight = \
            prior_between_example_sparsity_weight
        self.prior_within_example_constant = prior_within_example_constant
        self.posterior_sparsity_loss_type = posterior_sparsity_loss_type
        self.posterior_within_example_sparsity_weight = \
            posterior_within_example_sparsity_weight
        self.posterior_between_example_sparsity_weight = \
            posterior_between_example_sparsity_weight
        self.part_caps_sparsity_weight = part_caps_sparsity_weight
        self.reconstruct_alternatives = reconstruct_alternatives
        self.fuse_loss_tail = True     # set False for the op-by-op loss
        # set by train_step.TrainStep: hand the two decoders detached leaves
        # of their inputs (res._phase_cut) so that backward can run in two
        # parts with a gradient all-reduce between them
        self.split_backward = False

    # -- forward -------------------------------------------------------------
    def forward(self, image):
        """image (B,C,H,W) -> AttrDict (stacked_capsule_auto_encoder.py:92-215)."""
        if self.vote_type not in ('enc', 'soft', 'hard'):
            raise ValueError(f'Invalid vote_type: {self.vote_type}')
        if self.presence_type not in ('enc', 'soft', 'hard'):
            raise ValueError(f'Invalid presence_type: {self.presence_type}')
        batch_size = image.shape[0]
        if image.is_cuda and not nn_utils.replaying():
            # the three presence-noise draws of one forward (part_encoder.py:106,
            # object_decoder.py:201) as ONE device RNG launch, handed to the
            # modules through the replay hook
            with nn_utils.fixed_noise(self._draw_noise(image)):
                return self._forward(image)
        return self._forward(image)

    def _draw_noise(self, image):
        enc, layer = self.part_encoder, self.obj_decoder.capsule_layer
        B = image.shape[0]
        shapes = []
        # (keyed on each consumer's own mode: a frozen part encoder in eval()
        # inside a training SCAE draws nothing, like the reference)
        if enc.training and getattr(enc, "noise_scale", 0.) > 0. \
                and hasattr(enc, "n_caps"):
            shapes.append((B, enc.n_caps))
        if getattr(layer, "noise_type", None) == 'uniform' \
                and hasattr(layer, "n_votes"):
            shapes += [(B, layer.n_caps, 1), (B, layer.n_caps, layer.n_votes)]
        if len(shapes) < 2:
            return []                  # nothing to merge: modules draw themselves
        sizes = [int(torch.Size(s).numel()) for s in shapes]
        # device-resident generator: no host-driven seed / offset updates
        # when the step is replayed from a HIP graph
        flat = ops.uniform(sum(sizes), image) if image.dtype == torch.float32 \
            else torch.rand(sum(sizes), device=image.device, dtype=image.dtype)
        return [c.view(s) for c, s in zip(flat.split(sizes), shapes)]

    def _forward(self, image):
        batch_size = image.shape[0]
        # (inside a training step the template generator's colour kernel rides
        # behind the part encoder's head, which makes the features it reads)
        self._offer_template_colours(image)
        parts = self.part_encoder(image)
        templates = self.template_generator(feature=parts.feature,
                                            batch_size=batch_size).templates
        ops.withdraw_colored_templates_offer()

        # object encoder input: [pose, 1 - presence | feature | templates];
        # pose / presence / templates detached, the feature skip-connection
        # is not (:105-124)
        in_pose, in_presence, in_templates = \
            parts.pose, parts.presence, templates
        if self.stop_grad_caps_input:
            in_pose, in_presence, in_templates = \
                in_pose.detach(), in_presence.detach(), in_templates.detach()
        # 1 - presence: already emitted (detached) by the fused capsule head
        absence = parts._absence if self.stop_grad_caps_input and \
            "_absence" in parts else 1. - in_presence.unsqueeze(-1)
        segments = [in_pose, absence]
        if parts.feature is not None:
            # (a second tensor with the same values when the fused encoder
            # provides one: its gradient is summed inside the head kernel)
            segments.append(parts.get("_feature_twin", parts.feature)
                            if parts.get("_feature_twin") is not None
                            else parts.feature)
        segments.append(in_templates.flatten(2))
        if hasattr(self.obj_encoder, "forward_segments"):
            # fused trunk: the concat is never materialised.  Inside a
            # training step (ops.step_fusion) the part decoder's likelihood of
            # the step's input -- which needs nothing the object encoder
            # produces -- rides in the trunk's launch
            self._offer_likelihood_rider(image, templates, parts)
            obj_encoding = self.obj_encoder.forward_segments(segments,
                                                             in_presence)
            ops.withdraw_log_prob_rider()
        else:
            obj_encoding = self.obj_encoder(torch.cat(segments, -1),
                                            in_presence)

        if obj_encoding.dtype != image.dtype:
            # bf16 autocast covers the object encoder only (SURVEY.md 8d cfg-3:
            # "bf16 autocast for attention/linears, fp32 accumulators for the
            # mixture"); everything behind it stays fp32
            obj_encoding = obj_encoding.to(image.dtype)
        target_pose, target_presence = parts.pose, parts.presence
        if self.stop_grad_caps_target:
            target_pose = target_pose.detach()
            target_presence = target_presence.detach()
        live_pose, live_presence = parts.pose, parts.presence
        phase_cut = None
        if self.split_backward and self.stop_grad_caps_target and \
                torch.is_grad_enabled() and \
                self.vote_type == self.presence_type == 'enc':
            # everything behind this point sees leaves: loss.backward() stops
            # at them, backward(srcs, [l.grad for l in leaves]) does the rest
            srcs = [obj_encoding, templates, live_pose, live_presence]
            leaves = [t.detach().requires_grad_(t.requires_grad) for t in srcs]
            obj_encoding, templates, live_pose, live_presence = leaves
            phase_cut = (srcs, leaves)
        # the forward's scalar outputs ride in its last kernel when there is one
        fused_probs = self._fused_class_probs(obj_encoding)
        res = self.obj_decoder(obj_encoding, target_pose, target_presence,
                               **(dict(_defer_sums=True) if fused_probs else {}))
        pending = res.pop("_pending_sums", None)
        res.part_presence = live_presence
        if phase_cut is not None:
            res._phase_cut = phase_cut

        dec_pose = {'enc': live_pose, 'soft': res.soft_winner,
                    'hard': res.winner}[self.vote_type]
        dec_presence = {'enc': live_presence,
                        'soft': res.soft_winner_presence,
                        'hard': res.winner_presence}[self.presence_type]
        res.rec = self.part_decoder(templates=templates, pose=dec_pose,
                                    presence=dec_presence)

        if self.reconstruct_alternatives:      # :164-195
            with torch.no_grad():
                res.bottom_up_rec = self.part_decoder(
                    templates=templates, pose=parts.pose,
                    presence=parts.presence)
                res.top_down_rec = self.part_decoder(
                    templates=templates, pose=res.winner,
                    presence=parts.presence)
                n_obj = res.vote.shape[1]
                td_presence = parts.presence.repeat_interleave(n_obj, dim=0) \
                    * res.vote_presence_binary.flatten(0, 1)
                # B*O virtual images, each group of O sharing its image's
                # templates (the kernels index them, nothing is repeated)
                res.top_down_per_caps_rec = self.part_decoder(
                    templates=templates, pose=res.vote.flatten(0, 1),
                    presence=td_presence)

        res.templates = templates
        res.template_presence = live_presence
        if isinstance(res.rec, LazyAttrDict):
            # part_decoder.lazy_render: the alias must not force the render
            lazy = LazyAttrDict(res)
            lazy["_lazy"] = dict(transformed_templates=lambda d: dict.__setitem__(
                d, "transformed_templates", d["rec"]["transformed_templates"]))
            res = lazy
        else:
            res.transformed_templates = res.rec.transformed_templates

        if self.n_classes is not None:
            assert self.prior_classifier is not None
            assert self.posterior_classifier is not None
            linear = self.prior_classifier[0]
            if fused_probs:
                # both heads (and the capsule-mass reduction) in one launch
                res.prior_cls_prob, res.posterior_cls_prob = ops.class_probs(
                    res.caps_presence, res._posterior_full, linear.weight,
                    linear.bias, extra_sums=pending)
                return res
            res.prior_cls_prob = self.prior_classifier(
                res.caps_presence.detach())
            # as in the reference (:211) the posterior probabilities also go
            # through prior_classifier
            res.posterior_cls_prob = self.prior_classifier(
                res.posterior_mixing_prob.sum(-1).detach())
        return res

    def _offer_template_colours(self, image):
        tg = self.template_generator
        mlp = getattr(tg, "templates_color_mlp", None)
        if not image.is_cuda or image.dtype != torch.float32 or \
                not getattr(tg, "colorize_templates", False) or mlp is None \
                or len(mlp) != 4 or tg._nonlin_names[0] not in ops._NONLIN_CODE \
                or tg._nonlin_names[1] not in ops._NONLIN_CODE:
            return
        ops.offer_colored_templates(tg.template_logits, mlp[0].weight,
                                    mlp[0].bias, mlp[2].weight, mlp[2].bias,
                                    *tg._nonlin_names)

    def _offer_likelihood_rider(self, image, templates, parts):
        """stacked_capsule_auto_encoder.py:146-162 + :220 ahead of :126-130:
        with 'enc' votes and presences the reconstruction's inputs are the
        part encoder's own outputs."""
        target = ops.fusion_target()
        dec = self.part_decoder
        if target is None or target.data_ptr() != image.data_ptr() or \
                target.shape != image.shape or not torch.is_grad_enabled() \
                or self.vote_type != 'enc' or self.presence_type != 'enc' \
                or not self.fuse_loss_tail or self.recon_mse_weight > 0 \
                or self.part_caps_sparsity_weight > 0 \
                or not hasattr(dec, "decoder_inputs") \
                or not getattr(dec, "use_alpha_channel", False) \
                or templates.shape[0] != parts.pose.shape[0]:
            return
        ops.offer_log_prob_rider(
            dec.decoder_inputs(templates, parts.pose, parts.presence), image)

    def _fused_class_probs(self, obj_encoding):
        from .object_decoder import CapsuleObjectDecoder
        return (self.n_classes is not None and obj_encoding.is_cuda
                and type(self.obj_decoder) is CapsuleObjectDecoder
                and self.prior_classifier is not None
                and len(self.prior_classifier) == 2
                and ops.class_probs_supported(self.obj_decoder.n_obj_capsules,
                                              self.n_classes))

    # -- loss ----------------------------------------------------------------
    def loss(self, res, reconstruction_target, label=None):
        """-> (loss, log dict)   (stacked_capsule_auto_encoder.py:217-287)."""
        log = dict()
        sparsity_on = (self.prior_within_example_sparsity_weight > 0
                       or self.prior_between_example_sparsity_weight > 0)
        fused_tail = self._fused_tail_ok(res, label)
        if not fused_tail:
            # the op-by-op terms read the class probabilities and the scalar
            # sums whose launch a fused step parks for the tail kernel
            ops.flush_pending_forward()

        def tail(rec_sums=None, reg=None):
            # one kernel for the capsule-likelihood, sparsity and
            # classification terms (and one for their backward)
            return ops.loss_tail_scalar(
                res._log_prob_per_point, res._posterior_full,
                res.caps_presence,
                self.prior_classifier[0].weight if label is not None else None,
                self.prior_classifier[0].bias if label is not None else None,
                label, self.n_classes, self.prior_sparsity_loss_type,
                self.posterior_sparsity_loss_type, sparsity_on,
                [self.caps_ll_weight,
                 self.prior_within_example_sparsity_weight,
                 self.prior_between_example_sparsity_weight,
                 self.posterior_within_example_sparsity_weight,
                 self.posterior_between_example_sparsity_weight],
                self.prior_within_example_constant, rec_sums=rec_sums, reg=reg,
                w_reg=self.cpr_dynamic_reg_weight)

        def tail_log(t):
            log.update(log_prob_loss=t[10])
            if sparsity_on:
                log.update(prior_within_sparsity_loss=t[2],
                           prior_between_sparsity_loss=t[3],
                           posterior_within_sparsity_loss=t[4],
                           posterior_between_sparsity_loss=t[5])
            log.update(cpr_dynamic_reg_loss=res.cpr_dynamic_reg_loss)
            if label is not None:
                log.update(prior_cls_xe=t[6], posterior_cls_xe=t[7])

        rec_sums = None
        if fused_tail and self.recon_mse_weight <= 0 and \
                self.part_caps_sparsity_weight <= 0:
            rec_sums = res.rec.pdf.log_prob_tile_sums(reconstruction_target)
        if rec_sums is not None:
            # the whole training scalar from the tail kernel: K1 hands it
            # per-tile sums of the reconstruction log-likelihood
            loss, t = tail(rec_sums, res.cpr_dynamic_reg_loss.reshape(1))
            log.update(rec_ll_loss=t[9])
            tail_log(t)
            return loss, log

        rec_ll_per_pixel = res.rec.pdf.log_prob(reconstruction_target)
        rec_ll = rec_ll_per_pixel.flatten(1).sum(-1).mean()
        loss = -rec_ll
        log.update(rec_ll_loss=-rec_ll)

        if self.recon_mse_weight > 0:
            mse = ((reconstruction_target - res.rec.pdf.mode()) ** 2) \
                .flatten(1).sum(-1).mean()
            loss = loss + self.recon_mse_weight * mse
            log.update(mse=mse)

        if self.part_caps_sparsity_weight > 0:
            part_caps_l1 = res.part_presence.sum(-1).mean()
            loss = loss + self.part_caps_sparsity_weight * part_caps_l1
            log.update(part_caps_loss=part_caps_l1)

        if fused_tail:
            tail_loss, t = tail()
            loss = loss + tail_loss
            loss = loss + self.cpr_dynamic_reg_weight * res.cpr_dynamic_reg_loss
            tail_log(t)
            return loss, log

        loss = loss - self.caps_ll_weight * res.log_prob
        log.update(log_prob_loss=-res.log_prob)

        # both sparsity blocks are gated by the PRIOR weights (:243, :258)
        if sparsity_on:
            within, between = sparsity_loss(
                self.prior_sparsity_loss_type, res.caps_presence,
                n_classes=self.n_classes,
                within_example_constant=self.prior_within_example_constant)
            loss = loss + (self.prior_within_example_sparsity_weight * within
                           + self.prior_between_example_sparsity_weight
                           * between)
            log.update(prior_within_sparsity_loss=within,
                       prior_between_sparsity_loss=between)

            n_points = res.posterior_mixing_prob.shape[-1]
            mass_explained_by_capsule = res.posterior_mixing_prob.sum(-1)
            within, between = sparsity_loss(
                self.posterior_sparsity_loss_type,
                mass_explained_by_capsule / n_points, n_classes=self.n_classes)
            loss = loss + (
                self.posterior_within_example_sparsity_weight * within
                + self.posterior_between_example_sparsity_weight * between)
            log.update(posterior_within_sparsity_loss=within,
                       posterior_between_sparsity_loss=between)

        loss = loss + self.cpr_dynamic_reg_weight * res.cpr_dynamic_reg_loss
        log.update(cpr_dynamic_reg_loss=res.cpr_dynamic_reg_loss)

        if label is not None:
            assert self.n_classes is not None
            # cross_entropy over probabilities, as the reference does (:281)
            prior_cls_xe = F.cross_entropy(res.prior_cls_prob, target=label)
            posterior_cls_xe = F.cross_entropy(res.posterior_cls_prob,
                                               target=label)
            loss = loss + prior_cls_xe + posterior_cls_xe
            log.update(prior_cls_xe=prior_cls_xe,
                       posterior_cls_xe=posterior_cls_xe)
        return loss, log

    def _fused_tail_ok(self, res, label):
        if not self.fuse_loss_tail or "_posterior_full" not in res:
            return False
        if label is not None and self.n_classes is None:
            return False
        if self.prior_sparsity_loss_type not in ("l2", "entropy", "kl") or \
                self.posterior_sparsity_loss_type not in ("l2", "entropy", "kl"):
            return False
        cp = res.caps_presence
        if not cp.is_cuda or (self.n_classes is None
                              and "l2" in (self.prior_sparsity_loss_type,
                                           self.posterior_sparsity_loss_type)):
            return False
        return ops.loss_tail_supported(cp.shape[0], cp.shape[1],
                                       self.n_classes)

    def calculate_accuracy(self, res, label: torch.Tensor):
        prior_acc = (res.prior_cls_prob.argmax(-1) == label).float().mean()
        posterior_acc = (res.posterior_cls_prob.argmax(-1)
                         == label).float().mean()
        return torch.max(prior_acc, posterior_acc)
