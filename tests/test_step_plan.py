"""The step plan (torch_scae_amd/step_plan.py): the table of merged launches and
the per-step holder of parked work.

CPU part: the plan's own logic on stand-in launches.  GPU part: the cases the
round-3 review found unguarded -- a decoder fed by the capsule likelihood's
(soft) winners (SCAE's constructor default, vote_type='soft'), an encoder
without the fused part-encoder node, two steps interleaving in one process --
each held to the oracle / to the un-fused step.
"""
import re
import threading

import numpy as np
import pytest
import torch

from torch_scae_amd import step_plan
from torch_scae_amd.step_plan import RIDES, StepPlan


class _Fake:
    def __init__(self, log, name):
        self.log, self.name = log, name

    def launch_alone(self):
        self.log.append(self.name)


def test_rides_table_is_consistent():
    assert [k for k, r in RIDES.items() if r.scope != "offer"][:2] == \
        ["class_probs", "combine"]
    # the folding products' backward reads the reduction: table order is
    # launch order
    kinds = list(RIDES)
    assert kinds.index("reduce") < kinds.index("fold_bwd")
    for kind, ride in RIDES.items():
        assert ride.scope in ("fusing", "deferring", "offer"), kind
        assert ride.carriers and ride.abi.startswith("scae_"), kind
        for node in ride.carriers + (() if ride.readers == step_plan.ANY
                                     else tuple(ride.readers)):
            assert re.fullmatch(r"_[A-Za-z0-9]+\.(forward|backward)", node), node


def test_ops_has_no_module_level_parked_state():
    """Round-3 review: nine ``_PENDING_*`` module globals in ops.py.  The plan
    is the only holder now."""
    import inspect

    from torch_scae_amd import ops
    src = inspect.getsource(ops)
    assert "global _PENDING" not in src and "_PENDING_" not in src
    assert "global _FUSION_TARGET" not in src and "global _DEFERRED" not in src
    # every node method the table names exists in ops
    for ride in RIDES.values():
        for node in ride.carriers:
            cls, meth = node.split(".")
            assert hasattr(getattr(ops, cls), meth), node


def test_park_take_flush_and_enter():
    log = []
    plan = StepPlan()
    plan.park("fold_bwd", _Fake(log, "fold"))
    plan.park("reduce", _Fake(log, "reduce"))
    plan.park("k1_bwd", _Fake(log, "k1"))
    # the carrier does not flush what it carries; any other node launches a
    # parked K1 before it starts (readers = ANY)
    plan.enter("_CapsuleLikelihood.backward")
    assert log == []
    plan.enter("_ColoredTemplates.backward")
    assert log == ["k1"] and not plan.holds("k1_bwd")
    # parameter-gradient launches wait for their carrier or the scope's end
    plan.enter("_SetEncoder.backward")
    assert log == ["k1"]
    assert plan.take("wgrads") is None
    plan.flush_scope("deferring")
    assert log == ["k1", "reduce", "fold"]       # table order, not park order
    assert plan.parked == {}
    # a second launch of a kind pushes the first one out
    plan.park("tc_bwd", _Fake(log, "tc1"))
    plan.park("tc_bwd", _Fake(log, "tc2"))
    assert log[-1] == "tc1"
    plan.enter("_CapsuleHead.backward")          # a reader of tc_bwd
    assert log[-1] == "tc2"


def test_scopes():
    log = []
    plan = StepPlan()
    assert not plan.fused and not plan.parking
    with plan.fusing("image"):
        assert plan.fused and not plan.parking
        plan.park("class_probs", _Fake(log, "cp"))
        with plan.deferring():
            assert plan.parking
            with plan.deferring():              # nested: the outermost flushes
                plan.park("wgrads", _Fake(log, "wg"))
            assert log == []
        assert log == ["wg"] and plan.deferred is None
    assert log == ["wg", "cp"] and not plan.fused
    # a block that raised launches nothing and leaves nothing behind
    with pytest.raises(RuntimeError):
        with plan.fusing("image"), plan.deferring():
            plan.park("wgrads", _Fake(log, "lost"))
            raise RuntimeError
    assert log == ["wg", "cp"] and plan.deferred is None
    with plan.fusing("image"):
        assert plan.parked == {}


def test_plans_do_not_share_state_and_follow_the_thread_of_the_node():
    a, b = StepPlan("a"), StepPlan("b")
    assert step_plan.current() is step_plan.ambient
    with a.active():
        assert step_plan.current() is a
        with b.active():
            assert step_plan.current() is b
        assert step_plan.current() is a
        # a worker thread (autograd's) does not inherit the context variable:
        # a node brings its own plan along (ops._bwd -> step_plan.running)
        seen = []

        def worker():
            seen.append(step_plan.current())
            with step_plan.running(b, "_SetEncoder.backward"):
                seen.append(step_plan.current())
            seen.append(step_plan.current())
        t = threading.Thread(target=worker)
        t.start()
        t.join()
        assert seen == [step_plan.ambient, b, step_plan.ambient]
    log = []
    a.park("k1_bwd", _Fake(log, "a.k1"))
    b.enter("_ColoredTemplates.backward")
    assert log == [] and a.holds("k1_bwd")
    a.flush()
    assert log == ["a.k1"]


# ---------------------------------------------------------------------------
# GPU: the guarded cases against the oracle
# ---------------------------------------------------------------------------
def _medium_cfg(**scae_params):
    """Reference-width encoder (the fused part-encoder node) on 32 x 32
    images: every fused launch of the full-size step, small enough for an
    unscreened oracle comparison."""
    return dict(image_shape=(1, 32, 32), n_classes=10, n_part_caps=8,
                n_obj_caps=6,
                scae_params=dict(reconstruct_alternatives=False,
                                 **scae_params))


def _filled_state(cfg, seed=1):
    from torch_scae_amd import factory
    np.random.seed(seed)
    torch.manual_seed(seed)
    proto = factory.make_scae(cfg)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in proto.parameters():
            if float(p.abs().sum()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    return {k: v.clone() for k, v in proto.state_dict().items()}, g


def _eager_step(cfg, sd, B, **kw):
    from torch_scae_amd import factory
    from torch_scae_amd.train_step import TrainStep
    np.random.seed(0)
    torch.manual_seed(0)
    model = factory.make_scae(cfg)
    model.load_state_dict(sd)
    model = model.cuda().train()
    return model, TrainStep(model, B, cfg["image_shape"], use_graph=False,
                            optimizer=False, **kw)


def _flat_grads(step):
    from torch_scae_amd import nn_ext
    buf = step.flat.flat_grad
    views = {id(p): buf[off:off + p.numel()].view(p.shape)
             for p, off in zip(step.flat.params, step.flat.offsets)}
    return nn_ext.named_reference_grads(step.model,
                                        grad_of=lambda p: views.get(id(p)))


def _assert_grads(got, ref_grads, bar, what):
    off, n = [], 0
    for k, ref in ref_grads.items():
        if ref is None:
            continue
        scale = float(ref.abs().max())
        g = got[k].detach().cpu()
        if scale == 0.0:
            assert float(g.abs().max()) <= 1e-6, (what, k)
            continue
        off.append((float((g - ref).abs().max()) / scale, k))
        n += 1
    off.sort(reverse=True)
    assert n > 50 and off[0][0] <= bar, (what, off[:6])
    return off[0]


@pytest.mark.gpu
@pytest.mark.parametrize("vote_type,presence_type", [
    ("soft", "enc"),          # SCAE's constructor default
    ("soft", "soft"), ("hard", "hard"), ("enc", "soft"), ("enc", "enc")])
def test_fused_train_step_vs_oracle_for_every_decoder_feed(vote_type,
                                                            presence_type):
    """inside a fused step the K1 backward used to be parked
    for the capsule likelihood's backward whatever fed the decoder; with
    'soft' / 'hard' votes or presences that node READS K1's pose / presence
    gradients, in the very launch that was still writing them.  Every feed,
    TrainStep's defaults (fuse_kernels=True), against the oracle; and the
    launch is shared exactly when the decoder is fed by the encoder."""
    from oracle import scae_oracle as O
    from tests.gate_screen import screened_scae_batch
    from torch_scae_amd import nn_utils, ops
    cfg = _medium_cfg(vote_type=vote_type, presence_type=presence_type)
    sd, g = _filled_state(cfg)
    B = 32
    image, label, noise = screened_scae_batch(O, cfg, sd, B, g)
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ocfg = O.prepare_model_params(**cfg)
    ref_loss, _, ref_grads = O.train_step(P, ocfg, image, label, noise)

    model, step = _eager_step(cfg, sd, B)
    assert step.fuse_kernels
    calls, real = [], ops._lib.call

    def spy(name, *a):
        calls.append(name)
        return real(name, *a)
    ops._lib.call = spy
    try:
        with nn_utils.fixed_noise([n.clone() for n in noise]):
            loss = step(image.cuda(), label.cuda())
        torch.cuda.synchronize()
    finally:
        ops._lib.call = real
    shared = "scae_render_gmm_sums_bwd_likelihood_f32" in calls
    assert shared == (vote_type == presence_type == "enc"), calls
    if not shared:
        assert "scae_render_gmm_sums_bwd_f32" in calls
        assert "scae_capsule_likelihood_bwd_f32" in calls
        # ... and K1's launch comes first: the likelihood reads its outputs
        assert calls.index("scae_render_gmm_sums_bwd_f32") < \
            calls.index("scae_capsule_likelihood_bwd_f32")
    assert abs(float(loss) - float(ref_loss)) <= 1e-4 * abs(float(ref_loss))
    # (a 'soft' / 'hard' presence reaches the decoder as a sum over capsule
    # posteriors and leaves through 1 / presence: a few 1e-4 of fp32 round-off
    # in the part encoder's gradients, measured 1.6e-4; a wrongly ordered
    # launch gives O(1))
    bar = 1e-4 if presence_type == "enc" else 3e-4
    worst = _assert_grads(_flat_grads(step), ref_grads, bar,
                          (vote_type, presence_type))
    print(vote_type, presence_type, "worst gradient entry", worst)


@pytest.mark.gpu
def test_fused_train_step_without_the_fused_part_encoder_vs_oracle():
    """with an encoder outside the fused part-encoder
    node (channel counts the implicit-GEMM kernels do not cover: MIOpen convs
    + ``ops.capsule_head``) ``parts.feature`` feeds BOTH the template generator
    and the object encoder, so autograd adds to the colour MLP's feature
    gradient as soon as that node returns: its launch must not be parked."""
    from oracle import scae_oracle as O
    from torch_scae_amd import nn_utils, ops
    cfg = dict(image_shape=(1, 16, 16), n_classes=4, n_part_caps=5,
               n_obj_caps=4,
               pcae_cnn_encoder_params=dict(out_channels=[24, 24],
                                            kernel_sizes=[3, 3],
                                            strides=[2, 1]),
               pcae_template_generator_params=dict(template_size=(5, 5)),
               ocae_encoder_set_transformer_params=dict(dim_hidden=8,
                                                        dim_out=64, n_layers=2),
               ocae_decoder_capsule_params=dict(dim_caps=4, hidden_sizes=(8,)),
               scae_params=dict(reconstruct_alternatives=False,
                                vote_type="enc", presence_type="enc"))
    sd, g = _filled_state(cfg, seed=4)
    B = 8
    image = torch.rand(B, 1, 16, 16, generator=g)
    label = torch.randint(0, 4, (B,), generator=g)
    noise = [torch.rand(B, 5, generator=g), torch.rand(B, 4, 1, generator=g),
             torch.rand(B, 4, 5, generator=g)]
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ocfg = O.prepare_model_params(**cfg)
    ref_loss, _, ref_grads = O.train_step(P, ocfg, image, label, noise)

    model, step = _eager_step(cfg, sd, B)
    assert not model.part_encoder.encoder._hip_stack
    calls, real = [], ops._lib.call

    def spy(name, *a):
        calls.append(name)
        return real(name, *a)
    ops._lib.call = spy
    try:
        with nn_utils.fixed_noise([n.clone() for n in noise]):
            loss = step(image.cuda(), label.cuda())
        torch.cuda.synchronize()
    finally:
        ops._lib.call = real
    assert "scae_capsule_head_bwd_f32" in calls
    assert "scae_capsule_head_bwd_tc_f32" not in calls
    assert calls.index("scae_template_color_bwd_f32") < \
        calls.index("scae_capsule_head_bwd_f32")
    assert abs(float(loss) - float(ref_loss)) <= \
        1e-4 * max(1.0, abs(float(ref_loss)))
    got = _flat_grads(step)
    for k, ref in ref_grads.items():
        if ref is None:
            continue
        d = float((got[k].cpu() - ref).abs().max())
        assert d <= 1e-4 * (1.0 + float(ref.abs().max())), (k, d)


@pytest.mark.gpu
def test_two_steps_interleaved_in_one_process_equal_the_steps_run_apart():
    """Two models, two TrainSteps, one process: forward(A), forward(B),
    backward(A), backward(B) -- each on its own step plan -- must leave in
    each flat gradient buffer exactly the bits of the same step run alone
    (parked launches, deferred sums, prologue buffers and noise generators are
    per plan; under module-level state B's forward would have dropped A's
    parked class-probability launch and A's backward carried B's K1)."""
    import contextlib
    cfgs = [_medium_cfg(vote_type="enc", presence_type="enc"),
            dict(_medium_cfg(vote_type="enc", presence_type="enc"),
                 n_part_caps=6, n_obj_caps=5)]
    B = 16
    g = torch.Generator().manual_seed(21)
    batches = [(torch.rand(B, 1, 32, 32, generator=g).cuda(),
                torch.randint(0, 10, (B,), generator=g).cuda())
               for _ in cfgs]
    states = [_filled_state(c, seed=7 + i)[0] for i, c in enumerate(cfgs)]

    def build():
        from torch_scae_amd import ops
        steps = []
        for cfg, sd in zip(cfgs, states):
            torch.manual_seed(99)
            ops.reset_noise()
            steps.append(_eager_step(cfg, sd, B)[1])
        return steps

    def forward(step, batch, stack):
        step._stage(*batch)
        step.flat.clear_grads()
        plan = step.plan
        with plan.active():
            stack.enter_context(step._lazy())
            stack.enter_context(plan.fusing(step.image))
            res = step.model(step.image)
            loss, _ = step.model.loss(res, step.image, step.label)
        return loss

    def backward(step, loss, stack):
        plan = step.plan
        with plan.active(), plan.deferring():
            loss.backward()
        stack.close()
        step.flat.gather_grads()
        torch.cuda.synchronize()
        return float(loss), step.flat.flat_grad.clone()

    # apart
    apart = []
    for step, batch in zip(build(), batches):
        with contextlib.ExitStack() as stack:
            apart.append(backward(step, forward(step, batch, stack), stack))
    # interleaved
    sa, sb = build()
    with contextlib.ExitStack() as ka, contextlib.ExitStack() as kb:
        la = forward(sa, batches[0], ka)
        lb = forward(sb, batches[1], kb)
        assert sa.plan.holds("class_probs") or sa.plan.holds("combine")
        ra = backward(sa, la, ka)
        rb = backward(sb, lb, kb)
    def differing(step, a, b):
        nm = {id(p): n for n, p in step.model.named_parameters()}
        return [(nm[id(p)], float((a[o:o + p.numel()] - b[o:o + p.numel()])
                                  .abs().max()))
                for p, o in zip(step.flat.params, step.flat.offsets)
                if not torch.equal(a[o:o + p.numel()], b[o:o + p.numel()])]

    for step, (l0, g0), (l1, g1) in zip((sa, sb), apart, (ra, rb)):
        assert l0 == l1, (l0, l1)
        assert torch.equal(g0, g1), ("interleaved", differing(step, g0, g1)[:8])
        assert float(g0.abs().max()) > 0
    # and the plain step of the class gives the same bits as the hand-run one
    for step, batch, (l0, g0) in zip(build(), batches, apart):
        loss = step(*batch)
        torch.cuda.synchronize()
        assert float(loss) == l0
        assert torch.equal(step.flat.flat_grad, g0), \
            ("plain", differing(step, step.flat.flat_grad, g0)[:8])


@pytest.mark.gpu
def test_fusing_and_deferring_on_a_model_without_gradient_slots_vs_oracle():
    """``ops.step_fusion`` + ``ops.deferred_param_sums``
    on a PLAIN model (no ``FlatParameters``: no gradient slots).  The K1
    backward used to be parked there although the column sums over its
    partial matrices (``templates_alpha``, the background scalars) are not
    deferrable without slots and launched at once -- on partials the parked
    launch had not written.  Now K1 launches directly unless every one of
    those sums waits too; every gradient against the oracle."""
    from oracle import scae_oracle as O
    from tests.gate_screen import screened_scae_batch
    from torch_scae_amd import factory, nn_ext, nn_utils, ops
    cfg = _medium_cfg(vote_type="enc", presence_type="enc")
    sd, g = _filled_state(cfg)
    B = 32
    image, label, noise = screened_scae_batch(O, cfg, sd, B, g)
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ocfg = O.prepare_model_params(**cfg)
    ref_loss, _, ref_grads = O.train_step(P, ocfg, image, label, noise)

    np.random.seed(0)
    torch.manual_seed(0)
    model = factory.make_scae(cfg)
    model.load_state_dict(sd)
    model = model.cuda().train()
    assert not any(hasattr(p, "_scae_grad_slot") for p in model.parameters())
    x, y = image.cuda(), label.cuda()
    calls, real = [], ops._lib.call

    def spy(name, *a):
        calls.append(name)
        return real(name, *a)
    ops._lib.call = spy
    try:
        with nn_utils.fixed_noise([n.clone() for n in noise]), \
                ops.step_fusion(x):
            res = model(x)
            loss, _ = model.loss(res, x, y)
            with ops.deferred_param_sums():
                loss.backward()
        torch.cuda.synchronize()
    finally:
        ops._lib.call = real
    # without slots nothing of K1's sums can wait, so K1 is not parked
    assert "scae_render_gmm_sums_bwd_likelihood_f32" not in calls, calls
    k1 = calls.index("scae_render_gmm_sums_bwd_f32")
    sums = [i for i, c in enumerate(calls) if c == "scae_sum_rows_multi_f32"]
    assert sums and min(sums) > k1, calls
    assert abs(float(loss) - float(ref_loss)) <= 1e-4 * abs(float(ref_loss))
    got = nn_ext.named_reference_grads(model)
    for k in ("part_decoder.templates_alpha", "part_decoder.bg_value",
              "part_decoder.bg_mixing_logit"):
        assert ref_grads[k] is not None and float(ref_grads[k].abs().max()) > 0
    worst = _assert_grads(got, ref_grads, 1e-4, "no slots")
    print("no slots: worst gradient entry", worst)
