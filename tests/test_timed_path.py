"""Parity of the path ``bench.py`` times: ``TrainStep`` with its defaults --
the step captured as a HIP graph and REPLAYED, batch hand-over + Philox noise
draws + folded attention weights + the encoder's image layer in the prologue
launch, lazy render, flat gradient slots with deferred column sums, fused
RMSprop -- at BASELINE.json's full batch sizes.

The reference's step is ``BaseExperiment.training_step`` + RMSprop
(torch_scae_experiments/base_experiment.py:44-77, :109-126).  Here each
replayed step is checked against the CPU oracle + stock ``torch.optim.RMSprop``
on the same parameters-before, the same batch and the step's OWN noise draws:
the device generator is counter based, so the draw of the next prologue launch
is predicted from its state, the batch is gate-screened against exactly that
noise (tests/gate_screen.py), and after the replay the step's noise buffer
must equal the prediction bit for bit.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import scae_oracle as O
from tests.golden_util import assert_close
from tests.test_hip_model import FULL, full_size_params

pytestmark = pytest.mark.gpu


def _noise_shapes(cfg, B):
    M, Oc = cfg["n_part_caps"], cfg["n_obj_caps"]
    return [(B, M), (B, Oc, 1), (B, Oc, M)]


def predict_noise(step):
    """What the step's next prologue launch will draw (flat device tensor):
    the same Philox kernel on a COPY of the generator state."""
    from torch_scae_amd import ops
    pro = step._pro
    assert pro is not None and pro.noise is not None
    state = pro.noise_state.clone()
    out = torch.empty_like(pro.noise)
    P = ctypes.c_void_p
    ops._lib.call("scae_uniform_f32", P(out.data_ptr()), out.numel(),
                  P(state.data_ptr()),
                  P(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return out


def split_noise(flat, cfg, B):
    shapes = _noise_shapes(cfg, B)
    sizes = [int(np.prod(s)) for s in shapes]
    assert flat.numel() == sum(sizes)
    return [c.view(s).cpu() for c, s in zip(flat.split(sizes), shapes)]


def flat_named(step, buf):
    """{reference state_dict key: that parameter's slice of the flat buffer
    ``buf``} (gradients: after a replay ``p.grad`` is whatever the capture
    left behind, the flat buffer is what the optimiser reads)."""
    from torch_scae_amd import nn_ext
    views = {id(p): buf[off:off + p.numel()].view(p.shape)
             for p, off in zip(step.flat.params, step.flat.offsets)}
    return nn_ext.named_reference_grads(step.model,
                                        grad_of=lambda p: views.get(id(p)))


def flat_grads(step):
    return flat_named(step, step.flat.flat_grad)


def build_step(cfg, B, sd, **kw):
    from torch_scae_amd import factory, ops
    from torch_scae_amd.train_step import TrainStep
    np.random.seed(0)
    torch.manual_seed(1234)
    ops.reset_noise()
    model = factory.make_scae(cfg)
    model.load_state_dict(sd)
    model = model.cuda().train()
    return model, TrainStep(model, B, cfg["image_shape"], **kw)


def check_replayed_steps(model, step, cfg, B, P, g, iters, lr=3e-5,
                         loss_rtol=1e-4, entry_bar=1e-4, l2_bar=None,
                         update_l2=5e-2, what="", min_tensors=200, screen=True,
                         entry_abs=0.0, fp64_judge=False, make_batch=None,
                         shared=None, impose=True):
    """``iters`` replays of ``step`` against the oracle + torch.optim.RMSprop
    started from the model's CURRENT state (``P``: leaf copies of it).
    ``entry_bar``: every entry of every parameter gradient within that fraction
    of its tensor's largest entry (+ ``entry_abs``); ``screen`` False: plain
    random batches (small configurations, where the gate screen finds no
    clean sample and the tests allow 1e-4 (1 + max) instead); ``fp64_judge``:
    for a state in which gradients are sums with heavy cancellation (a trained
    model whose capsules are off) the fp32 oracle is itself off from its fp64
    evaluation by more than 1e-4 of some small tensors: an entry may then be as
    far from the fp64 oracle as 1e-4 of its tensor's largest + 4 x the fp32
    oracle's own worst error on that tensor, and tensors whose gradient is
    below 1e-10 everywhere (dust next to RMSprop's eps) only have to be dust;
    ``impose`` (fp32 steps on the implicit-GEMM encoder): the gates of the
    convolution layers and of the per-capsule MLPs are not screened -- the HIP
    side's own pattern is imposed on the oracle (tests/gate_screen.py, round
    5), so the images are arbitrary as far as 97 % of the units go; only the
    units inside fused kernels that keep no activations (colour MLP,
    set-transformer feed-forward, relu1) are still screened;
    ``shared``: a dict through which two runs from the same state and noise
    (fp32 and bf16 operands) share the screened batch and the oracle's result; ``l2_bar``: additionally a relative-L2 bound
    per tensor (bf16 operands); ``update_l2`` None: no per-step update check."""
    ocfg = O.prepare_model_params(**cfg)
    ropt = torch.optim.RMSprop(list(P.values()), lr=lr, alpha=0.99,
                               momentum=0.9, eps=1e-2 / B ** 2)
    from tests.gate_screen import (hip_gates, imposed_gates,
                                   screened_batch_for_noise)
    enc = model.part_encoder.encoder
    impose = bool(impose and getattr(enc, "_hip_stack", False)
                  and step.autocast_dtype is None)
    n_conv = len(enc.strides) if impose else 0
    for it in range(iters):
        # the SAME state-before on both sides, every step: RMSprop's update is
        # ~10 lr sign(g) wherever |g| >> eps, so an entry whose gradient is at
        # round-off level may move 1e-4 apart in ONE step without either side
        # being wrong -- left alone, step 2 would compare gradients taken at
        # different parameters
        now = model.state_dict()
        sq = flat_named(step, step.opt.square_avg)
        mom = flat_named(step, step.opt.buf)
        with torch.no_grad():
            for k, p in P.items():
                p.copy_(now[k])
        flat_noise = predict_noise(step)
        noise = split_noise(flat_noise, cfg, B)
        reuse = shared.get(it) if shared is not None else None
        # the capsule MLPs' gates leave the screen only if they WILL be imposed: whether the
        # model takes the kernel that keeps them (K7b) is asked of the model itself, on a
        # probe image, before the batch is drawn (a model on the GEMM-by-GEMM
        # form has its capsule gates screened like every other unit)
        caps_imposed = False
        if impose and reuse is None:
            probe = torch.rand(B, *cfg["image_shape"],
                               generator=torch.Generator().manual_seed(it))
            caps_imposed = hip_gates(model, probe.cuda(), noise)[1] is not None
        if reuse is not None:
            assert torch.equal(reuse["noise"], flat_noise.cpu())
            image, label = reuse["image"], reuse["label"]
        elif screen:
            image, label = screened_batch_for_noise(
                O, cfg, {k: p.detach() for k, p in P.items()}, noise, g,
                n_classes=cfg["n_classes"], skip=n_conv, skip_caps=caps_imposed)
        elif make_batch is not None:
            image, label = make_batch(it)
        else:
            image = torch.rand(B, *cfg["image_shape"], generator=g)
            label = torch.randint(0, cfg["n_classes"], (B,), generator=g)
        if reuse is not None:
            ref_loss, ref_grads = reuse["loss"], reuse["grads"]
            for k, p in P.items():
                p.grad = ref_grads[k]
        elif impose:
            # the branch of every convolution unit as the HIP kernels decide it
            # (same parameters, same image: the step has not run yet)
            conv, caps = hip_gates(model, image.cuda(), noise)
            assert (caps is not None) == caps_imposed
            with imposed_gates(O, conv, caps, f"{what} step {it}"):
                ref_loss, _, ref_grads = O.train_step(P, ocfg, image, label,
                                                      noise)
        else:
            ref_loss, _, ref_grads = O.train_step(P, ocfg, image, label, noise)
        if reuse is None:
            if shared is not None:
                shared[it] = dict(noise=flat_noise.cpu(), image=image, label=label,
                                  loss=ref_loss.detach(),
                                  grads={k: None if v is None else v.clone()
                                         for k, v in ref_grads.items()})
        slack = {}
        if fp64_judge:
            P64 = {k: p.detach().double().requires_grad_(True)
                   for k, p in P.items()}
            _, _, g64 = O.train_step(P64, ocfg, image.double(), label,
                                     [n.double() for n in noise])
            for k, ref in ref_grads.items():
                if ref is not None:
                    slack[k] = 4.0 * float((ref.double() - g64[k]).abs().max())
            del P64, g64
        ref_before = {k: p.detach().clone() for k, p in P.items()}
        ropt.zero_grad(set_to_none=True)
        with torch.no_grad():
            for k, p in P.items():
                p.grad = ref_grads[k]
                if p.grad is None:
                    continue
                # torch.optim.RMSprop's own state of this parameter, set to
                # the fused optimiser's (zeros at a fresh start; the running
                # averages when the step continues a trained state)
                st = ropt.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32)
                    st["square_avg"] = torch.zeros_like(p)
                    st["momentum_buffer"] = torch.zeros_like(p)
                st["square_avg"].copy_(sq[k])
                st["momentum_buffer"].copy_(mom[k])
        ropt.step()

        before = {k: v.clone() for k, v in model.state_dict().items()}
        loss = step(image.cuda(), label.cuda())
        torch.cuda.synchronize()
        # the replay consumed exactly the predicted draws
        assert torch.equal(step._pro.noise, flat_noise), (what, it)
        assert abs(float(loss) - float(ref_loss)) <= \
            loss_rtol * abs(float(ref_loss)), (what, it, float(loss),
                                               float(ref_loss))
        # every entry of every parameter gradient within ``entry_bar`` of its
        # tensor's largest entry (the screened batch has no borderline gate)
        grads, off, l2, n, dust = flat_grads(step), [], [], 0, set()
        for k, ref in ref_grads.items():
            if ref is None:
                continue
            scale = float(ref.abs().max())
            got = grads[k].detach().cpu()
            if scale < (1e-10 if fp64_judge else 1e-30):
                # exactly zero, or dust (the HIP side flushes denormals).  In the
                # trained state (fp64_judge) the background's mixing-logit gradient is
                # sum_pixels g (w_post - w_prior) with both responsibilities exactly 1
                # in torch's log_softmax; K1's backward recomputes w_post from the
                # forward's saved log-sum-exp, one rounding (6e-8 of ~5) away from it:
                # 2e5 terms of g = 1 / B with a 5e-7 relative wobble sum to ~1e-5
                # (9e-6 measured; 2e-6 if the wobbles were independent, 8e-4 if they
                # all had one sign) where the alive state has O(1) -- 1e-9 of the
                # terms' magnitudes.  The bar leaves the trajectory room to move.
                assert float(got.abs().max()) <= \
                    max(5e-5 if fp64_judge else 1e-6, entry_bar * entry_abs), \
                    (what, it, k, float(got.abs().max()))
                dust.add(k)
                continue
            err = max(0.0, float((got - ref).abs().max()) - slack.get(k, 0.0))
            off.append((err / (scale + entry_abs), k, scale))
            l2.append((float((got - ref).double().norm()) / float(ref.double().norm()), k))
            n += 1
        assert n > min_tensors
        off.sort(reverse=True)
        l2.sort(reverse=True)
        assert off[0][0] <= entry_bar, (what, it, off[:6])
        if l2_bar is not None:
            assert l2[0][0] <= l2_bar, (what, it, l2[:6])
        if update_l2 is None:
            continue
        # the fused RMSprop moved every tensor like torch.optim.RMSprop did:
        # a step is ~10 lr sign(g) for |g| >> eps, so the few entries with a
        # gradient at round-off level differ; 5 % relative L2 per tensor
        # (a wrong gradient or optimiser state gives O(1))
        after = model.state_dict()
        for k in ref_before:
            d_ref = P[k].detach() - ref_before[k]
            d_hip = (after[k] - before[k]).cpu()
            if fp64_judge and k in dust:
                # a gradient that is dust on the reference side and <= 5e-5 on the HIP side
                # (above) is still up to 80 x RMSprop's eps: the optimiser turns it into a
                # step of its usual size.  All that can be asked is that it IS one step
                # (momentum 0.9 on g / sqrt(v) <= 10: at most 100 lr).
                assert float(d_hip.abs().max()) <= 100 * lr, (what, it, k)
                continue
            nr = float(d_ref.norm())
            if nr == 0.0:
                assert float(d_hip.abs().max()) == 0.0, (what, it, k)
                continue
            if nr < 1e-2 * lr:   # dust: the WHOLE tensor moved by less than 1e-3 of what
                # one entry with |g| >> eps moves (~10 lr) -- both sides negligible
                assert float(d_hip.norm()) < 1e-1 * lr, (what, it, k)
                continue
            assert float((d_hip - d_ref).norm()) <= update_l2 * nr, \
                (what, it, k, float((d_hip - d_ref).norm()), nr)
    return off[0][:2], l2[0]


@pytest.mark.parametrize("name", ["cfg2", "cfg5", "mnist_40_32"])
def test_replayed_train_step_vs_oracle(name):
    cfg, B, sd, g = full_size_params(name)
    model, step = build_step(cfg, B, sd)      # bench.py's defaults
    assert step.use_graph and step._pro is not None and step.opt is not None
    assert step._lazy_dec is not None and not step.collective
    step.capture()
    assert step.graph is not None and step._pro.noise is not None
    assert step._pro.fold_outs is not None and step._pro.first_outs is not None

    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lr = 3e-5
    check_replayed_steps(model, step, cfg, B, P, g, 3, lr=lr, what=name)
    final = model.state_dict()
    for k, p in P.items():
        assert_close(final[k].cpu(), p.detach(), 1e-4, 2e-3, "param " + k)
        # ... and tighter, against what three such steps can move at all
        assert float((final[k].cpu() - p.detach()).abs().max()) <= \
            0.05 * 3 * 10 * lr, k


def test_replayed_cfg3_step_vs_oracle():
    """BASELINE.json configs[2]'s shape (48 / 64 capsules, B = 1024) as
    ``bench.py`` times it: the REPLAYED step, whose kernel selections differ
    from B = 128 (``gemm_multi_kernel<0>``, separate ``pool_bwd`` / ``tc_bwd``,
    ``saw_bwd_kernel<4>``, the large-batch loss tail, 32-row capsule-MLP
    workgroups).  fp32: the 1e-4 bars of the other configurations.  ``--bf16``
    (``autocast_dtype=torch.bfloat16``: bf16 operands on the matrix cores,
    fp32 accumulation): loss 2^-7 relative, every gradient tensor 5e-2
    relative L2 AND every entry within 2^-3 of its tensor's largest entry,
    against the fp32 oracle.  (Measured worst entry: 0.06 of its tensor's
    largest, in the per-capsule MLP weights -- operands rounded to 2^-9 flip
    the ReLU gates of the ~0.3 % of units within that rounding of zero, which
    moves single entries of a 1024-sample sum by a sample's share; 2^-5 is
    not what bf16 operands deliver.)  Both precisions start from the same
    state and draw the same noise, so they share one screened batch and one
    oracle evaluation."""
    cfg, B, sd, g = full_size_params("cfg3_shape")
    shared = {}
    for bf16 in (False, True):
        kw = dict(autocast_dtype=torch.bfloat16) if bf16 else {}
        model, step = build_step(cfg, B, sd, **kw)
        step.capture()
        assert step.graph is not None and step._pro.noise is not None
        P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        if bf16:
            worst, worst_l2 = check_replayed_steps(
                model, step, cfg, B, P, g, 1, loss_rtol=2.0 ** -7,
                entry_bar=2.0 ** -3, l2_bar=5e-2, update_l2=None,
                what="cfg3 bf16", shared=shared)
        else:
            worst, worst_l2 = check_replayed_steps(
                model, step, cfg, B, P, g, 1, what="cfg3 fp32", shared=shared)
        print(f"cfg3 {'bf16' if bf16 else 'fp32'} replayed step: worst entry "
              f"{worst}, worst L2 {worst_l2}")
        del model, step
        torch.cuda.empty_cache()


def test_fifty_replayed_steps_vs_oracle_and_torch_rmsprop():
    """SURVEY.md 8f.2 over a longer run: 50 replayed steps of ``TrainStep``
    (graph, prologue noise, fused RMSprop; lr large enough that the loss moves)
    on a small full-width model, EVERY step held to the oracle + stock
    ``torch.optim.RMSprop`` (base_experiment.py:44-77, :109-126) from the
    state the HIP trajectory has reached: loss 1e-4, the update 5 % L2 per
    tensor, every gradient entry 2e-3 (1 + its tensor's largest).  The entry
    bar is that of an UNSCREENED batch of 8 images x 5 capsules (the gate
    screen finds no clean sample for a model this small) of a model in
    motion: the pose non-linearities (theta x 2 pi, tanh(5 x)) amplify the
    fp32 round-off of the pooled features tenfold and more
    (tools/diag_fifty.py: at lr 1e-3 the pooled features of the two sides
    were 5e-5 apart after 17 steps, their poses 4e-4, the gradients at the
    encoder's output 0.3 % -- with K1 alone, on identical inputs, at 1e-6),
    and one ReLU gate within round-off of zero is an eighth of a sample's
    share here.  1e-5 .. 5e-4 observed; a wrong kernel gives O(1).  Both sides start
    each step from the same state because a free-running comparison measures
    the optimiser, not the kernels: RMSprop's update is ~lr sign(g) wherever
    |g| >> eps, so the oracle itself, evaluated in fp32 and in fp64, is
    1.2e-3 apart in the loss after 11 such steps and O(1) apart in the
    parameters whose gradients hover around zero after 50."""
    cfg = dict(image_shape=(1, 16, 16), n_classes=4, n_part_caps=5,
               n_obj_caps=4,
               pcae_cnn_encoder_params=dict(out_channels=[64, 64],
                                            kernel_sizes=[3, 3],
                                            strides=[2, 1]),
               pcae_template_generator_params=dict(template_size=(5, 5)),
               ocae_encoder_set_transformer_params=dict(dim_hidden=8,
                                                        dim_out=64, n_layers=2),
               ocae_decoder_capsule_params=dict(dim_caps=4, hidden_sizes=(8,)),
               scae_params=dict(reconstruct_alternatives=False))
    from torch_scae_amd import factory
    np.random.seed(3)
    torch.manual_seed(3)
    proto = factory.make_scae(cfg)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for p in proto.parameters():
            if float(p.abs().sum()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    sd = {k: v.clone() for k, v in proto.state_dict().items()}
    B, lr = 8, 1e-4
    model, step = build_step(cfg, B, sd, lr=lr)
    step.capture()
    from torch_scae_amd.data import stroke_batches as _sb
    warm = _sb(1, B, cfg["image_shape"], seed=8, n_classes=4)
    first = float(step(warm[0][0].cuda(), warm[1][0].cuda()))
    P = {k: v.detach().cpu().clone().requires_grad_(True)
         for k, v in model.state_dict().items()}
    # structured images (data.stroke_batches): on noise a model this small has
    # switched its capsules off -- most gradients exactly zero -- after 20 steps
    from torch_scae_amd.data import stroke_batches
    pool = stroke_batches(50, B, cfg["image_shape"], seed=9, n_classes=4)
    worst, _ = check_replayed_steps(
        model, step, cfg, B, P, g, 50, lr=lr, what="50 steps", min_tensors=25,
        screen=False, entry_abs=1.0, entry_bar=2e-3,
        make_batch=lambda it: (pool[0][it], pool[1][it]))
    last = float(step.loss)
    print(f"50 steps: loss {first:.3f} -> {last:.3f}, worst gradient entry "
          f"{worst}")
    moved = max(float((model.state_dict()[k].cpu() - sd[k]).abs().max())
                for k in sd)
    assert moved > 20 * lr, moved       # the trajectory went somewhere


def test_replayed_step_on_the_state_the_bench_ends_in():
    """``bench.py`` trains on U[0,1) noise images for thousands of steps before
    and while it times; on such data the part capsules switch off (DESIGN.md
    section 5): presences fall below log_safe's 1e-16 and the pose scale
    collapses, so that a whole image falls into one or two texel cells of a
    template -- the regime in which the K1 backward splits a cell over (row
    slice x row segment) lanes, and in which a component with such a presence
    gets exactly zero gradients without its pixel loop.  The bench's own model
    (its seed, its batches, TrainStep's defaults) is stepped until it is there
    (bench.py reports it after 600 steps), then the replayed step is held to the
    oracle from THAT state.  The entry bar there is 5e-4 of a tensor's largest
    entry, not the 1e-4 of the live states: which dead state a trajectory ends in
    depends on every bit of every kernel, and over ten such states (five sets of
    batches x two forms of the second layer's data gradient, round 6,
    profiles/r06/trained_state_spread.txt) the worst entry ranged from 1.2e-6 to
    2.1e-4 whatever the kernels -- presences of 1e-29 .. 1e-31 put products on
    either side of the denormal line, which the HIP side flushes and torch's CPU
    kernels keep, and 1 / presence brings them back; the tensors concerned have
    gradients of 1e-5 .. 5e-4, the errors are 5e-9 .. 1e-7 absolute, below
    RMSprop's eps (6e-7).  The bar is relative to the tensor's largest entry +
    100 eps: an error of 5e-4 x 100 eps = 5 % of eps cannot change an RMSprop
    step (lr g / (sqrt(v) + eps)) by more than 5 % of lr, whatever the tensor's
    own scale (a later trajectory had a tensor of scale 2.7e-6 = 4 eps 1.4e-9
    off).  SCAE_TEST_SEED / SCAE_TEST_EXTRA_STEPS move the trajectory (other
    batches / more steps) for such a sweep."""
    import bench
    cfg_b = bench.CONFIGS["mnist_24_24_bs128"]
    cfg, B = cfg_b["model"], cfg_b["batch"]
    from torch_scae_amd import ops
    torch.manual_seed(1234)
    ops.reset_noise()
    step = bench.make_step(cfg_b, torch.device("cuda", 0))
    model = step.model
    step.capture()
    images, labels = bench.synthetic_batches(
        cfg_b, torch.device("cuda", 0), 1000 + int(os.environ.get("SCAE_TEST_SEED", "0")))
    state = None
    for n_steps in range(500, 4001, 500):
        for i in range(500 + int(os.environ.get("SCAE_TEST_EXTRA_STEPS", "0"))):
            step(images[i % 8], labels[i % 8])
        torch.cuda.synchronize()
        state = bench.capsule_state(model, images[0])
        print(f"after {n_steps} steps: {state}")
        if state["presence_below_1e-16"] == 1.0:
            break
    # the regime is the trained one: every capsule off, poses collapsed
    assert state["presence_below_1e-16"] == 1.0 and \
        state["pose_scale_median"] < 0.1, state
    g = torch.Generator().manual_seed(5)
    P = {k: v.detach().cpu().clone().requires_grad_(True)
         for k, v in model.state_dict().items()}
    # (with every part capsule off, the gradients that pass through the part
    # decoder are exactly zero on both sides: fewer tensors carry an error)
    worst, _ = check_replayed_steps(model, step, cfg, B, P, g, 2,
                                    what="trained state", min_tensors=50,
                                    fp64_judge=True, entry_bar=5e-4,
                                    entry_abs=100 * 1e-2 / B ** 2)
    print("worst gradient entry in the trained state:", worst)


def _set_counter(step, value):
    """Put the step's device generator at launch ``value`` (arrivals 0)."""
    st = step._pro.noise_state
    st[1:] = torch.tensor([value, 0], dtype=st.dtype, device=st.device)
    step._pro.noise_fresh = False


def _three_steps(step, images, labels):
    losses, draws = [], []
    for i in range(3):
        losses.append(float(step(images[i], labels[i])))
        draws.append(step._pro.noise.clone())
    torch.cuda.synchronize()
    # per parameter NAME: the flat order depends on the layout (front block)
    names = {id(p): n for n, p in step.model.named_parameters()}
    state = {}
    for p, off in zip(step.flat.params, step.flat.offsets):
        for what, buf in (("param", step.flat.flat_param),
                          ("grad", step.flat.flat_grad),
                          ("square_avg", step.opt.square_avg),
                          ("buf", step.opt.buf)):
            state[what + "/" + names[id(p)]] = \
                buf[off:off + p.numel()].clone()
    return losses, draws, state


def test_replayed_step_equals_eager_step_bitwise():
    """graph == eager at cfg-2, B = 128, WITH the prologue's Philox noise: both
    runs start their generator at the same launch counter, take three steps on
    the same batches and must agree bit for bit in noise, loss, parameters,
    gradients and optimiser state."""
    cfg, B, sd, g = full_size_params("cfg2")
    images = torch.rand(3, B, *cfg["image_shape"], generator=g).cuda()
    labels = torch.randint(0, 10, (3, B), generator=g).cuda()
    out = {}
    for use_graph in (False, True):
        model, step = build_step(cfg, B, sd, use_graph=use_graph)
        if use_graph:
            step.capture()
        else:
            step._fwd_bwd()       # establishes the prologue's buffers
        assert step._pro.noise is not None
        _set_counter(step, 1000)
        out[use_graph] = _three_steps(step, images, labels)
    (l0, d0, s0), (l1, d1, s1) = out[False], out[True]
    for a, b in zip(d0, d1):
        assert torch.equal(a, b)
    assert not torch.equal(d0[0], d0[1])
    assert l0 == l1, (l0, l1)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


@pytest.mark.parametrize("knob,off,on,may_differ", [
    ("SCAE_K8_DGX", "0", "500", ("part_encoder.encoder.network.0.",)),
    ("SCAE_K8_DGK", "0", "1", ("part_encoder.encoder.network.",)),
])
def test_dma_data_gradient_tiles_change_only_the_encoders_gradients(knob, off, on, may_differ,
                                                                   monkeypatch):
    """The replayed cfg-2 step with a convolution layer's data gradient on a DMA-fed tile
    (conv_mfma.hip; DMODE 4 for the second layer, the default for a layer of >= 500 such tiles;
    DMODE 5, the K loop split over the waves, for the third and fourth) and on the
    first-generation tiles (SCAE_K8_DGX=0 / SCAE_K8_DGK=0): same state, noise and batch.  A
    data gradient only feeds the layers below it (and DMODE 5's launch sums its weight-gradient
    partials in other groups), so the loss and every gradient outside the convolution stack --
    the riders of the same launches included -- are the same bits, and the stack's own agree to
    round-off: nothing else is touched."""
    cfg, B, sd, g = full_size_params("cfg2")
    images = torch.rand(1, B, *cfg["image_shape"], generator=g).cuda()
    labels = torch.randint(0, 10, (1, B), generator=g).cuda()
    out = {}
    for mode in (off, on):
        monkeypatch.setenv(knob, mode)
        model, step = build_step(cfg, B, sd)
        step.capture()
        _set_counter(step, 1000)
        loss = float(step(images[0], labels[0]))
        torch.cuda.synchronize()
        names = {id(p): n for n, p in model.named_parameters()}
        out[mode] = (loss, {names[id(p)]: step.flat.flat_grad[o:o + p.numel()].clone()
                            for p, o in zip(step.flat.params, step.flat.offsets)})
    (l0, g0), (l1, g1) = out[off], out[on]
    assert l0 == l1
    allowed = [k for k in g0 if k.startswith(may_differ)]
    differ = [k for k in g0 if not torch.equal(g0[k], g1[k])]
    assert allowed and differ and set(differ) <= set(allowed), (differ, allowed)
    for k in allowed:
        top = float(g0[k].abs().max())
        assert float((g0[k] - g1[k]).abs().max()) <= 1e-5 * top, k


def test_two_graph_collective_step_equals_plain_step_bitwise_cfg2(nccl_group):
    """BASELINE.json configs[3]'s step shape on one GPU: a 1-rank RCCL group
    drives TrainStep's two-graph mode (bucket 1 all-reduced between the
    graphs) at cfg-2, B = 128, with Philox noise; it must end on exactly the
    state of the collective-free replayed step."""
    cfg, B, sd, g = full_size_params("cfg2")
    images = torch.rand(3, B, *cfg["image_shape"], generator=g).cuda()
    labels = torch.randint(0, 10, (3, B), generator=g).cuda()
    out = {}
    for mode, kw in (("plain", {}),
                     ("2 buckets", dict(force_collective=True)),
                     ("1 bucket", dict(force_collective=True,
                                       overlap=False)),
                     ("in graph", dict(force_collective=True,
                                       collective_mode="in graph"))):
        model, step = build_step(cfg, B, sd, **kw)
        step.capture()
        if mode == "plain":
            assert step.collective_mode is None and step.graph_b is None
        elif mode == "2 buckets":
            assert step.split and step.graph_b is not None
            assert 0 < step.flat.n_front < step.flat.numel
        elif mode == "in graph":
            assert step.in_graph_collective and not step.split
        else:
            assert step.collective and not step.split
        _set_counter(step, 1000)
        out[mode] = _three_steps(step, images, labels)
    l0, d0, s0 = out["plain"]
    for mode in ("2 buckets", "1 bucket", "in graph"):
        l, d, s = out[mode]
        assert all(torch.equal(a, b) for a, b in zip(d0, d)), mode
        assert l == l0, (mode, l, l0)
        assert set(s) == set(s0)
        for k in s0:
            assert torch.equal(s0[k], s[k]), (mode, k)


def test_likelihood_riding_in_the_trunk_launch_changes_nothing():
    """TrainStep(fuse_kernels=True): the part decoder's likelihood forward runs
    as a second block range of the object encoder's trunk launch
    (csrc/trunk_logprob.hip) instead of a launch of its own, and the capsule
    likelihood's backward as the first block range of the part decoder's
    likelihood backward (csrc/render_bwd_likelihood.hip).  Same kernels on
    the same inputs: three replayed steps at cfg-2, B = 128, end bit for bit
    on the state of the step with the two launches (noise on, generators
    aligned), and the step really takes the shared launch."""
    from torch_scae_amd import ops
    cfg, B, sd, g = full_size_params("cfg2")
    images = torch.rand(3, B, *cfg["image_shape"], generator=g).cuda()
    labels = torch.randint(0, 10, (3, B), generator=g).cuda()
    out, calls = {}, {}
    real = ops._lib.call
    for fuse in (True, False):
        seen = calls[fuse] = []

        def spy(name, *a, _seen=seen):
            _seen.append(name)
            return real(name, *a)
        ops._lib.call = spy
        try:
            model, step = build_step(cfg, B, sd, fuse_kernels=fuse)
            step.capture()
        finally:
            ops._lib.call = real
        _set_counter(step, 1000)
        out[fuse] = _three_steps(step, images, labels)
    # the template generator's colour kernel rides behind the part encoder's head
    assert "scae_template_color_fwd_f32" in calls[False]
    assert "scae_template_color_fwd_f32" not in calls[True]
    assert "scae_set_encoder_fwd_logprob_f32" in calls[True]
    assert "scae_render_gmm_logprob_sums_fwd_f32" not in calls[True]
    assert "scae_set_encoder_fwd_logprob_f32" not in calls[False]
    assert "scae_render_gmm_logprob_sums_fwd_f32" in calls[False]
    # ... and the backward mirror image: the capsule likelihood's backward
    # rides in the launch of the (parked) reconstruction likelihood's backward
    assert "scae_render_gmm_sums_bwd_likelihood_f32" in calls[True]
    assert "scae_render_gmm_sums_bwd_f32" not in calls[True]
    assert "scae_capsule_likelihood_bwd_f32" not in calls[True]
    assert "scae_render_gmm_sums_bwd_f32" in calls[False]
    assert "scae_capsule_likelihood_bwd_f32" in calls[False]
    (l1, d1, s1), (l0, d0, s0) = out[True], out[False]
    assert all(torch.equal(a, b) for a, b in zip(d0, d1))
    assert l0 == l1, (l0, l1)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def test_recorded_launch_list_equals_graph_replay_bitwise():
    """``TrainStep(replay="launches")``: the captured step re-issued from the library's own record of the kernel launches
    the capture made (``scae_launch_list_*``: kernel, grid, block, LDS, argument bytes; one
    hipLaunchKernel each) instead of as a HIP-graph replay -- the same launches, so loss,
    parameters and optimiser state after three steps must agree bit for bit with the
    graph's (cfg-2, B = 128, noise on)."""
    cfg, B, sd, g = full_size_params("cfg2")
    images = torch.rand(3, B, *cfg["image_shape"], generator=g).cuda()
    labels = torch.randint(0, cfg["n_classes"], (3, B), generator=g).cuda()
    out = []
    for mode in ("graph", "launches"):
        model, step = build_step(cfg, B, sd, replay=mode)
        losses = []
        for i in range(3):
            losses.append(float(step(images[i], labels[i])))
        torch.cuda.synchronize()
        assert step.graph is not None and len(step._launches) >= 15
        # every column sum of the step rides in the optimiser's launch: none on its own
        # (the warm-ups' held sums must not be flushed INTO the capture)
        names = [getattr(fn, "__name__", "?") for fn, _, _ in step._launches]
        assert "scae_rmsprop_sums_step_f32" in names, names
        assert "scae_sum_rows_multi_f32" not in names, names
        if mode == "launches":
            from torch_scae_amd import _lib
            # the list is taken because the graph holds exactly its launches, nothing else;
            # (a C-ABI call is at least one kernel launch)
            assert step._klist, step.graph_nodes
            nodes, kernels, recorded = step.graph_nodes
            assert nodes == kernels == recorded >= len(step._launches), step.graph_nodes
            assert _lib.load().scae_launch_list_size(step._klist) == recorded
        else:
            assert step._klist is None
        out.append((losses, step.flat.flat_param.clone(), step.opt.square_avg.clone(),
                    step.opt.buf.clone(), step.flat.flat_grad.clone()))
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    for a, b, what in zip(out[0][1:], out[1][1:], ("param", "square_avg", "buf", "grad")):
        assert torch.equal(a, b), what
    assert float(out[0][4].abs().max()) > 0


def test_two_lane_launch_replay_equals_graph_replay_bitwise():
    """``TrainStep(replay="launches", two_lanes=True)``: K1's backward on a second stream,
    as whole grid and from a fixed number of resident workgroups
    (``scae_decoder_desc.bwd_resident``), the fork / join edges re-issued from the list's own
    record.  Whatever the lanes' relative timing, loss, parameters, gradients and optimiser
    state after three steps equal the graph replay's bit for bit (cfg-2, B = 128, noise on)."""
    from torch_scae_amd import _lib
    cfg, B, sd, g = full_size_params("cfg2")
    images = torch.rand(3, B, *cfg["image_shape"], generator=g).cuda()
    labels = torch.randint(0, cfg["n_classes"], (3, B), generator=g).cuda()
    out = []
    for mode, resident in (("graph", None), ("launches", 0), ("launches", 384)):
        model, step = build_step(cfg, B, sd, replay=mode, two_lanes=mode == "launches")
        if resident is not None:
            step.plan.side_resident = resident
        losses = [float(step(images[i], labels[i])) for i in range(3)]
        torch.cuda.synchronize()
        if mode == "launches":
            lib = _lib.load()
            assert step._klist, step.graph_nodes
            # K1's backward alone on the side lane, the capsule likelihood's on its own
            assert lib.scae_launch_list_side_size(step._klist) == 1
            names = [getattr(fn, "__name__", "?") for fn, _, _ in step._launches]
            assert "scae_render_gmm_sums_bwd_f32" in names, names
            assert "scae_capsule_likelihood_bwd_f32" in names, names
        out.append((losses, step.flat.flat_param.clone(), step.opt.square_avg.clone(),
                    step.opt.buf.clone(), step.flat.flat_grad.clone()))
    for other in out[1:]:
        assert out[0][0] == other[0], (out[0][0], other[0])
        for a, b, what in zip(out[0][1:], other[1:], ("param", "square_avg", "buf", "grad")):
            assert torch.equal(a, b), what


def test_launch_replay_steps_aside_for_a_graph_with_other_nodes():
    """``training_step()`` computes the accuracy and copies the log values with
    torch kernels -- nodes of the captured graph that the library's launch list does not
    hold.  ``replay="launches"`` must then replay the GRAPH (the list would freeze the log at
    its capture-time values): the log of three steps on three batches equals the graph
    mode's, entry for entry, and changes from step to step.  Two steps built in one process,
    each with its own recording."""
    cfg, B, sd, g = full_size_params("cfg2_bs32")
    images = torch.rand(3, B, *cfg["image_shape"], generator=g).cuda()
    labels = torch.randint(0, cfg["n_classes"], (3, B), generator=g).cuda()
    logs = {}
    steps = []
    for mode in ("graph", "launches"):
        model, step = build_step(cfg, B, sd, replay=mode)
        steps.append(step)             # (both stay alive: two captured steps in one process)
        rows = []
        for i in range(3):
            out = step.training_step(images[i], labels[i])
            torch.cuda.synchronize()
            rows.append({k: float(v) for k, v in out["log"].items()})
        logs[mode] = rows
        if mode == "launches":
            assert step._klist is None, "a graph with torch nodes must not be replayed as a list"
            assert step.graph_nodes is None or step.graph_nodes[0] > step.graph_nodes[2]
    assert logs["graph"] == logs["launches"], (logs["graph"], logs["launches"])
    assert logs["graph"][0]["loss"] != logs["graph"][1]["loss"]
    # the plain step of the same process still takes its list
    model, step = build_step(cfg, B, sd, replay="launches")
    step(images[0], labels[0])
    torch.cuda.synchronize()
    assert step._klist, step.graph_nodes


def test_reset_noise_restarts_an_already_built_step():
    """``torch.manual_seed(s); ops.reset_noise()`` has to restart the noise
    sequence of a step that is already built and captured (its generator state lives on the
    step's own plan, and inside the graph by address): the same three draws again."""
    from torch_scae_amd import ops
    cfg, B, sd, g = full_size_params("cfg2_bs32")
    model, step = build_step(cfg, B, sd)          # (manual_seed(1234) + reset_noise inside)
    image = torch.rand(B, *cfg["image_shape"], generator=g).cuda()
    label = torch.randint(0, cfg["n_classes"], (B,), generator=g).cuda()

    def draws(n):
        out = []
        for _ in range(n):
            step(image, label)
            torch.cuda.synchronize()
            out.append(step._pro.noise.clone())
        return out
    first = draws(3)
    assert not torch.equal(first[0], first[1])
    torch.manual_seed(1234)
    ops.reset_noise()
    again = draws(3)
    # (the capture's warm-ups consumed draws before `first`: the restarted sequence begins at
    # the generator's origin, so compare it with a second restart, and require that it differs
    # from simply continuing)
    torch.manual_seed(1234)
    ops.reset_noise()
    third = draws(3)
    for a, b in zip(again, third):
        assert torch.equal(a, b)
    assert not torch.equal(again[0], first[0]) or not torch.equal(again[1], first[1])
    torch.manual_seed(4321)
    ops.reset_noise()
    other = draws(1)
    assert not torch.equal(other[0], again[0])
