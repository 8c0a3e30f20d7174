"""torch.autograd bridges to the HIP kernels of libscae_hip.so.

Each Function here is the host side of one C-ABI launcher pair in
``include/scae_hip.h``.  PyTorch only owns device memory and streams: buffers
are ``torch.empty`` allocations, kernels are enqueued on the calling thread's
current HIP stream (so everything composes with ``torch.cuda.graph`` capture
and with autograd's backward thread).  No op has a CPU or eager fallback.
"""
import ctypes
import functools

import numpy as np
import torch

from . import _lib, step_plan
from ._lib import DecoderDesc, ScaeHipError
from .step_plan import StepPlan, current as _plan

__all__ = ["geometric_transform", "qkv_attention", "set_encoder", "grouped_mlp", "seed_attention", "seed_attention_supported", "seed_fold", "seed_fold_supported", "loss_tail", "loss_tail_scalar", "loss_tail_supported", "capsule_votes",
           "capsule_likelihood", "colored_templates", "template_color_supported", "attention_conv_pool", "attention_pool_supported", "capsule_head", "part_encoder", "conv_stack", "conv_stack_supported",
           "uniform", "reset_noise", "pack_params", "render_templates", "render_gmm_log_prob", "render_gmm_log_prob_sums",
           "gmm_log_prob", "gmm_mean", "gmm_mode", "ScaeHipError"]


def _need_hip(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise ScaeHipError(
                "torch_scae_amd ops run only on a HIP device (got a "
                f"{t.device} tensor); there is no CPU path.")
        if t.dtype != torch.float32:
            raise ScaeHipError(f"float32 expected, got {t.dtype}")


def _c(t):
    return None if t is None else t.contiguous()


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(ref):
    return ctypes.c_void_p(torch.cuda.current_stream(ref.device).cuda_stream)


def _detached(t):
    return None if t is None else t.detach()


def _fwd(f):
    """``forward`` of a plan-aware autograd node: the step plan of the calling
    thread (step_plan.current) is kept on the node for its backward."""
    node = f.__qualname__

    @functools.wraps(f)
    def forward(ctx, *args):
        ctx.plan = plan = _plan()
        plan.enter(node)
        return f(ctx, *args)
    return staticmethod(forward)


def _bwd(f):
    """``backward`` of a plan-aware autograd node.  Autograd's worker thread
    does not inherit context variables: the node's own plan is made current
    while it runs, and whatever is parked on that plan and read by this node
    (step_plan.RIDES) is launched first."""
    node = f.__qualname__

    side = node in step_plan.SIDE_NODES

    @functools.wraps(f)
    def backward(ctx, *grads):
        reads = ()
        if side and ctx.plan.side_stream is not None:
            # (everything the node reads that autograd hands it: incoming
            # gradients and saved tensors)
            reads = tuple(grads) + tuple(ctx.saved_tensors)
        with step_plan.running(ctx.plan, node, reads):
            return f(ctx, *grads)
    return staticmethod(backward)


def _slot(t, ctx=None):
    """The flat-gradient slot of a parameter (data_parallel.GradSlot) or None
    (``ctx``: the autograd node asking, unused)."""
    return getattr(t, "_scae_grad_slot", None) if t is not None else None


def _grad_out(slot, like, shape=None):
    """Buffer for a gradient of ``shape`` (default ``like.shape``): the
    parameter's slot in the flat gradient buffer when it has an unclaimed one,
    else a fresh tensor on ``like``'s device."""
    shape = tuple(like.shape if shape is None else shape)
    if slot is not None and tuple(slot.shape) == shape:
        v = slot.take()
        if v is not None:
            # first gradient of this parameter in this backward: a column sum
            # into the slot may be deferred (``_in_slot``)
            v._scae_in_slot = True
            return v
        # a SECOND gradient of the parameter (shared parameter, module applied
        # twice): autograd is about to add this buffer to the slot's contents,
        # so whatever sum into the slot -- or parked launch that writes it --
        # is still waiting has to run first
        flush_param_sums()
        flush_pending_backward()
    return torch.empty(shape, device=like.device, dtype=like.dtype)


def _in_slot(t):
    """True for a gradient buffer ``_grad_out`` took from a parameter's slot."""
    return t is not None and getattr(t, "_scae_in_slot", False)


def _sum_jobs(jobs):
    arr = (_lib.ScaledSum * max(1, len(jobs)))()
    for a, (src, scale, dst) in zip(arr, jobs):
        assert src.is_contiguous() and dst.numel() == 1
        a.src, a.n, a.scale, a.dst = src.data_ptr(), src.numel(), scale, \
            dst.data_ptr()
    return arr


def scaled_sums(jobs):
    """[(src, scale, dst)]: dst[()] = scale * src.sum() for up to 8 tensors in
    ONE launch (the scalar outputs of the forward pass)."""
    _lib.call("scae_scaled_sums_f32", _sum_jobs(jobs), len(jobs),
              _stream(jobs[0][0]))


# BASELINE.json configs[2] ("bs=1024 bf16"): inside ``mfma_bf16()`` the GEMM-shaped kernels
# (K7 batched GEMMs, K8 convolutions) round their fp32 operands to bf16 on the way into LDS
# and multiply on v_mfma_f32_32x32x16_bf16 (fp32 accumulate, fp32 results) wherever the
# problem is large enough for the 128 x 128 tiles; everything else stays fp32.  The
# context has to cover the backward pass as well (train_step.TrainStep does that).
# The flag is a field of the current step plan (step_plan.StepPlan.bf16).
def mfma_bf16(enabled=True):
    return _plan().precision(enabled)


def _bf16():
    return _plan().bf16


def _prec(name):
    """C entry point of a GEMM-shaped launcher for the current precision."""
    return name.replace("_f32", "_bf16") if _plan().bf16 else name


# where a training step's folding products of the output attention ride: "conv" = the second
# convolution layer's forward launch, "prologue" = the step prologue's (measured per kernel
# generation: DESIGN.md section 5)
_FOLD_RIDES = __import__("os").environ.get("SCAE_FOLD_RIDES", "conv")


class StepPrologue:
    """Buffers of a training step's prologue launch (csrc/step_prologue.hip):
    the step's noise draws and the folded output-attention weights live in
    persistent tensors that ONE launch ahead of the step fills -- together with
    the batch hand-over -- instead of a launch each inside it.

    Inside ``with step_prologue(pro):`` the first ``uniform`` / ``seed_fold``
    call allocates its persistent outputs and launches as usual; from then on
    ``pro.launch(...)`` refreshes them and the calls return the buffers without
    launching (until they have been consumed once: a second forward without a
    new ``launch`` computes its own)."""

    def __init__(self):
        self.noise = self.noise_state = None
        self.fold_inputs = self.fold_outs = self.fold_dims = None
        # the encoder's image layer: (image, weights, biases, strides) it was
        # last computed from, its persistent outputs (act, wfs, wds)
        self.first_inputs = self.first_outs = None
        # ... and, for the bf16-resident encoder (csrc/conv_bf16.hip), the bf16
        # tensors the launch writes instead / as well: (act_h, wfh, wdh)
        self.first_half = None
        self.noise_fresh = self.fold_fresh = self.first_fresh = False
        # the folding products ride in the encoder's second conv launch once
        # that launch has shown it can carry them (``_conv_stack_fwd``): the
        # prologue then leaves them out (they were its longest part)
        self.fold_rides_conv = False
        # False: that launch cannot -- or (SCAE_FOLD_RIDES=prologue) shall not -- carry them
        self.fold_conv_ok = False if _FOLD_RIDES == "prologue" else None

    def _first_desc(self, image):
        """scae_first_layer_desc of the registered image layer over ``image``."""
        _, weights, biases, strides = self.first_inputs
        act, wfs, wds = self.first_outs
        d = _lib.FirstLayerDesc()
        d.img, d.w, d.bias, d.out = image.data_ptr(), weights[0].data_ptr(), \
            biases[0].data_ptr(), act.data_ptr()
        d.B, d.Cin, d.IH, d.IW = image.shape
        d.Cout, d.stride, d.n_layers = weights[0].shape[0], strides[0], len(wfs)
        for i, (w, wf, wd) in enumerate(zip(weights[1:], wfs, wds)):
            d.rw[i], d.rwf[i], d.rwd[i] = w.data_ptr(), wf.data_ptr(), \
                wd.data_ptr()
            d.rCout[i], d.rCin[i] = w.shape[0], w.shape[1]
        if self.first_half is not None:
            act_h, wfh, wdh = self.first_half
            d.out_h = act_h.data_ptr()
            for i, (a, b) in enumerate(zip(wfh, wdh)):
                d.rwfh[i], d.rwdh[i] = a.data_ptr(), b.data_ptr()
        return d

    def launch(self, dst_image=None, src_image=None, dst_label=None,
               src_label=None, stream_ref=None):
        """One launch: batch hand-over (when the four tensors are given) +
        whatever of noise / folding this prologue has buffers for."""
        stage = dst_image is not None
        n_noise = 0 if self.noise is None else self.noise.numel()
        # the image layer runs here when the batch it will be asked for is the
        # one at hand: the hand-over's destination, read at its source
        first = self.first_outs is not None and (
            not stage or (dst_image.data_ptr() == self.first_inputs[0].data_ptr()
                          and dst_image.shape == self.first_inputs[0].shape))
        if not (stage or n_noise or self.fold_outs is not None or first):
            return
        ref = dst_image if stage else (
            self.noise if n_noise else (self.fold_outs[0] if self.fold_outs
                                        is not None else self.first_outs[0]))
        _need_hip(ref)
        P = ctypes.c_void_p
        desc = fdesc = None
        fold_here = self.fold_outs is not None and not self.fold_rides_conv
        if fold_here:
            desc = ctypes.byref(_fold_desc(self.fold_inputs, self.fold_outs,
                                           *self.fold_dims))
        if first:
            fdesc = ctypes.byref(self._first_desc(
                src_image if stage else self.first_inputs[0]))
        _lib.call("scae_step_prologue_first_f32",
                  _p(dst_image), _p(src_image),
                  dst_image.numel() if stage else 0,
                  P(dst_label.data_ptr()) if stage else None,
                  P(src_label.data_ptr()) if stage else None,
                  dst_label.numel() if stage else 0,
                  _p(self.noise), n_noise, _p(self.noise_state), desc, fdesc,
                  _stream(ref))
        self.noise_fresh = self.noise is not None
        self.fold_fresh = fold_here
        self.first_fresh = first


def step_prologue(pro):
    """``with step_prologue(pro):`` -- see ``StepPrologue`` (``pro`` None:
    no-op); a field of the current step plan."""
    return _plan().with_prologue(pro)


def reset_noise():
    """Restart the device noise generators from ``torch.initial_seed()`` (call
    after ``torch.manual_seed`` to replay a noise sequence; not inside a graph
    capture).  Every step plan has its own generators -- a ``TrainStep`` owns
    one -- and ALL live plans restart, so ``torch.manual_seed(s);
    ops.reset_noise()`` replays the sequence of an already built step too.
    (A step whose graph is captured keeps drawing from its device-resident
    generator state: that state is re-seeded in place, see ``uniform``.)"""
    for plan in step_plan.live_plans():
        for key, (seed, state) in list(plan.noise.items()):
            # in place: captured graphs and prologues hold this tensor's address
            fresh = int(torch.initial_seed()) & ((1 << 63) - 1)
            state.copy_(torch.tensor([fresh, 0, 0], dtype=torch.int64))
            plan.noise[key] = (fresh, state)


def uniform(n, ref):
    """n floats ~ U[0,1) on ``ref``'s device from the device-resident Philox
    generator (csrc/noise.hip).  Seeded from ``torch.initial_seed()``; the
    generator state is per (device, stream) and restarts when that seed
    changes (``torch.manual_seed``) outside a graph capture."""
    _need_hip(ref)
    seed = int(torch.initial_seed()) & ((1 << 63) - 1)
    key = (ref.device, torch.cuda.current_stream(ref.device).cuda_stream)
    plan = _plan()
    state = plan.noise.get(key)
    if state is None or (state[0] != seed and
                         not torch.cuda.is_current_stream_capturing()):
        state = plan.noise[key] = (seed, torch.tensor(
            [seed, 0, 0], dtype=torch.int64).to(ref.device))
    pro = plan.prologue
    if pro is not None:
        if pro.noise is not None and pro.noise.numel() == int(n) \
                and pro.noise_state is state[1] and pro.noise_fresh:
            pro.noise_fresh = False     # drawn by the step's prologue launch
            return pro.noise
        # (re-)establish the persistent buffer; this draw is launched here
        pro.noise = torch.empty(int(n), device=ref.device, dtype=torch.float32)
        pro.noise_state, pro.noise_fresh = state[1], False
        out = pro.noise
    else:
        out = torch.empty(int(n), device=ref.device, dtype=torch.float32)
    _lib.call("scae_uniform_f32", _p(out), int(n), _p(state[1]), _stream(ref))
    return out


def _sum_rows(partial, shapes, starts=None, period=0, outs=None, transpose=0,
              defer=False):
    """Column sums of ``partial`` (rows, cols) scattered into fresh contiguous
    tensors of the given shapes; consecutive column ranges unless ``starts``
    gives each one's first column.  With ``period`` the columns form blocks
    of that width and output i gathers columns [starts[i], starts[i] + w_i) of
    every block (w_i = numel / number of blocks).  With ``transpose`` = W each
    output's window is an (n x W) matrix that is written transposed (W x n).
    One launch."""
    return _sum_rows_multi([dict(partial=partial, shapes=shapes, starts=starts,
                                 period=period, outs=outs, transpose=transpose,
                                 defer=defer)])[0]


_SUMS_PER_LAUNCH = 16


def deferred_param_sums():
    """Inside this context the column-sum jobs marked ``defer`` -- those whose
    outputs are parameter gradients, which nothing but the optimiser / the
    gradient all-reduce reads -- are queued (on the current step plan) instead
    of launched; ``flush_param_sums()`` (and leaving the context) runs
    everything queued in one launch per 16 jobs.  A training step's backward
    then ends in ONE column-sum launch instead of one per op."""
    return _plan().deferring()


def flush_param_sums():
    _plan().flush_sums()


def _sum_job_array(chunk):
    """struct scae_sum_job[len(chunk)] of column-sum units (<= 16)."""
    arr = (_lib.SumJob * len(chunk))()
    for a, (partial, rows, cols, segp, n, _keep) in zip(arr, chunk):
        a.src, a.rows, a.cols, a.segments, a.n_segments = \
            partial.data_ptr(), rows, cols, segp, n
    # (the job array points into the segment arrays: whoever keeps it -- a recorded
    # launch list, _lib.recorder -- keeps them and the partial matrices too)
    arr._keep = [(unit[0], unit[-1]) for unit in chunk]
    return arr


def _launch_sum_units(units):
    for k in range(0, len(units), _SUMS_PER_LAUNCH):
        chunk = units[k:k + _SUMS_PER_LAUNCH]
        _lib.call("scae_sum_rows_multi_f32", _sum_job_array(chunk), len(chunk),
                  _stream(chunk[0][0]))


def _sum_rows_multi(jobs):
    """Several ``_sum_rows`` jobs (dicts of its arguments, different partial
    matrices) in as few launches as possible (16 jobs of <= 8 outputs each per
    launch); returns the list of output lists.  Jobs with ``defer=True`` wait
    for ``flush_param_sums`` while ``deferred_param_sums`` is active (the
    queue keeps the partial matrix alive but no reference to the outputs:
    autograd only adopts a gradient tensor nobody else holds) -- but only when
    EVERY output is a slot view this backward took itself (``_grad_out``): a
    fresh buffer (the slot was already taken: shared parameter, second use in
    one backward) is accumulated by autograd as soon as the node returns, so
    its sum must have been launched by then -- and ``_grad_out`` flushes the
    queue before it hands out such a buffer, so the slot's own sum has run
    too.  (Not covered: a parameter that ALSO feeds an op outside this
    package; the models of this package have none.)"""
    units, later, results = [], [], []
    plan = _plan()
    for job in jobs:
        partial, shapes = job["partial"], job["shapes"]
        starts, period = job.get("starts"), job.get("period", 0)
        given, transpose = job.get("outs"), job.get("transpose", 0)
        rows, cols = partial.shape
        assert partial.is_contiguous()
        nblk = cols // period if period else 1
        outs, segs = [], (_lib.SumSegment * len(shapes))()
        pos = 0
        for i, shape in enumerate(shapes):
            o = given[i] if given is not None and given[i] is not None else \
                torch.empty(shape, device=partial.device, dtype=partial.dtype)
            if starts is not None:
                pos = starts[i]
            width = o.numel() // nblk
            segs[i].dst, segs[i].begin, segs[i].end = o.data_ptr(), pos, \
                pos + width
            segs[i].period = -transpose if transpose else period
            pos += width
            outs.append(o)
        results.append(outs)
        dst = later if (plan.deferred is not None and job.get("defer")
                        and all(_in_slot(o) for o in outs)) else units
        for k in range(0, len(shapes), 8):
            dst.append((partial, rows, cols, ctypes.cast(
                ctypes.byref(segs, k * ctypes.sizeof(_lib.SumSegment)),
                ctypes.POINTER(_lib.SumSegment)), min(8, len(shapes) - k), segs))
    if later:
        plan.deferred.extend(later)
    if units:
        _launch_sum_units(units)
    return results


# ----------------------------------------------------------------------------
# K5 geometric_transform (cv_ops.py:20-76)
# ----------------------------------------------------------------------------
class _GeometricTransform(torch.autograd.Function):
    @_fwd
    def forward(ctx, pose, similarity, nonlinear, as_matrix):
        _need_hip(pose)
        pose = pose.contiguous()
        n = pose.numel() // 6
        out = torch.empty(*pose.shape[:-1], 9 if as_matrix else 6,
                          device=pose.device, dtype=pose.dtype)
        if n:
            _lib.call("scae_geometric_transform_fwd_f32", _p(pose), _p(out), n,
                      int(similarity), int(nonlinear), int(as_matrix),
                      _stream(pose))
        ctx.save_for_backward(pose)
        ctx.flags = (int(similarity), int(nonlinear), int(as_matrix))
        if as_matrix:
            out = out.view(*pose.shape[:-1], 3, 3)
        return out

    @_bwd
    def backward(ctx, gout):
        (pose,) = ctx.saved_tensors
        n = pose.numel() // 6
        gpose = torch.empty_like(pose)
        if n:
            gout = gout.contiguous()
            _lib.call("scae_geometric_transform_bwd_f32", _p(pose), _p(gout),
                      _p(gpose), n, *ctx.flags, _stream(pose))
        return gpose, None, None, None


def geometric_transform(pose, similarity=False, nonlinear=True,
                        as_matrix=False):
    if pose.shape[-1] != 6:
        raise ValueError("pose tensor must have 6 entries in its last dim")
    return _GeometricTransform.apply(pose, similarity, nonlinear, as_matrix)


class _Mat3Mul(torch.autograd.Function):
    """out[..., v, :, :] = left[..., 0, :, :] @ right[..., v, :, :] -- the
    3 x 3 products of the hierarchical CapsuleLayer.forward
    (object_decoder.py:184-191)."""

    @_fwd
    def forward(ctx, left, right):
        _need_hip(left, right)
        V = right.shape[-3]
        left, right = left.contiguous(), right.contiguous()
        n_caps = right.numel() // (9 * V)
        if left.numel() != n_caps * 9:
            raise ScaeHipError("mat3_mul: one left matrix per capsule expected")
        out = torch.empty_like(right)
        _lib.call("scae_mat3_mul_fwd_f32", _p(left), _p(right), _p(out), n_caps,
                  V, _stream(right))
        ctx.save_for_backward(left, right)
        return out

    @_bwd
    def backward(ctx, gout):
        left, right = ctx.saved_tensors
        V = right.shape[-3]
        n_caps = right.numel() // (9 * V)
        gright = torch.empty_like(right)
        gleft = torch.empty_like(left) if ctx.needs_input_grad[0] else None
        _lib.call("scae_mat3_mul_bwd_f32", _p(left), _p(right),
                  _p(gout.contiguous()), _p(gleft), _p(gright), n_caps, V,
                  _stream(right))
        return gleft, gright


def mat3_mul(left, right):
    """left (..., 1, 3, 3) x right (..., V, 3, 3) -> (..., V, 3, 3)."""
    return _Mat3Mul.apply(left, right)


# ----------------------------------------------------------------------------
# K2 qkv_attention (set_transformer.py:24-47)
# ----------------------------------------------------------------------------
class _QKVAttention(torch.autograd.Function):
    """fp32: both passes on the fp32 MFMA kernels.  bf16 (q, k, v all bfloat16,
    as under ``torch.autocast``): forward on the bf16 MFMA kernel (fp32
    softmax; output bf16, probabilities kept in fp32), backward on the fp32
    kernel with upcast operands."""

    @_fwd
    def forward(ctx, q, k, v, presence):
        bf16 = q.dtype == torch.bfloat16
        if bf16:
            if not (k.dtype == v.dtype == torch.bfloat16 and q.is_cuda):
                raise ScaeHipError("bf16 attention wants q, k and v in bf16 "
                                   "on a HIP device")
            presence = None if presence is None else presence.float()
            _need_hip(presence)
        else:
            _need_hip(q, k, v, presence)
        q, k, v, presence = _c(q), _c(k), _c(v), _c(presence)
        HB, N, dk = q.shape
        M, dv = v.shape[1], v.shape[2]
        out = torch.empty(HB, N, dv, device=q.device, dtype=q.dtype)
        probs = torch.empty(HB, N, M, device=q.device, dtype=torch.float32)
        sqrt_dk = float(np.float32(np.sqrt(dk)))
        if bf16 and (N > 64 or M > 64):
            # the bf16 kernel has the 64-element tile limits of the matrix-core
            # forms: larger sets take the general fp32 kernel on upcast
            # operands (fp32 accumulate as before; output rounded to bf16)
            q32, k32, v32 = q.float(), k.float(), v.float()
            out32 = torch.empty(HB, N, dv, device=q.device, dtype=torch.float32)
            _lib.call("scae_qkv_attention_fwd_f32", _p(q32), _p(k32), _p(v32),
                      _p(presence), _p(out32), _p(probs), HB, N, M, dk, dv,
                      sqrt_dk, _stream(q))
            out.copy_(out32)
        else:
            _lib.call("scae_qkv_attention_fwd_bf16" if bf16 else
                      "scae_qkv_attention_fwd_f32", _p(q), _p(k), _p(v),
                      _p(presence), _p(out), _p(probs), HB, N, M, dk, dv,
                      sqrt_dk, _stream(q))
        ctx.save_for_backward(q, k, v, probs)
        ctx.has_presence = presence is not None
        ctx.sqrt_dk = sqrt_dk
        return out

    @_bwd
    def backward(ctx, gout):
        q, k, v, probs = ctx.saved_tensors
        dtype = q.dtype
        if dtype != torch.float32:       # bf16 forward: fp32 backward
            q, k, v = q.float(), k.float(), v.float()
        HB, N, dk = q.shape
        M, dv = v.shape[1], v.shape[2]
        gout = gout.float().contiguous()
        gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        gp = None
        if ctx.has_presence and ctx.needs_input_grad[3]:
            gp = torch.empty(HB, M, device=q.device, dtype=q.dtype)
        _lib.call("scae_qkv_attention_bwd_f32", _p(q), _p(k), _p(v), _p(probs),
                  _p(gout), _p(gq), _p(gk), _p(gv), _p(gp), HB, N, M, dk, dv,
                  ctx.sqrt_dk, _stream(q))
        if dtype != torch.float32:
            gq, gk, gv = gq.to(dtype), gk.to(dtype), gv.to(dtype)
        return gq, gk, gv, gp


def qkv_attention(queries, keys, values, presence=None):
    return _QKVAttention.apply(queries, keys, values, presence)


# ----------------------------------------------------------------------------
# K2b fused set-transformer trunk (set_transformer.py:212-219)
# ----------------------------------------------------------------------------
def _seg_arrays(segs):
    n = len(segs)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in segs])
    widths = (ctypes.c_int * n)(*[t.shape[2] for t in segs])
    rs = (ctypes.c_int * n)(*[t.stride(1) for t in segs])
    bs = (ctypes.c_int64 * n)(*[t.stride(0) for t in segs])
    return ptrs, widths, rs, bs


# ----------------------------------------------------------------------------
# Independent kernels sharing launches.  Inside a training step
# (``step_fusion(target)``, opened by train_step.TrainStep on its own step plan)
# a launch may be PARKED by the node that owns it and carried by a later node's
# launch; step_plan.RIDES is the table of these merges and the step plan the
# only holder of parked work.  The classes below are the parked launches: the
# arguments of the stand-alone launcher (``launch_alone``) which the carrier's
# merged launcher takes as well.  Same kernels, same values, fewer launches.
#
# Forward: the part decoder's likelihood of the step's own input image only
# needs the part encoder's outputs, like the object encoder's trunk -- and the
# trunk leaves three quarters of the SIMDs idle (csrc/trunk_logprob.hip).
# SCAE.forward prepares the likelihood as a RIDER of the trunk's launch;
# ``render_gmm_log_prob_sums`` later finds the result instead of launching.
# ----------------------------------------------------------------------------
def step_fusion(target):
    """``with step_fusion(image):`` -- the reconstruction target of the loss
    that will follow the forward inside the block (None: no fusion), on the
    current step plan."""
    return _plan().fusing(target)


def fusion_target():
    return _plan().target


class LogProbRider:
    """The tile-sum likelihood of ``x`` under the mixture of ``inputs``
    (DecoderInputs), prepared for the trunk's launch to carry."""

    def __init__(self, inputs, x):
        t = _prep_decoder(inputs.tensors())
        _need_hip(x)
        self.x = x.detach().contiguous()
        self.desc, (B, M, C, th, tw, H, W) = _make_desc(t, inputs.output_size)
        self._keep = t          # the descriptor holds raw pointers
        dev, dt = t[0].device, t[0].dtype
        tiles = _lib.load().scae_render_gmm_logprob_tiles(ctypes.byref(self.desc))
        self.sums = torch.empty(B, tiles, device=dev, dtype=dt)
        self.lse_post = torch.empty(B, C, H, W, device=dev, dtype=dt)
        self.lse_prior = torch.empty(B, 1 if t[1] is not None else C, H, W,
                                     device=dev, dtype=dt)
        self.key = self.key_of(inputs, x)
        self.launched = False

    @staticmethod
    def key_of(inputs, x):
        return (x.data_ptr(), tuple(x.shape), tuple(inputs.output_size)) + tuple(
            None if v is None else (v.data_ptr(), tuple(v.shape))
            for v in inputs.tensors())


def offer_log_prob_rider(inputs, x):
    """Called by SCAE.forward ahead of the object encoder: the next fused trunk
    launch carries this likelihood (if it can; else nothing happens)."""
    plan = _plan()
    plan.rider = LogProbRider(inputs, x)
    return plan.rider


def withdraw_log_prob_rider():
    """After the object encoder: a rider nobody launched is dropped."""
    plan = _plan()
    if plan.rider is not None and not plan.rider.launched:
        plan.rider = None


class _PendingK1Backward:
    """RIDES['k1_bwd']: the reconstruction likelihood's backward (K1), parked
    for the capsule likelihood's backward -- an independent
    one-workgroup-per-image kernel -- to carry."""

    def __init__(self, desc, keep, inputs, outputs, stream_ref):
        """``inputs`` / ``keep``: tensors kept alive until the launch;
        ``outputs``: the gradient buffers.  Those handed to autograd are kept
        as ADDRESSES only: a leaf's AccumulateGrad adopts a gradient it holds
        the sole reference to and clones -- here: copies an unfilled buffer --
        otherwise; autograd itself keeps them alive until their consumers
        run, which is after this launch (every plan-aware node launches a
        parked K1 before it starts, step_plan.RIDES)."""
        self.desc, self.keep, self.ref = desc, (keep, inputs), stream_ref
        self.ptrs = [ctypes.c_void_p(t.data_ptr()) if t is not None else None
                     for t in inputs] + \
                    [None if t is None else ctypes.c_void_p(
                        t if isinstance(t, int) else t.data_ptr())
                     for t in outputs]
        self.keep_out = [t for t in outputs if not isinstance(t, int)]

    def launch_alone(self):
        _lib.call("scae_render_gmm_sums_bwd_f32", ctypes.byref(self.desc),
                  *self.ptrs, _stream(self.ref))


class _PendingWeightGemms:
    """RIDES['wgrads']: the weight-gradient GEMMs of the capsule MLPs.
    Nothing reads a weight gradient before the optimiser does, so their launch
    waits for a later backward node with CUs to spare -- the output
    attention's (one workgroup per set), which like them depends on the MLPs'
    data-gradient chain only.  ``descs`` holds device addresses; ``keep`` the
    INPUT tensors (the outputs are the gradients handed to autograd, see
    ``_PendingK1Backward``)."""

    def __init__(self, descs, n, keep, stream_ref):
        self.descs, self.n, self.keep, self.ref = descs, n, keep, stream_ref

    def launch_alone(self):
        _lib.call("scae_gemm_multi_f32", self.descs, self.n, _stream(self.ref))


class _PendingTcBackward:
    """RIDES['tc_bwd']: the template colour MLP's backward, parked for the
    part-capsule head's backward launch (same workgroup decomposition; the
    head reads its ``g_feature``, at ``gf_ptr``).  ``keep``: inputs only."""

    def __init__(self, ptrs, bm, dims, keep, gf_ptr, stream_ref):
        self.ptrs, self.bm, self.dims = ptrs, bm, dims
        self.keep, self.gf_ptr, self.ref = keep, gf_ptr, stream_ref

    def launch_alone(self):
        _lib.call("scae_template_color_bwd_f32", *self.ptrs, *self.bm,
                  *self.dims, _stream(self.ref))


class _PendingReduce:
    """RIDES['reduce']: the partial-row reduction of the output attention's
    backward (its outputs feed the folding products' backward only): rides in
    the part encoder's conv backward, one launch before
    ``_PendingFoldBackward``."""

    def __init__(self, args, keep, stream_ref):
        self.args, self.keep, self.ref = args, keep, stream_ref

    def launch_alone(self):
        _lib.call("scae_seed_attention_mfma_reduce_f32", *self.args,
                  _stream(self.ref))


class _PendingFoldBackward:
    """RIDES['fold_bwd']: the backward of the output attention's folding
    products.  It writes parameter gradients only, and waits for the part
    encoder's first conv backward launch (``_conv_stack_bwd``) -- 256-thread
    workgroups and tens of microseconds of matrix tiles to hide its dependent
    chain behind."""

    def __init__(self, desc, grads, keep, stream_ref):
        self.desc, self.grads, self.keep, self.ref = desc, grads, keep, stream_ref

    def launch_alone(self):
        _lib.call("scae_seed_fold_bwd_f32", ctypes.byref(self.desc),
                  ctypes.byref(self.grads), _stream(self.ref))


def flush_pending_backward():
    """Launch what is parked and nobody carried: the K1 backward (e.g. no
    gradient reached the capsule likelihood), the capsule MLPs' weight
    gradients (e.g. an object encoder without the matrix-core attention), ..."""
    _plan().flush_scope("deferring")


class _SetEncoder(torch.autograd.Function):
    @_fwd
    def forward(ctx, presence, packed, meta, *segs):
        D, Dout, L, layer_norm = meta
        _need_hip(presence, packed, *segs)
        segs = [t if t.stride(2) == 1 else t.contiguous() for t in segs]
        presence, packed = _c(presence), packed.contiguous()
        B, N = segs[0].shape[:2]
        Din = sum(t.shape[2] for t in segs)
        lib = _lib.load()
        assert packed.numel() == lib.scae_set_encoder_param_count(
            D, Din, Dout, L, int(layer_norm))
        z = torch.empty(B, N, Dout if Dout else D, device=packed.device,
                        dtype=packed.dtype)
        hsave = torch.empty(B, L + 1, N, D, device=packed.device,
                            dtype=packed.dtype)
        ptrs, widths, rs, bs = _seg_arrays(segs)
        # configs[2]'s precision (inside ``mfma_bf16()``): bf16 operands for the
        # attention products of every block, forward and backward
        ctx.bf16 = bool(_bf16() and lib.scae_set_encoder_bf16_supported(
            N, D, Din, Dout, L, int(layer_norm)))
        rider = ctx.plan.rider
        if rider is not None and not rider.launched \
                and rider.x.device == packed.device:
            # the part decoder's likelihood rides in this launch (where the
            # launcher finds that it pays; two launches otherwise)
            _lib.call("scae_set_encoder_fwd_logprob_bf16" if ctx.bf16 else
                      "scae_set_encoder_fwd_logprob_f32", len(segs), ptrs,
                      widths, rs, bs, _p(presence), _p(packed), _p(z),
                      _p(hsave), B, N, D, Din, Dout, L, int(layer_norm),
                      ctypes.byref(rider.desc), _p(rider.x), _p(rider.sums),
                      _p(rider.lse_post), _p(rider.lse_prior), _stream(packed))
            rider.launched = True
        else:
            _lib.call("scae_set_encoder_fwd_bf16" if ctx.bf16 else
                      "scae_set_encoder_fwd_f32", len(segs), ptrs, widths, rs,
                      bs, _p(presence), _p(packed), _p(z), _p(hsave), B, N, D,
                      Din, Dout, L, int(layer_norm), _stream(packed))
        ctx.save_for_backward(packed, hsave, *segs,
                              *([presence] if presence is not None else []))
        ctx.has_presence = presence is not None
        ctx.nseg = len(segs)
        ctx.dims = (B, N, D, Din, Dout, L, int(layer_norm))
        ctx.slot = _slot(packed, ctx)
        return z

    @_bwd
    def backward(ctx, gz):
        packed, hsave = ctx.saved_tensors[:2]
        segs = list(ctx.saved_tensors[2:2 + ctx.nseg])
        presence = ctx.saved_tensors[2 + ctx.nseg] if ctx.has_presence else None
        B, N, D, Din, Dout, L, ln = ctx.dims
        gz = gz.contiguous()
        grid = _lib.load().scae_set_encoder_grid(B)
        partial = torch.empty(grid, packed.numel(), device=packed.device,
                              dtype=packed.dtype)
        gsegs = [torch.empty(B, N, t.shape[2], device=t.device, dtype=t.dtype)
                 if ctx.needs_input_grad[3 + i] else None
                 for i, t in enumerate(segs)]
        ptrs, widths, rs, bs = _seg_arrays(segs)
        gptrs = (ctypes.c_void_p * len(segs))(
            *[None if g is None else g.data_ptr() for g in gsegs])
        _lib.call("scae_set_encoder_bwd_bf16" if ctx.bf16 else
                  "scae_set_encoder_bwd_f32", len(segs), ptrs, widths, rs, bs,
                  gptrs, _p(presence), _p(packed), _p(hsave), _p(gz),
                  _p(partial), B, N, D, Din, Dout, L, ln, _stream(packed))
        taken_before = ctx.slot is not None and ctx.slot.taken
        if ctx.slot is not None and not taken_before and \
                any(sl.taken for sl in getattr(ctx.slot, "parts", ())):
            # a part's slot already holds another op's gradient (a trunk
            # parameter shared with a second slot-aware op): writing the packed
            # block would overwrite it -- leave the slots to autograd's sum
            ctx.slot.taken = taken_before = True
        gpacked = _grad_out(ctx.slot, packed)
        if ctx.slot is not None and not taken_before:
            for sl in getattr(ctx.slot, "parts", ()):
                sl.taken = True   # the parts' slots are written through it
        # (in its slot the packed block is parameters only: its sum can wait)
        return (None, _sum_rows(partial, [packed.shape], outs=[gpacked],
                                defer=True)[0], None, *gsegs)


def set_encoder(segments, presence, packed_params, dim_hidden, dim_out,
                n_layers, layer_norm):
    """fc1 -> n_layers x SAB -> fc2 on a set given as column segments
    (each (B, N, w_i)); returns (B, N, dim_out)."""
    if presence is not None and presence.requires_grad:
        raise ScaeHipError("the fused trunk treats presence as a constant")
    return _SetEncoder.apply(presence, packed_params,
                             (dim_hidden, dim_out, n_layers, bool(layer_norm)),
                             *segments)


class _PackParams(torch.autograd.Function):
    """Concatenation of flattened parameters.  When they already lie back to
    back in memory in this order (FlatParameters places a module's
    ``_flat_param_groups`` that way) the result aliases them -- no copy -- and
    the gradient is handed back as views of one buffer."""

    @_fwd
    def forward(ctx, *parts):
        ctx.shapes = [tuple(p.shape) for p in parts]
        adjacent = all(p.is_contiguous() for p in parts) and all(
            a.data_ptr() + 4 * a.numel() == b.data_ptr()
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            for a, b in zip(parts, parts[1:]))
        if adjacent:
            total = sum(p.numel() for p in parts)
            return torch.empty(0, device=parts[0].device, dtype=parts[0].dtype) \
                .set_(parts[0].untyped_storage(), parts[0].storage_offset(),
                      (total,), (1,))
        return torch.cat([p.reshape(-1) for p in parts])

    @_bwd
    def backward(ctx, g):
        out, off = [], 0
        for shape in ctx.shapes:
            n = int(np.prod(shape))
            out.append(g[off:off + n].view(shape))
            off += n
        return tuple(out)


def pack_params(parts):
    """One flat buffer of ``parts`` (see ``_PackParams``).  When the parts also
    own back-to-back slots of the flat gradient buffer, the result carries a
    slot covering all of them, so the consumer's backward writes the packed
    gradient straight into place."""
    parts = list(parts)
    packed = _PackParams.apply(*parts)
    slots = [_slot(p) for p in parts]
    if packed.data_ptr() == parts[0].data_ptr() and all(
            sl is not None and not sl.taken for sl in slots) and all(
            a.flat_grad is b.flat_grad and a.offset + a.numel == b.offset
            for a, b in zip(slots, slots[1:])):
        packed._scae_grad_slot = type(slots[0])(
            slots[0].flat_grad, slots[0].offset, (packed.numel(),))
        packed._scae_grad_slot.parts = slots
    return packed


# ----------------------------------------------------------------------------
# K2c output attention with folded projections (set_transformer.py:218-223)
# ----------------------------------------------------------------------------
def seed_attention_supported(N, O, D, C):
    lib = _lib.load()
    return bool(lib.scae_seed_attention_mfma_supported(N, O, D, C)
                or lib.scae_seed_attention_supported(N, O, D, C))


class _SeedAttention(torch.autograd.Function):
    @_fwd
    def forward(ctx, h, q, wk, bk, wv, bv, presence, from_fold=False):
        _need_hip(h, q, wk, bk, wv, bv, presence)
        ctx.from_fold = bool(from_fold)
        h, q, wk, bk, wv, bv, presence = (_c(t) for t in (h, q, wk, bk, wv, bv,
                                                          presence))
        B, N, D = h.shape
        O, C = q.shape
        out = torch.empty(B, O, C, device=h.device, dtype=h.dtype)
        ctx.mfma = bool(_lib.load().scae_seed_attention_mfma_supported(N, O, D,
                                                                       C))
        if ctx.mfma:     # wave-per-tile kernels on the matrix cores
            _lib.call(_prec("scae_seed_attention_mfma_fwd_f32"), _p(h), _p(q), _p(wk),
                      _p(wv), _p(bv), _p(presence), _p(out), B, N, O, C,
                      _stream(h))
        else:
            _lib.call("scae_seed_attention_fwd_f32", _p(h), _p(q), _p(wk),
                      _p(bk), _p(wv), _p(bv), _p(presence), _p(out), None, B, N,
                      O, D, C, _stream(h))
        ctx.save_for_backward(h, q, wk, bk, wv, bv,
                              *([presence] if presence is not None else []))
        ctx.has_presence = presence is not None
        return out

    @_bwd
    def backward(ctx, gout):
        h, q, wk, bk, wv, bv = ctx.saved_tensors[:6]
        presence = ctx.saved_tensors[6] if ctx.has_presence else None
        B, N, D = h.shape
        O, C = q.shape
        lib = _lib.load()
        if ctx.mfma:
            rows = lib.scae_seed_attention_mfma_rows(B)
            new = lambda *shape: torch.empty(*shape, device=h.device,  # noqa: E731
                                             dtype=h.dtype)
            partial, gh = new(rows, O * D + C * D + C), new(B, N, D)
            args = (_p(h), _p(q), _p(wk), _p(wv), _p(presence),
                    _p(gout.contiguous()), _p(gh), _p(partial), B, N, O, C)
            plan = ctx.plan
            parked, carried = plan.take("wgrads"), False
            if parked is not None:
                # the capsule MLPs' weight-gradient tiles as the tail of this launch
                rc = getattr(lib, _prec("scae_seed_attention_mfma_bwd_gemm_f32"))(
                    *args, parked.descs, parked.n, _stream(h))
                if rc == _lib.ERR_UNSUPPORTED:   # large GEMMs: on their own
                    parked.launch_alone()
                else:
                    _lib.check(rc, "scae_seed_attention_mfma_bwd_gemm")
                    carried = True
            if not carried:
                _lib.call(_prec("scae_seed_attention_mfma_bwd_f32"), *args,
                          _stream(h))
            # column sums of the partials, expanded to the operands' gradients
            gq, gwk, gbk, gwv, gbv = new(O, C), new(C, D), new(C), new(C, D), \
                new(C)
            red = (_p(partial), rows, _p(q), _p(wk), _p(gq), _p(gwk), _p(gbk),
                   _p(gwv), _p(gbv), O, C)
            if ctx.from_fold and plan.parking and not plan.bf16:
                # only the folding products' backward reads these: the launch
                # waits for a carrier (RIDES['reduce']: launched ahead of
                # ``_SeedFold.backward``, which parks itself behind it)
                plan.park("reduce", _PendingReduce(red, (partial, q, wk), h))
            else:
                _lib.call("scae_seed_attention_mfma_reduce_f32", *red,
                          _stream(h))
            return gh, gq, gwk, gbk, gwv, gbv, None, None
        grid, S = lib.scae_seed_attention_grid(B, O), \
            lib.scae_seed_attention_splits(B, O)
        npar = O * C + 2 * C * D + 2 * C
        partial = torch.empty(grid, npar, device=h.device, dtype=h.dtype)
        gh = torch.empty(S, B, N, D, device=h.device, dtype=h.dtype)
        _lib.call("scae_seed_attention_bwd_f32", _p(h), _p(q), _p(wk), _p(bk),
                  _p(wv), _p(bv), _p(presence), _p(gout.contiguous()), _p(gh),
                  _p(partial), B, N, O, D, C, _stream(h))
        # one h-gradient slab per query group
        jobs = [dict(partial=partial,
                     shapes=[(O, C), (C, D), (C,), (C, D), (C,)])]
        if S > 1:
            jobs.append(dict(partial=gh.view(S, -1), shapes=[(B, N, D)]))
        res = _sum_rows_multi(jobs)
        gq, gwk, gbk, gwv, gbv = res[0]
        gh = gh[0] if S == 1 else res[1][0]
        return gh, gq, gwk, gbk, gwv, gbv, None, None


def seed_attention(h, q, wk, bk, wv, bv, presence=None):
    """out (B,O,C) = softmax((q K'^T - (1-presence) 1e32)/sqrt(C)) V' with
    K' = h wk^T + bk, V' = h wv^T + bv."""
    if presence is not None and presence.requires_grad:
        raise ScaeHipError("seed_attention treats presence as a constant")
    # (do the five folded operands all come from ONE seed_fold call?  Then their
    # gradients feed that node's backward only, and inside a fused step the
    # reduction that produces them may wait for a carrier together with it)
    fns = {t.grad_fn for t in (q, wk, bk, wv, bv)}
    fn = next(iter(fns)) if len(fns) == 1 else None
    from_fold = fn is not None and type(fn).__name__ == "_SeedFoldBackward"
    return _SeedAttention.apply(h, q, wk, bk, wv, bv, presence, from_fold)


# ----------------------------------------------------------------------------
# K2d weight folding for the output attention (set_transformer.py:218-223)
# ----------------------------------------------------------------------------
def seed_fold_supported(O, C, D):
    return bool(_lib.load().scae_seed_fold_supported(O, C, D))


_FOLD_INPUTS = ("seeds", "wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "w2",
                "b2")


def _fold_desc(inputs, outputs, O, C, D):
    d = _lib.SeedFoldDesc()
    for name, t in zip(_FOLD_INPUTS, inputs):
        setattr(d, name, t.data_ptr())
    for name, t in zip(("q", "wkf", "bkf", "wvf", "bvf", "wv2e", "wowv"),
                       outputs):
        setattr(d, name, t.data_ptr())
    d.O, d.C, d.D = O, C, D
    return d


class _SeedFold(torch.autograd.Function):
    """(seeds, Wq, bq, Wk, bk, Wv, bv, Wo, bo, W2, b2) -> (q, wkf, bkf, wvf,
    bvf) consumed by seed_attention."""

    @_fwd
    def forward(ctx, *inputs):
        _need_hip(*inputs)
        ctx.slots = [_slot(t, ctx) for t in inputs]
        inputs = tuple(t.contiguous() for t in inputs)
        seeds, w2 = inputs[0], inputs[9]
        O, C = seeds.shape[-2:]           # (O, C) or the parameter's (1, O, C)
        D = w2.shape[1]
        new = lambda *shape: torch.empty(*shape, device=seeds.device,
                                         dtype=seeds.dtype)
        pro, launch = ctx.plan.prologue, True
        if pro is not None:
            ptrs = [t.data_ptr() for t in inputs]
            if pro.fold_outs is not None and pro.fold_dims == (O, C, D) and \
                    [t.data_ptr() for t in pro.fold_inputs] == ptrs:
                # persistent outputs: filled by the step's prologue launch
                # (fresh), else by the launch below
                outs, launch = pro.fold_outs, not pro.fold_fresh
            else:
                outs = (new(O, C), new(C, D), new(C), new(C, D), new(C),
                        new(C, D + 1), new(C, C))
                pro.fold_inputs, pro.fold_outs, pro.fold_dims = \
                    inputs, outs, (O, C, D)
            pro.fold_fresh = False
        else:
            outs = (new(O, C), new(C, D), new(C), new(C, D), new(C),
                    new(C, D + 1), new(C, C))
        if launch:
            desc = _fold_desc(inputs, outs, O, C, D)
            _lib.call("scae_seed_fold_fwd_f32", ctypes.byref(desc),
                      _stream(seeds))
        ctx.save_for_backward(*inputs, *outs)
        return tuple(t.view_as(t) for t in outs[:5]) if pro is not None \
            else outs[:5]

    @_bwd
    def backward(ctx, g_q, g_wkf, g_bkf, g_wvf, g_bvf):
        inputs, outs = ctx.saved_tensors[:11], ctx.saved_tensors[11:]
        seeds, w2 = inputs[0], inputs[9]
        O, C = seeds.shape[-2:]
        D = w2.shape[1]
        zeros = lambda ref: torch.zeros_like(ref)
        incoming = [g if g is not None else zeros(o) for g, o in
                    zip((g_q, g_wkf, g_bkf, g_wvf, g_bvf), outs)]
        incoming = [g.contiguous() for g in incoming]
        grads = [_grad_out(sl, t) for sl, t in zip(ctx.slots, inputs)]
        desc = _fold_desc(inputs, outs, O, C, D)
        g = _lib.SeedFoldGrads()
        for name, t in zip(("g_q", "g_wkf", "g_bkf", "g_wvf", "g_bvf"),
                           incoming):
            setattr(g, name, t.data_ptr())
        for name, t in zip(_FOLD_INPUTS, grads):
            setattr(g, "d_" + name, t.data_ptr())
        # (bf16 mode has no carrier: the conv backward takes another tile form,
        # and a launch that only waits runs later, on colder caches)
        plan = ctx.plan
        reduce = plan.take("reduce")   # (this launch reads what that one writes)
        if plan.parking and not plan.bf16 and \
                all(_in_slot(t) for t in grads):
            # parameter gradients only: the launch waits for a carrier, behind
            # the reduction it reads (table order)
            if reduce is not None:
                plan.park("reduce", reduce)
            plan.park("fold_bwd", _PendingFoldBackward(
                desc, g, (inputs, outs, incoming), seeds))
        else:
            if reduce is not None:
                reduce.launch_alone()
            _lib.call("scae_seed_fold_bwd_f32", ctypes.byref(desc),
                      ctypes.byref(g), _stream(seeds))
        return tuple(grads)


def seed_fold(seeds, wq, bq, wk, bk, wv, bv, wo, bo, w2, b2):
    """Parameter-only products feeding seed_attention:
    q = seeds wq^T + bq, wkf = wk w2, bkf = wk b2 + bk, wvf = wo wv w2,
    bvf = wo (wv b2 + bv) + bo."""
    return _SeedFold.apply(seeds, wq, bq, wk, bk, wv, bv, wo, bo, w2, b2)


# ----------------------------------------------------------------------------
# K8 CNN encoder: 3x3 valid conv + ReLU stack (part_encoder.py:26-44)
# ----------------------------------------------------------------------------
def conv_stack_supported(in_channels, out_channels, kernel_sizes, strides):
    """Shapes the implicit-GEMM kernels are built for."""
    chans = [in_channels] + list(out_channels)
    return (1 <= len(out_channels) <= 9 and in_channels <= 4
            and all(k == 3 for k in kernel_sizes)
            and all(s in (1, 2) for s in strides)
            and all(c % 64 == 0 for c in out_channels)
            and all(c <= 1024 for c in chans))


# K8r takes a layer when the ring-pipelined tiles would leave the chip short of work:
# fewer than ~3 of their 32 x 64 tiles per CU (measured at B = 128 / 1024)
_CONV_RESIDENT_MAX_TILES = 768


def _conv_resident(B, ih, iw, ci, co, s):
    oh, ow = (ih - 3) // s + 1, (iw - 3) // s + 1
    tiles = -(-B * oh * ow // 32) * (co // 64)
    return tiles < _CONV_RESIDENT_MAX_TILES and bool(
        _lib.load().scae_conv3x3_fwd_res_supported(B, ih, iw, ci, co, s))


def _conv_stack_fwd(image, strides, weights, biases, post_bias=None):
    """-> (acts, wds, x_post): the NHWC ReLU outputs of every layer, the
    re-laid-out filters the data-gradient kernels read, and (with
    ``post_bias`` (C,OH,OW), needs >= 2 layers) acts[-1] + post_bias."""
    L = len(strides)
    B, C0, H, W = image.shape
    dev, st = image.device, _stream(image)
    new = lambda *shape: torch.empty(*shape, device=dev, dtype=image.dtype)
    c1, s = weights[0].shape[0], strides[0]
    oh, ow = (H - 3) // s + 1, (W - 3) // s + 1
    pro, launch = _plan().prologue, True
    key = lambda ts: [(t.data_ptr(), tuple(t.shape)) for t in ts]
    # configs[2]'s precision with bf16-resident operands: the image layer then
    # writes bf16 (and bf16 copies of the re-laid-out filters)
    resident = _conv_bf16_resident(B, oh, ow, weights, strides)
    halves = None
    if pro is not None and pro.first_outs is not None and \
            key([image, *weights, *biases]) == key(
                [pro.first_inputs[0], *pro.first_inputs[1],
                 *pro.first_inputs[2]]) and \
            tuple(strides) == pro.first_inputs[3] and \
            (pro.first_half is not None) == resident:
        # persistent outputs: filled by the step's prologue launch (fresh),
        # else by the launch below
        act, wfs, wds = pro.first_outs
        halves = pro.first_half
        launch = not pro.first_fresh
    else:
        act = new(B, oh, ow, c1) if not resident else new(1)
        wds, wfs = [], []
        for l in range(1, L):
            co, ci = weights[l].shape[0], weights[l].shape[1]
            # (Cout,9,Cin) + its fragment-major copy, three bf16 planes: 2.5 x
            # (scae_conv3x3_wf_floats; include/scae_hip.h, K8)
            wfs.append(new(3, co, 9, ci))
            wds.append(new(ci, 9, co))
        if resident:
            half = lambda *shape: torch.empty(*shape, device=dev,
                                              dtype=torch.bfloat16)
            halves = (half(B, oh, ow, c1),
                      [half(w.shape[0], 9, w.shape[1]) for w in weights[1:]],
                      [half(w.shape[1], 9, w.shape[0]) for w in weights[1:]])
        if pro is not None:
            pro.first_inputs = (image, list(weights), list(biases),
                                tuple(strides))
            pro.first_outs = (act, wfs, wds)
            pro.first_half = halves
    if pro is not None:
        pro.first_fresh = False
    acts = [act]
    # the image layer; the (parameter-only) filter re-layouts of the other
    # layers ride in the same launch
    n = L - 1
    if launch:
        arr = lambda ts: (ctypes.c_void_p * max(n, 1))(
            *[t.data_ptr() for t in ts])
        ints = lambda v: (ctypes.c_int * max(n, 1))(*v)
        if resident:
            _lib.call("scae_conv3x3_first_fwd_relayout_bf16", _p(image),
                      _p(weights[0]), _p(biases[0]), _p(halves[0]), B, C0, H, W,
                      c1, s, n, arr(weights[1:]), arr(wfs), arr(wds),
                      arr(halves[1]), arr(halves[2]),
                      ints([w.shape[0] for w in weights[1:]]),
                      ints([w.shape[1] for w in weights[1:]]), st)
        else:
            _lib.call("scae_conv3x3_first_fwd_relayout_f32", _p(image),
                      _p(weights[0]), _p(biases[0]), _p(act), B, C0, H, W, c1, s,
                      n, arr(weights[1:]), arr(wfs), arr(wds),
                      ints([w.shape[0] for w in weights[1:]]),
                      ints([w.shape[1] for w in weights[1:]]), st)
    x_post = None
    if resident:
        return _conv_stack_fwd_bf16r(halves, weights, biases, strides,
                                     post_bias, image.dtype)
    for l in range(1, L):
        w, s = weights[l], strides[l]
        co, ci = w.shape[0], w.shape[1]
        ih, iw = act.shape[1], act.shape[2]
        out = new(B, (ih - 3) // s + 1, (iw - 3) // s + 1, co)
        if l == L - 1 and post_bias is not None:
            x_post = torch.empty_like(out)
        conv = (_p(act), _p(wfs[l - 1]), _p(biases[l]), _p(out),
                _p(post_bias if x_post is not None else None), _p(x_post), B,
                ih, iw, ci, co, s)
        carried = False
        if l == 1 and pro is not None and pro.fold_outs is not None:
            # a training step's folding products ride in this launch (the
            # largest early one); the prologue leaves them out from then on
            if _bf16() or pro.fold_conv_ok is False:
                pro.fold_rides_conv = False
            elif pro.fold_fresh:
                pro.fold_rides_conv = True    # (from the next prologue launch)
            else:
                rc = _lib.load().scae_conv3x3_fwd_fold_f32(
                    *conv, ctypes.byref(_fold_desc(
                        pro.fold_inputs, pro.fold_outs, *pro.fold_dims)), st)
                if rc == _lib.ERR_UNSUPPORTED:
                    pro.fold_conv_ok = pro.fold_rides_conv = False
                else:
                    _lib.check(rc, "scae_conv3x3_fwd_fold_f32")
                    carried = pro.fold_fresh = pro.fold_rides_conv = True
        if not carried:
            # small layers: the input images resident in LDS (K8r)
            if not _bf16() and _conv_resident(B, ih, iw, ci, co, s):
                _lib.call("scae_conv3x3_fwd_res_f32", _p(act),
                          _p(wfs[l - 1][1]), *conv[2:], 0, st)
            else:
                _lib.call(_prec("scae_conv3x3_fwd_f32"), *conv, st)
        acts.append(out)
        act = out
    return acts, wds, x_post


_CONV_BF16R = __import__("os").environ.get("SCAE_CONV_BF16R", "1") != "0"



def _conv_bf16_resident(B, ih, iw, weights, strides):
    """configs[2]'s precision with every GEMM operand of layers 1.. kept as
    bf16 in HBM (csrc/conv_bf16.hip): inside ``mfma_bf16()`` when every such
    layer has the kernels' shapes."""
    if not (_bf16() and _CONV_BF16R and len(strides) >= 2):
        return False
    lib = _lib.load()
    for l in range(1, len(strides)):
        co, ci, s = weights[l].shape[0], weights[l].shape[1], strides[l]
        if not lib.scae_conv3x3_bf16r_supported(B, ih, iw, ci, co, s):
            return False
        ih, iw = (ih - 3) // s + 1, (iw - 3) // s + 1
    return True


def _cvt_bf16(pairs, ref):
    """bf16 copies of fp32 tensors, up to 8 per launch: [(src, dst)]."""
    for k in range(0, len(pairs), 8):
        chunk = pairs[k:k + 8]
        n = len(chunk)
        _lib.call("scae_cvt_bf16_batch", n,
                  (ctypes.c_void_p * n)(*[a.data_ptr() for a, _ in chunk]),
                  (ctypes.c_void_p * n)(*[b.data_ptr() for _, b in chunk]),
                  (ctypes.c_int64 * n)(*[a.numel() for a, _ in chunk]),
                  _stream(ref))


def _conv_stack_fwd_bf16r(halves, weights, biases, strides, post_bias, dt):
    """Layers 1.. of the stack on the bf16-resident kernels.  ``halves``: the
    image layer's output and the re-laid-out filters as bf16 (act_h, wfh,
    wdh).  -> (acts, wds, x_post) like ``_conv_stack_fwd``, with the
    activations of layers 0 .. L-2 and the data-gradient filters as bf16
    tensors (what ``_conv_stack_bwd`` then reads); the last layer's output
    stays fp32 (the attention conv and its ReLU gate read it)."""
    act_h, wfh, wdh = halves
    L = len(strides)
    B, dev, st = act_h.shape[0], act_h.device, _stream(act_h)
    half = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.bfloat16)
    acts, x_post = [act_h], None
    for l in range(1, L):
        co, ci, s = weights[l].shape[0], weights[l].shape[1], strides[l]
        ih, iw = act_h.shape[1], act_h.shape[2]
        oh, ow = (ih - 3) // s + 1, (iw - 3) // s + 1
        out_h = half(B, oh, ow, co)
        last = l == L - 1
        out_f = torch.empty(B, oh, ow, co, device=dev, dtype=dt) \
            if last else None
        if last and post_bias is not None:
            x_post = torch.empty_like(out_f)
        _lib.call("scae_conv3x3_fwd_bf16r", _p(act_h), _p(wfh[l - 1]),
                  _p(biases[l]), _p(out_h), _p(out_f),
                  _p(post_bias if x_post is not None else None), _p(x_post),
                  B, ih, iw, ci, co, s, st)
        acts.append(out_f if last else out_h)
        act_h = out_h
    return acts, wdh, x_post


def _conv_stack_bwd(image, acts, wds, strides, wshapes, dpre, gout,
                    defer=False):
    """Weight / bias gradients of the stack from ``dpre``, the (B,OH,OW,C)
    gradient w.r.t. the last layer's pre-activation; ``gout(i)`` supplies the
    buffer of weight i (0..L-1) / bias i (L..2L-1)."""
    L = len(strides)
    B, C0, H, W = image.shape
    dev, dt, st = image.device, image.dtype, _stream(image)
    new = lambda *shape: torch.empty(*shape, device=dev, dtype=dt)
    gws, gbs = [None] * L, [None] * L
    pending = []       # (partial, gw, gb, co, ci, splits): reduced in one launch
    lib, plan = _lib.load(), _plan()
    resident = L >= 2 and acts[0].dtype == torch.bfloat16
    if resident:
        # bf16-resident operands (``_conv_stack_fwd_bf16r``): the incoming
        # gradient once to bf16, then per layer the weight-gradient partials
        # and the gated data gradient (bf16; fp32 for the image layer's kernel)
        dpre_h = torch.empty(dpre.shape, device=dev, dtype=torch.bfloat16)
        _cvt_bf16([(dpre.contiguous(), dpre_h)], image)
    for l in (range(L - 1, 0, -1) if resident else ()):
        co, ci = wshapes[l][0], wshapes[l][1]
        xin, s = acts[l - 1], strides[l]
        ih, iw, oh, ow = xin.shape[1], xin.shape[2], dpre_h.shape[1], dpre_h.shape[2]
        splits = lib.scae_conv3x3_wgrad_bf16r_splits(B, oh, ow, ci, co)
        partial = new(splits * (9 * co * ci + co))
        gw, gb = gout(l), gout(L + l)
        _lib.call("scae_conv3x3_wgrad_bf16r", _p(dpre_h), _p(xin), _p(partial),
                  B, ih, iw, ci, co, s, st)
        din_h = torch.empty(B, ih, iw, ci, device=dev, dtype=torch.bfloat16) \
            if l > 1 else None
        din_f = new(B, ih, iw, ci) if l == 1 else None
        _lib.call("scae_conv3x3_dgrad_bf16r", _p(dpre_h), _p(wds[l - 1]),
                  _p(xin), _p(din_h), _p(din_f), B, ih, iw, ci, co, s, st)
        pending.append((partial, gw, gb, co, ci, splits))
        gws[l], gbs[l] = gw, gb
        dpre_h = din_h
        dpre = din_f
    for l in (range(L - 1, 0, -1) if not resident else ()):
        co, ci = wshapes[l][0], wshapes[l][1]
        xin, s = acts[l - 1], strides[l]
        ih, iw, oh, ow = xin.shape[1], xin.shape[2], dpre.shape[1], dpre.shape[2]
        splits = _lib.load().scae_conv3x3_wgrad_splits(B, oh, ow, ci, co)
        partial = new(splits * (9 * co * ci + co))
        gw, gb = gout(l), gout(L + l)
        din = new(B, ih, iw, ci)
        # weight-gradient partials and the (ReLU-gated) data gradient both
        # only wait for dpre: one launch
        pair = (_p(dpre), _p(wds[l - 1]), _p(xin), _p(din), _p(partial), B, ih,
                iw, ci, co, s)
        # parked parameter-gradient launches of the object encoder ride here:
        # the attention's reduction in the third layer's launch, the folding
        # products' backward (which reads it) in the second's, the largest
        carried = False
        if not plan.bf16 and l <= 2:
            parked = plan.take("reduce")
            if parked is not None:
                rc = lib.scae_conv3x3_bwd_pair_reduce_f32(
                    *pair, *parked.args, st) if l == 2 else _lib.ERR_UNSUPPORTED
                if rc == _lib.ERR_UNSUPPORTED:
                    parked.launch_alone()
                else:
                    _lib.check(rc, "scae_conv3x3_bwd_pair_reduce_f32")
                    carried = True
            parked = None if carried or l != 1 else plan.take("fold_bwd")
            if parked is not None:
                rc = lib.scae_conv3x3_bwd_pair_fold_f32(
                    *pair, ctypes.byref(parked.desc),
                    ctypes.byref(parked.grads), st)
                if rc == _lib.ERR_UNSUPPORTED:
                    parked.launch_alone()
                else:
                    _lib.check(rc, "scae_conv3x3_bwd_pair_fold_f32")
                    carried = True
        if not carried:
            _lib.call(_prec("scae_conv3x3_bwd_pair_f32"), *pair, st)
        pending.append((partial, gw, gb, co, ci, splits))
        gws[l], gbs[l] = gw, gb
        dpre = din
    c1 = wshapes[0][0]
    k1 = C0 * 9 + 1
    partial = new(_lib.load().scae_conv3x3_first_wgrad_rows(B, c1), c1, k1)
    # the image layer's weight-gradient partials; the split reductions of the
    # other layers' partials ride in the same launch
    n = len(pending)
    arr = lambda k: (ctypes.c_void_p * max(n, 1))(
        *[p[k].data_ptr() for p in pending])
    ints = lambda k: (ctypes.c_int * max(n, 1))(*[p[k] for p in pending])
    _lib.call("scae_conv3x3_first_wgrad_reduce_f32", _p(dpre), _p(image),
              _p(partial), B, C0, H, W, c1, strides[0], n, arr(0), arr(1),
              arr(2), ints(3), ints(4), ints(5), st)
    gws[0], gbs[0] = _sum_rows(partial.view(partial.shape[0], -1),
                               [(c1, C0, 3, 3), (c1,)],
                               outs=[gout(0), gout(L)], defer=defer)
    return gws, gbs


class _ConvStack(torch.autograd.Function):
    """relu(conv3x3(.. relu(conv3x3(image)) ..)): image (B, C, H, W) NCHW ->
    (B, C_last, OH, OW) as a channels-last view.  Intermediates are NHWC."""

    @_fwd
    def forward(ctx, image, strides, *wb):
        _need_hip(image, *wb)
        L = len(strides)
        image = image.contiguous()
        weights = [w.contiguous() for w in wb[:L]]
        biases = [b.contiguous() for b in wb[L:]]
        acts, wds, _ = _conv_stack_fwd(image, strides, weights, biases)
        ctx.save_for_backward(image, *acts, *wds)
        ctx.meta = (tuple(strides), [tuple(w.shape) for w in weights])
        ctx.slots = [_slot(t, ctx) for t in wb]
        ctx.refs = [tuple(t.shape) for t in wb]
        return acts[-1].permute(0, 3, 1, 2)

    @_bwd
    def backward(ctx, gy):
        strides, wshapes = ctx.meta
        L = len(strides)
        image = ctx.saved_tensors[0]
        acts = ctx.saved_tensors[1:1 + L]
        wds = ctx.saved_tensors[1 + L:]
        dpre = torch.ops.aten.threshold_backward(
            gy.permute(0, 2, 3, 1).contiguous(), acts[-1], 0.0)
        gws, gbs = _conv_stack_bwd(
            image, acts, wds, strides, wshapes, dpre,
            lambda i: _grad_out(ctx.slots[i], image, ctx.refs[i]))
        return (None, None, *gws, *gbs)


def conv_stack(image, weights, biases, strides):
    """ReLU(conv3x3) stack of the part-capsule encoder on the HIP kernels.
    The image receives no gradient (it is the data)."""
    return _ConvStack.apply(image, tuple(int(s) for s in strides), *weights,
                            *biases)


# ----------------------------------------------------------------------------
# K9 part-capsule head: 1x1 attention conv + attention pooling
# (part_encoder.py:71-74, nn_ext.py:76-101)
# ----------------------------------------------------------------------------
def attention_pool_supported(HW, A, P):
    return bool(_lib.load().scae_attention_pool_supported(HW, A, P))


# longest reduction (pixels) of one weight-gradient group of the 1x1 conv
_CONV1X1_KMAX = int(__import__("os").environ.get("SCAE_CONV1X1_KMAX", "1024"))


def _conv1x1_fwd(x, weight, bias):
    """y (B, HW, AP) = x (B, HW, C) weight^T + bias on the K7 GEMM."""
    B, HW, C = x.shape
    AP = weight.shape[0]
    y = torch.empty(B, HW, AP, device=x.device, dtype=x.dtype)
    _gemm(_p(x), _p(weight), _p(y), 1, B * HW, AP, C, True, C, 0, True, C, 0,
          AP, 0, bias=_p(bias), bias_ld=1, ref=x)
    return y


def _conv1x1_bwd(x, weight, dy, gate=None, outs=None, raw_sum=None,
                 defer=False):
    """-> (dx, dweight, dbias) of the 1x1 conv given dy (B, HW, AP).  With
    ``gate`` (the ReLU output x was made from, layout of x) -> (dx zeroed
    where gate <= 0, dweight, dbias, ungated dx).  ``outs``: buffers for
    (dweight, dbias).  ``raw_sum`` = (shape, out): also the batch sum of the
    ungated dx, written transposed ((HW, C) -> (C, HW)) into ``out``, in the
    same launch as the weight / bias sums."""
    B, HW, C = x.shape
    AP = weight.shape[0]
    # weight / bias gradient: dy^T x split over groups of images (the launch
    # also emits the column sums of dy), summed afterwards
    gsz = max(d for d in range(1, B + 1)
              if B % d == 0 and (d == 1 or HW * d <= _CONV1X1_KMAX))
    S, kper, slab = B // gsz, HW * gsz, AP * C + AP
    part = torch.empty(S, slab, device=x.device, dtype=x.dtype)
    dx = torch.empty_like(x)
    dgrad = _gemm_desc(_p(dy), _p(weight), _p(dx), 1, B * HW, C, AP, True, AP,
                       0, False, C, 0, C, 0)
    raw = None
    if gate is not None:
        raw = torch.empty_like(x)
        dgrad.mask, dgrad.ldmask, dgrad.c_nomask = gate.data_ptr(), C, \
            raw.data_ptr()
    _gemm_pair(
        _gemm_desc(_p(dy), _p(x), _p(part), S, AP, C, kper, False, AP,
                   kper * AP, False, C, kper * C, C, slab,
                   asum=_off(part, AP * C), asum_b=slab), dgrad, x)
    jobs = [dict(partial=part, shapes=[tuple(weight.shape), (AP,)], outs=outs,
                 defer=defer)]
    if raw_sum is not None:
        jobs.append(dict(partial=raw.view(B, HW * C), shapes=[raw_sum[0]],
                         transpose=C, outs=[raw_sum[1]], defer=defer))
    gw, gb = _sum_rows_multi(jobs)[0]
    return (dx, gw, gb) if gate is None else (dx, gw, gb, raw)


class _AttentionConvPool(torch.autograd.Function):
    """x (B, HW, C) NHWC features, weight (A*P, C), bias (A*P) -> (B, A, P-1):
    y = x weight^T + bias, then per capsule the softmax-over-pixels of its
    last channel pools its other P-1 channels."""

    @_fwd
    def forward(ctx, x, weight, bias, n_caps):
        _need_hip(x, weight, bias)
        x, weight, bias = x.contiguous(), weight.contiguous(), bias.contiguous()
        B, HW, C = x.shape
        P = weight.shape[0] // n_caps
        y = _conv1x1_fwd(x, weight, bias)
        out = torch.empty(B, n_caps, P - 1, device=x.device, dtype=x.dtype)
        _lib.call("scae_attention_pool_fwd_f32", _p(y), _p(out), B, HW, n_caps,
                  P, _stream(x))
        ctx.save_for_backward(x, weight, y)
        ctx.n_caps = n_caps
        return out

    @_bwd
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        B, HW, C = x.shape
        A = ctx.n_caps
        dy = torch.empty_like(y)
        _lib.call("scae_attention_pool_bwd_f32", _p(y), _p(g.contiguous()),
                  _p(dy), B, HW, A, weight.shape[0] // A, _stream(x))
        return (*_conv1x1_bwd(x, weight, dy), None)


def _capsule_head_fwd(x, weight, bias, noise_u, n_caps, noise_scale,
                      similarity):
    """1x1 attention conv + capsule head forward -> (y, pooled, pose,
    presence, feature or None, absence): one launch where the fused kernel
    covers the shape and its per-image weight reads stay cheap, the K7 GEMM +
    the head kernel otherwise."""
    B, HW, C = x.shape
    P = weight.shape[0] // n_caps
    F = P - 8
    new = lambda *shape: torch.empty(*shape, device=x.device, dtype=x.dtype)
    pooled, pose, presence = new(B, n_caps, P - 1), new(B, n_caps, 6), \
        new(B, n_caps)
    feature = new(B, n_caps, F) if F > 0 else None
    absence = new(B, n_caps, 1)
    head = (_p(noise_u), float(noise_scale), int(similarity), _p(pooled),
            _p(pose), _p(presence), _p(feature), _p(absence), B, HW, n_caps, P,
            _stream(x))
    if _lib.load().scae_capsule_head_conv_preferred(B, HW, n_caps, P, C) and \
            (x.data_ptr() | weight.data_ptr()) % 16 == 0:
        # the 1x1 conv inside the pooling workgroups (small batches)
        y = new(B, HW, n_caps * P)
        offer = _plan().offers.get("tc_fwd")
        if offer is not None and feature is not None and offer.result is None \
                and offer.M == n_caps and offer.F == F and offer.ok:
            # RIDES['tc_fwd']: the template generator's colour MLP behind the
            # head, workgroup by workgroup (it reads ``feature``, made here)
            raw = new(1, n_caps, offer.C, *offer.hw)
            templates = new(B, n_caps, offer.C, *offer.hw)
            color = new(B, n_caps, offer.C)
            if _lib.try_call(
                    "scae_capsule_head_conv_fwd_tc_f32", _p(x), _p(weight),
                    _p(bias), C, _p(y), *head[:-1],
                    *[_p(t) for t in offer.params], _p(raw), _p(templates),
                    _p(color), offer.C, offer.hw[0] * offer.hw[1], F, offer.H1,
                    *offer.codes, head[-1]):
                offer.result, offer.feature = (raw, templates, color), feature
                return y, pooled, pose, presence, feature, absence
            offer.ok = False
        _lib.call("scae_capsule_head_conv_fwd_f32", _p(x), _p(weight),
                  _p(bias), C, _p(y), *head)
    else:
        y = _conv1x1_fwd(x, weight, bias)
        _lib.call("scae_capsule_head_fwd_f32", _p(y), *head)
    return y, pooled, pose, presence, feature, absence


class _CapsuleHead(torch.autograd.Function):
    """The whole head of CapsuleImageEncoder.forward (part_encoder.py:71-92,
    n_poses = 6): 1x1 conv, attention pooling, split, presence noise +
    sigmoid, geometric_transform -> (pose (B,A,6), presence (B,A), feature
    (B,A,F) or None)."""

    @_fwd
    def forward(ctx, x, weight, bias, noise_u, n_caps, noise_scale, similarity):
        _need_hip(x, weight, bias, noise_u)
        x, weight, bias = x.contiguous(), weight.contiguous(), bias.contiguous()
        noise_u = _c(noise_u)
        B, HW, C = x.shape
        P = weight.shape[0] // n_caps
        F = P - 8
        new = lambda *shape: torch.empty(*shape, device=x.device, dtype=x.dtype)
        y, pooled, pose, presence, feature, absence = _capsule_head_fwd(
            x, weight, bias, noise_u, n_caps, noise_scale, similarity)
        ctx.save_for_backward(x, weight, y, pooled,
                              *([noise_u] if noise_u is not None else []))
        ctx.meta = (n_caps, float(noise_scale), int(similarity),
                    noise_u is not None)
        ctx.set_materialize_grads(False)
        if feature is None:
            feature = new(B, n_caps, 0)
            ctx.mark_non_differentiable(feature)
        ctx.mark_non_differentiable(absence)
        return pose, presence, feature, absence

    @_bwd
    def backward(ctx, g_pose, g_presence, g_feature, _g_absence):
        x, weight, y, pooled = ctx.saved_tensors[:4]
        A, noise_scale, similarity, has_noise = ctx.meta
        noise_u = ctx.saved_tensors[4] if has_noise else None
        B, HW, C = x.shape
        dy = torch.empty_like(y)
        # (a parked colour-MLP backward, whose g_feature this launch may read,
        # was launched on entry: RIDES['tc_bwd'].readers)
        _lib.call("scae_capsule_head_bwd_f32", _p(y), _p(pooled), _p(noise_u),
                  noise_scale, similarity, _p(_c(g_pose)), _p(_c(g_presence)),
                  _p(_c(g_feature)), None, _p(dy), B, HW, A,
                  weight.shape[0] // A, _stream(x))
        return (*_conv1x1_bwd(x, weight, dy), None, None, None, None)


def capsule_head(x, weight, bias, n_caps, noise_u=None, noise_scale=0.,
                 similarity=False):
    """Fused part-capsule head; see ``_CapsuleHead``.  feature is None when
    the capsules have no special features.  Also returns 1 - presence
    (B,A,1), detached -- the set-transformer input of SCAE.forward."""
    pose, presence, feature, absence = _CapsuleHead.apply(
        x, weight, bias, noise_u, n_caps, noise_scale, similarity)
    return pose, presence, (feature if feature.shape[-1] > 0 else None), absence


class _PartEncoder(torch.autograd.Function):
    """The whole CapsuleImageEncoder.forward (part_encoder.py:86-113) as ONE
    autograd node: conv stack (K8) with the embedding bias in the last
    layer's epilogue, 1x1 attention conv (K7), capsule head (K9).  As one node
    the backward needs no stand-alone glue launches: the ReLU gate of the last
    conv layer rides in the epilogue of the 1x1 conv's data-gradient GEMM
    (which also keeps the ungated values for the embedding-bias gradient), and
    the two consumers of ``feature`` get separate outputs whose gradients the
    head kernel adds."""

    @_fwd
    def forward(ctx, image, strides, post_bias, att_w, att_b, noise_u, n_caps,
                noise_scale, similarity, *wb):
        _need_hip(image, post_bias, att_w, att_b, noise_u, *wb)
        L = len(strides)
        image = image.contiguous()
        weights = [w.contiguous() for w in wb[:L]]
        biases = [b.contiguous() for b in wb[L:]]
        acts, wds, x = _conv_stack_fwd(image, strides, weights, biases,
                                       post_bias.contiguous())
        B, OH, OW, C = x.shape
        HW = OH * OW
        x = x.view(B, HW, C)
        att_w2 = att_w.contiguous().view(-1, C)
        P = att_w2.shape[0] // n_caps
        F = P - 8
        noise_u = _c(noise_u)
        new = lambda *shape: torch.empty(*shape, device=x.device, dtype=x.dtype)
        y, pooled, pose, presence, feature, absence = _capsule_head_fwd(
            x, att_w2, att_b.contiguous(), noise_u, n_caps, noise_scale,
            similarity)
        ctx.save_for_backward(image, *acts, *wds, x, att_w2, y, pooled,
                              *([noise_u] if noise_u is not None else []))
        ctx.meta = (tuple(strides), [tuple(w.shape) for w in weights], n_caps,
                    float(noise_scale), int(similarity), noise_u is not None,
                    tuple(post_bias.shape), tuple(att_w.shape))
        ctx.slots = [_slot(t, ctx) for t in (post_bias, att_w, att_b, *wb)]
        ctx.refs = [tuple(t.shape) for t in wb]
        ctx.set_materialize_grads(False)
        if feature is None:
            feature = new(B, n_caps, 0)
            ctx.mark_non_differentiable(feature)
        # the same values as a second tensor object (same memory): one output
        # per consumer, so autograd never has to add their gradients
        twin = torch.empty(0, device=x.device, dtype=x.dtype).set_(
            feature.untyped_storage(), feature.storage_offset(),
            feature.shape, feature.stride())
        if F <= 0:
            ctx.mark_non_differentiable(twin)
        ctx.mark_non_differentiable(absence)
        return pose, presence, feature, twin, absence

    @_bwd
    def backward(ctx, g_pose, g_presence, g_feature, g_twin, _g_absence):
        (strides, wshapes, A, noise_scale, similarity, has_noise, pb_shape,
         attw_shape) = ctx.meta
        L = len(strides)
        saved = ctx.saved_tensors
        image, acts, wds = saved[0], saved[1:1 + L], saved[1 + L:2 * L]
        x, att_w2, y, pooled = saved[2 * L:2 * L + 4]
        noise_u = saved[2 * L + 4] if has_noise else None
        B, HW, C = x.shape
        dy = torch.empty_like(y)
        P_ = att_w2.shape[0] // A
        head = (_p(y), _p(pooled), _p(noise_u), noise_scale, similarity,
                _p(_c(g_pose)), _p(_c(g_presence)))
        parked, carried = ctx.plan.take("tc_bwd"), False
        if parked is not None:
            # the colour MLP's backward in front of the head's, workgroup by
            # workgroup -- if its g_feature is one of the two gradients here
            mine = [t for t in (g_twin, g_feature) if t is not None
                    and t.is_contiguous() and t.data_ptr() == parked.gf_ptr]
            rc = _lib.ERR_UNSUPPORTED
            if len(mine) == 1 and parked.bm == (B, A):
                other = g_feature if mine[0] is g_twin else g_twin
                rc = _lib.load().scae_capsule_head_bwd_tc_f32(
                    *head, _p(_c(other)), _p(dy), B, HW, A, P_, *parked.ptrs,
                    *parked.dims, _stream(x))
            if rc == _lib.ERR_UNSUPPORTED:
                parked.launch_alone()
            else:
                _lib.check(rc, "scae_capsule_head_bwd_tc_f32")
                carried = True
        if not carried:
            _lib.call("scae_capsule_head_bwd_f32", *head, _p(_c(g_feature)),
                      _p(_c(g_twin)), _p(dy), B, HW, A, P_, _stream(x))
        act = acts[-1]
        g_attw = _grad_out(ctx.slots[1], x, attw_shape)
        g_attb = _grad_out(ctx.slots[2], x, (att_w2.shape[0],))
        # embedding-bias gradient: batch sum of the ungated dx, (HW, C) -> (C, HW)
        g_pb = _grad_out(ctx.slots[0], x, pb_shape)
        # parameters with a slot in the flat gradient buffer are leaves: the
        # sums that produce their gradients can wait for the step's one
        # column-sum launch (ops.deferred_param_sums)
        dpre, _, _, _ = _conv1x1_bwd(x, att_w2, dy, gate=act.view(B, HW, C),
                                     outs=[g_attw, g_attb],
                                     raw_sum=(pb_shape, g_pb),
                                     defer=all(sl is not None
                                               for sl in ctx.slots[:3]))
        gws, gbs = _conv_stack_bwd(
            image, acts, wds, strides, wshapes, dpre.view(act.shape),
            lambda i: _grad_out(ctx.slots[3 + i], image, ctx.refs[i]),
            defer=ctx.slots[3] is not None and ctx.slots[3 + L] is not None)
        return (None, None, g_pb, g_attw, g_attb, None, None, None, None,
                *gws, *gbs)


def part_encoder(image, weights, biases, strides, post_bias, att_weight,
                 att_bias, n_caps, noise_u=None, noise_scale=0.,
                 similarity=False):
    """CapsuleImageEncoder.forward as one autograd node (``_PartEncoder``):
    -> (pose (B,A,6), presence (B,A), feature (B,A,F) or None, feature twin
    (the same values for a second consumer), 1 - presence (B,A,1) detached).
    Needs a >= 2-layer ``conv_stack_supported`` stack, n_poses == 6 and
    ``attention_pool_supported`` shapes."""
    pose, presence, feature, twin, absence = _PartEncoder.apply(
        image, tuple(int(s) for s in strides), post_bias, att_weight, att_bias,
        noise_u, n_caps, noise_scale, similarity, *weights, *biases)
    if feature.shape[-1] == 0:
        feature = twin = None
    return pose, presence, feature, twin, absence


def attention_conv_pool(x, weight, bias, n_caps):
    """1x1 conv (as a GEMM over NHWC pixels) + multiple_attention_pooling_2d."""
    return _AttentionConvPool.apply(x, weight, bias, n_caps)


# ----------------------------------------------------------------------------
# class probabilities of SCAE.forward (stacked_capsule_auto_encoder.py:205-212)
# ----------------------------------------------------------------------------
class _PendingClassProbs:
    """RIDES['class_probs']: the class-probability launch of SCAE.forward,
    parked inside a fused step until ``_LossTail.forward`` carries it (or
    ``flush_pending_forward`` launches it: a step without the fused loss
    tail)."""

    def __init__(self, args, keep, stream_ref):
        self.args, self.keep, self.ref = args, keep, stream_ref

    def launch_alone(self):
        _lib.call("scae_class_probs_f32", *self.args, _stream(self.ref))


class _PendingCombine:
    """RIDES['combine']: the batch combine of a loss-tail forward launched
    with ``defer_combine``: carried by the tail's backward launch, or launched
    on its own (``scae_loss_tail_combine_f32``) when no backward follows."""

    def __init__(self, tail_args, keep, loss_ptr, out_ptr, stream_ref):
        self.tail, self.keep, self.ref = tail_args, keep, stream_ref
        self.loss_ptr, self.out_ptr = loss_ptr, out_ptr

    def launch_alone(self):
        ex = self.keep[-1]
        ex.defer_combine = 0
        _lib.call("scae_loss_tail_combine_f32", *self.tail, _stream(self.ref))


def flush_pending_forward():
    """Launch the forward launches still parked on the current plan (the
    op-by-op loss reads what they write)."""
    _plan().flush_scope("fusing")


def class_probs_supported(O, ncls):
    return bool(_lib.load().scae_class_probs_supported(O, ncls))


class _ClassProbs(torch.autograd.Function):
    """(caps_presence, posterior (B,O+1,M)) detached -> softmax(linear(.)) for
    the prior and the posterior capsule activations, both through the same
    classifier.  Differentiable w.r.t. the classifier only (as in the
    reference, whose inputs are detached); the rarely-needed backward runs the
    plain ops."""

    @_fwd
    def forward(ctx, caps_presence, posterior, weight, bias, extra_sums):
        _need_hip(caps_presence, posterior, weight, bias)
        cp, post = caps_presence.detach().contiguous(), \
            posterior.detach().contiguous()
        weight, bias = weight.contiguous(), bias.contiguous()
        B, O1, M = post.shape
        ncls = weight.shape[0]
        prior = torch.empty(B, ncls, device=cp.device, dtype=cp.dtype)
        posterior_prob = torch.empty_like(prior)
        extra = list(extra_sums or ())
        args = (_p(cp), _p(post), _p(weight), _p(bias), _p(prior),
                _p(posterior_prob), B, O1 - 1, M, ncls, _sum_jobs(extra),
                len(extra))
        if ctx.plan.fused:
            # inside a fused step nothing reads these before SCAE.loss: the
            # launch waits for the loss tail's per-image launch to carry it
            ctx.plan.flush_scope("fusing")
            ctx.plan.park("class_probs", _PendingClassProbs(
                args, (cp, post, weight, bias, prior, posterior_prob, extra),
                cp))
        else:
            _lib.call("scae_class_probs_f32", *args, _stream(cp))
        ctx.save_for_backward(cp, post, weight, bias)
        ctx.set_materialize_grads(False)
        return prior, posterior_prob

    @_bwd
    def backward(ctx, g_prior, g_post):
        cp, post, weight, bias = ctx.saved_tensors
        with torch.enable_grad():
            w, b = weight.detach().requires_grad_(), bias.detach().requires_grad_()
            outs, gs = [], []
            for x, g in ((cp, g_prior), (post[:, :-1].sum(-1), g_post)):
                if g is not None:
                    outs.append(torch.softmax(
                        torch.nn.functional.linear(x, w, b), -1))
                    gs.append(g)
            gw, gb = torch.autograd.grad(outs, (w, b), gs) if outs else (None,
                                                                         None)
        return None, None, gw, gb, None


def class_probs(caps_presence, posterior, weight, bias, extra_sums=None):
    """-> (prior_cls_prob, posterior_cls_prob).  ``extra_sums``: pending
    ``scaled_sums`` jobs (<= 8) that ride in the same launch."""
    return _ClassProbs.apply(caps_presence, posterior, weight, bias, extra_sums)


# ----------------------------------------------------------------------------
# K10 coloured templates (part_decoder.py:78-110)
# ----------------------------------------------------------------------------
_NONLIN_CODE = {"sigmoid": 0, "relu1": 1}


def template_color_supported(M, C, F, H1, template_nonlin, color_nonlin):
    return (template_nonlin in _NONLIN_CODE and color_nonlin in _NONLIN_CODE
            and bool(_lib.load().scae_template_color_supported(M, C, F, H1)))


class _TcOffer:
    """RIDES['tc_fwd']: the template generator's colour-MLP forward, offered
    (by SCAE.forward, ahead of the part encoder) to the part-capsule head's
    launch, which produces the ``feature`` it reads."""

    def __init__(self, logits, w1, b1, w2, b2, tnl, cnl):
        _need_hip(logits, w1, b1, w2, b2)
        self.params = [t.detach().contiguous() for t in (logits, w1, b1, w2, b2)]
        _, self.M, self.C, th, tw = logits.shape
        self.hw, self.F, self.H1 = (th, tw), w1.shape[1], w1.shape[0]
        self.codes = (tnl, cnl)
        self.result = self.feature = None
        self.ok = True

    def serves(self, logits, feature, w1, b1, w2, b2, tnl, cnl):
        return self.result is not None and self.codes == (tnl, cnl) and \
            self.feature.data_ptr() == feature.data_ptr() and \
            self.feature.shape == feature.shape and all(
                a.data_ptr() == b.data_ptr() and a.shape == b.shape
                for a, b in zip(self.params, (logits, w1, b1, w2, b2)))


def offer_colored_templates(template_logits, w1, b1, w2, b2, template_nonlin,
                            color_nonlin):
    """Inside a fused step: let the part encoder's head launch carry the
    coloured-template kernel that the next ``colored_templates`` call with
    these parameters (and that encoder's ``feature``) would launch."""
    plan = _plan()
    if plan.fused and torch.is_grad_enabled() and template_logits.is_cuda:
        plan.offer("tc_fwd", _TcOffer(template_logits, w1, b1, w2, b2,
                                      _NONLIN_CODE[template_nonlin],
                                      _NONLIN_CODE[color_nonlin]))


def withdraw_colored_templates_offer():
    _plan().claim("tc_fwd")


class _ColoredTemplates(torch.autograd.Function):
    @_fwd
    def forward(ctx, logits, feature, w1, b1, w2, b2, tnl, cnl,
                feature_node=None):
        _need_hip(logits, feature, w1, b1, w2, b2)
        logits, feature, w1, b1, w2, b2 = (t.contiguous() for t in (
            logits, feature, w1, b1, w2, b2))
        _, M, C, th, tw = logits.shape
        B, F, H1 = feature.shape[0], feature.shape[2], w1.shape[0]
        new = lambda *shape: torch.empty(*shape, device=logits.device,
                                         dtype=logits.dtype)
        offer = ctx.plan.claim("tc_fwd")
        if offer is not None and offer.serves(logits, feature, w1, b1, w2, b2,
                                              tnl, cnl):
            raw, templates, color = offer.result     # made in the head's launch
        else:
            raw, templates, color = new(1, M, C, th, tw), \
                new(B, M, C, th, tw), new(B, M, C)
            _lib.call("scae_template_color_fwd_f32", _p(logits), _p(feature),
                      _p(w1), _p(b1), _p(w2), _p(b2), _p(raw), _p(templates),
                      _p(color), B, M, C, th * tw, F, H1, tnl, cnl,
                      _stream(logits))
        ctx.save_for_backward(logits, feature, w1, b1, w2, b2, color)
        ctx.codes = (tnl, cnl)
        ctx.slots = [_slot(t, ctx) for t in (logits, w1, b1, w2, b2)]
        ctx.set_materialize_grads(False)
        # may the backward's launch wait for the part-capsule head's?  Only
        # when that node is the SOLE reader of g_feature: the fused part
        # encoder hands every consumer of ``feature`` its own output (the
        # twin), so autograd never adds anything to this one's gradient
        ctx.feature_node = feature_node
        return raw, templates

    @_bwd
    def backward(ctx, g_raw, g_templates):
        logits, feature, w1, b1, w2, b2, color = ctx.saved_tensors
        _, M, C, th, tw = logits.shape
        B, F, H1 = feature.shape[0], feature.shape[2], w1.shape[0]
        if g_templates is None:
            g_templates = torch.zeros(B, M, C, th, tw, device=logits.device,
                                      dtype=logits.dtype)
        n1, n2, n3 = H1 * F, H1 * F + H1, H1 * F + H1 + C * H1
        g_logits = _grad_out(ctx.slots[0], logits)
        g_feature = torch.empty_like(feature)
        rows = _lib.load().scae_template_color_partial_rows(B, M)
        partial = torch.empty(rows, n3 + C, device=logits.device,
                              dtype=logits.dtype)
        g_templates, g_raw = g_templates.contiguous(), _c(g_raw)
        ptrs = (_p(logits), _p(feature), _p(w1), _p(b1), _p(w2), _p(b2),
                _p(color), _p(g_templates), _p(g_raw), _p(g_logits),
                _p(g_feature), _p(partial))
        dims = (C, th * tw, F, H1, *ctx.codes)
        outs = [_grad_out(sl, t) for sl, t in zip(ctx.slots[1:],
                                                  (w1, b1, w2, b2))]
        # (parked only when the column sums of ``partial`` wait too)
        if ctx.plan.parking and ctx.feature_node == "_PartEncoderBackward" \
                and _in_slot(g_logits) and all(_in_slot(o) for o in outs):
            # the part-capsule head's backward, the only reader of g_feature,
            # runs the same (image, capsule group) workgroups: this launch
            # waits for it (``_PartEncoder.backward`` launches it first if it
            # cannot carry it)
            ctx.plan.park("tc_bwd", _PendingTcBackward(
                ptrs, (B, M), dims, (logits, feature, w1, b1, w2, b2, color,
                                     g_templates, g_raw, partial),
                g_feature.data_ptr(), logits))
        else:
            _lib.call("scae_template_color_bwd_f32", *ptrs, B, M, *dims,
                      _stream(logits))
        gw1, gb1, gw2, gb2 = _sum_rows(
            partial, [(H1, F), (H1,), (C, H1), (C,)], outs=outs,
            defer=all(sl is not None for sl in ctx.slots[1:5]))
        return g_logits, g_feature, gw1, gb1, gw2, gb2, None, None, None


def colored_templates(template_logits, feature, w1, b1, w2, b2,
                      template_nonlin, color_nonlin):
    """-> (raw_templates (1,M,C,h,w), templates (B,M,C,h,w)); nonlin names
    'sigmoid' | 'relu1'."""
    # (which node made ``feature``: see _ColoredTemplates.forward)
    node = type(feature.grad_fn).__name__ if feature.grad_fn is not None \
        else None
    return _ColoredTemplates.apply(template_logits, feature, w1, b1, w2, b2,
                                   _NONLIN_CODE[template_nonlin],
                                   _NONLIN_CODE[color_nonlin], node)


# ----------------------------------------------------------------------------
# K7 per-capsule MLPs on the batched MFMA GEMM (object_decoder.py:137-158)
# ----------------------------------------------------------------------------
def _gemm(A, B, C, batch, M, N, K, a_k, lda, a_b, b_k, ldb, b_b, ldc, c_b,
          bias=None, bias_ld=1, bias_b=0, mask=None, ldmask=0, mask_b=0,
          relu=False, asum=None, asum_b=0, asum_ld=1, ref=None):
    _lib.call(_prec("scae_gemm_f32"), A, B, C, bias, mask, asum, batch, M, N, K,
              int(a_k), lda, a_b, int(b_k), ldb, b_b, ldc, c_b, bias_ld, bias_b,
              ldmask, mask_b, asum_b, asum_ld, int(relu), _stream(ref))


def _gemm_desc(A, B, C, batch, M, N, K, a_k, lda, a_b, b_k, ldb, b_b, ldc, c_b,
               bias=None, bias_ld=1, bias_b=0, mask=None, ldmask=0, mask_b=0,
               relu=False, asum=None, asum_b=0, asum_ld=1):
    """The arguments of ``_gemm`` as a struct scae_gemm_desc (A .. asum are
    ctypes pointers as returned by ``_p`` / ``_off``)."""
    d = _lib.GemmDesc()
    for name, ptr in (("A", A), ("B", B), ("C", C), ("bias", bias),
                      ("mask", mask), ("asum", asum)):
        setattr(d, name, None if ptr is None else ptr.value)
    d.batch, d.M, d.N, d.K = batch, M, N, K
    d.a_kcontig, d.lda, d.a_batch = int(a_k), lda, a_b
    d.b_kcontig, d.ldb, d.b_batch = int(b_k), ldb, b_b
    d.ldc, d.c_batch = ldc, c_b
    d.bias_ld, d.bias_batch = bias_ld, bias_b
    d.ldmask, d.mask_batch, d.asum_batch, d.relu = ldmask, mask_b, asum_b, \
        int(relu)
    d.asum_ld = asum_ld
    return d


def _gemm_pair(first, second, ref):
    """Two independent GEMMs (``_gemm_desc``) in one launch."""
    _lib.call(_prec("scae_gemm_pair_f32"), ctypes.byref(first), ctypes.byref(second),
              _stream(ref))


def _off(t, nfloats=0):
    return ctypes.c_void_p(t.data_ptr() + 4 * nfloats)


class _GroupedMLP(torch.autograd.Function):
    """relu(.. relu(x W0^T + b0) .. Wn^T + bn) for G independent groups.
    x (B, G, Kin) with unit last stride; W_l (G, N_l, K_l [+1 if ones_input and
    l == 0]); b_l (G, N_l) or None; returns (B, G, N_last) contiguous.

    Two private contracts let a chain of these MLPs (and the K3 vote kernel
    behind it) skip the stand-alone ReLU-gate launches of the backward pass:
    ``grad_pregated``: the consumer hands back the gradient w.r.t. the last
    PRE-activation (already zeroed where the output is 0); ``x_is_relu``: x is
    the ReLU output of a producer with ``grad_pregated`` -- the returned input
    gradient is gated by x > 0 in the epilogue of the data-gradient GEMM."""

    @_fwd
    def forward(ctx, x, ones_input, n_layers, grad_pregated, x_is_relu,
                pad_out, *wb):
        _need_hip(x, *wb)
        if x.stride(2) != 1:
            x = x.contiguous()
        weights = [w.contiguous() for w in wb[:n_layers]]
        biases = [None if b is None else b.contiguous() for b in wb[n_layers:]]
        B, G, Kin = x.shape
        if ones_input and biases[0] is not None:
            raise ScaeHipError("ones_input with a bias is not built")
        acts = []                      # post-ReLU outputs of every layer
        cur, cur_ld, cur_b, K = x, x.stride(0), x.stride(1), Kin
        for l, (w, b) in enumerate(zip(weights, biases)):
            N, ldb = w.shape[1], w.shape[2]
            last = l == n_layers - 1
            if last:
                # pad_out: group rows padded to a multiple of 4 floats, so that
                # this GEMM's stores and the consumers' loads are 16-byte wide
                Np = (N + 3) // 4 * 4 if pad_out else N
                out = torch.empty(B, G, Np, device=x.device,
                                  dtype=x.dtype)[:, :, :N]
                ldc, c_b = G * Np, Np
            else:
                out = torch.empty(G, B, N, device=x.device, dtype=x.dtype)
                ldc, c_b = N, B * N
            if l == 0 and ones_input:   # implicit trailing 1.0 input column
                bias, bias_ld, bias_b = _off(w, Kin), ldb, N * ldb
            elif b is not None:
                bias, bias_ld, bias_b = _p(b), 1, N
            else:
                bias, bias_ld, bias_b = None, 1, 0
            _gemm(_p(cur), _p(w), _p(out), G, B, N, K, True, cur_ld, cur_b, True,
                  ldb, N * ldb, ldc, c_b, bias, bias_ld, bias_b, relu=True,
                  ref=x)
            acts.append(out)
            cur, cur_ld, cur_b, K = out, N, B * N, N
        ctx.save_for_backward(x, *weights, *acts)
        ctx.meta = (bool(ones_input), n_layers,
                    [b is not None for b in biases], bool(grad_pregated),
                    bool(x_is_relu))
        ctx.slots = [_slot(t, ctx) for t in wb]
        return acts[-1]

    @_bwd
    def backward(ctx, gy):
        ones_input, L, has_bias, pregated, x_is_relu = ctx.meta
        x = ctx.saved_tensors[0]
        weights = ctx.saved_tensors[1:1 + L]
        acts = ctx.saved_tensors[1 + L:]
        B, G, Kin = x.shape
        dev, dt = x.device, x.dtype
        # gradient w.r.t. the last pre-activation, layout (B, G, N); the group
        # rows may be padded (stride(1) >= N) when the consumer made them so
        gpre = gy
        if not pregated:
            gpre = torch.ops.aten.threshold_backward(gpre.contiguous(),
                                                     acts[-1].contiguous(), 0.0)
        if not (gpre.stride(2) == 1 and gpre.stride(1) >= gpre.shape[2]
                and gpre.stride(0) == G * gpre.stride(1)):
            gpre = gpre.contiguous()
        g_ld, g_b = gpre.stride(0), gpre.stride(1)
        gws, gbs = [None] * L, [None] * L
        gx = None
        for l in range(L - 1, -1, -1):
            w = weights[l]
            N, ldb = w.shape[1], w.shape[2]
            K = ldb - (1 if (l == 0 and ones_input) else 0)
            if l == 0:
                xin, x_ld, x_b = x, x.stride(0), x.stride(1)
            else:
                xin, x_ld, x_b = acts[l - 1], K, B * K
            gw = _grad_out(ctx.slots[l], w)
            # gW[g] (N x K) = gpre^T x : both operands k(=batch)-strided; the
            # bias gradient sum_b gpre is emitted by the same launch
            asum_ld = 1
            if has_bias[l]:
                gsum = _grad_out(ctx.slots[L + l], w, (G, N))
                asum, asum_b = _p(gsum), N
            elif l == 0 and ones_input:
                # the gradient of the implicit ones column IS that column sum:
                # written straight into column K of gW
                gsum, asum, asum_b, asum_ld = None, _off(gw, K), N * ldb, ldb
            else:
                gsum, asum, asum_b = None, None, 0
            wgrad = _gemm_desc(_p(gpre), _p(xin), _p(gw), G, N, K, B, False,
                               g_ld, g_b, False, x_ld, x_b, ldb, N * ldb,
                               asum=asum, asum_b=asum_b, asum_ld=asum_ld)
            # the data gradient of the layer only waits for gpre as well:
            # both GEMMs go out in one launch
            dgrad = gnext = None
            if l > 0:
                # g wrt previous pre-activation = (gpre W) gated by its ReLU
                gnext = torch.empty(G, B, K, device=dev, dtype=dt)
                dgrad = _gemm_desc(_p(gpre), _p(w), _p(gnext), G, B, K, N, True,
                                   g_ld, g_b, False, ldb, N * ldb, K, B * K,
                                   mask=_p(acts[l - 1]), ldmask=K, mask_b=B * K)
            elif ctx.needs_input_grad[0]:
                gx = torch.empty(B, G, K, device=dev, dtype=dt)
                gate = dict(mask=_p(x), ldmask=x.stride(0),
                            mask_b=x.stride(1)) if x_is_relu else {}
                dgrad = _gemm_desc(_p(gpre), _p(w), _p(gx), G, B, K, N, True,
                                   g_ld, g_b, False, ldb, N * ldb, G * K, K,
                                   **gate)
            if dgrad is not None:
                _gemm_pair(wgrad, dgrad, x)
            else:
                _gemm(_p(gpre), _p(xin), _p(gw), G, N, K, B, False, g_ld, g_b,
                      False, x_ld, x_b, ldb, N * ldb, asum=asum, asum_b=asum_b,
                      asum_ld=asum_ld, ref=x)
            if has_bias[l]:
                gbs[l] = gsum
            gws[l] = gw
            if l > 0:
                gpre, g_ld, g_b = gnext, K, B * K
        return (gx, None, None, None, None, None, *gws, *gbs)


class _Linear(torch.autograd.Function):
    """y = x W^T + b on K7 (fp32 MFMA): the projections / feed-forward layers of
    the set-transformer blocks that run module by module (multi-head, ISAB,
    PMA; set_transformer.py:56-104, :107-133).  Forward one launch (bias in the
    epilogue), backward one (weight + bias gradient and data gradient as a
    pair)."""

    @_fwd
    def forward(ctx, x, weight, bias):
        _need_hip(x, weight, bias)
        N, K = weight.shape
        x2 = x.reshape(-1, K)
        if x2.stride(1) != 1 or x2.stride(0) != K:
            x2 = x2.contiguous()
        weight = weight.contiguous()
        M = x2.shape[0]
        y = torch.empty(M, N, device=x.device, dtype=x.dtype)
        _gemm(_p(x2), _p(weight), _p(y), 1, M, N, K, True, K, 0, True, K, 0, N,
              0, bias=_p(_c(bias)), bias_ld=1, bias_b=0, ref=x)
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        ctx.x_shape = x.shape
        ctx.slots = (_slot(weight, ctx), _slot(bias, ctx) if bias is not None else None)
        return y.view(*x.shape[:-1], N)

    @_bwd
    def backward(ctx, gy):
        x2, weight = ctx.saved_tensors
        N, K = weight.shape
        M = x2.shape[0]
        gy2 = gy.reshape(M, N)
        if gy2.stride(1) != 1 or gy2.stride(0) != N:
            gy2 = gy2.contiguous()
        gw = _grad_out(ctx.slots[0], weight)
        gb = _grad_out(ctx.slots[1], weight, (N,)) if ctx.has_bias else None
        # gW (N x K) = gy^T x: both operands strided along the contraction (rows);
        # the bias gradient is A's column sum, emitted by the same tiles
        wgrad = _gemm_desc(_p(gy2), _p(x2), _p(gw), 1, N, K, M, False, N, 0,
                           False, K, 0, K, 0, asum=_p(gb), asum_b=0, asum_ld=1)
        if ctx.needs_input_grad[0]:
            gx = torch.empty(M, K, device=x2.device, dtype=x2.dtype)
            dgrad = _gemm_desc(_p(gy2), _p(weight), _p(gx), 1, M, K, N, True, N,
                               0, False, K, 0, K, 0)
            _gemm_pair(wgrad, dgrad, x2)
            gx = gx.view(ctx.x_shape)
        else:
            gx = None
            _lib.call("scae_gemm_multi_f32", (_lib.GemmDesc * 1)(wgrad), 1,
                      _stream(x2))
        return gx, gw, gb


def linear(x, weight, bias=None):
    """``F.linear`` on the batched MFMA GEMM (K7)."""
    return _Linear.apply(x, weight, bias)


class HipLinear(torch.nn.Linear):
    """``nn.Linear`` (same parameters, same state_dict keys) whose fp32 CUDA
    forward / backward run on K7 instead of the BLAS library."""

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and \
                self.weight.dtype == torch.float32 and not _bf16() and \
                not torch.is_autocast_enabled("cuda"):
            return linear(x, self.weight, self.bias)
        return torch.nn.functional.linear(x, self.weight, self.bias)


class _LayerNorm(torch.autograd.Function):
    """nn.LayerNorm(d) of the module-by-module set-transformer blocks
    (set_transformer.py:114-131): one wave per row; the weight / bias gradients
    leave as per-workgroup partial rows summed by ``_sum_rows``."""

    @_fwd
    def forward(ctx, x, weight, bias, eps):
        _need_hip(x, weight, bias)
        d = x.shape[-1]
        x2 = x.reshape(-1, d).contiguous()
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, device=x.device, dtype=x.dtype)
        rstd = torch.empty_like(mean)
        _lib.call("scae_layer_norm_fwd_f32", _p(x2), _p(_c(weight)),
                  _p(_c(bias)), _p(y), _p(mean), _p(rstd), rows, d, float(eps),
                  _stream(x))
        ctx.save_for_backward(x2, mean, rstd, *([weight] if weight is not None
                                                else []))
        ctx.has = (weight is not None, bias is not None)
        ctx.x_shape = x.shape
        ctx.slots = (_slot(weight, ctx) if weight is not None else None,
                     _slot(bias, ctx) if bias is not None else None)
        return y.view(x.shape)

    @_bwd
    def backward(ctx, gy):
        x2, mean, rstd = ctx.saved_tensors[:3]
        weight = ctx.saved_tensors[3] if ctx.has[0] else None
        rows, d = x2.shape
        gy2 = gy.reshape(rows, d).contiguous()
        gx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        want_p = ctx.has[0] or ctx.has[1]
        partial = torch.empty(_lib.load().scae_layer_norm_rows(rows), 2 * d,
                              device=x2.device, dtype=x2.dtype) \
            if want_p else None
        _lib.call("scae_layer_norm_bwd_f32", _p(x2), _p(_c(weight)), _p(mean),
                  _p(rstd), _p(gy2), _p(gx), _p(partial), rows, d, _stream(x2))
        gw = gb = None
        if want_p:
            outs = [_grad_out(ctx.slots[0], x2, (d,)) if ctx.has[0] else
                    torch.empty(d, device=x2.device, dtype=x2.dtype),
                    _grad_out(ctx.slots[1], x2, (d,)) if ctx.has[1] else
                    torch.empty(d, device=x2.device, dtype=x2.dtype)]
            gw, gb = _sum_rows(partial, [(d,), (d,)], outs=outs)
            gw = gw if ctx.has[0] else None
            gb = gb if ctx.has[1] else None
        return (None if gx is None else gx.view(ctx.x_shape)), gw, gb, None


class HipLayerNorm(torch.nn.LayerNorm):
    """``nn.LayerNorm`` over the last dimension (same parameters / state_dict
    keys) on the HIP kernel for fp32 CUDA inputs."""

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and \
                len(self.normalized_shape) == 1 and \
                self.normalized_shape[0] <= 1024 and \
                not torch.is_autocast_enabled("cuda"):
            return _LayerNorm.apply(x, self.weight, self.bias, self.eps)
        return super().forward(x)


def mlp_chain_supported(x, layers):
    """The one-launch chain (csrc/mlp_chain.hip) covers fp32 tensors (inside
    ``mfma_bf16()`` its products take bf16 operands), <= 4 layers, widths <=
    scae_mlp_chain_max_width()."""
    if not x.is_cuda or x.dtype != torch.float32 or \
            not 1 <= len(layers) <= 4:
        return False
    wmax = _lib.load().scae_mlp_chain_max_width()
    return x.shape[2] <= wmax and all(
        w.shape[1] <= wmax and w.shape[2] <= wmax + 1 and
        not (ones and b is not None) for w, b, ones in layers)


def _chain_forward_desc(x, ones_flags, weights, biases):
    """-> (struct scae_mlp_chain_desc of the forward chain, [layer outputs]);
    hidden outputs (G, B, N), the last one (B, G, N) with rows padded to a
    multiple of 4 floats."""
    L = len(ones_flags)
    B, G, Kin = x.shape
    d = _lib.MlpChainDesc()
    d.n_layers, d.in_, d.in_gs, d.in_bs, d.in_dim, d.B, d.G = \
        L, x.data_ptr(), x.stride(1), x.stride(0), Kin, B, G
    d.row_tile = _CHAIN_ROW_TILE
    d.bf16 = int(_bf16())
    acts, K = [], Kin
    for l, (w, b) in enumerate(zip(weights, biases)):
        N, ldw = w.shape[1], w.shape[2]
        assert ldw == K + (1 if ones_flags[l] else 0)
        if l == L - 1:
            Np = (N + 3) // 4 * 4
            out = torch.empty(B, G, Np, device=x.device,
                              dtype=x.dtype)[:, :, :N]
            out_gs, out_bs = Np, G * Np
        else:
            out = torch.empty(G, B, N, device=x.device, dtype=x.dtype)
            out_gs, out_bs = B * N, N
        y = d.layer[l]
        y.w, y.w_gs, y.ldw, y.K, y.N = w.data_ptr(), N * ldw, ldw, K, N
        if ones_flags[l]:          # implicit trailing 1.0 input column
            y.bias, y.bias_gs, y.bias_ld = w.data_ptr() + 4 * K, N * ldw, ldw
        elif b is not None:
            y.bias, y.bias_gs, y.bias_ld = b.data_ptr(), N, 1
        y.out, y.out_gs, y.out_bs, y.relu = out.data_ptr(), out_gs, out_bs, 1
        acts.append(out)
        K = N
    return d, acts


# batch rows per K7b workgroup handed to the launcher (scae_mlp_chain_desc.row_tile):
# 0 = by shape; the chain tests set 16 / 32 to run both forms at every size
_CHAIN_ROW_TILE = 0


def _chain_backward(x, weights, acts, ones_flags, has_bias, x_is_relu, slots,
                    need_gx, gpre=None, votes=None, park=False):
    """The backward of the chain: the data-gradient chain (one launch; with
    ``votes`` -- a struct scae_votes_desc whose gall_param(_gated) rows are
    ``gpre`` -- the vote kernel's backward rides at its head), then every
    weight gradient in one launch.  -> (gx, gws, gbs)."""
    L = len(ones_flags)
    B, G, Kin = x.shape
    dev, dt = x.device, x.dtype
    # 1. data-gradient chain: g_{l-1} = gate_{l-1}(g_l W_l), last layer first
    d = _lib.MlpChainDesc()
    d.n_layers, d.in_, d.in_gs, d.in_bs, d.in_dim, d.B, d.G = \
        L, gpre.data_ptr(), gpre.stride(1), gpre.stride(0), gpre.shape[2], B, G
    d.row_tile = _CHAIN_ROW_TILE
    d.bf16 = int(_bf16())
    gx = torch.empty(B, G, Kin, device=dev, dtype=dt) if need_gx else None
    gs = [None] * L          # gradient w.r.t. every pre-activation
    gs[L - 1] = gpre
    n_chain = 0
    for i, l in enumerate(range(L - 1, -1, -1)):
        w = weights[l]
        N, ldw = w.shape[1], w.shape[2]
        K = ldw - (1 if ones_flags[l] else 0)
        if l == 0 and not need_gx:
            break
        y = d.layer[i]
        y.w, y.w_gs, y.ldw, y.K, y.N = w.data_ptr(), N * ldw, ldw, N, K
        if l > 0:
            gs[l - 1] = torch.empty(G, B, K, device=dev, dtype=dt)
            y.gate, y.gate_gs, y.gate_bs = acts[l - 1].data_ptr(), B * K, K
            y.out, y.out_gs, y.out_bs = gs[l - 1].data_ptr(), B * K, K
        else:
            if x_is_relu:
                y.gate, y.gate_gs, y.gate_bs = x.data_ptr(), x.stride(1), \
                    x.stride(0)
            y.out, y.out_gs, y.out_bs = gx.data_ptr(), K, G * K
        n_chain += 1
    d.n_layers = n_chain
    if votes is not None:
        assert n_chain >= 1
        _lib.call("scae_mlp_chain_votes_bwd_f32", ctypes.byref(d),
                  ctypes.byref(votes), _stream(x))
    elif n_chain:
        _lib.call("scae_mlp_chain_bwd_f32", ctypes.byref(d), _stream(x))
    # 2. all weight gradients gW_l[g] (N x K) = g_l^T act_{l-1} in one launch
    descs = (_lib.GemmDesc * L)()
    gws, gbs = [None] * L, [None] * L
    for l in range(L):
        w = weights[l]
        N, ldw = w.shape[1], w.shape[2]
        K = ldw - (1 if ones_flags[l] else 0)
        if l == 0:
            xin, x_ld, x_b = x, x.stride(0), x.stride(1)
        else:
            xin, x_ld, x_b = acts[l - 1], K, B * K
        g = gs[l]
        if l == L - 1:
            g_ld, g_b = g.stride(0), g.stride(1)
        else:
            g_ld, g_b = N, B * N
        gw = _grad_out(slots[l], w)
        asum_ld = 1
        if has_bias[l]:
            gsum = _grad_out(slots[L + l], w, (G, N))
            asum, asum_b = _p(gsum), N
            gbs[l] = gsum
        elif ones_flags[l]:
            asum, asum_b, asum_ld = _off(gw, K), N * ldw, ldw
        else:
            asum, asum_b = None, 0
        descs[l] = _gemm_desc(_p(g), _p(xin), _p(gw), G, N, K, B, False,
                              g_ld, g_b, False, x_ld, x_b, ldw, N * ldw,
                              asum=asum, asum_b=asum_b, asum_ld=asum_ld)
        gws[l] = gw
    # (parked only when every output is a slot view taken here: a fresh buffer
    # is accumulated by autograd as soon as this node returns)
    if park and not _bf16() and L <= 4 and all(_in_slot(t) for t in gws) and \
            all(_in_slot(t) for t in gbs if t is not None):
        _plan().park("wgrads", _PendingWeightGemms(
            descs, L, (x, gs, acts, [w for w in weights]), x))
    else:
        _lib.call(_prec("scae_gemm_multi_f32"), descs, L, _stream(x))
    return gx, gws, gbs


class _MLPChain(torch.autograd.Function):
    """relu(W_n .. relu(W_0 x + b_0) .. + b_n) for G groups, all layers in ONE
    launch (K7b); a layer with ``ones`` sees its input extended by a constant
    1.0 column (object_decoder.py:149).  x (B, G, Kin); W_l (G, N_l, K_l [+1]);
    returns (B, G, N_last) -- group rows padded to a multiple of 4 floats.
    The consumer hands back the gradient w.r.t. the last PRE-activation
    (``_GroupedMLP``'s ``grad_pregated`` contract); backward = one launch for
    the data-gradient chain + one for all weight-gradient GEMMs."""

    @_fwd
    def forward(ctx, x, ones_flags, x_is_relu, *wb):
        L = len(ones_flags)
        _need_hip(x, *wb)
        if x.stride(2) != 1:
            x = x.contiguous()
        weights = [w.contiguous() for w in wb[:L]]
        biases = [None if b is None else b.contiguous() for b in wb[L:]]
        d, acts = _chain_forward_desc(x, ones_flags, weights, biases)
        _lib.call("scae_mlp_chain_fwd_f32", ctypes.byref(d), _stream(x))
        ctx.save_for_backward(x, *weights, *acts)
        ctx.meta = (tuple(ones_flags), [b is not None for b in biases],
                    bool(x_is_relu))
        ctx.slots = [_slot(t, ctx) for t in wb]
        return acts[-1]

    @_bwd
    def backward(ctx, gy):
        ones_flags, has_bias, x_is_relu = ctx.meta
        L = len(ones_flags)
        x = ctx.saved_tensors[0]
        weights = ctx.saved_tensors[1:1 + L]
        acts = ctx.saved_tensors[1 + L:]
        G = x.shape[1]
        gpre = gy
        if not (gpre.stride(2) == 1 and gpre.stride(1) >= gpre.shape[2]
                and gpre.stride(0) == G * gpre.stride(1)):
            gpre = gpre.contiguous()
        gx, gws, gbs = _chain_backward(x, weights, acts, ones_flags, has_bias,
                                       x_is_relu, ctx.slots,
                                       ctx.needs_input_grad[0], gpre=gpre)
        return (gx, None, None, *gws, *gbs)


class _ChainVotes(torch.autograd.Function):
    """CapsuleLayer.forward (object_decoder.py:120-236) from the object
    encoding to the votes in ONE launch: the MLP chain of ``_MLPChain`` with the
    vote kernel of ``_CapsuleVotes`` at its end; backward: the vote kernel's
    backward at the head of the data-gradient chain (one launch), all weight
    gradients (one launch), the bias column sums (deferred).  Outputs as
    ``capsule_votes``."""

    @_fwd
    def forward(ctx, x, ones_flags, vflags, noise_scale, cpr_static, b_cvr,
                b_caps, b_vote, b_scale, noise_caps, noise_vote, *wb):
        L = len(ones_flags)
        similarity, learn_vote_scale, allow_deformations, defer_reg = vflags
        _need_hip(x, cpr_static, b_cvr, b_caps, b_vote, b_scale, noise_caps,
                  noise_vote, *wb)
        if x.stride(2) != 1:
            x = x.contiguous()
        weights = [w.contiguous() for w in wb[:L]]
        biases = [None if b is None else b.contiguous() for b in wb[L:]]
        vargs = [_c(t) for t in (cpr_static, b_cvr, b_caps, b_vote, b_scale,
                                 noise_caps, noise_vote)]
        d, acts = _chain_forward_desc(x, ones_flags, weights, biases)
        all_param = acts[-1]
        B, O, A = all_param.shape
        V, ldp = (A - 7) // 8, all_param.stride(1)
        assert A == 8 * V + 7
        dev, dt = x.device, x.dtype
        new = lambda *shape: torch.empty(*shape, device=dev, dtype=dt)  # noqa: E731
        vote, scale, vp = new(B, O, V, 6), new(B, O, V), new(B, O, V)
        lc, lv, reg = new(B, O, 1), new(B, O, V), new(B, O)
        reg_loss = torch.empty((), device=dev, dtype=dt)   # l2_loss(.)/B, :170
        caps_presence = new(B, O)
        caps_arg = torch.empty(B, O, device=dev, dtype=torch.int32)
        v = _lib.VotesDesc()
        for name, t in zip(("cpr_static", "bias_cvr", "bias_caps", "bias_vote",
                            "bias_scale", "noise_caps", "noise_vote"), vargs):
            setattr(v, name, None if t is None else t.data_ptr())
        v.noise_scale, v.V, v.ld_param = float(noise_scale), V, ldp
        v.similarity, v.learn_vote_scale, v.allow_deformations = \
            int(similarity), int(learn_vote_scale), int(allow_deformations)
        for name, t in (("vote", vote), ("scale", scale), ("vote_presence", vp),
                        ("logit_caps", lc), ("logit_vote", lv),
                        ("reg_partial", reg), ("caps_presence", caps_presence),
                        ("caps_arg", caps_arg)):
            setattr(v, name, t.data_ptr())
        _lib.call("scae_mlp_chain_votes_fwd_f32", ctypes.byref(d),
                  ctypes.byref(v), _stream(x))
        if not defer_reg:
            scaled_sums([(reg, 0.5 / B, reg_loss)])
        ctx.save_for_backward(x, caps_arg, *weights, *acts,
                              *[t for t in vargs if t is not None])
        ctx.has_noise = (vargs[5] is not None, vargs[6] is not None)
        ctx.meta = (tuple(ones_flags), [b is not None for b in biases],
                    float(noise_scale), (V, ldp, int(similarity),
                                         int(learn_vote_scale),
                                         int(allow_deformations)))
        ctx.slots = [_slot(t, ctx) for t in wb]
        ctx.vslots = [_slot(t, ctx) for t in (cpr_static, b_cvr, b_caps, b_vote,
                                         b_scale)]
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(reg)
        return vote, scale, vp, lc, lv, reg_loss, caps_presence, reg

    @_bwd
    def backward(ctx, gvote, gscale, gvp, glc, glv, greg, gcp, _greg_partial):
        ones_flags, has_bias, noise_scale, (V, ldp, sim, lvs, adef) = ctx.meta
        L = len(ones_flags)
        saved = list(ctx.saved_tensors)
        x, caps_arg = saved[0], saved[1]
        weights = saved[2:2 + L]
        acts = saved[2 + L:2 + 2 * L]
        rest = saved[2 + 2 * L:]
        vargs = rest[:5]
        rest = rest[5:]
        noise_caps = rest.pop(0) if ctx.has_noise[0] else None
        noise_vote = rest.pop(0) if ctx.has_noise[1] else None
        all_param = acts[-1]
        B, O, A = all_param.shape
        dev, dt = x.device, x.dtype
        padded = lambda: torch.empty(B, O, ldp, device=dev,   # noqa: E731
                                     dtype=dt)[:, :, :A]
        gall, ggated = padded(), padded()
        gin = torch.empty(B, O, V, 6, device=dev, dtype=dt)
        grads = [_c(g) for g in (gvote, gscale, gvp, glc, glv, greg, gcp)]
        v = _lib.VotesDesc()
        for name, t in zip(("all_param", "cpr_static", "bias_cvr", "bias_caps",
                            "bias_vote", "bias_scale", "noise_caps",
                            "noise_vote"),
                           [all_param, *vargs, noise_caps, noise_vote]):
            setattr(v, name, None if t is None else t.data_ptr())
        v.noise_scale, v.V, v.ld_param = noise_scale, V, ldp
        v.similarity, v.learn_vote_scale, v.allow_deformations = sim, lvs, adef
        for name, t in zip(("gvote", "gscale", "gvote_presence", "glogit_caps",
                            "glogit_vote", "greg", "gcaps_presence"), grads):
            setattr(v, name, None if t is None else t.data_ptr())
        v.caps_arg = caps_arg.data_ptr()
        v.gall_param, v.gcpr_in, v.gall_param_gated = \
            gall.data_ptr(), gin.data_ptr(), ggated.data_ptr()
        # (inside a fused step the weight-gradient launch waits for the output
        # attention's backward to carry it)
        gx, gws, gbs = _chain_backward(
            x, weights, acts, ones_flags, has_bias, False, ctx.slots, True,
            gpre=ggated, votes=v,
            park=ctx.plan.parking)
        # bias gradients: batch sums of column blocks of gall (B, O*A)
        outs = [_grad_out(sl, t) for sl, t in zip(ctx.vslots, vargs)]
        gall_rows = torch.as_strided(gall, (B, O * ldp), (O * ldp, 1))
        (g_static,), (g_cvr, g_caps, g_vote, g_scale) = _sum_rows_multi([
            dict(partial=gin.view(B, -1), shapes=[vargs[0].shape],
                 outs=outs[:1], defer=ctx.vslots[0] is not None),
            dict(partial=gall_rows, shapes=[t.shape for t in vargs[1:5]],
                 starts=[6 * V, 6 * V + 6, 6 * V + 7, 7 * V + 7], period=ldp,
                 outs=outs[1:],
                 defer=all(sl is not None for sl in ctx.vslots[1:5]))])
        return (gx, None, None, None, g_static, g_cvr, g_caps, g_vote, g_scale,
                None, None, *gws, *gbs)


def chain_votes(x, layers, cpr_static, bias_cvr, bias_caps, bias_vote,
                bias_scale, noise_caps=None, noise_vote=None, noise_scale=0.,
                similarity=False, learn_vote_scale=True,
                allow_deformations=True, defer_reg=False):
    """``capsule_votes(mlp_chain(x, layers), ...)`` as one launch forward and
    one launch for the vote + data-gradient half of the backward; returns what
    ``capsule_votes`` returns."""
    ws = [w for w, _, _ in layers]
    bs = [b for _, b, _ in layers]
    return _ChainVotes.apply(
        x, tuple(bool(o) for _, _, o in layers),
        (bool(similarity), bool(learn_vote_scale), bool(allow_deformations),
         bool(defer_reg)), float(noise_scale), cpr_static, bias_cvr, bias_caps,
        bias_vote, bias_scale, noise_caps, noise_vote, *ws, *bs)


def mlp_chain(x, layers, x_is_relu=False):
    """``layers``: [(weight (G,N,K[+1]), bias (G,N) or None, ones_input)] -- up
    to 4 ReLU layers of G independent groups evaluated in one launch; x
    (B, G, Kin) -> (B, G, N_last) (a view of rows padded to a multiple of 4
    floats).  The returned tensor's gradient must be the gradient w.r.t. the
    last PRE-activation (zero where the output is zero), as ``capsule_votes(
    param_is_relu=True)`` provides it."""
    ws = [w for w, _, _ in layers]
    bs = [b for _, b, _ in layers]
    return _MLPChain.apply(x, tuple(bool(o) for _, _, o in layers),
                           bool(x_is_relu), *ws, *bs)


def grouped_mlp(x, weights, biases, ones_input=False, grad_pregated=False,
                x_is_relu=False, pad_out=False):
    """ReLU MLPs of G independent groups; x (B, G, Kin) -> (B, G, N_last).
    ``grad_pregated`` / ``x_is_relu``: see ``_GroupedMLP``.  ``pad_out``: the
    result's group rows are padded to a multiple of 4 floats (a strided view;
    for consumers that take a row stride, like ``capsule_votes``)."""
    biases = list(biases) if biases is not None else [None] * len(weights)
    return _GroupedMLP.apply(x, ones_input, len(weights), grad_pregated,
                             x_is_relu, pad_out, *weights, *biases)


# ----------------------------------------------------------------------------
# K3 capsule votes (object_decoder.py:160-225)
# ----------------------------------------------------------------------------
class _CapsuleVotes(torch.autograd.Function):
    @_fwd
    def forward(ctx, all_param, cpr_static, b_cvr, b_caps, b_vote, b_scale,
                noise_caps, noise_vote, noise_scale, similarity,
                learn_vote_scale, allow_deformations, param_is_relu,
                defer_reg):
        _need_hip(all_param, cpr_static, b_cvr, b_caps, b_vote, b_scale,
                  noise_caps, noise_vote)
        B, O, A = all_param.shape
        V = (A - 7) // 8
        # all_param may come with padded capsule rows (stride ldp >= A, see
        # grouped_mlp(pad_out=True)); anything else is made dense
        ldp = all_param.stride(1)
        if not (all_param.stride(2) == 1 and ldp >= A
                and all_param.stride(0) == O * ldp):
            all_param, ldp = all_param.contiguous(), A
        args = [all_param] + [_c(t) for t in (cpr_static, b_cvr, b_caps, b_vote,
                                              b_scale, noise_caps, noise_vote)]
        dev, dt = all_param.device, all_param.dtype
        vote = torch.empty(B, O, V, 6, device=dev, dtype=dt)
        scale = torch.empty(B, O, V, device=dev, dtype=dt)
        vp = torch.empty(B, O, V, device=dev, dtype=dt)
        lc = torch.empty(B, O, 1, device=dev, dtype=dt)
        lv = torch.empty(B, O, V, device=dev, dtype=dt)
        reg = torch.empty(B, O, device=dev, dtype=dt)
        reg_loss = torch.empty((), device=dev, dtype=dt)   # l2_loss(.)/B, :170
        caps_presence = torch.empty(B, O, device=dev, dtype=dt)
        caps_arg = torch.empty(B, O, device=dev, dtype=torch.int32)
        flags = (B, O, V, ldp, int(similarity), int(learn_vote_scale),
                 int(allow_deformations))
        _lib.call("scae_capsule_votes_fwd_f32", *[_p(t) for t in args],
                  float(noise_scale), _p(vote), _p(scale), _p(vp), _p(lc),
                  _p(lv), _p(reg), _p(caps_presence), _p(caps_arg), *flags,
                  _stream(all_param))
        if not defer_reg:   # else the caller folds `reg` into reg_loss later
            scaled_sums([(reg, 0.5 / B, reg_loss)])
        ctx.save_for_backward(caps_arg, *[t for t in args if t is not None])
        ctx.has_noise = (args[6] is not None, args[7] is not None)
        ctx.noise_scale = float(noise_scale)
        ctx.flags = flags
        ctx.param_is_relu = bool(param_is_relu)
        ctx.slots = [_slot(t, ctx) for t in (cpr_static, b_cvr, b_caps, b_vote,
                                        b_scale)]
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(reg)
        return vote, scale, vp, lc, lv, reg_loss, caps_presence, reg

    @_bwd
    def backward(ctx, gvote, gscale, gvp, glc, glv, greg, gcp, _greg_partial):
        saved = list(ctx.saved_tensors)
        caps_arg = saved.pop(0)
        args = saved[:6]
        args.append(saved.pop(6) if ctx.has_noise[0] else None)
        args.append(saved[6] if ctx.has_noise[1] else None)
        all_param = args[0]
        B, O, V, ldp = ctx.flags[:4]
        A = all_param.shape[2]
        # gradients with all_param's (possibly padded) capsule rows
        padded = lambda: torch.empty(B, O, ldp, device=all_param.device,  # noqa: E731
                                     dtype=all_param.dtype)[:, :, :A]
        gall = padded()
        # all_param = relu(.): hand the producer the gradient w.r.t. its
        # pre-activation (``grouped_mlp(grad_pregated=True)``); the bias
        # gradients below still need the ungated one
        ggated = padded() if ctx.param_is_relu else None
        gin = torch.empty(B, O, V, 6, device=all_param.device,
                          dtype=all_param.dtype)
        grads = [_c(g) for g in (gvote, gscale, gvp, glc, glv, greg, gcp)]
        _lib.call("scae_capsule_votes_bwd_f32", *[_p(t) for t in args],
                  ctx.noise_scale, *[_p(g) for g in grads], _p(caps_arg),
                  _p(gall), _p(gin), _p(ggated), *ctx.flags,
                  _stream(all_param))
        # bias gradients: batch sums of column blocks of gall (B, O*A); each
        # capsule's block lands in its row of the (.., O, ..) parameter
        outs = [_grad_out(sl, t) for sl, t in zip(ctx.slots, args[1:6])]
        gall_rows = torch.as_strided(gall, (B, O * ldp), (O * ldp, 1))
        (g_static,), (g_cvr, g_caps, g_vote, g_scale) = _sum_rows_multi([
            dict(partial=gin.view(B, -1), shapes=[args[1].shape],
                 outs=outs[:1], defer=ctx.slots[0] is not None),
            dict(partial=gall_rows,
                 shapes=[t.shape for t in args[2:6]],
                 starts=[6 * V, 6 * V + 6, 6 * V + 7, 7 * V + 7], period=ldp,
                 outs=outs[1:],
                 defer=all(sl is not None for sl in ctx.slots[1:5]))])
        return (gall if ggated is None else ggated, g_static, g_cvr, g_caps,
                g_vote, g_scale, None, None, None, None, None, None, None, None)


def capsule_votes(all_param, cpr_static, bias_cvr, bias_caps, bias_vote,
                  bias_scale, noise_caps=None, noise_vote=None, noise_scale=0.,
                  similarity=False, learn_vote_scale=True,
                  allow_deformations=True, param_is_relu=False,
                  defer_reg=False):
    """-> vote (B,O,V,6), scale, vote_presence, presence_logit_per_caps
    (B,O,1), presence_logit_per_vote (B,O,V), cpr_dynamic_reg_loss (),
    caps_presence (B,O) = vote_presence.max(-1).  ``param_is_relu``:
    all_param is the output of ``grouped_mlp(grad_pregated=True)`` and gets
    the gradient w.r.t. that MLP's last pre-activation.  Also returns the
    (B,O) partials of the reg loss; with ``defer_reg`` the reg-loss scalar is
    left for the caller to fill (``scaled_sums([(partial, .5/B, reg_loss)])``,
    together with other scalars in one launch)."""
    return _CapsuleVotes.apply(all_param, cpr_static, bias_cvr, bias_caps,
                               bias_vote, bias_scale, noise_caps, noise_vote,
                               noise_scale, similarity, learn_vote_scale,
                               allow_deformations, param_is_relu, defer_reg)


# ----------------------------------------------------------------------------
# K4 capsule likelihood (object_decoder.py:257-372)
# ----------------------------------------------------------------------------
class _CapsuleLikelihood(torch.autograd.Function):
    @_fwd
    def forward(ctx, vote, scale, vp, dummy_vote, x, presence, defer_sum):
        _need_hip(vote, scale, vp, dummy_vote, x, presence)
        vote, scale, vp, dummy_vote, x, presence = (
            _c(t) for t in (vote, scale, vp, dummy_vote, x, presence))
        B, O, M, _ = vote.shape
        dev, dt = vote.device, vote.dtype
        f = lambda *s: torch.empty(*s, device=dev, dtype=dt)  # noqa: E731
        lpp, binary = f(B, M), f(B, O, M)
        winner, winner_p = f(B, M, 6), f(B, M)
        widx = torch.empty(B, M, device=dev, dtype=torch.int64)
        from_caps = torch.empty(B, M, device=dev, dtype=torch.int64)
        soft, soft_p = f(B, M, 6), f(B, M)
        post, mlp, mlogit = f(B, O + 1, M), f(B, O + 1, M), f(B, O + 1, M)
        log_prob = torch.empty((), device=dev, dtype=dt)
        _lib.call("scae_capsule_likelihood_fwd_f32", _p(vote), _p(scale),
                  _p(vp), _p(dummy_vote), _p(x), _p(presence), _p(lpp),
                  _p(binary), _p(winner), _p(winner_p), _p(widx),
                  _p(from_caps), _p(soft), _p(soft_p), _p(post), _p(mlp),
                  _p(mlogit), B, O, M, _stream(vote))
        if not defer_sum:   # else the caller fills log_prob (scaled_sums)
            scaled_sums([(lpp, 1.0 / B, log_prob)])
        saved = [vote, scale, vp, dummy_vote, x, post, widx]
        if presence is not None:
            saved.append(presence)
        ctx.save_for_backward(*saved)
        ctx.has_presence = presence is not None
        ctx.dims = (B, O, M)
        ctx.dummy_shape = dummy_vote.shape
        ctx.mark_non_differentiable(binary, widx, from_caps)
        ctx.set_materialize_grads(False)
        return (lpp, binary, winner, winner_p, widx, from_caps, soft, soft_p,
                post, mlp, mlogit, log_prob)

    @_bwd
    def backward(ctx, g_lpp, _gb, g_w, g_wp, _gi, _gf, g_s, g_sp, g_post,
                 g_mlp, g_mlogit, g_log_prob):
        saved = ctx.saved_tensors
        vote, scale, vp, dummy_vote, x, post, widx = saved[:7]
        presence = saved[7] if ctx.has_presence else None
        B, O, M = ctx.dims
        if g_log_prob is not None:     # log_prob = sum(lpp) / B  (op-by-op loss)
            spread = (g_log_prob / B).expand(B, M)
            g_lpp = spread if g_lpp is None else g_lpp + spread
        gvote, gscale, gvp = (torch.empty_like(t) for t in (vote, scale, vp))
        gx = torch.empty_like(x)
        gpres = torch.empty(B, M, device=x.device, dtype=x.dtype) \
            if presence is not None else None
        gdummy = torch.empty(B, M, 6, device=x.device, dtype=x.dtype)
        gin = [_c(g) for g in (g_lpp, g_w, g_wp, g_s, g_sp, g_post, g_mlp,
                               g_mlogit)]
        pending = ctx.plan.take("k1_bwd")
        if pending is not None and any(
                g is not None for g in (g_w, g_wp, g_s, g_sp)):
            # a decoder fed by the (soft) winners ('soft' / 'hard' votes or
            # presences): these gradients are the parked launch's outputs
            pending.launch_alone()
            pending = None
        if pending is not None and pending.ref.device == vote.device:
            # the parked K1 backward and this kernel in one launch
            k = _lib.LikelihoodBwdDesc()
            for name, t in zip(
                    ("vote", "scale", "vote_presence", "dummy_vote", "x",
                     "presence", "posterior", "winner_idx", "g_lpp",
                     "g_winner", "g_winner_presence", "g_soft_winner",
                     "g_soft_winner_presence", "g_posterior",
                     "g_mixing_log_prob", "g_mixing_logit", "gvote", "gscale",
                     "gvote_presence", "gx", "gpresence", "gdummy_partial"),
                    (vote, scale, vp, dummy_vote, x, presence, post, widx,
                     *gin, gvote, gscale, gvp, gx, gpres, gdummy)):
                setattr(k, name, None if t is None else t.data_ptr())
            k.B, k.O, k.M = B, O, M
            _lib.call("scae_render_gmm_sums_bwd_likelihood_f32",
                      ctypes.byref(pending.desc), *pending.ptrs,
                      ctypes.byref(k), _stream(vote))
        else:
            if pending is not None:
                pending.launch_alone()
            _lib.call("scae_capsule_likelihood_bwd_f32", _p(vote), _p(scale),
                      _p(vp), _p(dummy_vote), _p(x), _p(presence), _p(post),
                      _p(widx), *[_p(g) for g in gin], _p(gvote), _p(gscale),
                      _p(gvp), _p(gx), _p(gpres), _p(gdummy), B, O, M,
                      _stream(vote))
        # the dummy vote only reaches the (soft) winner outputs: without a
        # gradient on those it gets none at all, as in the reference
        g_dummy = None
        if g_w is not None or g_s is not None:
            g_dummy = _sum_rows(gdummy.view(B, -1), [ctx.dummy_shape])[0]
        return gvote, gscale, gvp, g_dummy, gx, gpres, None


def capsule_likelihood(vote, scale, vote_presence, dummy_vote, x,
                       presence=None, defer_sum=False):
    """-> (log_prob_per_point (B,M), vote_presence_binary, winner,
    winner_presence, winner_idx, is_from_capsule, soft_winner,
    soft_winner_presence, posterior (B,O+1,M), mixing_log_prob, mixing_logit,
    log_prob () = sum(log_prob_per_point) / B -- left for the caller to fill
    with ``scaled_sums([(lpp, 1/B, log_prob)])`` when ``defer_sum``)"""
    return _CapsuleLikelihood.apply(vote, scale, vote_presence, dummy_vote, x,
                                    presence, defer_sum)


# ----------------------------------------------------------------------------
# K6 fused tail of SCAE.loss (stacked_capsule_auto_encoder.py:238-285)
# ----------------------------------------------------------------------------
_SPARSITY_TYPES = {"l2": 0, "entropy": 1, "kl": 2}


def loss_tail_supported(B, O, n_classes):
    return bool(_lib.load().scae_loss_tail_supported(B, O, n_classes or 0))


class _LossTail(torch.autograd.Function):
    @_fwd
    def forward(ctx, lpp, posterior, caps_presence, cls_w, cls_b, label,
                rec_sums, reg, cfg):
        _need_hip(lpp, posterior, caps_presence, cls_w, cls_b, rec_sums, reg)
        ctx.slots = (_slot(cls_w, ctx), _slot(cls_b, ctx))
        lpp, posterior, caps_presence = _c(lpp), _c(posterior), _c(caps_presence)
        cls_w, cls_b, label = _c(cls_w), _c(cls_b), _c(label)
        rec_sums, reg = _c(rec_sums), _c(reg)
        B, O1, M = posterior.shape
        O = O1 - 1
        ncls = 0 if cls_w is None else cls_w.shape[0]
        (n_classes_cfg, prior_type, post_type, sparsity_on, weights,
         within_const, w_reg) = cfg
        w5 = (ctypes.c_float * 5)(*weights)
        ints = (B, O, M, ncls, int(n_classes_cfg or 0), prior_type, post_type,
                int(sparsity_on))
        # the two outputs are views of ONE buffer that the node (and a parked
        # combine) keeps alive: a deferred combine writes them during the
        # backward pass, by address -- also when the caller has dropped the log
        # vector by then (the outputs themselves cannot be held: they point
        # back at this node)
        both = torch.empty(16, device=lpp.device, dtype=lpp.dtype)
        out = both[:12]
        wc = float("nan") if within_const is None else float(within_const)
        lab = None if label is None else ctypes.c_void_p(label.data_ptr())
        ex = _lib.LossExtras()
        if rec_sums is not None:
            ex.rec_sums, ex.n_rec = rec_sums.data_ptr(), rec_sums.numel()
        if reg is not None:
            ex.reg, ex.w_reg = reg.data_ptr(), float(w_reg)
        loss = both[12]
        ex.loss = loss.data_ptr()
        # The whole training scalar inside a fused step: nothing reads the
        # scalars before the backward has run, so the batch combine (one
        # workgroup, a dependent launch of its own) becomes a workgroup of the
        # backward launch.  A forward no backward follows is completed by
        # ``flush_pending_forward`` (the step's exit).
        defer = ctx.plan.fused and rec_sums is not None and \
            any(ctx.needs_input_grad) and \
            bool(_lib.load().scae_loss_tail_defer_preferred(B, O))
        ex.defer_combine = int(defer)
        # per-image / per-column statistics the backward kernel reads back
        ws = torch.empty(_lib.load().scae_loss_tail_workspace_floats(
            ints[0], ints[1], ints[3]), device=lpp.device, dtype=lpp.dtype)
        tail = (_p(lpp), _p(posterior), _p(caps_presence), _p(cls_w), _p(cls_b),
                lab, ctypes.byref(ex), _p(out), _p(ws), *ints, w5, wc)
        parked = ctx.plan.take("class_probs")
        if parked is not None:    # SCAE.forward's class probabilities ride along
            _lib.call("scae_loss_tail_fwd_class_probs_f32", *tail, *parked.args,
                      _stream(lpp))
        else:
            _lib.call("scae_loss_tail_fwd_f32", *tail, _stream(lpp))
        ctx.save_for_backward(lpp, posterior, caps_presence, ws,
                              *([cls_w, cls_b, label] if label is not None
                                else []),
                              *([rec_sums] if rec_sums is not None else []),
                              *([reg] if reg is not None else []))
        ctx.has = (label is not None, rec_sums is not None, reg is not None)
        ctx.call = (ints, tuple(weights), wc, float(w_reg))
        ctx.set_materialize_grads(False)
        ctx.deferred = None
        if defer:
            # (addresses only: ``loss`` / ``out`` are this node's outputs)
            ctx.deferred = _PendingCombine(
                tail, (lpp, posterior, caps_presence, cls_w, cls_b, label,
                       rec_sums, reg, ws, w5, both, ex), loss.data_ptr(),
                out.data_ptr(), lpp)
            ctx.plan.park("combine", ctx.deferred)
        return loss, out

    @_bwd
    def backward(ctx, g_loss, gout):
        saved = list(ctx.saved_tensors)
        lpp, posterior, cp, ws = saved[:4]
        rest = saved[4:]
        has_label, has_rec, has_reg = ctx.has
        cls_w = cls_b = label = rec_sums = reg = None
        if has_label:
            cls_w, cls_b, label = rest[:3]
            rest = rest[3:]
        if has_rec:
            rec_sums, rest = rest[0], rest[1:]
        if has_reg:
            reg = rest[0]
        ints, weights, wc, w_reg = ctx.call
        g_lpp, g_post, g_cp = (torch.empty_like(t) for t in (lpp, posterior, cp))
        g_w = _grad_out(ctx.slots[0], cls_w) if cls_w is not None else None
        g_b = _grad_out(ctx.slots[1], cls_b) if cls_b is not None else None
        g_rec = torch.empty_like(rec_sums) if has_rec else None
        g_reg = torch.empty_like(reg) if has_reg else None
        ex = _lib.LossExtras()
        if has_rec:
            ex.rec_sums, ex.n_rec = rec_sums.data_ptr(), rec_sums.numel()
            ex.g_rec_sums = g_rec.data_ptr()
        if has_reg:
            ex.reg, ex.w_reg, ex.g_reg = reg.data_ptr(), w_reg, g_reg.data_ptr()
        if g_loss is None and gout is None:
            gout = torch.zeros(12, device=lpp.device, dtype=lpp.dtype)
        if g_loss is not None:
            g_loss = g_loss.contiguous()
            ex.g_loss = g_loss.data_ptr()
        pend = ctx.deferred
        if pend is not None and ctx.plan.holds("combine", pend):
            # the forward's combine workgroup rides in this launch
            ctx.plan.take("combine")
            ex.defer_combine, ex.loss, ex.out12 = 1, pend.loss_ptr, pend.out_ptr
        w5 = (ctypes.c_float * 5)(*weights)
        lab = None if label is None else ctypes.c_void_p(label.data_ptr())
        _lib.call("scae_loss_tail_bwd_f32", _p(lpp), _p(posterior), _p(cp),
                  _p(cls_w), _p(cls_b), lab, ctypes.byref(ex),
                  _p(_c(gout)), _p(ws), _p(g_lpp), _p(g_post), _p(g_cp),
                  _p(g_w), _p(g_b), *ints, w5, wc, _stream(lpp))
        return g_lpp, g_post, g_cp, g_w, g_b, None, g_rec, g_reg, None


def loss_tail_scalar(lpp, posterior, caps_presence, cls_w, cls_b, label,
                     n_classes, prior_type, post_type, sparsity_on, weights,
                     within_const=None, rec_sums=None, reg=None, w_reg=0.):
    """-> (loss (), tensor (12)).  The 12-vector: [loss, log_prob,
    prior_within, prior_between, posterior_within, posterior_between,
    prior_cls_xe, posterior_cls_xe, rec_ll, -rec_ll, -log_prob, reg]; loss
    also carries -rec_ll (from K1's tile sums ``rec_sums``) and w_reg * reg
    when those are given.  The loss is returned as its own 0-dim output too, so
    that ``loss.backward()`` seeds one scalar instead of scattering into the
    vector."""
    if label is None:
        cls_w = cls_b = None
    cfg = (n_classes, _SPARSITY_TYPES[prior_type], _SPARSITY_TYPES[post_type],
           bool(sparsity_on), [float(w) for w in weights], within_const,
           float(w_reg))
    return _LossTail.apply(lpp, posterior, caps_presence, cls_w, cls_b, label,
                           rec_sums, reg, cfg)


def loss_tail(*args, **kwargs):
    """``loss_tail_scalar(...)[1]``: the 12-vector only."""
    return loss_tail_scalar(*args, **kwargs)[1]


# ----------------------------------------------------------------------------
# K1 template render + mixture likelihood (part_decoder.py:174-237,
# distributions.py:34-47)
# ----------------------------------------------------------------------------
class DecoderInputs:
    """The compact description of one TemplateBasedImageDecoder call."""

    FIELDS = ("templates", "templates_alpha", "pose", "presence", "bg_image",
              "bg_value", "bg_mixing_logit", "temperature_logit", "out_scale")

    def __init__(self, output_size, **tensors):
        self.output_size = tuple(output_size)
        for f in self.FIELDS:
            setattr(self, f, tensors.get(f))

    def tensors(self):
        return [getattr(self, f) for f in self.FIELDS]


def _make_desc(tensors, output_size):
    (templates, alpha, pose, presence, bg_image, bg_value, bg_ml, temp,
     out_scale) = tensors
    B0, M, C, th, tw = templates.shape
    B = pose.shape[0]            # B0 template sets may serve B = r * B0 images
    if B % B0:
        raise ValueError(f"{B} poses cannot share {B0} template sets")
    H, W = output_size
    d = DecoderDesc(_p(templates), _p(alpha), _p(pose), _p(presence),
                    _p(bg_image), _p(bg_value), _p(bg_ml), _p(temp),
                    _p(out_scale), B, M, C, th, tw, H, W, B // B0)
    return d, (B, M, C, th, tw, H, W)


def _decoder_backward(ctx_tensors, output_size, needs, x, lse_post, lse_prior,
                      g_lp, g_tt, g_ml, g_tile=None, slots=None,
                      alpha_shape=None, park=False):
    (templates, alpha, pose, presence, bg_image, bg_value, bg_ml, temp,
     out_scale) = ctx_tensors
    d, (B, M, C, th, tw, H, W) = _make_desc(ctx_tensors, output_size)
    # on a step's second lane the backward shares the chip with the main
    # lane's launches: a fixed number of resident workgroups (step_plan)
    lane_plan = _plan()
    if lane_plan.side_stream is not None and lane_plan.side_open:
        d.bwd_resident = int(lane_plan.side_resident)
    dev, dt = templates.device, templates.dtype
    g_templates = torch.empty_like(templates)
    g_alpha_p = torch.empty(B, M, th, tw, device=dev, dtype=dt) \
        if alpha is not None else None
    g_pose = torch.empty_like(pose)
    g_presence = torch.empty_like(presence) if presence is not None else None
    g_bg_image = torch.empty_like(bg_image) if bg_image is not None else None
    g_scal = torch.empty(B, M + 1, 4, device=dev, dtype=dt)
    slots = slots or [None] * 9
    # batch sums of the alpha partials and of the four scalar parameters'
    # (B*(M+1), 4) partials: one launch
    jobs, g_alpha, gs = [], None, [None] * 4
    if alpha is not None:
        ashape = tuple(alpha_shape or alpha.shape)
        jobs.append(dict(partial=g_alpha_p.view(B, -1), shapes=[ashape],
                         outs=[_grad_out(slots[1], alpha, ashape)],
                         defer=slots[1] is not None))
    scal = [(i, t) for i, t in enumerate((bg_value, bg_ml, temp, out_scale))
            if t is not None]
    if scal:
        jobs.append(dict(partial=g_scal.view(-1, 4),
                         shapes=[t.shape for _, t in scal],
                         starts=[i for i, _ in scal],
                         outs=[_grad_out(slots[5 + i], t) for i, t in scal],
                         defer=all(slots[5 + i] is not None
                                   for i, _ in scal)))
    # a parked launch fills the partial matrices LATER: only when every column
    # sum over them waits too (deferred, i.e. all of its outputs are slot views
    # this backward took itself -- _sum_rows_multi's own rule); a sum that
    # would launch now would read partials nobody has written yet
    if park:
        plan = _plan()
        park = plan.deferred is not None and all(
            job["defer"] and all(_in_slot(o) for o in job["outs"])
            for job in jobs)
    if g_tile is not None and park:
        # inside a training step: parked for the capsule likelihood's backward
        # to carry (both only wait for the loss tail); the gradient buffers are
        # handed to autograd now and filled by that launch -- their consumers
        # (template generator, part encoder, the deferred column sums) all run
        # after it
        plan.flush_scope("deferring")
        addr = lambda t: None if t is None else t.data_ptr()   # noqa: E731
        plan.park("k1_bwd", _PendingK1Backward(
            d, ctx_tensors, (x, lse_post, lse_prior, g_tile),
            (addr(g_templates), g_alpha_p, addr(g_pose), addr(g_presence),
             addr(g_bg_image), g_scal), templates))
    elif g_tile is not None:
        _lib.call("scae_render_gmm_sums_bwd_f32", ctypes.byref(d), _p(x),
                  _p(lse_post), _p(lse_prior), _p(g_tile), _p(g_templates),
                  _p(g_alpha_p), _p(g_pose), _p(g_presence), _p(g_bg_image),
                  _p(g_scal), _stream(templates))
    else:
        _lib.call("scae_render_gmm_bwd_f32", ctypes.byref(d), _p(x),
                  _p(lse_post), _p(lse_prior), _p(g_lp), _p(g_tt), _p(g_ml),
                  _p(g_templates), _p(g_alpha_p), _p(g_pose), _p(g_presence),
                  _p(g_bg_image), _p(g_scal), _stream(templates))
    res = _sum_rows_multi(jobs) if jobs else []
    if alpha is not None:
        g_alpha = res[0][0]
    if scal:
        for (i, _), o in zip(scal, res[-1]):
            gs[i] = o
    return (g_templates, g_alpha, g_pose, g_presence, g_bg_image, *gs)


def _prep_decoder(tensors):
    _need_hip(*tensors)
    tensors = [_c(_detached(t)) for t in tensors]
    if tensors[1] is not None:                      # templates_alpha (1,M,1,h,w)
        tensors[1] = tensors[1].reshape(-1, *tensors[1].shape[-2:])
    return tensors


class _RenderTemplates(torch.autograd.Function):
    """materialising path: (transformed_templates, mixing_logits)."""

    @_fwd
    def forward(ctx, output_size, *tensors):
        t = _prep_decoder(tensors)
        d, (B, M, C, th, tw, H, W) = _make_desc(t, output_size)
        dev, dt = t[0].device, t[0].dtype
        Cm = 1 if t[1] is not None else C
        tt = torch.empty(B, M + 1, C, H, W, device=dev, dtype=dt)
        ml = torch.empty(B, M + 1, Cm, H, W, device=dev, dtype=dt)
        _lib.call("scae_template_render_fwd_f32", ctypes.byref(d), _p(tt),
                  _p(ml), _stream(t[0]))
        ctx.save_for_backward(*[x for x in t if x is not None])
        ctx.present = [x is not None for x in t]
        ctx.output_size = output_size
        ctx.alpha_shape = None if tensors[1] is None else tensors[1].shape
        ctx.slots = [_slot(v, ctx) for v in tensors]
        ctx.set_materialize_grads(False)
        return tt, ml

    @_bwd
    def backward(ctx, g_tt, g_ml):
        if g_tt is None and g_ml is None:
            return (None,) * 10
        it = iter(ctx.saved_tensors)
        t = [next(it) if p else None for p in ctx.present]
        grads = _decoder_backward(t, ctx.output_size, ctx.needs_input_grad,
                                  None, None, None, None, _c(g_tt), _c(g_ml),
                                  slots=ctx.slots, alpha_shape=ctx.alpha_shape)
        grads = list(grads)
        if grads[1] is not None:
            grads[1] = grads[1].view(ctx.alpha_shape)
        return (None, *grads)


class _RenderGmmLogProb(torch.autograd.Function):
    """fused path: log_prob(x) from the compact decoder inputs."""

    @_fwd
    def forward(ctx, output_size, x, *tensors):
        t = _prep_decoder(tensors)
        _need_hip(x)
        x = x.detach().contiguous()
        d, (B, M, C, th, tw, H, W) = _make_desc(t, output_size)
        dev, dt = t[0].device, t[0].dtype
        Cm = 1 if t[1] is not None else C
        lp = torch.empty(B, C, H, W, device=dev, dtype=dt)
        lse_post = torch.empty(B, C, H, W, device=dev, dtype=dt)
        lse_prior = torch.empty(B, Cm, H, W, device=dev, dtype=dt)
        _lib.call("scae_render_gmm_logprob_fwd_f32", ctypes.byref(d), _p(x),
                  _p(lp), _p(lse_post), _p(lse_prior), _stream(x))
        ctx.save_for_backward(x, lse_post, lse_prior,
                              *[v for v in t if v is not None])
        ctx.present = [v is not None for v in t]
        ctx.output_size = output_size
        ctx.alpha_shape = None if tensors[1] is None else tensors[1].shape
        ctx.slots = [_slot(v, ctx) for v in tensors]
        return lp

    @_bwd
    def backward(ctx, g_lp):
        x, lse_post, lse_prior = ctx.saved_tensors[:3]
        it = iter(ctx.saved_tensors[3:])
        t = [next(it) if p else None for p in ctx.present]
        grads = list(_decoder_backward(t, ctx.output_size, ctx.needs_input_grad,
                                       x, lse_post, lse_prior,
                                       g_lp.contiguous(), None, None,
                                       slots=ctx.slots,
                                       alpha_shape=ctx.alpha_shape))
        if grads[1] is not None:
            grads[1] = grads[1].view(ctx.alpha_shape)
        # the reconstruction target gets no gradient on this path (the
        # reference feeds it the input image, which never requires grad)
        return (None, None, *grads)


class _RenderGmmLogProbSums(torch.autograd.Function):
    """fused path for the training loss: per (image, pixel tile) sums of
    log_prob(x) instead of the per-pixel map -> (B, tiles)."""

    @_fwd
    def forward(ctx, output_size, x, rider, parkable, *tensors):
        ctx.parkable = bool(parkable)
        t = _prep_decoder(tensors)
        _need_hip(x)
        x = x.detach().contiguous()
        if rider is not None:     # computed in the object encoder's launch
            sums, lse_post, lse_prior = rider.sums, rider.lse_post, \
                rider.lse_prior
        else:
            d, (B, M, C, th, tw, H, W) = _make_desc(t, output_size)
            dev, dt = t[0].device, t[0].dtype
            Cm = 1 if t[1] is not None else C
            tiles = _lib.load().scae_render_gmm_logprob_tiles(ctypes.byref(d))
            sums = torch.empty(B, tiles, device=dev, dtype=dt)
            lse_post = torch.empty(B, C, H, W, device=dev, dtype=dt)
            lse_prior = torch.empty(B, Cm, H, W, device=dev, dtype=dt)
            _lib.call("scae_render_gmm_logprob_sums_fwd_f32", ctypes.byref(d),
                      _p(x), _p(sums), _p(lse_post), _p(lse_prior), _stream(x))
        ctx.save_for_backward(x, lse_post, lse_prior,
                              *[v for v in t if v is not None])
        ctx.present = [v is not None for v in t]
        ctx.output_size = output_size
        ctx.alpha_shape = None if tensors[1] is None else tensors[1].shape
        ctx.slots = [_slot(v, ctx) for v in tensors]
        return sums

    @_bwd
    def backward(ctx, g_sums):
        x, lse_post, lse_prior = ctx.saved_tensors[:3]
        it = iter(ctx.saved_tensors[3:])
        t = [next(it) if p else None for p in ctx.present]
        grads = list(_decoder_backward(t, ctx.output_size, ctx.needs_input_grad,
                                       x, lse_post, lse_prior, None, None, None,
                                       g_tile=g_sums.contiguous(),
                                       slots=ctx.slots,
                                       alpha_shape=ctx.alpha_shape,
                                       # (only where every consumer of the
                                       # gradients runs later: sums deferred,
                                       # inside a fused training step, pose
                                       # and presence not the capsule
                                       # likelihood's own outputs)
                                       # (with a second lane the launch goes
                                       # there at once: step_plan.SIDE_NODES)
                                       park=ctx.plan.parking
                                       and ctx.parkable
                                       and ctx.plan.side_stream is None))
        if grads[1] is not None:
            grads[1] = grads[1].view(ctx.alpha_shape)
        return (None, None, None, None, *grads)


_K1_GRADIENT_READERS = ("_ColoredTemplatesBackward", "_PartEncoderBackward",
                        "_CapsuleHeadBackward")


def render_gmm_log_prob_sums(inputs: "DecoderInputs", x):
    """(B, tiles) partial sums of the mixture log-likelihood of ``x``; their
    total is sum_{b,c,h,w} log_prob."""
    M, C = inputs.templates.shape[1:3]
    B = inputs.pose.shape[0]
    if tuple(x.shape) != (B, C, *inputs.output_size):
        raise ValueError(f"log_prob target must be {(B, C, *inputs.output_size)}"
                         f", got {tuple(x.shape)}")
    if x.requires_grad:
        raise ScaeHipError("fused log_prob does not differentiate w.r.t. its "
                           "target; use the materialised mixture for that")
    plan = _plan()
    rider = plan.rider
    if rider is not None and rider.launched and \
            rider.key == LogProbRider.key_of(inputs, x):
        plan.rider = None         # consumed
    else:
        rider = None
    # the backward's launch may wait for the capsule likelihood's backward to
    # carry it (RIDES['k1_bwd']) when every reader of its outputs is a
    # plan-aware node (which launches a parked K1 before it starts) or a leaf
    # -- and not that carrier itself: a decoder fed with the likelihood's
    # (soft) winners, vote_type / presence_type 'soft' or 'hard'
    # (stacked_capsule_auto_encoder.py:146-156), is read by it
    parkable = all(
        t is None or t.grad_fn is None
        or type(t.grad_fn).__name__ in _K1_GRADIENT_READERS
        for t in (inputs.templates, inputs.pose, inputs.presence))
    return _RenderGmmLogProbSums.apply(inputs.output_size, x, rider, parkable,
                                       *inputs.tensors())


def render_templates(inputs: DecoderInputs):
    return _RenderTemplates.apply(inputs.output_size, *inputs.tensors())


def render_gmm_log_prob(inputs: DecoderInputs, x):
    M, C = inputs.templates.shape[1:3]
    B = inputs.pose.shape[0]
    if tuple(x.shape) != (B, C, *inputs.output_size):
        raise ValueError(f"log_prob target must be {(B, C, *inputs.output_size)}"
                         f", got {tuple(x.shape)}")
    if x.requires_grad:
        raise ScaeHipError("fused log_prob does not differentiate w.r.t. its "
                           "target; use the materialised mixture for that")
    return _RenderGmmLogProb.apply(inputs.output_size, x, *inputs.tensors())


# ----------------------------------------------------------------------------
# generic mixture over materialised tensors (distributions.py:34-77)
# ----------------------------------------------------------------------------
def _gmm_dims(loc, ml):
    B, K, C = loc.shape[:3]
    Cm = ml.shape[2]
    P = int(np.prod(loc.shape[3:])) if loc.dim() > 3 else 1
    if ml.shape[0] != B or ml.shape[1] != K or ml.shape[3:] != loc.shape[3:] \
            or Cm not in (1, C):
        raise ScaeHipError(f"unsupported mixture shapes loc {tuple(loc.shape)}"
                           f" logits {tuple(ml.shape)}")
    return B, K, C, Cm, P


class _GmmLogProb(torch.autograd.Function):
    @_fwd
    def forward(ctx, loc, ml, sigma, x):
        _need_hip(loc, ml, sigma, x)
        loc, ml, sigma, x = _c(loc), _c(ml), _c(sigma), _c(x)
        B, K, C, Cm, P = _gmm_dims(loc, ml)
        out = torch.empty_like(x)
        _lib.call("scae_gmm_log_prob_fwd_f32", _p(loc), _p(ml), _p(sigma),
                  _p(x), _p(out), B, K, C, Cm, P, _stream(loc))
        ctx.save_for_backward(loc, ml, sigma, x)
        return out

    @_bwd
    def backward(ctx, g):
        loc, ml, sigma, x = ctx.saved_tensors
        B, K, C, Cm, P = _gmm_dims(loc, ml)
        g = g.contiguous()
        g_loc, g_ml = torch.empty_like(loc), torch.empty_like(ml)
        g_sig = torch.empty(B, device=loc.device, dtype=loc.dtype)
        g_x = torch.empty_like(x) if ctx.needs_input_grad[3] else None
        _lib.call("scae_gmm_log_prob_bwd_f32", _p(loc), _p(ml), _p(sigma),
                  _p(x), _p(g), _p(g_loc), _p(g_ml), _p(g_sig), _p(g_x), B, K,
                  C, Cm, P, _stream(loc))
        return g_loc, g_ml, g_sig.sum().view_as(sigma), g_x


def _check_gmm_x(loc, x):
    if tuple(x.shape) != (loc.shape[0], *loc.shape[2:]):
        raise ScaeHipError(f"x {tuple(x.shape)} does not match loc "
                           f"{tuple(loc.shape)}")


def gmm_log_prob(loc, mixing_logits, sigma, x):
    """loc (B,K,C,*), mixing_logits (B,K,1|C,*), sigma 1-element tensor."""
    if sigma.numel() != 1:
        raise ScaeHipError("only a scalar mixture scale is supported")
    _check_gmm_x(loc, x)
    return _GmmLogProb.apply(loc, mixing_logits, sigma.reshape(1), x)


class _GmmMean(torch.autograd.Function):
    """sum_k softmax(logits)_k * loc_k (distributions.py:37-39), differentiable
    w.r.t. both operands; the backward of this inspection path is composed
    from device tensor ops."""

    @_fwd
    def forward(ctx, loc, ml):
        loc, ml = _c(loc), _c(ml)
        B, K, C, Cm, P = _gmm_dims(loc, ml)
        out = torch.empty(B, *loc.shape[2:], device=loc.device, dtype=loc.dtype)
        _lib.call("scae_gmm_mean_f32", _p(loc), _p(ml), _p(out), B, K, C, Cm, P,
                  _stream(loc))
        ctx.save_for_backward(loc, ml)
        return out

    @_bwd
    def backward(ctx, g):
        loc, ml = ctx.saved_tensors
        prob = torch.softmax(ml, 1)
        g = g.unsqueeze(1)
        g_loc = prob * g if ctx.needs_input_grad[0] else None
        g_ml = None
        if ctx.needs_input_grad[1]:
            t = g * loc                               # (B,K,C,P)
            if ml.shape[2] != loc.shape[2]:
                t = t.sum(2, keepdim=True)
            g_ml = prob * (t - (prob * t).sum(1, keepdim=True))
        return g_loc, g_ml


def gmm_mean(loc, mixing_logits):
    _need_hip(loc, mixing_logits)
    return _GmmMean.apply(loc, mixing_logits)


class _GmmMode(torch.autograd.Function):
    """loc of the component with the largest mixing log-prob
    (distributions.py:50-77 without the straight-through estimator):
    differentiable w.r.t. loc only -- the incoming gradient goes to the winning
    component -- exactly like the reference's sum(one_hot * loc)."""

    @_fwd
    def forward(ctx, loc, ml, sigma, maximum):
        loc, ml = _c(loc), _c(ml)
        B, K, C, Cm, P = _gmm_dims(loc, ml)
        out = torch.empty(B, *loc.shape[2:], device=loc.device, dtype=loc.dtype)
        _lib.call("scae_gmm_mode_f32", _p(loc), _p(ml), _p(_c(sigma)),
                  _p(out), int(maximum), B, K, C, Cm, P, _stream(loc))
        ctx.save_for_backward(ml)
        ctx.loc_shape = loc.shape
        return out

    @_bwd
    def backward(ctx, g):
        (ml,) = ctx.saved_tensors
        # (with `maximum` and a shared scale the added density is the same
        # for every component: the winner is the arg-max of the logits)
        idx = ml.argmax(1, keepdim=True)
        idx = idx.expand(ml.shape[0], 1, *ctx.loc_shape[2:])
        g_loc = torch.zeros(ctx.loc_shape, device=g.device, dtype=g.dtype)
        g_loc.scatter_(1, idx, g.unsqueeze(1))
        return g_loc, None, None, None


def gmm_mode(loc, mixing_logits, sigma, maximum=False):
    _need_hip(loc, mixing_logits, sigma)
    B, K, C, Cm, P = _gmm_dims(loc, mixing_logits)
    if maximum and Cm == 1 and C > 1:
        # same failure as the reference's in-place broadcast, distributions.py:65
        raise RuntimeError(f"output with shape {list(mixing_logits.shape)} "
                           f"doesn't match the broadcast shape "
                           f"{list(loc.shape)}")
    return _GmmMode.apply(loc, mixing_logits.detach(), sigma.detach(),
                          bool(maximum))
