"""One SCAE training step = forward + SCAE.loss + backward (+ gradient
all-reduce + optimiser), optionally captured once into a HIP graph and
replayed: at B=128 the step is launch-latency bound, so
removing the per-launch host cost matters more than any single kernel.

Mirrors the semantics of the reference's BaseExperiment.training_step
(torch_scae_experiments/base_experiment.py:109-126)."""
import contextlib
import os

import torch
import torch.distributed as dist

from . import ops
from .data_parallel import (FlatParameters, RMSpropFlat, all_reduce_gradients,
                            broadcast_parameters, world)


# parameters whose gradients are final once backward has come down through
# the two decoders (the object decoder's capsule MLPs alone are 66 % of the
# model): the first all-reduce bucket
EARLY_PREFIXES = ("obj_decoder.", "part_decoder.", "prior_classifier.",
                  "posterior_classifier.")


def _side_stream(device, spec):
    """The second lane's stream.  ``spec`` True: a stream of torch's pool; an int n (or the
    string "n" / "n:stride"): a stream restricted to n compute units
    (hipExtStreamCreateWithCUMask; every stride-th CU of the mask's numbering) -- the side
    lane's kernel then leaves the other CUs to the main lane's launches."""
    if spec is True:
        return torch.cuda.Stream(device)
    import ctypes
    n, _, stride = str(spec).partition(":")
    n, stride = int(n), int(stride or 1)
    total = torch.cuda.get_device_properties(device).multi_processor_count
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    cu, taken = 0, 0
    while taken < min(n, total):
        if not mask[cu // 32] >> (cu % 32) & 1:
            mask[cu // 32] |= 1 << (cu % 32)
            taken += 1
        cu = (cu + stride) % total
        if stride > 1 and cu < stride and mask[cu // 32] >> (cu % 32) & 1:
            cu += 1        # (wrapped onto a taken CU: shift the comb)
    hip = ctypes.CDLL("libamdhip64.so")
    st = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), words, mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed with {rc}")
    return torch.cuda.ExternalStream(st.value, device=device)


class TrainStep:
    """collective modes (world > 1, or ``force_collective`` in a 1-rank group):

    * ``"2 buckets"`` (default when the model exposes the backward cut): the
      step is captured as TWO HIP graphs sharing one memory pool -- A: forward,
      loss, backward down to the inputs of the two decoders; B: the backward of
      the object encoder, template generator and part encoder.  The decoders'
      gradient block (flat[:n_front]) is all-reduced on RCCL's stream while B
      replays; the rest after B; then the fused RMSprop.  (Branches INSIDE one
      replayed graph serialise on this stack, DESIGN.md section 5, hence two
      graphs with the collective between them.)
    * ``"1 bucket"``: one graph, one all-reduce of the whole flat buffer after
      it (``overlap=False``, or a model without the cut).
    * ``"in graph"`` (``collective_mode="in graph"`` or env
      ``SCAE_GRAPH_ALLREDUCE=1``): the all-reduce and the RMSprop step captured
      inside the one graph -- no host-side stream hand-off at all, nothing
      overlapped.
    ``collective_mode``: one of the three names (``"2 buckets"`` falls back to
    ``"1 bucket"`` for a model without the backward cut), ``"off"`` (no
    collective even with ranks: measurements only -- the ranks' parameters
    drift apart) or None (two buckets where possible).  ``bench.py`` measures
    all three on the ranks it is given and takes the fastest.
    """

    MODES = ("2 buckets", "1 bucket", "in graph")

    def __init__(self, model, batch_size, image_shape, lr=3e-5, use_graph=True,
                 optimizer=True, momentum=0.9, weight_decay=0.0,
                 lr_decay_rate=0.997, autocast_dtype=None,
                 force_collective=False, overlap=True, lazy_render=True,
                 prologue=True, fuse_kernels=True, collective_mode=None,
                 replay="graph", two_lanes=False):
        self.model = model
        self.device = next(model.parameters()).device
        self.world = world()[1]
        if collective_mode is None and \
                os.environ.get("SCAE_GRAPH_ALLREDUCE", "0") == "1":
            collective_mode = "in graph"
        if collective_mode not in (None, "off") + self.MODES:
            raise ValueError(f"collective_mode must be one of {self.MODES}, "
                             f"'off' or None, got {collective_mode!r}")
        self.collective = collective_mode != "off" and (
            self.world > 1 or (force_collective
                               and torch.distributed.is_initialized()))
        self.in_graph_collective = self.collective and use_graph and \
            collective_mode == "in graph"
        overlap = overlap and collective_mode in (None, "2 buckets")
        self.split = bool(self.collective and overlap
                          and not self.in_graph_collective
                          # the conditions under which SCAE._forward cuts
                          # the backward in two (res._phase_cut)
                          and getattr(model, "stop_grad_caps_target", False)
                          and getattr(model, "vote_type", None) == "enc"
                          and getattr(model, "presence_type", None) == "enc")
        self.flat = FlatParameters(
            model, front=(lambda n: n.startswith(EARLY_PREFIXES))
            if self.split else None)
        if self.split and self.flat.n_front in (0, self.flat.numel):
            self.split = False
        if hasattr(model, "split_backward"):
            model.split_backward = self.split
        # loss-only step: the (B, M+1, ., H, W) reconstruction tensors, which
        # neither SCAE.loss nor its backward read, render on first access.
        # Scoped to the step's own forward (_lazy): the user's model keeps
        # returning plain AttrDicts outside it
        dec = getattr(model, "part_decoder", None)
        self._lazy_dec = dec if lazy_render and hasattr(dec, "lazy_render") \
            else None
        # the step's noise draws and parameter-only folding products ride with
        # the batch hand-over in ONE launch ahead of the step (ops.StepPrologue)
        self._pro = ops.StepPrologue() if prologue and \
            self.device.type == "cuda" else None
        # this step's launch plan (step_plan.py): the only holder of its
        # parked launches, deferred column sums, prologue buffers and noise
        # generators -- nothing of a step lives in module state, so several
        # steps (models) can interleave in one process
        self.plan = ops.StepPlan("train step", prologue=self._pro)
        self.collective_mode = None if not self.collective else \
            "in graph" if self.in_graph_collective else \
            "2 buckets, the first overlapping the encoder backward" \
            if self.split else "1 bucket after the backward"
        broadcast_parameters(self.flat)
        # eps = 1e-2 / bs**2 as in configs/optimizer/rmsprop.yaml
        self.opt = RMSpropFlat(self.flat, lr=lr, momentum=momentum,
                               eps=1e-2 / float(batch_size) ** 2,
                               weight_decay=weight_decay) \
            if optimizer else None
        self.lr_decay_rate = lr_decay_rate
        # the backward's last column sums (parameter gradients only) ride in the optimiser's
        # launch: one launch less on the step's dependent chain.  Not with a collective (the
        # all-reduce reads the finished gradient buffer first) nor with weight decay.
        self.plan.sums_to_optimizer = bool(
            self.opt is not None and fuse_kernels and not self.collective
            and weight_decay == 0 and self.device.type == "cuda")
        # torch.bfloat16 (BASELINE.json configs[2]): the GEMM-shaped kernels --
        # K8 convolutions, K7 capsule-MLP / 1x1-conv GEMMs, forward and backward
        # -- take bf16 operands with fp32 accumulation (ops.mfma_bf16); the
        # mixture likelihood, the capsule likelihood, the fused object encoder
        # and all reductions stay fp32
        if autocast_dtype not in (None, torch.bfloat16):
            raise ValueError("autocast_dtype must be None or torch.bfloat16")
        self.autocast_dtype = autocast_dtype
        self.log = None          # device tensors of the last step's log dict
        self.image = torch.zeros(batch_size, *image_shape, device=self.device)
        self.label = torch.zeros(batch_size, dtype=torch.long,
                                 device=self.device)
        self.loss = torch.zeros((), device=self.device)
        self._one = torch.ones((), device=self.device)   # d loss / d loss
        self.use_graph = use_graph
        # how a captured step is re-issued: "graph" (hipGraphLaunch: ~10 us of host time and
        # ~8.6 us of device time between two replays, tools/graph_gap_probe.py) or "launches"
        # (the library's record of the captured launches -- scae_launch_list_run: a
        # hipLaunchKernel each on the current stream, no per-replay device cost: 0-5 us per
        # step at cfg-2 depending on the host -- ~18 us of host time per launch leave little
        # room beside a 550 us step --, tools/launch_list_probe.py; single-rank steps only,
        # and only when the captured graph holds nothing but those launches: otherwise --
        # e.g. training_step()'s log outputs, computed by torch kernels -- the graph replays)
        if replay not in ("graph", "launches"):
            raise ValueError("replay must be 'graph' or 'launches'")
        self.replay = replay
        # replay="launches", two_lanes: the part decoder's likelihood backward (K1) on a
        # second stream beside the object path's backward (step_plan.SIDE_NODES): kernels of
        # two plain streams do overlap on this stack (branches of a replayed graph do not),
        # and the recorded list re-issues the two lanes with their fork / join edges.
        # Measured at cfg-2 (profiles/r06/lanes.txt): the lanes overlap and the step gets
        # SLOWER -- 0.540 - 0.558 against 0.5205 ms: beside K1's workgroups the capsule-MLP
        # chain runs 57 - 68 us instead of 31, the attention's backward 32 instead of 22;
        # what the object path's launches leave idle is not what K1 needs.  Not the default.
        self.plan.side_stream = _side_stream(self.device, two_lanes) if (
            two_lanes and replay == "launches" and not self.collective
            and self.device.type == "cuda") else None
        self._launches = None
        self._klist = None
        self.graph_nodes = None  # (graph nodes, kernel nodes, recorded launches) of a capture
        # independent kernels of the step sharing launches (ops.step_fusion:
        # the reconstruction likelihood rides with the object encoder's trunk)
        self.fuse_kernels = fuse_kernels
        self.skip_collective = False
        self._capturing = False
        self._stream = None
        self._with_log = False
        self.graph = None        # the whole step, or its part A when split
        self.graph_b = None
        self._cut = None

    @contextlib.contextmanager
    def _lazy(self):
        dec = self._lazy_dec
        if dec is None:
            yield
            return
        prev, dec.lazy_render = dec.lazy_render, True
        try:
            yield
        finally:
            dec.lazy_render = prev

    # -- the step in two parts ------------------------------------------------
    def _part_a(self):
        """forward + loss + backward (split: down to the decoders' inputs)."""
        plan = self.plan
        stale = plan.take_held_sums()
        if stale and not self._capturing:
            # (a backward nobody followed by the optimiser; inside a capture they are the
            # warm-ups' -- whose gradients nobody reads -- and are dropped: launched here they
            # would become a launch of every replay)
            ops._launch_sum_units(stale)
        self.flat.clear_grads()
        with plan.active(), plan.precision(self.autocast_dtype is not None), \
                self._lazy(), \
                plan.fusing(self.image if self.fuse_kernels else None):
            res = self.model(self.image)
            loss, info = self.model.loss(res, self.image, self.label)
            # a resident seed: no ones_like fill per step; the column sums that
            # only produce parameter gradients wait for ONE launch at the end
            with plan.deferring():
                loss.backward(self._one)
        self._cut = res.get("_phase_cut") if self.split else None
        self.flat.gather_grads(None if self._cut is None else 0)
        if self._capturing:
            # the captured loss tensor lives in the graph's private pool at a
            # fixed address: expose it instead of copying it out every replay
            self.loss = loss.detach()
        else:
            self.loss.copy_(loss.detach())
        if self._with_log:
            # the `log` dict of BaseExperiment.training_step (:118-125)
            acc = self.model.calculate_accuracy(res, self.label) \
                if self.model.n_classes is not None else None
            fresh = dict(loss=loss.detach(), **{k: v.detach()
                                                for k, v in info.items()})
            if acc is not None:
                fresh["accuracy"] = acc.detach()
            if self.log is None:
                self.log = {k: torch.zeros_like(v) for k, v in fresh.items()}
            for k, v in fresh.items():
                self.log[k].copy_(v)

    def _part_b(self):
        """split only: the backward below the decoders' inputs."""
        if self._cut is None:
            return
        srcs, leaves = self._cut
        self._cut = None
        keep = [(t, l.grad) for t, l in zip(srcs, leaves)
                if l.grad is not None]
        plan = self.plan
        with plan.active(), plan.precision(self.autocast_dtype is not None), \
                plan.deferring():
            torch.autograd.backward([t for t, _ in keep],
                                    [g for _, g in keep])
        self.flat.gather_grads(1)

    def _fwd_bwd(self):
        self._part_a()
        self._part_b()

    def _reduce(self, which=None, async_op=False):
        # SUM all-reduce; the 1/world scale rides in the optimiser kernel
        if self.skip_collective:      # measurement only (bench.py's comm leg)
            return None
        return all_reduce_gradients(self.flat, average=self.opt is None,
                                    which=which, async_op=async_op,
                                    force=self.collective)

    def _finish(self):
        if self.collective:
            self._reduce()
        if self.opt is not None:
            self.opt.step(grad_scale=1.0 / self.world,
                          sum_units=self.plan.take_held_sums())

    def _run(self, part_a, part_b):
        """One step from its two parts (graph replays or eager calls)."""
        if not self.split:
            part_a()
            return
        part_a()
        w0 = self._reduce(0, async_op=True)     # overlaps part B
        part_b()
        w1 = self._reduce(1, async_op=True)
        for w in (w0, w1):
            if w is not None:
                w.wait()                        # stream-side wait, not a host one

    def _capture(self):
        # warm up on a side stream (allocator, lazy init), then capture; the
        # same stream for every (re-)capture: autograd keeps each parameter's
        # AccumulateGrad node, and with it the stream it first ran on
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        s = self._stream
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                self._refresh_prologue()
                self._fwd_bwd()
            self._refresh_prologue()    # what the capture below consumes
        torch.cuda.current_stream().wait_stream(s)
        # capture on the SAME stream the warm-up ran on: autograd caches each
        # parameter's AccumulateGrad node together with its stream
        # (keep_graph: the captured hipGraph_t stays readable -- _graph_is_only_launches)
        try:
            self.graph = torch.cuda.CUDAGraph(keep_graph=self.replay == "launches")
        except TypeError:            # (a torch without keep_graph: graph replay only)
            self.graph = torch.cuda.CUDAGraph()
        # With a process group alive its watchdog thread polls HIP events at
        # any time; under the default "global" capture mode such a poll during
        # the capture is an error that aborts the process.  Then (and only
        # then) the check is narrowed to the capturing thread.
        mode = "thread_local" if dist.is_available() and dist.is_initialized() \
            else "global"
        import ctypes
        from . import _lib
        lib = _lib.load()
        self._free_list()
        # the library's record of the launches this capture issues: bound to the capturing
        # stream, so another step's (or an eager forward's) launches on other streams are
        # not in it (include/scae_hip.h, launch lists)
        klist = lib.scae_launch_list_begin(ctypes.c_void_p(s.cuda_stream))
        if klist and self.plan.side_stream is not None:
            lib.scae_launch_list_side_stream(
                klist, ctypes.c_void_p(self.plan.side_stream.cuda_stream))
        self._capturing = True
        ok = False
        try:
            with torch.cuda.graph(self.graph, stream=s, capture_error_mode=mode), \
                    _lib.recorder() as launches:
                self._part_a()
                if not self.split:
                    self._part_b()
                    if not self.collective or self.in_graph_collective:
                        self._finish()
            if klist:
                lib.scae_launch_list_end(klist)
            # The step as a plain list of kernel launches (scae_launch_list_*: kernel, grid,
            # block, LDS, argument bytes), re-issued by replay_launches() -- but only when the
            # graph holds exactly these: a captured torch kernel (training_step's accuracy
            # and log copies), a memset inside a launcher or a collective is a node of the
            # graph that the list does not have, and replaying the list would silently drop
            # it.  `launches` (the C-ABI calls with their ctypes arguments) keeps the
            # buffers the list points into alive.
            self._launches = launches if not self.collective else None
            if klist and not self.collective and not self.split \
                    and self.replay == "launches" \
                    and self._graph_is_only_launches(lib.scae_launch_list_size(klist)):
                self._klist, klist = klist, None
            if self.split:
                # part B allocates from part A's pool: the tensors A left for it
                # (saved activations, the cut gradients) are alive across the two
                # captures, and the graphs are always replayed A, B, A, B, ...
                self.graph_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_b, stream=s,
                                      pool=self.graph.pool(),
                                      capture_error_mode=mode):
                    self._part_b()
            ok = True
        finally:
            # (a capture that raised -- out of memory, an op that cannot be captured -- must
            # not leave the step in capture mode: eager calls would drop held column sums
            # and alias self.loss to a pool tensor)
            self._capturing = False
            if klist:
                lib.scae_launch_list_free(klist)
            if not ok:
                self.graph = self.graph_b = None
                self._launches = None
                self._free_list()

    def _graph_is_only_launches(self, n_launches):
        """True when the captured graph's nodes are exactly ``n_launches`` kernel nodes (what
        the library recorded): only then is the launch list the whole step.  Anything that
        cannot be verified counts as a mismatch (the step then replays its graph)."""
        import ctypes
        try:
            raw = self.graph.raw_cuda_graph()
            hip = ctypes.CDLL("libamdhip64.so")
            n = ctypes.c_size_t(0)
            if hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n)) != 0:
                return False
            nodes = (ctypes.c_void_p * max(1, n.value))()
            if hip.hipGraphGetNodes(ctypes.c_void_p(raw), nodes, ctypes.byref(n)) != 0:
                return False
            kernels = other = 0
            for i in range(n.value):
                t = ctypes.c_int(-1)
                if hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t)) != 0:
                    return False
                kernels += t.value == 0          # hipGraphNodeTypeKernel
                # (what a forked capture adds carries no work: empty / event nodes -- the
                # list re-issues the stream edges from its own record of them)
                other += t.value not in (0, 5, 6, 7)   # ... Empty, WaitEvent, EventRecord
            self.graph_nodes = (n.value - (n.value - kernels - other), kernels, n_launches)
            return other == 0 and kernels == n_launches
        except Exception:       # (no raw graph in this torch build, no HIP runtime handle)
            return False

    def replay_launches(self):
        """The captured step re-issued launch by launch on the current stream instead of as a
        graph replay (single-rank steps): no per-replay graph cost on the device, one C call
        (a hipLaunchKernel per recorded launch) on the host."""
        import ctypes
        from . import _lib
        side = self.plan.side_stream
        _lib.call("scae_launch_list_run2", self._klist, ctypes.c_void_p(
            torch.cuda.current_stream(self.device).cuda_stream),
            None if side is None else ctypes.c_void_p(side.cuda_stream))

    def _free_list(self):
        if getattr(self, "_klist", None):
            from . import _lib
            _lib.load().scae_launch_list_free(self._klist)
        self._klist = None

    def __del__(self):
        try:
            self._free_list()
        except Exception:      # (interpreter shutdown)
            pass

    def capture(self):
        """Build the step's HIP graph(s) now instead of at the first call
        (no-op without ``use_graph`` or when already built).  Runs the
        forward / backward warm-ups on whatever the resident input buffers
        hold; parameters and optimiser state are not touched."""
        if self.use_graph and self.graph is None:
            self._capture()
            self._refresh_prologue()   # the capture consumed the last one

    def prepare(self, image, label):
        """Stage a batch and build the step's graph(s) WITHOUT stepping: what
        a measurement does before it snapshots the state it wants to time."""
        self._stage(image, label)
        self.capture()

    def snapshot(self):
        """The training state this step advances -- parameters, the
        optimiser's two moment buffers and its learning rate -- as clones (a
        few device copies; not for use inside a timed region).  ``restore``
        puts it back, under an already captured graph too: the graph reads
        and writes the same flat buffers, and everything derived from the
        parameters (folding products, filter re-layouts) is recomputed by
        every step's own prologue / graph."""
        snap = {"param": self.flat.flat_param.clone()}
        if self.opt is not None:
            snap.update(square_avg=self.opt.square_avg.clone(),
                        buf=self.opt.buf.clone(), lr=self.opt.lr)
        return snap

    @torch.no_grad()
    def restore(self, snap):
        self.flat.flat_param.copy_(snap["param"])
        if self.opt is not None:
            self.opt.square_avg.copy_(snap["square_avg"])
            self.opt.buf.copy_(snap["buf"])
            if self.opt.lr != snap["lr"]:
                self.opt.set_lr(snap["lr"])

    def _refresh_prologue(self):
        """Noise + folding products for the next forward (no batch)."""
        if self._pro is not None:
            with self.plan.active():
                self._pro.launch(stream_ref=self.image)

    def _stage(self, image, label):
        """The batch into the resident input buffers: one launch when both
        tensors already live on the device in the buffers' layout -- the
        step's prologue launch when there is one."""
        direct = image.is_cuda and label.is_cuda \
            and image.dtype == self.image.dtype \
            and label.dtype == self.label.dtype and image.is_contiguous() \
            and label.is_contiguous() and image.shape == self.image.shape \
            and label.shape == self.label.shape \
            and image.device == self.device == label.device
        if self._pro is not None:
            if direct:
                with self.plan.active():
                    self._pro.launch(self.image, image, self.label, label)
            else:
                self.image.copy_(image, non_blocking=True)
                self.label.copy_(label, non_blocking=True)
                self._refresh_prologue()
            return
        if image.is_cuda and label.is_cuda and image.dtype == self.image.dtype \
                and label.dtype == self.label.dtype and image.is_contiguous() \
                and label.is_contiguous() and image.shape == self.image.shape \
                and label.shape == self.label.shape \
                and image.device == self.device == label.device:
            import ctypes
            from . import _lib
            P = ctypes.c_void_p
            _lib.call("scae_stage_batch", P(self.image.data_ptr()),
                      P(image.data_ptr()), image.numel(),
                      P(self.label.data_ptr()), P(label.data_ptr()),
                      label.numel(),
                      P(torch.cuda.current_stream(self.device).cuda_stream))
            return
        self.image.copy_(image, non_blocking=True)
        self.label.copy_(label, non_blocking=True)

    def __call__(self, image, label):
        """image / label may be device tensors; copied into the static inputs."""
        self._stage(image, label)
        if self.use_graph:
            self.capture()
            if self.split:
                self._run(self.graph.replay, self.graph_b.replay)
                if self.opt is not None:
                    self.opt.step(grad_scale=1.0 / self.world)
            else:
                if self.replay == "launches" and self._klist:
                    self.replay_launches()
                else:
                    self.graph.replay()
                if self.collective and not self.in_graph_collective:
                    self._finish()
        elif self.split:
            self._run(self._part_a, self._part_b)
            if self.opt is not None:
                self.opt.step(grad_scale=1.0 / self.world)
        else:
            self._fwd_bwd()
            self._finish()
        return self.loss

    def training_step(self, image, label):
        """-> {'loss': tensor, 'log': {...}} like BaseExperiment.training_step
        (base_experiment.py:109-126); the log values are device tensors that
        the next call overwrites.  Builds the step with the log outputs on
        first use (costs a few extra small kernels per step)."""
        if not self._with_log:
            self._with_log, self.graph, self.graph_b = True, None, None
        loss = self(image, label)
        return dict(loss=loss, log=self.log)

    def end_epoch(self):
        """Per-epoch ExponentialLR step (base_experiment.py:73-76)."""
        if self.opt is not None and self.lr_decay_rate:
            self.opt.decay_lr(self.lr_decay_rate)
