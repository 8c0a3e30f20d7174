"""The launch plan of one training step.

At B = 128 an SCAE step is bound by its launch count:
independent kernels are merged into one launch by hand, which means that a
launch is often PARKED by the autograd node that owns it and carried by a later
node's launch.  This module is the only holder of such parked work.

* ``RIDES`` is the table of every such merge: which node parks what, which
  node's launch carries it, and which nodes read what it writes (it is
  launched on its own before any of those starts).  A new merge is a new row.
* ``StepPlan`` is one step's state: the fusion target, the likelihood offered
  to the trunk's launch, the parked launches, the queue of deferred column
  sums, the step prologue's buffers, the precision, the noise generators.
  ``train_step.TrainStep`` owns one; ops called outside any step share the
  process-wide ``ambient`` plan, which parks nothing unless a test opens
  ``ops.step_fusion`` on it.
* An op finds its plan through a context variable in the forward pass
  (``current()``) and keeps it on its autograd context; autograd's backward
  thread -- which does not inherit context variables -- is handed the node's
  own plan for the duration of the node (``ops._bwd``).  Two steps on two
  models therefore never see each other's parked work, in whatever order
  their forwards and backwards interleave.
"""
import collections
import contextlib
import contextvars
import weakref

ANY = "*"
_LIVE = weakref.WeakSet()     # every live StepPlan


class Ride:
    """One row of ``RIDES``.

    carriers: qualified names of the autograd-node methods whose launch can
              host the parked work (they ``take`` it);
    readers:  node methods that read what the parked launch writes -- or whose
              results autograd may add to it -- so the launch must have been
              issued before they start (``ANY``: every node but the carriers);
    scope:    the ``StepPlan`` scope whose end launches whatever nobody
              carried ("fusing" / "deferring"); "offer": a launch OFFERED ahead
              of its own node to an earlier node's launch (``StepPlan.offer`` /
              ``claim``) -- its node finds the result, or launches as usual;
    abi:      the merged C-ABI launcher (documentation; include/scae_hip.h)."""

    __slots__ = ("carriers", "readers", "scope", "abi", "what")

    def __init__(self, what, carriers, readers=(), scope="deferring", abi=""):
        self.what, self.carriers, self.readers = what, tuple(carriers), readers
        self.scope, self.abi = scope, abi


# Ordered: a row may read what an earlier row writes (the folding products'
# backward reads the attention's reduction), so flushes walk the table in order.
RIDES = collections.OrderedDict([
    ("tc_fwd", Ride(
        "the colour MLP's forward (K10)",
        carriers=("_PartEncoder.forward",), scope="offer",
        abi="scae_capsule_head_conv_fwd_tc_f32")),
    ("logprob_fwd", Ride(
        "the reconstruction likelihood's forward (K1)",
        carriers=("_SetEncoder.forward",), scope="offer",
        abi="scae_set_encoder_fwd_logprob_f32")),
    ("class_probs", Ride(
        "SCAE.forward's two classifier heads (one wave per image)",
        carriers=("_LossTail.forward",), scope="fusing",
        abi="scae_loss_tail_fwd_class_probs_f32")),
    ("combine", Ride(
        "the loss tail's batch combine (one workgroup)",
        carriers=("_LossTail.backward",), scope="fusing",
        abi="scae_loss_tail_bwd_f32 (scae_loss_extras.defer_combine)")),
    ("k1_bwd", Ride(
        "the reconstruction likelihood's backward (K1)",
        carriers=("_CapsuleLikelihood.backward",), readers=ANY,
        abi="scae_render_gmm_sums_bwd_likelihood_f32")),
    ("wgrads", Ride(
        "the capsule MLPs' four weight-gradient GEMMs",
        carriers=("_SeedAttention.backward",),
        abi="scae_seed_attention_mfma_bwd_gemm_f32")),
    ("tc_bwd", Ride(
        "the colour MLP's backward (K10)",
        carriers=("_PartEncoder.backward",),
        readers=("_CapsuleHead.backward",),
        abi="scae_capsule_head_bwd_tc_f32")),
    ("reduce", Ride(
        "the output attention's partial-row reduction",
        carriers=("_PartEncoder.backward", "_ConvStack.backward"),
        # (its one reader, _SeedFold.backward, takes it itself: it either parks
        # behind it -- the next row -- or launches it first)
        abi="scae_conv3x3_bwd_pair_reduce_f32")),
    ("fold_bwd", Ride(
        "the folding products' backward (K2d)",
        carriers=("_PartEncoder.backward", "_ConvStack.backward"),
        abi="scae_conv3x3_bwd_pair_fold_f32")),
])


# The second lane (round 6).  Below the loss tail a step's backward has two chains that meet
# again at the part-capsule head: the part decoder's likelihood backward (K1: 3 200
# workgroups of VALU work, ~50 us at cfg-2, needs nothing but the loss tail's seed) and the
# object path (capsule likelihood -> capsule-MLP chain -> output attention -> trunk: ~90 us
# of latency-bound launches of one workgroup per image / set / capsule block, whose input
# gradient for the special features the head's backward adds to K1's).  A forked HIP graph
# replays its branches one after the other on this stack, but kernels of two plain streams
# do overlap (tools/probes/stream_overlap.cpp), so a step that is re-issued launch by launch
# (train_step.TrainStep(replay="launches")) gives K1's backward a stream of its own: the
# launches and allocations of the nodes below happen on ``StepPlan.side_stream``.
SIDE_NODES = frozenset(("_RenderGmmLogProbSums.backward",))
# Plan-aware nodes of the main lane that do not read what a side node returned (the object
# path).  Any OTHER node that starts while the side lane is open -- the readers of K1's
# gradients: the template generator, the part-capsule head, the part encoder -- makes the
# main lane wait for the side lane first.
MAIN_NODES = frozenset((
    "_LossTail.backward", "_ClassProbs.backward", "_CapsuleLikelihood.backward",
    "_ChainVotes.backward", "_MLPChain.backward", "_GroupedMLP.backward",
    "_CapsuleVotes.backward", "_SeedAttention.backward", "_SetEncoder.backward",
    "_SeedFold.backward", "_PackParams.backward", "_Linear.backward",
    "_LayerNorm.backward", "_QKVAttention.backward"))


import os as _os
_DEBUG = bool(_os.environ.get("SCAE_LANES_DEBUG"))


def _note_order(later, earlier):
    """Tell the library's open launch recordings about a stream dependency the caller has
    just created (include/scae_hip.h, scae_launch_list_order)."""
    import ctypes
    from . import _lib
    _lib.load().scae_launch_list_order(ctypes.c_void_p(later.cuda_stream),
                                       ctypes.c_void_p(earlier.cuda_stream))


class StepPlan:
    def __init__(self, name="step", prologue=None):
        _LIVE.add(self)
        self.name = name
        self.target = None        # the loss's reconstruction target while fusing
        self.rider = None         # ops.LogProbRider offered to the trunk launch
        self.offers = {}          # kind -> launch offered to an earlier node's launch
        self.parked = {}          # kind -> parked launch (has .launch_alone())
        self.deferred = None      # queued column-sum units while deferring
        # the step's last column sums ride in the optimiser's launch (train_step.TrainStep
        # sets this; data_parallel.RMSpropFlat.step(sum_units=...) takes them): the outermost
        # ``deferring`` exit then HOLDS what is queued instead of launching it
        self.sums_to_optimizer = False
        self.held_sums = []
        self.prologue = prologue  # ops.StepPrologue or None
        self.bf16 = False         # configs[2]'s operand precision
        self.noise = {}           # (device, stream) -> (seed, generator state)
        self.side_stream = None   # torch.cuda.Stream of the second lane (None: one lane)
        # resident workgroups of the side lane's K1 backward (scae_decoder_desc.bwd_resident;
        # 0: one workgroup per (component, image) pair, which floods the chip)
        self.side_resident = int(_os.environ.get("SCAE_SIDE_RESIDENT", "512"))
        self._main = None         # the stream the side lane forked from, while it is open

    # -- the second lane ----------------------------------------------------
    @contextlib.contextmanager
    def side_lane(self, reads=()):
        """The calling node's launches and allocations go to the side stream, which first
        waits for everything the main stream has been given so far.  ``reads``: the
        tensors the node reads -- those allocated on the main stream are marked as in use
        by the side stream (inside a graph capture their memory is then not handed out
        again before the capture ends)."""
        import torch
        side = self.side_stream
        if side is None:
            yield
            return
        main = torch.cuda.current_stream(side.device)
        if main == side:          # (already there: a nested node)
            yield
            return
        if _DEBUG:
            print("[lanes] fork", flush=True)
        side.wait_stream(main)
        _note_order(side, main)
        self._main = main
        for t in reads:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(side)
        with torch.cuda.stream(side):
            yield

    def join_side(self):
        """The main stream waits for the side lane (before its first launch that reads what
        the side lane wrote; before the optimiser)."""
        import torch
        if self.side_stream is None or self._main is None:
            return
        if _DEBUG:
            import traceback
            print("[lanes] join from", traceback.extract_stack(limit=3)[0][2:4], flush=True)
        main, self._main = self._main, None
        cur = torch.cuda.current_stream(self.side_stream.device)
        if cur != self.side_stream:     # (called from the main lane: the usual case)
            main = cur
        main.wait_stream(self.side_stream)
        _note_order(main, self.side_stream)

    @property
    def side_open(self):
        return self._main is not None

    # -- who is current ---------------------------------------------------
    @contextlib.contextmanager
    def active(self):
        """Make this the plan the ops of the calling thread find."""
        token = _CURRENT.set(self)
        try:
            yield self
        finally:
            _CURRENT.reset(token)

    @property
    def fused(self):
        return self.target is not None

    @property
    def parking(self):
        """Parameter-gradient launches may wait: inside a fused step whose
        column sums wait too."""
        return self.target is not None and self.deferred is not None

    # -- parked launches --------------------------------------------------
    def offer(self, kind, work):
        assert RIDES[kind].scope == "offer", kind
        self.offers[kind] = work

    def claim(self, kind):
        return self.offers.pop(kind, None)

    def park(self, kind, work):
        assert kind in RIDES and RIDES[kind].scope != "offer", kind
        stale = self.parked.pop(kind, None)
        if stale is not None:        # (never within one step of one model)
            stale.launch_alone()
        self.parked[kind] = work

    def take(self, kind):
        return self.parked.pop(kind, None)

    def holds(self, kind, work=None):
        held = self.parked.get(kind)
        return held is not None and (work is None or held is work)

    def flush(self, *kinds):
        """Launch, in table order, what is parked of ``kinds`` (all: none
        given)."""
        for kind in RIDES:
            if (not kinds or kind in kinds) and kind in self.parked:
                self.parked.pop(kind).launch_alone()

    def flush_scope(self, scope):
        self.flush(*[k for k, r in RIDES.items() if r.scope == scope])

    def enter(self, node):
        """Called at the top of every plan-aware autograd-node method: what is
        parked and read by ``node`` is launched first."""
        if not self.parked:
            return
        for kind in RIDES:
            if kind not in self.parked:
                continue
            ride = RIDES[kind]
            if node in ride.carriers:
                continue
            if ride.readers == ANY or node in ride.readers:
                self.parked.pop(kind).launch_alone()

    # -- deferred column sums ----------------------------------------------
    def flush_sums(self):
        if self.deferred:
            self.join_side()     # (the units' partial matrices may be the side lane's)
            units = list(self.deferred)
            del self.deferred[:]
            from . import ops
            ops._launch_sum_units(units)

    def take_held_sums(self):
        units, self.held_sums = self.held_sums, []
        return units

    # -- scopes -------------------------------------------------------------
    @contextlib.contextmanager
    def fusing(self, target):
        """The forward + loss (+ backward) of one step whose loss will take
        ``target`` as its reconstruction target (None: nothing is fused)."""
        prev = (self.target, self.rider)
        self.target, self.rider = target, None
        self.parked.clear()      # (a backward that raised may have left some)
        self.offers.clear()
        ok = False
        try:
            yield self
            ok = True
        finally:
            self.target, self.rider = prev
            self.offers.clear()
            if ok:
                self.flush_scope("fusing")   # (a block without the fused tail)

    @contextlib.contextmanager
    def deferring(self):
        """Column sums that only produce parameter gradients are queued; the
        outermost exit launches what is still parked, then all of them."""
        outer = self.deferred
        if outer is None:
            self.deferred = []
        ok = False
        try:
            yield self
            ok = True
        finally:
            if outer is None:
                try:
                    if ok:
                        self.join_side()
                        self.flush_scope("deferring")
                        if self.sums_to_optimizer:
                            self.held_sums.extend(self.deferred)
                            del self.deferred[:]
                        else:
                            self.flush_sums()
                finally:
                    self.deferred = None

    @contextlib.contextmanager
    def with_prologue(self, pro):
        prev, self.prologue = self.prologue, pro
        try:
            yield pro
        finally:
            self.prologue = prev

    @contextlib.contextmanager
    def precision(self, bf16):
        prev, self.bf16 = self.bf16, bool(bf16)
        try:
            yield self
        finally:
            self.bf16 = prev


def live_plans():
    """Every StepPlan that is still alive (``ops.reset_noise`` restarts the noise
    generators of all of them)."""
    return list(_LIVE)


_CURRENT = contextvars.ContextVar("scae_step_plan", default=None)
ambient = StepPlan("ambient")


def current():
    plan = _CURRENT.get()
    return ambient if plan is None else plan


@contextlib.contextmanager
def running(plan, node, reads=()):
    """For autograd's backward thread: ``plan`` is current while ``node`` (a
    qualified method name) runs, and what ``node`` reads is launched first.  A
    node of the object path's backward (``SIDE_NODES``) runs on the plan's side
    stream when it has one; ``reads``: the tensors it reads."""
    token = _CURRENT.set(plan)
    try:
        plan.enter(node)
        if plan.side_stream is not None and node in SIDE_NODES:
            with plan.side_lane(reads):
                yield plan
        else:
            if plan.side_open and node not in MAIN_NODES:
                if _DEBUG:
                    print(f"[lanes] join before {node}", flush=True)
                plan.join_side()
            yield plan
    finally:
        _CURRENT.reset(token)
