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
def running(plan, node):
    """For autograd's backward thread: ``plan`` is current while ``node`` (a
    qualified method name) runs, and what ``node`` reads is launched first."""
    token = _CURRENT.set(plan)
    try:
        plan.enter(node)
        yield plan
    finally:
        _CURRENT.reset(token)
