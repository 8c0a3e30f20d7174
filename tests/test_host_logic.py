"""CPU-only checks of the host side: the C-ABI library loads and exports every
symbol the header declares, the nn.Module surface / checkpoint schema match
the reference, and the product refuses to run without a HIP device."""
import os
import re

import numpy as np
import pytest
import torch

from tests.golden_util import load, model_names, sub

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from torch_scae_amd import _lib
    header = open(os.path.join(ROOT, "include", "scae_hip.h")).read()
    declared = set(re.findall(r"\b(scae_[a-z0-9_]+)\s*\(", header))
    declared -= {"scae_decoder_desc"}
    assert declared, "no declarations found"
    assert declared == set(_lib.SIGNATURES), \
        declared.symmetric_difference(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.scae_abi_version() == _lib.ABI_VERSION == 2
    assert lib.scae_conv3x3_wf_floats(128, 128) == 5 * 128 * 9 * 128 // 2   # (+ the packed planes)
    assert lib.scae_conv3x3_wf_floats(8, 8) == 8 * 9 * 8
    assert lib.scae_conv3x3_wf_floats(0, 8) == 0
    assert b"limits" in lib.scae_error_string(-2)


def test_launch_list_bookkeeping_without_a_gpu():
    """scae_launch_list_*: a recording is a handle bound to a stream -- several may be open at
    once (two steps capturing in one process), each is ended and freed on its own; an empty
    recording is a list of size 0, running it is a no-op (no HIP call is made), an open one
    refuses to run."""
    import ctypes
    from torch_scae_amd import _lib
    lib = _lib.load()
    P = ctypes.c_void_p
    a = lib.scae_launch_list_begin(P(0x1000))           # (stream handles are only compared)
    b = lib.scae_launch_list_begin(P(0x2000))
    assert a and b and a != b
    assert lib.scae_launch_list_run(P(a), None) == -1   # still recording
    assert lib.scae_launch_list_end(P(a)) == 0
    assert lib.scae_launch_list_end(P(a)) == -1         # not open any more
    assert lib.scae_launch_list_size(P(a)) == 0
    assert lib.scae_launch_list_run(P(a), None) == 0
    lib.scae_launch_list_free(P(a))
    lib.scae_launch_list_free(P(b))                     # (free also ends an open recording)
    c = lib.scae_launch_list_begin(None)                # the null stream is a stream too
    assert c and lib.scae_launch_list_end(P(c)) == 0
    lib.scae_launch_list_free(P(c))
    assert lib.scae_launch_list_end(None) == -1
    assert lib.scae_launch_list_run(None, None) == -1
    assert lib.scae_launch_list_size(None) == 0


def test_launch_list_lanes_and_order_edges_without_a_gpu():
    """A recording with a side stream: stream-order edges the caller reports
    (scae_launch_list_order) are noted by the recordings that hold BOTH streams, an edge
    that is already implied (same edge, nothing given to the earlier lane since) is not
    noted twice, and a list without launches runs as a no-op on one stream or two."""
    import ctypes
    from torch_scae_amd import _lib
    lib = _lib.load()
    P = ctypes.c_void_p
    a = lib.scae_launch_list_begin(P(0x10))
    b = lib.scae_launch_list_begin(P(0x30))             # knows nothing of 0x20
    assert lib.scae_launch_list_side_stream(P(a), P(0x20)) == 0
    assert lib.scae_launch_list_side_stream(P(a), P(0x40)) == -1   # one side lane per list
    assert lib.scae_launch_list_side_stream(P(b), P(0x30)) == -1   # not its own stream
    assert lib.scae_launch_list_order(P(0x20), P(0x10)) == 1       # side waits for main: a
    assert lib.scae_launch_list_order(P(0x20), P(0x10)) == 1       # (implied: not noted again)
    assert lib.scae_launch_list_order(P(0x10), P(0x20)) == 1       # main waits for side
    assert lib.scae_launch_list_order(P(0x20), P(0x30)) == 0       # no list holds both
    assert lib.scae_launch_list_end(P(a)) == 0 and lib.scae_launch_list_end(P(b)) == 0
    assert lib.scae_launch_list_order(P(0x20), P(0x10)) == 0       # nothing open
    assert lib.scae_launch_list_size(P(a)) == 0 and lib.scae_launch_list_side_size(P(a)) == 0
    assert lib.scae_launch_list_lane(P(a), 0) == -1
    assert lib.scae_launch_list_run2(P(a), None, None) == 0        # (one stream: edges skipped)
    assert lib.scae_launch_list_side_stream(P(a), P(0x50)) == -1   # closed
    lib.scae_launch_list_free(P(a))
    lib.scae_launch_list_free(P(b))


def test_bf16_resident_conv_entry_points_reject_bad_arguments_without_a_gpu():
    """csrc/conv_bf16.hip: shape predicate, split rule and argument checks happen before any
    HIP call."""
    import ctypes
    from torch_scae_amd import _lib
    lib = _lib.load()
    P = ctypes.c_void_p
    d = P(0x1000)
    assert lib.scae_conv3x3_bf16r_supported(1024, 19, 19, 128, 128, 2) == 1
    assert lib.scae_conv3x3_bf16r_supported(128, 9, 9, 256, 128, 1) == 1
    assert lib.scae_conv3x3_bf16r_supported(128, 9, 9, 64, 128, 1) == 0      # channels % 128
    assert lib.scae_conv3x3_bf16r_supported(128, 9, 9, 128, 128, 3) == 0     # stride
    assert lib.scae_conv3x3_bf16r_supported(128, 2, 9, 128, 128, 1) == 0     # no output row
    assert lib.scae_conv3x3_bf16r_supported(1 << 16, 40, 40, 128, 128, 1) == 0   # >= 2 GiB
    assert lib.scae_conv3x3_fwd_bf16r(None, d, d, d, None, None, None, 4, 9, 9, 128, 128, 1,
                                      None) == -1
    assert lib.scae_conv3x3_fwd_bf16r(d, d, d, d, None, None, d, 4, 9, 9, 128, 128, 1,
                                      None) == -1      # out_post without its bias
    assert lib.scae_conv3x3_fwd_bf16r(d, d, d, d, None, None, None, 4, 9, 9, 64, 128, 1,
                                      None) == -2
    assert lib.scae_conv3x3_dgrad_bf16r(d, d, None, None, None, 4, 9, 9, 128, 128, 1, None) == -1
    assert lib.scae_conv3x3_wgrad_bf16r(d, None, d, 4, 9, 9, 128, 128, 1, None) == -1
    # two workgroups per CU over the 9 taps, at least four 64-pixel chunks each
    assert lib.scae_conv3x3_wgrad_bf16r_splits(1024, 9, 9, 128, 128) == 56
    assert lib.scae_conv3x3_wgrad_bf16r_splits(2, 7, 7, 128, 128) == 1
    assert lib.scae_conv3x3_wgrad_bf16r_splits(2, 7, 7, 64, 128) == 0
    assert lib.scae_cvt_bf16_batch(0, None, None, None, None) == -1
    assert lib.scae_cvt_bf16_batch(1, (P * 1)(0x1000), (P * 1)(0x2000),
                                   (ctypes.c_int64 * 1)(12), None) == -1     # n % 8


def test_sum_jobs_reject_a_ragged_periodic_window():
    """a periodic segment over a column count that is not a multiple of its period
    has no well-defined destination range (the optimiser launch derives the elements a sum
    workgroup owns from cols / period): rejected before any HIP call."""
    import ctypes
    from torch_scae_amd import _lib
    lib = _lib.load()
    seg = (_lib.SumSegment * 1)(_lib.SumSegment(ctypes.c_void_p(0x1000), 0, 4, 8))
    assert lib.scae_sum_rows_f32(ctypes.c_void_p(0x2000), 4, 20, seg, 1, None) == -1
    assert lib.scae_sum_rows_f32(None, 4, 24, seg, 1, None) == -1      # (null source)


def test_launchers_reject_bad_arguments_without_a_gpu():
    """Argument validation happens before any HIP call."""
    from torch_scae_amd import _lib
    lib = _lib.load()
    assert lib.scae_geometric_transform_fwd_f32(None, None, 4, 0, 1, 0,
                                                None) == -1
    # sets beyond the matrix-core tiles: the general kernels take them (fp32);
    # the bf16 forward keeps the tile limit
    assert lib.scae_qkv_attention_fwd_f32(None, None, None, None, None, None,
                                          1, 65, 4, 4, 4, 2.0, None) == -1
    assert lib.scae_qkv_attention_fwd_bf16(None, None, None, None, None, None,
                                           1, 65, 4, 4, 4, 2.0, None) == -2
    d = _lib.DecoderDesc()
    assert lib.scae_template_render_fwd_f32(d, None, None, None) == -1
    # the launch-merging entry points: null pointers / empty job lists / bad sizes
    assert lib.scae_qkv_attention_fwd_bf16(None, None, None, None, None, None,
                                           1, 4, 4, 4, 4, 2.0, None) == -1
    assert lib.scae_uniform_f32(None, 16, None, None) == -1
    assert lib.scae_scaled_sums_f32(None, 1, None) == -1
    assert lib.scae_sum_rows_multi_f32(None, 1, None) == -1
    jobs = (_lib.SumJob * 5)()
    assert lib.scae_sum_rows_multi_f32(jobs, 5, None) == -1          # > 4 jobs
    assert lib.scae_stage_batch(None, None, 4, None, None, 0, None) == -1
    assert lib.scae_step_prologue_f32(None, None, 4, None, None, 0, None, 0,
                                      None, None, None) == -1
    assert lib.scae_step_prologue_f32(None, None, 0, None, None, 0, None, 0,
                                      None, None, None) == -1   # nothing to do
    first = _lib.FirstLayerDesc()                                # null image / filters
    assert lib.scae_step_prologue_first_f32(None, None, 0, None, None, 0, None, 0,
                                            None, None, first, None) == -1
    assert lib.scae_gemm_multi_f32(None, 2, None) == -1
    assert lib.scae_layer_norm_fwd_f32(None, None, None, None, None, None, 4, 16, 1e-5, None) == -1
    assert lib.scae_layer_norm_bwd_f32(None, None, None, None, None, None, None, 4, 16, None) == -1
    assert lib.scae_layer_norm_rows(0) == 0 and lib.scae_layer_norm_rows(1000) == 250
    assert lib.scae_gemm_multi_f32((_lib.GemmDesc * 5)(), 5, None) == -1   # > 4
    chain = _lib.MlpChainDesc()
    assert lib.scae_mlp_chain_fwd_f32(None, None) == -1
    assert lib.scae_mlp_chain_fwd_f32(chain, None) == -1          # no input, no layers
    assert lib.scae_mlp_chain_bwd_f32(chain, None) == -1
    assert lib.scae_mlp_chain_max_width() >= 512
    assert lib.scae_mlp_chain_votes_fwd_f32(chain, None, None) == -1
    assert lib.scae_mlp_chain_votes_bwd_f32(chain, _lib.VotesDesc(), None) == -1
    assert lib.scae_conv3x3_bwd_pair_f32(None, None, None, None, None, 2, 9, 9,
                                         64, 64, 1, None) == -1
    assert lib.scae_conv3x3_first_fwd_relayout_f32(
        None, None, None, None, 2, 1, 8, 8, 64, 1, 0, None, None, None, None,
        None, None) == -1
    assert lib.scae_conv3x3_first_wgrad_reduce_f32(
        None, None, None, 2, 1, 8, 8, 64, 1, 0, None, None, None, None, None,
        None, None) == -1
    assert lib.scae_loss_tail_workspace_floats(128, 24, 10) == \
        128 * 8 + 128 * 24 + 128 * 2 * 10 + 2 * 24
    assert lib.scae_loss_tail_workspace_floats(0, 24, 10) == 0
    assert lib.scae_class_probs_f32(None, None, None, None, None, None, 4, 8, 8,
                                    4, None, 0, None) == -1


def test_ops_refuse_cpu_tensors():
    from torch_scae_amd import cv_ops, ops
    with pytest.raises(ops.ScaeHipError):
        cv_ops.geometric_transform(torch.zeros(3, 6))
    from torch_scae_amd.set_transformer import qkv_attention
    with pytest.raises(ops.ScaeHipError):
        qkv_attention(torch.zeros(1, 2, 4), torch.zeros(1, 2, 4),
                      torch.zeros(1, 2, 4))


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "torch_scae_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("oracle-", ""), fn


@pytest.mark.parametrize("name", model_names())
def test_state_dict_schema_matches_reference(name):
    """Checkpoint keys / shapes / order equal the reference's (captured in the
    golden files), including the per-capsule MLP keys."""
    from torch_scae_amd import factory
    blob, meta = load(name)
    np.random.seed(0)
    torch.manual_seed(0)
    model = factory.make_scae(meta["config"])
    ref = sub(blob, "param/")
    sd = model.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k in ref:
        assert tuple(sd[k].shape) == tuple(ref[k].shape), k
    model.load_state_dict(ref)
    sd = model.state_dict()
    for k in ref:
        assert torch.equal(sd[k], ref[k]), k


def test_default_config_counts():
    from torch_scae_amd import factory
    np.random.seed(0)
    torch.manual_seed(0)
    model = factory.make_scae(dict(image_shape=(1, 40, 40), n_classes=10,
                                   n_part_caps=24, n_obj_caps=24))
    sd = model.state_dict()
    assert len(sd) == 228
    assert sum(v.numel() for v in sd.values()) == 2414879
    assert model.reconstruct_alternatives is True      # SCAE ctor default
    assert tuple(sd["part_encoder.img_embedding_bias"].shape) == (128, 5, 5)
    assert tuple(sd["obj_decoder.capsule_layer.caps_mlps.0.2.weight"].shape) \
        == (199, 128)


def test_prepare_model_params_matches_oracle_defaults():
    from oracle import scae_oracle as O
    from torch_scae_amd import factory
    kw = dict(image_shape=(3, 32, 32), n_classes=10, n_part_caps=32,
              n_obj_caps=32)
    a, b = factory.prepare_model_params(**kw), O.prepare_model_params(**kw)
    for sec in a:
        if isinstance(a[sec], dict):
            for k, v in a[sec].items():
                bv = b[sec][k]
                assert (tuple(v) == tuple(bv)) if isinstance(v, (list, tuple)) \
                    else v == bv, (sec, k)
    assert a["ocae_encoder_set_transformer"]["dim_in"] == 386
    with pytest.raises(AssertionError):       # derived keys are protected
        factory.prepare_model_params(pcae_decoder_params=dict(n_templates=3),
                                     **kw)


def test_attr_dict_behaviour():
    from torch_scae_amd.general_utils import AttrDict
    d = AttrDict(a=1)
    d.b = 2
    assert d["b"] == 2 and d.a == 1
    del d.a
    assert "a" not in d
    d.update(c=3)
    assert d.get("c") == 3 and d.get("zz") is None


def test_grouped_mlp_has_no_eager_form():
    """No CPU fallback anywhere on the hot path: the stacked per-capsule MLPs
    raise on a CPU tensor like every other op (their parity with the
    reference's loop of MLPs is a GPU test, test_hip_ops.py)."""
    from torch_scae_amd.nn_ext import GroupedMLP
    from torch_scae_amd.ops import ScaeHipError
    gm = GroupedMLP(3, [6, 9, 4])
    with pytest.raises(ScaeHipError):
        gm(torch.randn(2, 3, 6))


def test_fixed_noise_replay_and_shape_check():
    from torch_scae_amd import nn_utils
    a = torch.rand(2, 3)
    with nn_utils.fixed_noise([a]):
        assert torch.equal(nn_utils.rand_like(torch.empty(2, 3)), a)
        # queue exhausted -> fresh noise
        assert nn_utils.rand_like(torch.empty(2, 3)).shape == (2, 3)
    with nn_utils.fixed_noise([a]):
        with pytest.raises(ValueError):
            nn_utils.rand_like(torch.empty(4, 4))


def test_checkpoint_interchange_roundtrip():
    """Lightning-prefixed reference checkpoints load, and export back."""
    from torch_scae_amd import checkpoint, factory
    cfg = dict(image_shape=(1, 28, 28), n_classes=3, n_part_caps=4,
               n_obj_caps=3)
    torch.manual_seed(0)
    a, b = factory.make_scae(cfg), factory.make_scae(cfg)
    ck = checkpoint.to_reference_checkpoint(a, epoch=7)
    assert all(k.startswith("scae.") for k in ck["state_dict"])
    # per-capsule keys of the reference layout, not the stacked ones
    assert any(".mlps.0.0.weight" in k for k in ck["state_dict"])
    checkpoint.load_reference_checkpoint(b, ck)
    for (ka, va), (kb, vb) in zip(a.state_dict().items(),
                                  b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    # a bare state_dict (no wrapper, no prefix) is accepted too
    checkpoint.load_reference_checkpoint(b, a.state_dict())


def test_mnist_pad_and_translate():
    """mnist/experiment.py:23-40: pad 28 -> 40, integer shifts within +-6."""
    from torch_scae_amd.data import pad_and_translate
    g = torch.Generator().manual_seed(0)
    digits = torch.randint(0, 256, (9, 1, 28, 28), generator=g,
                           dtype=torch.uint8)
    shifts = torch.tensor([[0, 0], [6, 6], [-6, -6], [3, -2], [-1, 5],
                           [6, -6], [0, 4], [-5, 0], [2, 2]])
    out = pad_and_translate(digits, (40, 40), shifts=shifts)
    assert out.shape == (9, 1, 40, 40) and out.dtype == torch.float32
    for b, (dy, dx) in enumerate(shifts.tolist()):
        want = torch.zeros(40, 40)
        want[6 + dy:34 + dy, 6 + dx:34 + dx] = digits[b, 0].float() / 255
        assert torch.equal(out[b, 0], want), b
    rnd = pad_and_translate(digits, (40, 40), generator=g)
    assert torch.allclose(rnd.flatten(1).sum(1),
                          digits.float().flatten(1).sum(1) / 255)
    assert torch.equal(pad_and_translate(digits[:, :, :, :], (28, 28)),
                       digits.float() / 255)


def test_lazy_attr_dict_materialises_on_first_access():
    from torch_scae_amd.general_utils import AttrDict, LazyAttrDict
    calls = []

    def fill(d):
        calls.append(1)
        dict.__setitem__(d, "a", 1)
        dict.__setitem__(d, "b", 2)
        d["_lazy"].clear()

    d = LazyAttrDict(pdf=3)
    d["_lazy"] = dict(a=fill, b=fill)
    assert "a" in d and "zz" not in d and d.get("zz", 7) == 7
    assert "a" not in list(d.keys()) and calls == []
    assert d.b == 2 and d["a"] == 1 and d.get("a") == 1 and calls == [1]
    assert {"a", "b", "pdf"} <= set(d.keys())
    with pytest.raises(AttributeError):
        d.nope
    with pytest.raises(KeyError):
        d["nope"]
    outer = LazyAttrDict(AttrDict(x=1, rec=d))
    outer["_lazy"] = dict(t=lambda dd: dict.__setitem__(dd, "t",
                                                        dd["rec"]["a"]))
    assert outer.t == 1 and outer.x == 1
    del outer.x                      # attribute deletion still works
    assert "x" not in outer


def test_factory_rejects_configurations_outside_the_kernel_limits():
    """shape limits of the kernels surface when the model is
    assembled, with a message, not as SCAE_ERR_UNSUPPORTED at the first forward."""
    from torch_scae_amd import factory
    base = dict(image_shape=(1, 40, 40), n_classes=10, n_part_caps=24,
                n_obj_caps=24)
    factory.check_kernel_limits(factory.prepare_model_params(**base))
    # (80 part / 72 object capsules -- beyond the 64-element tiles of the
    # matrix-core kernels -- are inside the limits since round 3)
    factory.check_kernel_limits(factory.prepare_model_params(
        **dict(base, n_part_caps=80, n_obj_caps=72)))
    for bad, needle in ((dict(n_part_caps=201), "n_part_caps"),
                        (dict(n_obj_caps=201, n_part_caps=10), "n_obj_caps"),
                        (dict(n_obj_caps=150, n_part_caps=100), "n_obj_caps * n_part_caps"),
                        # (inside the product limit, outside the
                        # capsule likelihood's own LDS budget)
                        (dict(n_obj_caps=65, n_part_caps=200), "n_obj_caps [+] 14"),
                        (dict(image_shape=(5, 40, 40)), "channels"),
                        (dict(pcae_template_generator_params=dict(
                            template_size=(64, 64))), "th*tw")):
        with pytest.raises(ValueError, match=needle.replace("*", r"\*")):
            factory.make_scae(dict(base, **bad))
    # more classes than the fused classifier heads take is NOT a limit: SCAE
    # falls back to nn.Linear heads + the op-by-op loss
    assert factory.make_scae(dict(base, n_classes=40)).n_classes == 40


def test_grad_slot_has_one_taker_per_step():
    """data_parallel.GradSlot: the first gradient of a parameter in a backward
    is written into its slot of the flat buffer (and only such a buffer may
    have its column sum deferred); a second one gets a fresh buffer that
    autograd accumulates."""
    import torch
    from torch_scae_amd import ops
    from torch_scae_amd.data_parallel import FlatParameters
    lin = torch.nn.Linear(3, 2)
    flat = FlatParameters(lin)
    slot = ops._slot(lin.weight)
    assert slot is lin.weight._scae_grad_slot and not slot.taken
    v = ops._grad_out(slot, lin.weight)
    assert ops._in_slot(v) and v.data_ptr() == flat.grad_views()[0].data_ptr()
    w = ops._grad_out(slot, lin.weight)          # second taker: fresh buffer
    assert not ops._in_slot(w) and w.data_ptr() != v.data_ptr()
    flat.clear_grads()
    assert not slot.taken and ops._in_slot(ops._grad_out(slot, lin.weight))


def test_likelihood_tiling_follows_the_workgroups_a_cu_holds():
    """scae_render_gmm_logprob_tiles (host logic, no launch): the wave form's
    tile count.  cfg-2 takes 4-wave workgroups (7 tiles of 256 pixels: four
    of them fit a CU's wave slots AND its LDS, so the launch shared with the
    object encoder's trunk runs in one round -- DESIGN.md section 5, round 4);
    shapes whose template planes do not fit four times (hydra 40 capsules,
    CIFAR's three channels) and large batches keep the >= 512-workgroup rule."""
    import ctypes
    from torch_scae_amd import _lib
    lib = _lib.load()
    dummy = ctypes.c_void_p(16)      # (never dereferenced by the host logic)

    def tiles(B, M, C, t, H):
        d = _lib.DecoderDesc(templates=dummy, templates_alpha=dummy, pose=dummy,
                             presence=dummy, bg_image=None, bg_value=dummy,
                             bg_mixing_logit=dummy, temperature_logit=None,
                             out_scale=None, B=B, M=M, C=C, th=t, tw=t, H=H, W=H,
                             template_repeat=1)
        return lib.scae_render_gmm_logprob_tiles(ctypes.byref(d))
    assert tiles(128, 24, 1, 11, 40) == 7       # cfg-2: 25 waves in 4-wave tiles
    assert tiles(128, 40, 1, 11, 40) == 4       # hydra: 62 KB of planes per workgroup
    assert tiles(256, 32, 3, 14, 32) == 2       # cfg-5
    assert tiles(1024, 48, 1, 11, 40) == 2      # cfg-3: the batch fills the device
