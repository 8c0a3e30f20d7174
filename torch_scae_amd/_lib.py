"""ctypes binding of libscae_hip.so (C ABI: include/scae_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C torch_scae_amd/csrc``.  There is NO fallback: if it is missing, or a
tensor is not on a HIP device, the ops raise.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int64,
                    c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SCAE_HIP_LIB: another build of the same library, for A/B measurements)
LIB_PATH = os.environ.get("SCAE_HIP_LIB") or os.path.join(_HERE, "lib",
                                                        "libscae_hip.so")

P = c_void_p  # device pointer


class DecoderDesc(Structure):
    """struct scae_decoder_desc"""
    _fields_ = [("templates", P), ("templates_alpha", P), ("pose", P),
                ("presence", P), ("bg_image", P), ("bg_value", P),
                ("bg_mixing_logit", P), ("temperature_logit", P),
                ("out_scale", P),
                ("B", c_int), ("M", c_int), ("C", c_int), ("th", c_int),
                ("tw", c_int), ("H", c_int), ("W", c_int),
                ("template_repeat", c_int), ("bwd_resident", c_int)]


class LikelihoodBwdDesc(Structure):
    """struct scae_likelihood_bwd_desc"""
    _fields_ = [(n, P) for n in (
        "vote", "scale", "vote_presence", "dummy_vote", "x", "presence",
        "posterior", "winner_idx", "g_lpp", "g_winner", "g_winner_presence",
        "g_soft_winner", "g_soft_winner_presence", "g_posterior",
        "g_mixing_log_prob", "g_mixing_logit", "gvote", "gscale",
        "gvote_presence", "gx", "gpresence", "gdummy_partial")] + [
        ("B", c_int), ("O", c_int), ("M", c_int)]


class GemmDesc(Structure):
    """struct scae_gemm_desc"""
    _fields_ = [("A", P), ("B", P), ("C", P), ("bias", P), ("mask", P),
                ("asum", P), ("batch", c_int), ("M", c_int), ("N", c_int),
                ("K", c_int), ("a_kcontig", c_int), ("lda", c_int),
                ("a_batch", c_int64), ("b_kcontig", c_int), ("ldb", c_int),
                ("b_batch", c_int64), ("ldc", c_int), ("c_batch", c_int64),
                ("bias_ld", c_int), ("bias_batch", c_int64), ("ldmask", c_int),
                ("mask_batch", c_int64), ("asum_batch", c_int64),
                ("relu", c_int), ("asum_ld", c_int), ("c_nomask", P)]


class MlpChainLayer(Structure):
    """struct scae_mlp_chain_layer"""
    _fields_ = [("w", P), ("w_gs", c_int64), ("ldw", c_int), ("K", c_int),
                ("N", c_int), ("bias", P), ("bias_gs", c_int64),
                ("bias_ld", c_int), ("gate", P), ("gate_gs", c_int64),
                ("gate_bs", c_int64), ("out", P), ("out_gs", c_int64),
                ("out_bs", c_int64), ("relu", c_int)]


class MlpChainDesc(Structure):
    """struct scae_mlp_chain_desc"""
    _fields_ = [("layer", MlpChainLayer * 4), ("n_layers", c_int), ("in_", P),
                ("in_gs", c_int64), ("in_bs", c_int64), ("in_dim", c_int),
                ("B", c_int), ("G", c_int), ("row_tile", c_int),
                ("bf16", c_int)]


class VotesDesc(Structure):
    """struct scae_votes_desc"""
    _fields_ = [(n, P) for n in ("all_param", "cpr_static", "bias_cvr",
                                 "bias_caps", "bias_vote", "bias_scale",
                                 "noise_caps", "noise_vote")] + \
        [("noise_scale", c_float), ("V", c_int), ("ld_param", c_int),
         ("similarity", c_int), ("learn_vote_scale", c_int),
         ("allow_deformations", c_int)] + \
        [(n, P) for n in ("vote", "scale", "vote_presence", "logit_caps",
                          "logit_vote", "reg_partial", "caps_presence",
                          "caps_arg", "gvote", "gscale", "gvote_presence",
                          "glogit_caps", "glogit_vote", "greg", "gcaps_presence",
                          "gall_param", "gcpr_in", "gall_param_gated")]


class SumSegment(Structure):
    """struct scae_sum_segment"""
    _fields_ = [("dst", P), ("begin", c_int64), ("end", c_int64),
                ("period", c_int64)]


class SumJob(Structure):
    """struct scae_sum_job"""
    _fields_ = [("src", P), ("rows", c_int64), ("cols", c_int64),
                ("segments", POINTER(SumSegment)), ("n_segments", c_int)]


class ScaledSum(Structure):
    """struct scae_scaled_sum"""
    _fields_ = [("src", P), ("n", c_int64), ("scale", c_float), ("dst", P)]


class LossExtras(Structure):
    """struct scae_loss_extras"""
    _fields_ = [("rec_sums", P), ("n_rec", c_int), ("reg", P),
                ("w_reg", c_float), ("g_rec_sums", P), ("g_reg", P),
                ("loss", P), ("g_loss", P), ("defer_combine", c_int),
                ("out12", P)]


class SeedFoldDesc(Structure):
    """struct scae_seed_fold_desc"""
    _fields_ = [(n, P) for n in (
        "seeds", "wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "w2", "b2",
        "q", "wkf", "bkf", "wvf", "bvf", "wv2e", "wowv")] + \
        [("O", c_int), ("C", c_int), ("D", c_int)]


class FirstLayerDesc(Structure):
    """struct scae_first_layer_desc"""
    _fields_ = [("img", P), ("w", P), ("bias", P), ("out", P)] + \
        [(n, c_int) for n in ("B", "Cin", "IH", "IW", "Cout", "stride",
                              "n_layers")] + \
        [("rw", P * 8), ("rwf", P * 8), ("rwd", P * 8),
         ("rCout", c_int * 8), ("rCin", c_int * 8),
         ("out_h", P), ("rwfh", P * 8), ("rwdh", P * 8)]


class SeedFoldGrads(Structure):
    """struct scae_seed_fold_grads"""
    _fields_ = [(n, P) for n in (
        "g_q", "g_wkf", "g_bkf", "g_wvf", "g_bvf", "d_seeds", "d_wq", "d_bq",
        "d_wk", "d_bk", "d_wv", "d_bv", "d_wo", "d_bo", "d_w2", "d_b2",
        "gv2e", "t1")]


# name -> argtypes; mirrors include/scae_hip.h one to one
SIGNATURES = {
    "scae_abi_version": [],
    "scae_error_string": [c_int],
    "scae_uniform_f32": [P, c_int64, P, P],
    "scae_geometric_transform_fwd_f32": [P, P, c_int64, c_int, c_int, c_int, P],
    "scae_geometric_transform_bwd_f32": [P, P, P, c_int64, c_int, c_int, c_int,
                                         P],
    "scae_mat3_mul_fwd_f32": [P, P, P, c_int64, c_int, P],
    "scae_mat3_mul_bwd_f32": [P, P, P, P, P, c_int64, c_int, P],
    "scae_qkv_attention_fwd_f32": [P, P, P, P, P, P, c_int, c_int, c_int,
                                   c_int, c_int, c_float, P],
    "scae_qkv_attention_fwd_bf16": [P, P, P, P, P, P, c_int, c_int, c_int,
                                    c_int, c_int, c_float, P],
    "scae_qkv_attention_bwd_f32": [P, P, P, P, P, P, P, P, P, c_int, c_int,
                                   c_int, c_int, c_int, c_float, P],
    "scae_set_encoder_param_count": [c_int] * 5,
    "scae_set_encoder_grid": [c_int],
    "scae_set_encoder_supported": [c_int] * 6,
    "scae_set_encoder_fwd_f32": [c_int, P, P, P, P, P, P, P, P] + [c_int] * 7 + [P],
    "scae_set_encoder_bwd_f32": [c_int, P, P, P, P, P, P, P, P, P, P]
                                + [c_int] * 7 + [P],
    "scae_render_gmm_sums_bwd_likelihood_f32": [P] * 13,
    "scae_set_encoder_fwd_logprob_f32": [c_int, P, P, P, P, P, P, P, P] + [c_int] * 7
                                        + [P] * 6,
    "scae_set_encoder_fwd_logprob_bf16": [c_int, P, P, P, P, P, P, P, P] + [c_int] * 7
                                         + [P] * 6,
    "scae_set_encoder_bf16_supported": [c_int] * 6,
    "scae_set_encoder_fwd_bf16": [c_int, P, P, P, P, P, P, P, P] + [c_int] * 7 + [P],
    "scae_set_encoder_bwd_bf16": [c_int, P, P, P, P, P, P, P, P, P, P]
                                 + [c_int] * 7 + [P],
    "scae_seed_attention_splits": [c_int] * 2,
    "scae_seed_attention_grid": [c_int] * 2,
    "scae_seed_attention_supported": [c_int] * 4,
    "scae_seed_attention_fwd_f32": [P] * 9 + [c_int] * 5 + [P],
    "scae_seed_attention_bwd_f32": [P] * 10 + [c_int] * 5 + [P],
    "scae_seed_attention_mfma_supported": [c_int] * 4,
    "scae_seed_attention_mfma_rows": [c_int],
    "scae_seed_attention_mfma_fwd_f32": [P] * 7 + [c_int] * 4 + [P],
    "scae_seed_attention_mfma_bwd_f32": [P] * 8 + [c_int] * 4 + [P],
    "scae_seed_attention_mfma_fwd_bf16": [P] * 7 + [c_int] * 4 + [P],
    "scae_seed_attention_mfma_bwd_bf16": [P] * 8 + [c_int] * 4 + [P],
    "scae_seed_attention_mfma_bwd_gemm_f32": [P] * 8 + [c_int] * 4
    + [POINTER(GemmDesc), c_int, P],
    "scae_seed_attention_mfma_bwd_gemm_bf16": [P] * 8 + [c_int] * 4
    + [POINTER(GemmDesc), c_int, P],
    "scae_seed_attention_mfma_reduce_f32": [P, c_int] + [P] * 7 + [c_int] * 2 + [P],
    "scae_seed_fold_supported": [c_int] * 3,
    "scae_seed_fold_fwd_f32": [POINTER(SeedFoldDesc), P],
    "scae_seed_fold_bwd_f32": [POINTER(SeedFoldDesc), POINTER(SeedFoldGrads), P],
    "scae_layer_norm_rows": [c_int64],
    "scae_layer_norm_fwd_f32": [P] * 6 + [c_int64, c_int, c_float, P],
    "scae_layer_norm_bwd_f32": [P] * 7 + [c_int64, c_int, P],
    "scae_gemm_f32": [P] * 6 + [c_int] * 6 + [c_int64, c_int, c_int, c_int64,
                                              c_int, c_int64, c_int, c_int64,
                                              c_int, c_int64, c_int64, c_int, c_int,
                                              P],
    "scae_gemm_bf16": [P] * 6 + [c_int] * 6 + [c_int64, c_int, c_int, c_int64,
                                              c_int, c_int64, c_int, c_int64,
                                              c_int, c_int64, c_int64, c_int, c_int,
                                              P],
    "scae_gemm_pair_f32": [POINTER(GemmDesc), POINTER(GemmDesc), P],
    # the library's own record of a step's kernel launches (train_step.TrainStep)
    "scae_launch_list_begin": [P],
    "scae_launch_list_end": [P],
    "scae_launch_list_size": [P],
    "scae_launch_list_run": [P, P],
    "scae_launch_list_side_stream": [P, P],
    "scae_launch_list_order": [P, P],
    "scae_launch_list_run2": [P, P, P],
    "scae_launch_list_side_size": [P],
    "scae_launch_list_lane": [P, c_int],
    "scae_launch_list_timeline": [P, P, P, POINTER(c_float), c_int],
    "scae_launch_list_free": [P],
    "scae_gemm_multi_f32": [POINTER(GemmDesc), c_int, P],
    "scae_gemm_multi_bf16": [POINTER(GemmDesc), c_int, P],
    "scae_mlp_chain_max_width": [],
    "scae_mlp_chain_fwd_f32": [POINTER(MlpChainDesc), P],
    "scae_mlp_chain_bwd_f32": [POINTER(MlpChainDesc), P],
    "scae_mlp_chain_votes_fwd_f32": [POINTER(MlpChainDesc), POINTER(VotesDesc), P],
    "scae_mlp_chain_votes_bwd_f32": [POINTER(MlpChainDesc), POINTER(VotesDesc), P],
    "scae_gemm_pair_bf16": [POINTER(GemmDesc), POINTER(GemmDesc), P],
    "scae_conv3x3_wf_floats": [c_int, c_int],
    "scae_conv3x3_relayout_f32": [P, P, P, c_int, c_int, P],
    "scae_conv3x3_bf16r_supported": [c_int] * 6,
    "scae_conv3x3_first_fwd_relayout_bf16": [P] * 4 + [c_int] * 7 + [P] * 8,
    "scae_cvt_bf16_batch": [c_int, P, P, P, P],
    "scae_conv3x3_fwd_bf16r": [P] * 7 + [c_int] * 6 + [P],
    "scae_conv3x3_dgrad_bf16r": [P] * 5 + [c_int] * 6 + [P],
    "scae_conv3x3_wgrad_bf16r_splits": [c_int] * 5,
    "scae_conv3x3_wgrad_bf16r": [P] * 3 + [c_int] * 6 + [P],
    "scae_conv3x3_relayout_batch_f32": [c_int, P, P, P, P, P, P],
    "scae_conv3x3_first_fwd_f32": [P] * 4 + [c_int] * 6 + [P],
    "scae_conv3x3_first_fwd_relayout_f32": [P] * 4 + [c_int] * 7 + [P] * 6,
    "scae_conv3x3_first_wgrad_reduce_f32": [P] * 3 + [c_int] * 7 + [P] * 7,
    "scae_conv3x3_first_wgrad_rows": [c_int] * 2,
    "scae_conv3x3_first_wgrad_f32": [P] * 3 + [c_int] * 6 + [P],
    "scae_conv3x3_fwd_f32": [P] * 6 + [c_int] * 6 + [P],
    "scae_conv3x3_fwd_fold_f32": [P] * 6 + [c_int] * 6 + [POINTER(SeedFoldDesc), P],
    "scae_conv3x3_fwd_res_supported": [c_int] * 6,
    "scae_conv3x3_fwd_res_f32": [P] * 6 + [c_int] * 7 + [P],
    "scae_conv3x3_fwd_bf16": [P] * 6 + [c_int] * 6 + [P],
    "scae_conv3x3_dgrad_f32": [P] * 4 + [c_int] * 6 + [P],
    "scae_conv3x3_bwd_pair_f32": [P] * 5 + [c_int] * 6 + [P],
    "scae_conv3x3_bwd_pair_reduce_f32": [P] * 5 + [c_int] * 6 + [P, c_int] + [P] * 7
    + [c_int] * 2 + [P],
    "scae_conv3x3_bwd_pair_fold_f32": [P] * 5 + [c_int] * 6
    + [POINTER(SeedFoldDesc), POINTER(SeedFoldGrads), P],
    "scae_conv3x3_bwd_pair_bf16": [P] * 5 + [c_int] * 6 + [P],
    "scae_conv3x3_wgrad_reduce_batch_f32": [c_int] + [P] * 7,
    "scae_conv3x3_wgrad_splits": [c_int] * 5,
    "scae_conv3x3_wgrad_f32": [P] * 5 + [c_int] * 6 + [P],
    "scae_attention_pool_supported": [c_int] * 3,
    "scae_attention_pool_fwd_f32": [P, P] + [c_int] * 4 + [P],
    "scae_attention_pool_bwd_f32": [P, P, P] + [c_int] * 4 + [P],
    "scae_stage_batch": [P, P, c_int64, P, P, c_int64, P],
    "scae_step_prologue_f32": [P, P, c_int64, P, P, c_int64, P, c_int64, P,
                               POINTER(SeedFoldDesc), P],
    "scae_step_prologue_first_f32": [P, P, c_int64, P, P, c_int64, P, c_int64,
                                     P, POINTER(SeedFoldDesc),
                                     POINTER(FirstLayerDesc), P],
    "scae_rmsprop_step_f32": [P, P, P, P, c_int64, c_float, P, c_float, c_float,
                              c_float, c_float, c_float, P],
    "scae_rmsprop_sums_step_f32": [P, P, P, P, c_int64, c_float, P, c_float, c_float,
                                   c_float, c_float, POINTER(SumJob), c_int, P],
    "scae_capsule_head_fwd_f32": [P, P, c_float, c_int, P, P, P, P, P] + [c_int] * 4 + [P],
    "scae_capsule_head_conv_supported": [c_int] * 4,
    "scae_capsule_head_conv_preferred": [c_int] * 5,
    "scae_capsule_head_conv_fwd_f32": [P, P, P, c_int, P, P, c_float, c_int, P, P, P, P, P]
    + [c_int] * 4 + [P],
    "scae_capsule_head_conv_fwd_tc_f32": [P, P, P, c_int, P, P, c_float, c_int, P, P, P, P, P]
    + [c_int] * 4 + [P] * 8 + [c_int] * 6 + [P],
    "scae_capsule_head_bwd_f32": [P, P, P, c_float, c_int, P, P, P, P, P] + [c_int] * 4 + [P],
    "scae_capsule_head_bwd_tc_f32": [P, P, P, c_float, c_int, P, P, P, P] + [c_int] * 4
    + [P] * 12 + [c_int] * 6 + [P],
    "scae_template_color_supported": [c_int] * 4,
    "scae_template_color_partial_rows": [c_int] * 2,
    "scae_template_color_fwd_f32": [P] * 9 + [c_int] * 8 + [P],
    "scae_template_color_bwd_f32": [P] * 12 + [c_int] * 8 + [P],
    "scae_sum_rows_f32": [P, c_int64, c_int64, POINTER(SumSegment), c_int, P],
    "scae_sum_rows_multi_f32": [POINTER(SumJob), c_int, P],
    "scae_scaled_sums_f32": [POINTER(ScaledSum), c_int, P],
    "scae_class_probs_supported": [c_int] * 2,
    "scae_class_probs_f32": [P] * 6 + [c_int] * 4 + [POINTER(ScaledSum), c_int, P],
    "scae_capsule_votes_fwd_f32": [P] * 8 + [c_float] + [P] * 8
                                  + [c_int] * 7 + [P],
    "scae_capsule_votes_bwd_f32": [P] * 8 + [c_float] + [P] * 11
                                  + [c_int] * 7 + [P],
    "scae_capsule_likelihood_fwd_f32": [P] * 17 + [c_int] * 3 + [P],
    "scae_capsule_likelihood_bwd_f32": [P] * 22 + [c_int] * 3 + [P],
    "scae_loss_tail_supported": [c_int] * 3,
    "scae_loss_tail_workspace_floats": [c_int] * 3,
    "scae_loss_tail_fwd_f32": [P] * 6 + [POINTER(LossExtras), P, P] + [c_int] * 8
    + [POINTER(c_float), c_float, P],
    "scae_loss_tail_defer_preferred": [c_int] * 2,
    "scae_loss_tail_combine_f32": [P] * 6 + [POINTER(LossExtras), P, P] + [c_int] * 8
    + [POINTER(c_float), c_float, P],
    "scae_loss_tail_fwd_class_probs_f32": [P] * 6 + [POINTER(LossExtras), P, P]
    + [c_int] * 8 + [POINTER(c_float), c_float] + [P] * 6 + [c_int] * 4
    + [POINTER(ScaledSum), c_int, P],
    "scae_loss_tail_bwd_f32": [P] * 6 + [POINTER(LossExtras)] + [P] * 7
    + [c_int] * 8 + [POINTER(c_float), c_float, P],
    "scae_template_render_fwd_f32": [POINTER(DecoderDesc), P, P, P],
    "scae_render_gmm_logprob_fwd_f32": [POINTER(DecoderDesc), P, P, P, P, P],
    "scae_render_gmm_bwd_f32": [POINTER(DecoderDesc)] + [P] * 12 + [P],
    "scae_render_gmm_logprob_tiles": [POINTER(DecoderDesc)],
    "scae_render_gmm_logprob_sums_fwd_f32": [POINTER(DecoderDesc)] + [P] * 4 + [P],
    "scae_render_gmm_sums_bwd_f32": [POINTER(DecoderDesc)] + [P] * 10 + [P],
    "scae_gmm_log_prob_fwd_f32": [P] * 5 + [c_int] * 4 + [c_int64, P],
    "scae_gmm_log_prob_bwd_f32": [P] * 9 + [c_int] * 4 + [c_int64, P],
    "scae_gmm_mean_f32": [P] * 3 + [c_int] * 4 + [c_int64, P],
    "scae_gmm_mode_f32": [P] * 4 + [c_int] * 5 + [c_int64, P],
}

# (everything else returns int)
_RESTYPES = {"scae_error_string": c_char_p,
             "scae_loss_tail_workspace_floats": c_int64,
             "scae_conv3x3_wf_floats": c_int64,
             "scae_launch_list_begin": P,
             "scae_launch_list_free": None}
ABI_VERSION = 2     # SCAE_ABI_VERSION of the include/scae_hip.h this binding mirrors

_lib = None


class ScaeHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises if not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ScaeHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import "
            f"__graft_entry__ as g; g.build()'` or `make -C torch_scae_amd/csrc`."
            " torch_scae_amd has no CPU / eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
        if argtypes and argtypes[-1] is P and not name.startswith("scae_launch_list"):
            setattr(lib, name, _recording(fn))     # a launcher: (..., void *stream)
    if lib.scae_abi_version() != ABI_VERSION:
        raise ScaeHipError(
            f"{LIB_PATH} has ABI version {lib.scae_abi_version()}, this package "
            f"binds version {ABI_VERSION}: rebuild it (make -C torch_scae_amd/csrc)")
    _lib = lib
    return lib


# Launch recording (train_step.TrainStep(replay="launches")): while a recorder is installed
# every successful launcher call is kept as (function, arguments); the list IS the step -- the
# same launches a captured HIP graph holds -- and can be re-issued on any stream without the
# Python above the C ABI.  The arguments keep what they point to alive (device buffers of the
# capture's private pool are owned by the graph object).
_RECORDER = None


def _recording(fn):
    def launcher(*args):
        rc = fn(*args)
        if _RECORDER is not None and rc == 0:
            # arguments converted ONCE to the C types of the prototype (a replay then only
            # passes ready ctypes objects: the per-call conversion of ~20 Python ints / floats
            # per launcher was most of a replay's host time)
            conv = []
            for t, a in zip(fn.argtypes, args):
                if a is None or isinstance(a, (int, float)):
                    a = t(a) if a is not None else t()
                conv.append(a)
            _RECORDER.append((fn, tuple(conv[:-1]), args))
        return rc
    launcher.__name__ = getattr(fn, "__name__", "launcher")
    return launcher


class recorder:
    """``with _lib.recorder() as launches:`` -- collects the launcher calls made inside."""

    def __enter__(self):
        global _RECORDER
        self.prev, self.launches = _RECORDER, []
        _RECORDER = self.launches
        return self.launches

    def __exit__(self, *exc):
        global _RECORDER
        _RECORDER = self.prev
        return False


def replay(launches, stream):
    """Re-issue recorded launches on ``stream`` (a ``c_void_p`` handle): every launcher's
    last argument is its stream."""
    for fn, cargs, _keep in launches:
        rc = fn(*cargs, stream)
        if rc != 0:
            check(rc, getattr(fn, "__name__", "launch"))


ERR_UNSUPPORTED = -2     # SCAE_ERR_UNSUPPORTED


def check(rc, what):
    if rc != 0:
        msg = load().scae_error_string(rc)
        raise ScaeHipError(f"{what} failed with code {rc}: "
                           f"{msg.decode() if msg else '?'}")


def call(name, *args):
    check(getattr(load(), name)(*args), name)


def try_call(name, *args):
    """``call`` for a merged launcher that may decline a shape: True when it
    ran, False for SCAE_ERR_UNSUPPORTED (the caller launches the parts in
    turn); any other error raises."""
    rc = getattr(load(), name)(*args)
    if rc == ERR_UNSUPPORTED:
        return False
    check(rc, name)
    return True
