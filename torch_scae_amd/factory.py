"""Hyper-parameter defaults and model assembly (reference:
torch_scae/factory.py)."""
from argparse import Namespace

from .object_decoder import CapsuleLayer, CapsuleObjectDecoder
from .part_decoder import TemplateBasedImageDecoder, TemplateGenerator
from .part_encoder import CNNEncoder, CapsuleImageEncoder
from .set_transformer import SetTransformer
from .stacked_capsule_auto_encoder import SCAE


def _section(defaults, overrides, derived=()):
    overrides = overrides or {}
    for key in derived:      # the reference asserts these are not overridden
        assert key not in overrides
    out = dict(defaults)
    out.update(overrides)
    return out


def prepare_model_params(image_shape, n_classes, n_part_caps, n_obj_caps,
                         pcae_cnn_encoder_params=None,
                         pcae_encoder_params=None,
                         pcae_template_generator_params=None,
                         pcae_decoder_params=None,
                         ocae_encoder_set_transformer_params=None,
                         ocae_decoder_capsule_params=None, scae_params=None):
    """Merge user overrides into the defaults of factory.py:33-135 and derive
    the cross-module sizes."""
    cnn = _section(dict(input_shape=image_shape, out_channels=[128] * 4,
                        kernel_sizes=[3, 3, 3, 3], strides=[2, 2, 1, 1],
                        activate_final=True),
                   pcae_cnn_encoder_params, ('input_shape',))
    enc = _section(dict(input_shape=image_shape, n_caps=n_part_caps, n_poses=6,
                        n_special_features=16, similarity_transform=False),
                   pcae_encoder_params, ('input_shape',))
    tgen = _section(dict(n_templates=enc['n_caps'], n_channels=image_shape[0],
                         template_size=(11, 11), template_nonlin='sigmoid',
                         dim_feature=enc['n_special_features'],
                         colorize_templates=True, color_nonlin='sigmoid'),
                    pcae_template_generator_params,
                    ('n_templates', 'n_channels', 'dim_feature'))
    dec = _section(dict(n_templates=tgen['n_templates'],
                        template_size=tgen['template_size'],
                        output_size=image_shape[1:], learn_output_scale=False,
                        use_alpha_channel=True, background_value=True),
                   pcae_decoder_params,
                   ('n_templates', 'template_size', 'output_size'))
    # factory.py:79-86 squares template_size[0] (not h*w); kept as is
    st_dim_in = (enc['n_poses'] + tgen['dim_feature'] + 1
                 + tgen['n_channels'] * tgen['template_size'][0]
                 * tgen['template_size'][0])
    st = _section(dict(n_layers=3, n_heads=1, dim_in=st_dim_in, dim_hidden=16,
                       dim_out=256, n_outputs=n_obj_caps, layer_norm=True),
                  ocae_encoder_set_transformer_params,
                  ('_ocae_st_dim_in', 'n_obj_caps'))
    caps = _section(dict(n_caps=st['n_outputs'], dim_feature=st['dim_out'],
                         n_votes=dec['n_templates'], dim_caps=32,
                         hidden_sizes=(128,), caps_dropout_rate=0.0,
                         learn_vote_scale=True, allow_deformations=True,
                         noise_type='uniform', noise_scale=4.,
                         similarity_transform=False),
                    ocae_decoder_capsule_params,
                    ('n_caps', 'dim_feature', 'n_votes'))
    scae = _section(dict(n_classes=n_classes, vote_type='enc',
                         presence_type='enc', stop_grad_caps_input=True,
                         stop_grad_caps_target=True, caps_ll_weight=1.,
                         cpr_dynamic_reg_weight=10,
                         prior_sparsity_loss_type='l2',
                         prior_within_example_sparsity_weight=2.0,
                         prior_between_example_sparsity_weight=0.35,
                         posterior_sparsity_loss_type='entropy',
                         posterior_within_example_sparsity_weight=0.7,
                         posterior_between_example_sparsity_weight=0.2),
                    scae_params, ('n_classes',))
    return dict(image_shape=image_shape, n_classes=n_classes,
                n_part_caps=n_part_caps, n_obj_caps=n_obj_caps,
                pcae_cnn_encoder=cnn, pcae_encoder=enc,
                pcae_template_generator=tgen, pcae_decoder=dec,
                ocae_encoder_set_transformer=st, ocae_decoder_capsule=caps,
                scae=scae)


# Hard shape limits of the HIP kernels behind the module surface (there is no
# eager fallback): checked when the model is assembled, not at the first forward.
KERNEL_LIMITS = dict(
    # up to 64 / 64 everything runs on the matrix-core kernels; beyond that the
    # general attention (set_attention_big.hip) and the two-pass capsule
    # likelihood take over, bounded by what one problem's dS (N x M floats) /
    # mixture statistics (3 x O x M floats) may occupy of a CU's 160 KB of LDS
    n_part_caps=200,        # K2: n_part^2 floats of dS per set
    n_obj_caps=200,
    caps_product=13000,     # K2's output attention: n_obj * n_part
    # K4 (capsule_likelihood_dev.h, lk_lds): (3 * n_obj + 14) * n_part floats
    # of one image's mixture statistics in the 160 KB of LDS
    likelihood_floats=40960,
    n_channels=4,           # K1 template planes
    template_texels=4096,   # K1: (C + 1) * th * tw floats of LDS per template
)


def check_kernel_limits(params: dict):
    """Raise ValueError for a configuration the kernels are not built for
    (``params``: the output of ``prepare_model_params``)."""
    lim = KERNEL_LIMITS
    n_part, n_obj = params["n_part_caps"], params["n_obj_caps"]
    C = params["image_shape"][0]
    th, tw = params["pcae_template_generator"]["template_size"]
    problems = []
    if n_part > lim["n_part_caps"]:
        problems.append(f"n_part_caps={n_part} > {lim['n_part_caps']} "
                        "(attention set size / capsule-likelihood groups)")
    if n_obj > lim["n_obj_caps"]:
        problems.append(f"n_obj_caps={n_obj} > {lim['n_obj_caps']} "
                        "(capsule likelihood, output attention seeds)")
    if n_obj * n_part > lim["caps_product"]:
        problems.append(f"n_obj_caps * n_part_caps = {n_obj * n_part} > "
                        f"{lim['caps_product']} (capsule likelihood: mixture "
                        "statistics of one image in LDS)")
    if (3 * n_obj + 14) * n_part > lim["likelihood_floats"]:
        problems.append(f"(3 * n_obj_caps + 14) * n_part_caps = "
                        f"{(3 * n_obj + 14) * n_part} > "
                        f"{lim['likelihood_floats']} (capsule likelihood: one "
                        "image's mixture statistics in LDS)")
    if C > lim["n_channels"]:
        problems.append(f"{C} image channels > {lim['n_channels']} "
                        "(template render / mixture likelihood)")
    if (C + 1) * th * tw > lim["template_texels"]:
        problems.append(f"(C+1)*th*tw = {(C + 1) * th * tw} > "
                        f"{lim['template_texels']} (template planes in LDS)")
    # (n_classes above the fused heads' limit is not a problem: SCAE then runs
    # the nn.Linear classifiers and the op-by-op loss,
    # SCAE._fused_class_probs / _fused_tail_ok)
    if problems:
        raise ValueError("torch_scae_amd's HIP kernels do not cover this "
                         "configuration: " + "; ".join(problems))


def make_scae(model_params: dict):
    """config dict -> wired SCAE (factory.py:152-178)."""
    params = prepare_model_params(**model_params)
    check_kernel_limits(params)
    cfg = Namespace(**params)
    part_encoder = CapsuleImageEncoder(
        encoder=CNNEncoder(**cfg.pcae_cnn_encoder), **cfg.pcae_encoder)
    obj_decoder = CapsuleObjectDecoder(CapsuleLayer(**cfg.ocae_decoder_capsule))
    return SCAE(part_encoder=part_encoder,
                template_generator=TemplateGenerator(
                    **cfg.pcae_template_generator),
                part_decoder=TemplateBasedImageDecoder(**cfg.pcae_decoder),
                obj_encoder=SetTransformer(**cfg.ocae_encoder_set_transformer),
                obj_decoder=obj_decoder, **cfg.scae)
