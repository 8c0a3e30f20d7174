"""torch_scae_amd -- the SCAE forward/backward hot path of bdsaglam/torch-scae,
rebuilt for MI355X (gfx950): hand-written HIP kernels behind the reference's
own nn.Module surface.  Module names mirror the reference package so that
``from torch_scae_amd import factory; factory.make_scae(cfg)`` is a drop-in
for ``torch_scae.factory.make_scae(cfg)``."""
from . import (cv_ops, distributions, factory, general_utils, math_ops,  # noqa
               nn_ext, nn_utils, object_decoder, ops, part_decoder,
               part_encoder, set_transformer, stacked_capsule_auto_encoder)
from .stacked_capsule_auto_encoder import SCAE  # noqa

__version__ = "0.1.0"
