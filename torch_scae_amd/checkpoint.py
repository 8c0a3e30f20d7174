"""Checkpoint interchange with the reference.

The reference trains under PyTorch-Lightning: its checkpoints are dicts with a
``state_dict`` entry whose keys carry the attribute name of the model inside
``BaseExperiment`` as a prefix (``scae.``, base_experiment.py:36).  The
modules here keep the reference's own parameter names (the stacked
per-capsule MLP weights map themselves to / from the per-capsule keys), so
interchange is a matter of the prefix and of the Lightning wrapper."""
import torch

PREFIX = "scae."


def reference_state_dict(obj, prefix=PREFIX):
    """Plain SCAE state_dict from a reference checkpoint: ``obj`` is a path, a
    Lightning checkpoint dict, or a (possibly prefixed) state_dict."""
    if isinstance(obj, (str, bytes)) or hasattr(obj, "__fspath__"):
        obj = torch.load(obj, map_location="cpu")
    if isinstance(obj, dict) and "state_dict" in obj and \
            not torch.is_tensor(obj["state_dict"]):
        obj = obj["state_dict"]
    if any(k.startswith(prefix) for k in obj):
        obj = {k[len(prefix):]: v for k, v in obj.items()
               if k.startswith(prefix)}
    return dict(obj)


def load_reference_checkpoint(model, obj, prefix=PREFIX, strict=True):
    """Load a reference checkpoint (see ``reference_state_dict``) into a
    ``torch_scae_amd`` SCAE; returns ``load_state_dict``'s result."""
    return model.load_state_dict(reference_state_dict(obj, prefix),
                                 strict=strict)


def to_reference_checkpoint(model, prefix=PREFIX, **extra):
    """Lightning-style checkpoint dict the reference's ``load_from_checkpoint``
    / ``load_state_dict`` accept: {'state_dict': {prefix + key: tensor}}."""
    sd = {prefix + k: v.detach().cpu().clone()
          for k, v in model.state_dict().items()}
    return dict(state_dict=sd, **extra)
