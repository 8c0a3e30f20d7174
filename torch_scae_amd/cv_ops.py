"""geometric_transform on the HIP kernel K5 (reference: torch_scae/cv_ops.py:20-76)."""
from . import ops


def geometric_transform(pose_tensor, similarity=False, nonlinear=True,
                        as_matrix=False):
    """Pose 6-vectors (sx, sy, theta, shear, tx, ty) -> affine / similarity
    transforms.

    Args mirror cv_ops.py:20-35.  Returns [..., 6], or [..., 3, 3] if
    ``as_matrix``.  One fused forward kernel and one backward kernel replace
    the reference's split / sigmoid / tanh / sin / cos / cat chain; unlike the
    reference the input is never modified in place (cv_ops.py:45 scales a view
    of it by 2*pi).
    """
    return ops.geometric_transform(pose_tensor, similarity, nonlinear,
                                   as_matrix)
