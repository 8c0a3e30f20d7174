// geometric_transform math (cv_ops.py:20-76) shared by K5 and the fused
// part-capsule head (K9): pose 6-vector -> rows of the 2x3 affine, and its
// vector-Jacobian product.
#pragma once
#include "common.h"

namespace scae_gt {
constexpr float kTwoPi = 6.283185307179586f;

// forward values + the intermediates the backward needs
struct GtState {
  float sx, sy, sh, c, s;       // transformed scale_x, scale_y, shear, cos, sin
  float dsx, dsy, dsh, dth, dtx, dty;  // d(transformed)/d(raw)
  float tx, ty;
};

__device__ __forceinline__ void gt_eval(const float *p, int nonlinear, GtState &g) {
  float sx = p[0], sy = p[1], th = p[2], sh = p[3], tx = p[4], ty = p[5];
  if (nonlinear) {  // cv_ops.py:40-45
    const float ex = scae::sigmoidf_(sx), ey = scae::sigmoidf_(sy);
    g.sx = ex + 1e-2f;
    g.sy = ey + 1e-2f;
    g.dsx = ex * (1.f - ex);
    g.dsy = ey * (1.f - ey);
    g.tx = tanhf(tx * 5.f);
    g.ty = tanhf(ty * 5.f);
    g.sh = tanhf(sh * 5.f);
    g.dtx = 5.f * (1.f - g.tx * g.tx);
    g.dty = 5.f * (1.f - g.ty * g.ty);
    g.dsh = 5.f * (1.f - g.sh * g.sh);
    th = th * kTwoPi;
    g.dth = kTwoPi;
  } else {  // cv_ops.py:46-47
    g.sx = fabsf(sx) + 1e-2f;
    g.sy = fabsf(sy) + 1e-2f;
    g.dsx = sx > 0.f ? 1.f : (sx < 0.f ? -1.f : 0.f);
    g.dsy = sy > 0.f ? 1.f : (sy < 0.f ? -1.f : 0.f);
    g.tx = tx;
    g.ty = ty;
    g.sh = sh;
    g.dtx = g.dty = g.dsh = 1.f;
    g.dth = 1.f;
  }
  g.c = cosf(th);
  g.s = sinf(th);
}

__device__ __forceinline__ void gt_rows(const GtState &g, int similarity, float *o) {
  if (similarity) {  // cv_ops.py:51-54
    o[0] = g.sx * g.c;
    o[1] = -g.sx * g.s;
    o[2] = g.tx;
    o[3] = g.sx * g.s;
    o[4] = g.sx * g.c;
    o[5] = g.ty;
  } else {  // cv_ops.py:56-63
    o[0] = g.sx * g.c + g.sh * g.sy * g.s;
    o[1] = -g.sx * g.s + g.sh * g.sy * g.c;
    o[2] = g.tx;
    o[3] = g.sy * g.s;
    o[4] = g.sy * g.c;
    o[5] = g.ty;
  }
}

__device__ __forceinline__ void gt_backward(const GtState &g, int similarity,
                                            const float *go, float *gp) {
  float gsx, gsy, gsh, gth;
  if (similarity) {
    gsx = go[0] * g.c - go[1] * g.s + go[3] * g.s + go[4] * g.c;
    gsy = 0.f;
    gsh = 0.f;
    gth = g.sx * (-go[0] * g.s - go[1] * g.c + go[3] * g.c - go[4] * g.s);
  } else {
    gsx = go[0] * g.c - go[1] * g.s;
    gsy = go[0] * g.sh * g.s + go[1] * g.sh * g.c + go[3] * g.s + go[4] * g.c;
    gsh = go[0] * g.sy * g.s + go[1] * g.sy * g.c;
    gth = go[0] * (-g.sx * g.s + g.sh * g.sy * g.c) + go[1] * (-g.sx * g.c - g.sh * g.sy * g.s) +
          go[3] * g.sy * g.c - go[4] * g.sy * g.s;
  }
  gp[0] = gsx * g.dsx;
  gp[1] = gsy * g.dsy;
  gp[2] = gth * g.dth;
  gp[3] = gsh * g.dsh;
  gp[4] = go[2] * g.dtx;
  gp[5] = go[5] * g.dty;
}

}  // namespace scae_gt
