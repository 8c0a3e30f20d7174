// K5 -- geometric_transform for gfx950: pose 6-vector -> 2x3 affine (or 3x3).
// Replaces cv_ops.py:20-76 (a chain of ~25 elementwise ATen ops on split
// views) with one memory-bound elementwise kernel, forward and backward.
#include "common.h"

namespace {
constexpr int NT = 256;
constexpr float kTwoPi = 6.283185307179586f;

struct GtOut {
  float o[6];
};

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

__global__ __launch_bounds__(NT) void gt_fwd_kernel(const float *__restrict__ pose,
                                                    float *__restrict__ out, int64_t n,
                                                    int similarity, int nonlinear,
                                                    int as_matrix) {
  const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  float p[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) p[j] = pose[i * 6 + j];
  GtState g;
  gt_eval(p, nonlinear, g);
  float o[6];
  gt_rows(g, similarity, o);
  const int stride = as_matrix ? 9 : 6;
#pragma unroll
  for (int j = 0; j < 6; ++j) out[i * stride + j] = o[j];
  if (as_matrix) {  // cv_ops.py:68-74
    out[i * 9 + 6] = 0.f;
    out[i * 9 + 7] = 0.f;
    out[i * 9 + 8] = 1.f;
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

__global__ __launch_bounds__(NT) void gt_bwd_kernel(const float *__restrict__ pose,
                                                    const float *__restrict__ gout,
                                                    float *__restrict__ gpose, int64_t n,
                                                    int similarity, int nonlinear,
                                                    int as_matrix) {
  const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  float p[6], go[6], gp[6];
  const int stride = as_matrix ? 9 : 6;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    p[j] = pose[i * 6 + j];
    go[j] = gout[i * stride + j];
  }
  GtState g;
  gt_eval(p, nonlinear, g);
  gt_backward(g, similarity, go, gp);
#pragma unroll
  for (int j = 0; j < 6; ++j) gpose[i * 6 + j] = gp[j];
}
}  // namespace

extern "C" int scae_geometric_transform_fwd_f32(const float *pose, float *out, int64_t n,
                                                int similarity, int nonlinear,
                                                int as_matrix, void *stream) {
  SCAE_REQUIRE(pose && out && n > 0);
  hipLaunchKernelGGL(gt_fwd_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0,
                     (hipStream_t)stream, pose, out, n, similarity, nonlinear, as_matrix);
  return scae_launch_status();
}

extern "C" int scae_geometric_transform_bwd_f32(const float *pose, const float *gout,
                                                float *gpose, int64_t n, int similarity,
                                                int nonlinear, int as_matrix,
                                                void *stream) {
  SCAE_REQUIRE(pose && gout && gpose && n > 0);
  hipLaunchKernelGGL(gt_bwd_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0,
                     (hipStream_t)stream, pose, gout, gpose, n, similarity, nonlinear,
                     as_matrix);
  return scae_launch_status();
}
