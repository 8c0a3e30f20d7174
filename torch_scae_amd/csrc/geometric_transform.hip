// K5 -- geometric_transform for gfx950: pose 6-vector -> 2x3 affine (or 3x3).
// Replaces cv_ops.py:20-76 (a chain of ~25 elementwise ATen ops on split
// views) with one memory-bound elementwise kernel, forward and backward.
#include "geometric_transform.h"

namespace {
using namespace scae_gt;
constexpr int NT = 256;

struct GtOut {
  float o[6];
};

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
