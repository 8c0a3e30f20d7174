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
  scae::launch(gt_fwd_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0,
                     (hipStream_t)stream, pose, out, n, similarity, nonlinear, as_matrix);
  return scae_launch_status();
}

extern "C" int scae_geometric_transform_bwd_f32(const float *pose, const float *gout,
                                                float *gpose, int64_t n, int similarity,
                                                int nonlinear, int as_matrix,
                                                void *stream) {
  SCAE_REQUIRE(pose && gout && gpose && n > 0);
  scae::launch(gt_bwd_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0,
                     (hipStream_t)stream, pose, gout, gpose, n, similarity, nonlinear,
                     as_matrix);
  return scae_launch_status();
}

// ---------------------------------------------------------------------------------------
// vote = parent (x) child: the 3 x 3 products of the hierarchical CapsuleLayer.forward
// (object_decoder.py:184-191: torch.matmul(cvr.repeat(1, 1, n_votes, 1, 1), cpr)); one
// left matrix per capsule against its V right matrices.  A thread per (capsule, vote);
// the backward's sum over the votes (gradient of the left matrix) is a wave per capsule.
namespace {
__global__ __launch_bounds__(NT) void mat3_fwd_kernel(const float *__restrict__ left,
                                                      const float *__restrict__ right,
                                                      float *__restrict__ out, int64_t n, int V) {
  const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  const float *a = left + (i / V) * 9, *b = right + i * 9;
  float *o = out + i * 9;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      o[3 * r + c] = a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c] + a[3 * r + 2] * b[6 + c];
}

// g_right = left^T g_out (per vote); g_left = sum_v g_out right^T (per capsule)
__global__ __launch_bounds__(NT) void mat3_bwd_kernel(const float *__restrict__ left,
                                                      const float *__restrict__ right,
                                                      const float *__restrict__ gout,
                                                      float *__restrict__ gleft,
                                                      float *__restrict__ gright, int64_t ncaps,
                                                      int V) {
  const int lane = threadIdx.x & 63;
  const int64_t cap = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (cap >= ncaps) return;   // (wave-uniform)
  const float *a = left + cap * 9;
  float acc[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) acc[e] = 0.f;
  for (int v = lane; v < V; v += 64) {
    const float *b = right + (cap * V + v) * 9, *g = gout + (cap * V + v) * 9;
    float *gb = gright + (cap * V + v) * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gb[3 * r + c] = a[r] * g[c] + a[3 + r] * g[3 + c] + a[6 + r] * g[6 + c];
        acc[3 * r + c] += g[3 * r] * b[3 * c] + g[3 * r + 1] * b[3 * c + 1] + g[3 * r + 2] * b[3 * c + 2];
      }
  }
#pragma unroll
  for (int e = 0; e < 9; ++e) acc[e] = scae::wave_sum(acc[e]);
  if (lane == 0 && gleft) {
#pragma unroll
    for (int e = 0; e < 9; ++e) gleft[cap * 9 + e] = acc[e];
  }
}
}  // namespace

extern "C" int scae_mat3_mul_fwd_f32(const float *left, const float *right, float *out,
                                     int64_t n_caps, int V, void *stream) {
  SCAE_REQUIRE(left && right && out && n_caps > 0 && V > 0);
  const int64_t n = n_caps * V;
  scae::launch(mat3_fwd_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0,
                     (hipStream_t)stream, left, right, out, n, V);
  return scae_launch_status();
}

extern "C" int scae_mat3_mul_bwd_f32(const float *left, const float *right, const float *gout,
                                     float *gleft, float *gright, int64_t n_caps, int V,
                                     void *stream) {
  SCAE_REQUIRE(left && right && gout && gright && n_caps > 0 && V > 0);
  const int per = NT / 64;
  scae::launch(mat3_bwd_kernel, dim3((unsigned)((n_caps + per - 1) / per)), dim3(NT), 0,
                     (hipStream_t)stream, left, right, gout, gleft, gright, n_caps, V);
  return scae_launch_status();
}
