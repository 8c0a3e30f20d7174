// K2d -- batch-invariant weight folding for the output attention (K2c).
//
// set_transformer.py:218-223 evaluates fc2 and the q/k/v/o projections of
// MultiHeadQKVAttention(seeds, z, z) per set element.  K2c instead consumes
//   q   = seeds Wq^T + bq                        (O, C)
//   wkf = Wk W2,          bkf = Wk b2 + bk       (C, D), (C)
//   wvf = Wo (Wv W2),     bvf = Wo (Wv b2 + bv) + bo
// which only depend on parameters.  With W2e = [W2 | b2] (C x (D+1)) these are
// three "C x C times C x (D+1)" products, a few MFLOP -- but as a chain of
// library GEMM / GEMV / add launches (and twice as many in the backward pass)
// they cost more than the attention itself.  Here: two launches forward, two
// backward.
//   fwd1: row c of Wk, Wv (one wave each row): [wkf|bkf], wv2e = Wv W2e + [0|bv], q[:,c]
//   fwd2: row c of Wo: [wvf|bvf] = Wo wv2e + [0|bo]
//   bwdA: column jobs, one workgroup per output column, threads over j:
//         gv2e = Wo^T [g_wvf|g_bvf],  t1 = Wk^T [g_wkf|g_bkf],  d_seeds = g_q Wq
//   bwdB: row jobs (outer products, K = O or D+1): d_Wq, d_Wk, d_Wo, d_Wv and
//         the four bias gradients; column jobs: [d_W2|d_b2] = t1 + Wv^T gv2e
#include "common.h"

namespace {
constexpr int CMAX = 512;  // lanes hold C/64 <= 8 elements of a weight row

// s_mat (C x DP, LDS) <- [mat (C x D) | col (C)] or a ready C x DP matrix
__device__ __forceinline__ void stage_ext(float *s_mat, const float *mat, const float *col, int C,
                                          int D) {
  const int DP = D + 1;
  if (col) {
    for (int e = threadIdx.x; e < C * D; e += blockDim.x) {
      const int j = e / D, d = e - j * D;
      s_mat[j * DP + d] = mat[e];
    }
    for (int j = threadIdx.x; j < C; j += blockDim.x) s_mat[j * DP + D] = col[j];
  } else {
    for (int e = threadIdx.x; e < C * DP; e += blockDim.x) s_mat[e] = mat[e];
  }
  __syncthreads();
}

__device__ __forceinline__ void load_row(float (&r)[CMAX / 64], const float *row, int C,
                                         int lane) {
#pragma unroll
  for (int i = 0; i < CMAX / 64; ++i) r[i] = lane + 64 * i < C ? row[lane + 64 * i] : 0.f;
}

// sum_j r[j] * s_mat[j][d]   (valid in every lane)
__device__ __forceinline__ float row_dot_col(const float (&r)[CMAX / 64], const float *s_mat,
                                             int DP, int d, int C, int lane) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CMAX / 64; ++i)
    if (lane + 64 * i < C) s = fmaf(r[i], s_mat[(lane + 64 * i) * DP + d], s);
  return scae::wave_sum(s);
}

__global__ __launch_bounds__(64) void fold_fwd1_kernel(scae_seed_fold_desc a) {
  extern __shared__ float s_mat[];
  const int c = blockIdx.x, lane = threadIdx.x, C = a.C, D = a.D, DP = D + 1;
  stage_ext(s_mat, a.w2, a.b2, C, D);
  float rk[CMAX / 64], rv[CMAX / 64], rq[CMAX / 64];
  load_row(rk, a.wk + (size_t)c * C, C, lane);
  load_row(rv, a.wv + (size_t)c * C, C, lane);
  load_row(rq, a.wq + (size_t)c * C, C, lane);
  for (int d = 0; d < DP; ++d) {
    const float sk = row_dot_col(rk, s_mat, DP, d, C, lane);
    const float sv = row_dot_col(rv, s_mat, DP, d, C, lane);
    if (lane == 0) {
      if (d < D) {
        a.wkf[c * D + d] = sk;
        a.wv2e[c * DP + d] = sv;
      } else {
        a.bkf[c] = sk + a.bk[c];
        a.wv2e[c * DP + D] = sv + a.bv[c];
      }
    }
  }
  for (int o = 0; o < a.O; ++o) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CMAX / 64; ++i)
      if (lane + 64 * i < C) s = fmaf(rq[i], a.seeds[(size_t)o * C + lane + 64 * i], s);
    s = scae::wave_sum(s);
    if (lane == 0) a.q[(size_t)o * C + c] = s + a.bq[c];
  }
}

__global__ __launch_bounds__(64) void fold_fwd2_kernel(scae_seed_fold_desc a) {
  extern __shared__ float s_mat[];
  const int c = blockIdx.x, lane = threadIdx.x, C = a.C, D = a.D, DP = D + 1;
  stage_ext(s_mat, a.wv2e, nullptr, C, D);
  float ro[CMAX / 64];
  load_row(ro, a.wo + (size_t)c * C, C, lane);
  for (int d = 0; d < DP; ++d) {
    const float s = row_dot_col(ro, s_mat, DP, d, C, lane);
    if (lane == 0) {
      if (d < D)
        a.wvf[c * D + d] = s;
      else
        a.bvf[c] = s + a.bo[c];
    }
  }
}

// out[j] = sum_c W[c][j] * g(c), one thread per j (coalesced rows of W)
template <class G>
__device__ __forceinline__ float col_dot(const float *W, int C, int j, G g) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int c = 0;
  for (; c + 4 <= C; c += 4) {
    s0 = fmaf(W[(size_t)c * C + j], g(c), s0);
    s1 = fmaf(W[(size_t)(c + 1) * C + j], g(c + 1), s1);
    s2 = fmaf(W[(size_t)(c + 2) * C + j], g(c + 2), s2);
    s3 = fmaf(W[(size_t)(c + 3) * C + j], g(c + 3), s3);
  }
  for (; c < C; ++c) s0 = fmaf(W[(size_t)c * C + j], g(c), s0);
  return (s0 + s1) + (s2 + s3);
}

// [g_w | g_b](c, d)
__device__ __forceinline__ float ext_at(const float *gw, const float *gb, int D, int c, int d) {
  return d < D ? gw[c * D + d] : gb[c];
}

// grid: 2*(D+1) + O workgroups of C threads
__global__ void fold_bwdA_kernel(scae_seed_fold_desc a, scae_seed_fold_grads g) {
  const int C = a.C, D = a.D, DP = D + 1, j = threadIdx.x;
  int job = blockIdx.x;
  if (job < DP) {  // gv2e[:, d] = Wo^T [g_wvf | g_bvf][:, d]
    const int d = job;
    g.gv2e[j * DP + d] =
        col_dot(a.wo, C, j, [&](int c) { return ext_at(g.g_wvf, g.g_bvf, D, c, d); });
    return;
  }
  job -= DP;
  if (job < DP) {  // t1[:, d] = Wk^T [g_wkf | g_bkf][:, d]
    const int d = job;
    g.t1[j * DP + d] =
        col_dot(a.wk, C, j, [&](int c) { return ext_at(g.g_wkf, g.g_bkf, D, c, d); });
    return;
  }
  const int o = job - DP;  // d_seeds[o, :] = g_q[o, :] Wq
  g.d_seeds[(size_t)o * C + j] =
      col_dot(a.wq, C, j, [&](int c) { return g.g_q[(size_t)o * C + c]; });
}

// grid: C row workgroups + (D+1) column workgroups, C threads each
__global__ void fold_bwdB_kernel(scae_seed_fold_desc a, scae_seed_fold_grads g) {
  const int C = a.C, D = a.D, DP = D + 1, j = threadIdx.x;
  if ((int)blockIdx.x >= C) {  // [d_W2 | d_b2][:, d] = t1[:, d] + Wv^T gv2e[:, d]
    const int d = blockIdx.x - C;
    const float s =
        g.t1[j * DP + d] + col_dot(a.wv, C, j, [&](int c) { return g.gv2e[c * DP + d]; });
    if (d < D)
      g.d_w2[j * D + d] = s;
    else
      g.d_b2[j] = s;
    return;
  }
  const int c = blockIdx.x;
  float dq = 0.f, sq = 0.f;
  for (int o = 0; o < a.O; ++o) {
    const float gq = g.g_q[(size_t)o * C + c];
    dq = fmaf(gq, a.seeds[(size_t)o * C + j], dq);
    sq += gq;
  }
  float dk = 0.f, dv = 0.f, dwo = 0.f;
  for (int d = 0; d < DP; ++d) {
    const float w2e = d < D ? a.w2[j * D + d] : a.b2[j];
    dk = fmaf(ext_at(g.g_wkf, g.g_bkf, D, c, d), w2e, dk);
    dv = fmaf(g.gv2e[c * DP + d], w2e, dv);
    dwo = fmaf(ext_at(g.g_wvf, g.g_bvf, D, c, d), a.wv2e[j * DP + d], dwo);
  }
  const size_t e = (size_t)c * C + j;
  g.d_wq[e] = dq;
  g.d_wk[e] = dk;
  g.d_wv[e] = dv;
  g.d_wo[e] = dwo;
  if (j == 0) {
    g.d_bq[c] = sq;
    g.d_bk[c] = g.g_bkf[c];
    g.d_bo[c] = g.g_bvf[c];
    g.d_bv[c] = g.gv2e[c * DP + D];
  }
}

int check(const scae_seed_fold_desc *a) {
  if (!a) return SCAE_ERR_BAD_ARG;
  if (!(a->seeds && a->wq && a->bq && a->wk && a->bk && a->wv && a->bv && a->wo && a->bo &&
        a->w2 && a->b2 && a->q && a->wkf && a->bkf && a->wvf && a->bvf && a->wv2e))
    return SCAE_ERR_BAD_ARG;
  if (a->O <= 0 || a->C <= 0 || a->D <= 0) return SCAE_ERR_BAD_ARG;
  if (!scae_seed_fold_supported(a->O, a->C, a->D)) return SCAE_ERR_UNSUPPORTED;
  return SCAE_OK;
}
}  // namespace

extern "C" int scae_seed_fold_supported(int O, int C, int D) {
  return O > 0 && D > 0 && C > 0 && C % 64 == 0 && C <= CMAX &&
         (size_t)C * (D + 1) * sizeof(float) <= 96 * 1024;
}

extern "C" int scae_seed_fold_fwd_f32(const scae_seed_fold_desc *desc, void *stream) {
  int rc = check(desc);
  if (rc) return rc;
  const size_t lds = (size_t)desc->C * (desc->D + 1) * sizeof(float);
  hipLaunchKernelGGL(fold_fwd1_kernel, dim3(desc->C), dim3(64), lds, (hipStream_t)stream, *desc);
  hipLaunchKernelGGL(fold_fwd2_kernel, dim3(desc->C), dim3(64), lds, (hipStream_t)stream, *desc);
  return scae_launch_status();
}

extern "C" int scae_seed_fold_bwd_f32(const scae_seed_fold_desc *desc,
                                      const scae_seed_fold_grads *grads, void *stream) {
  int rc = check(desc);
  if (rc) return rc;
  const scae_seed_fold_grads *g = grads;
  SCAE_REQUIRE(g && g->g_q && g->g_wkf && g->g_bkf && g->g_wvf && g->g_bvf && g->d_seeds &&
               g->d_wq && g->d_bq && g->d_wk && g->d_bk && g->d_wv && g->d_bv && g->d_wo &&
               g->d_bo && g->d_w2 && g->d_b2 && g->gv2e && g->t1);
  const int DP = desc->D + 1;
  hipLaunchKernelGGL(fold_bwdA_kernel, dim3(2 * DP + desc->O), dim3(desc->C), 0,
                     (hipStream_t)stream, *desc, *g);
  hipLaunchKernelGGL(fold_bwdB_kernel, dim3(desc->C + DP), dim3(desc->C), 0,
                     (hipStream_t)stream, *desc, *g);
  return scae_launch_status();
}
