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
//   fwd1: row c of Wk, Wv, Wq (one workgroup per row): [wkf|bkf], wv2e = Wv W2e + [0|bv], q[:,c]
//   fwd2: row c of Wo: [wvf|bvf] = Wo wv2e + [0|bo]
//   bwdA: column jobs, one workgroup per output column, threads over j:
//         gv2e = Wo^T [g_wvf|g_bvf],  t1 = Wk^T [g_wkf|g_bkf],  d_seeds = g_q Wq
//   bwdB: row jobs (outer products, K = O or D+1): d_Wq, d_Wk, d_Wo, d_Wv and
//         the four bias gradients; column jobs: [d_W2|d_b2] = t1 + Wv^T gv2e
#include "common.h"

namespace {
constexpr int NT = 256;

// Row job (workgroup = one row c of the C x C matrices, 256 threads over j):
// sums[v][d] = sum_j rows[v][j] * ext[j][d] for NV rows at once, ext = [mat | col]
// (C x (D+1)); the D+1 per-thread products are reduced wave-wide, then across
// the 4 waves through `red` (NV * (D+1) * 4 floats).  Results valid for
// threads t < NV*(D+1): value index t = v*(D+1) + d.
template <int D, int NV>
__device__ __forceinline__ float rows_times_ext(const float *const (&rows)[NV], const float *mat,
                                                int mat_ld, const float *col, int C, float *red) {
  constexpr int DP = D + 1;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  float acc[NV][DP];
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int d = 0; d < DP; ++d) acc[v][d] = 0.f;
  for (int j = t; j < C; j += NT) {
    float e[DP];
#pragma unroll
    for (int d = 0; d < D; ++d) e[d] = mat[(size_t)j * mat_ld + d];
    e[D] = col ? col[j] : mat[(size_t)j * mat_ld + D];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const float r = rows[v][j];
#pragma unroll
      for (int d = 0; d < DP; ++d) acc[v][d] = fmaf(r, e[d], acc[v][d]);
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      const float s = scae::wave_sum(acc[v][d]);
      if (lane == 0) red[(v * DP + d) * 4 + wave] = s;
    }
  __syncthreads();
  float out = 0.f;
  if (t < NV * DP) out = (red[t * 4] + red[t * 4 + 1]) + (red[t * 4 + 2] + red[t * 4 + 3]);
  return out;
}

template <int D>
__global__ __launch_bounds__(NT) void fold_fwd1_kernel(scae_seed_fold_desc a) {
  constexpr int DP = D + 1;
  __shared__ float red[2 * DP * 4];
  const int c = blockIdx.x, t = threadIdx.x, C = a.C;
  const float *const rows[2] = {a.wk + (size_t)c * C, a.wv + (size_t)c * C};
  const float s = rows_times_ext<D, 2>(rows, a.w2, D, a.b2, C, red);
  if (t < 2 * DP) {
    const int v = t / DP, d = t - v * DP;
    if (v == 0) {
      if (d < D)
        a.wkf[c * D + d] = s;
      else
        a.bkf[c] = s + a.bk[c];
    } else {
      a.wv2e[c * DP + d] = d < D ? s : s + a.bv[c];
    }
  }
  // q[:, c] = seeds Wq[c, :]^T + bq[c]: 8 lanes per seed o, each an 8-strided
  // slice of j, so that all loads of the dot product are in flight at once
  const float *wq = a.wq + (size_t)c * C;
  const int sub = t & 7;
  for (int o = t >> 3; o < a.O; o += NT / 8) {
    const float *sd = a.seeds + (size_t)o * C;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    for (int j = sub; j < C; j += 32) {  // C % 64 == 0
      p0 = fmaf(wq[j], sd[j], p0);
      p1 = fmaf(wq[j + 8], sd[j + 8], p1);
      p2 = fmaf(wq[j + 16], sd[j + 16], p2);
      p3 = fmaf(wq[j + 24], sd[j + 24], p3);
    }
    float r = (p0 + p1) + (p2 + p3);
    r += __shfl_xor(r, 1, 64);
    r += __shfl_xor(r, 2, 64);
    r += __shfl_xor(r, 4, 64);
    if (sub == 0) a.q[(size_t)o * C + c] = r + a.bq[c];
  }
}

template <int D>
__global__ __launch_bounds__(NT) void fold_fwd2_kernel(scae_seed_fold_desc a) {
  constexpr int DP = D + 1;
  __shared__ float red[DP * 4];
  const int c = blockIdx.x, t = threadIdx.x, C = a.C;
  const float *const rows[1] = {a.wo + (size_t)c * C};
  const float s = rows_times_ext<D, 1>(rows, a.wv2e, DP, nullptr, C, red);
  if (t < DP) {
    if (t < D)
      a.wvf[c * D + t] = s;
    else
      a.bvf[c] = s + a.bo[c];
  }
}

// Column job: out[j] = sum_c W[c][j] * g(c).  Workgroup (C, parts): thread
// (j, part) walks a quarter of the rows (coalesced W rows, 8 loads in flight),
// the parts are summed through LDS.  Result valid for threadIdx.y == 0.
template <class G>
__device__ __forceinline__ float col_dot(const float *W, int C, float *red, G g) {
  const int j = threadIdx.x, part = threadIdx.y, parts = blockDim.y;
  const int per = (C + parts - 1) / parts, cb = part * per, ce = min(C, cb + per);
  float s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s[u] = 0.f;
  int c = cb;
  for (; c + 8 <= ce; c += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = fmaf(W[(size_t)(c + u) * C + j], g(c + u), s[u]);
  }
  for (; c < ce; ++c) s[0] = fmaf(W[(size_t)c * C + j], g(c), s[0]);
  const float tot = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  red[part * C + j] = tot;
  __syncthreads();
  float out = 0.f;
  if (part == 0)
    for (int p = 0; p < parts; ++p) out += red[p * C + j];
  return out;
}

// [g_w | g_b](c, d)
__device__ __forceinline__ float ext_at(const float *gw, const float *gb, int D, int c, int d) {
  return d < D ? gw[c * D + d] : gb[c];
}

// grid: 2*(D+1) + O workgroups of (C, parts) threads; LDS parts*C floats
__global__ void fold_bwdA_kernel(scae_seed_fold_desc a, scae_seed_fold_grads g) {
  extern __shared__ float red[];
  const int C = a.C, D = a.D, DP = D + 1, j = threadIdx.x;
  const bool lead = threadIdx.y == 0;
  int job = blockIdx.x;
  if (job < DP) {  // gv2e[:, d] = Wo^T [g_wvf | g_bvf][:, d]
    const int d = job;
    const float s =
        col_dot(a.wo, C, red, [&](int c) { return ext_at(g.g_wvf, g.g_bvf, D, c, d); });
    if (lead) g.gv2e[j * DP + d] = s;
    return;
  }
  job -= DP;
  if (job < DP) {  // t1[:, d] = Wk^T [g_wkf | g_bkf][:, d]
    const int d = job;
    const float s =
        col_dot(a.wk, C, red, [&](int c) { return ext_at(g.g_wkf, g.g_bkf, D, c, d); });
    if (lead) g.t1[j * DP + d] = s;
    return;
  }
  const int o = job - DP;  // d_seeds[o, :] = g_q[o, :] Wq
  const float s = col_dot(a.wq, C, red, [&](int c) { return g.g_q[(size_t)o * C + c]; });
  if (lead) g.d_seeds[(size_t)o * C + j] = s;
}

// grid: C row workgroups + (D+1) column workgroups, (C, parts) threads each.
// Row c: the four outer-product rows d_Wq, d_Wk, d_Wv, d_Wo[c, :] are shared
// out over the parts.
__global__ void fold_bwdB_kernel(scae_seed_fold_desc a, scae_seed_fold_grads g) {
  extern __shared__ float red[];
  const int C = a.C, D = a.D, DP = D + 1, j = threadIdx.x;
  if ((int)blockIdx.x >= C) {  // [d_W2 | d_b2][:, d] = t1[:, d] + Wv^T gv2e[:, d]
    const int d = blockIdx.x - C;
    const float s = col_dot(a.wv, C, red, [&](int c) { return g.gv2e[c * DP + d]; });
    if (threadIdx.y == 0) {
      if (d < D)
        g.d_w2[j * D + d] = s + g.t1[j * DP + d];
      else
        g.d_b2[j] = s + g.t1[j * DP + d];
    }
    return;
  }
  const int c = blockIdx.x;
  const size_t e = (size_t)c * C + j;
  for (int which = threadIdx.y; which < 4; which += blockDim.y) {
    float acc = 0.f;
    if (which == 0) {
      float sq = 0.f;
#pragma unroll 8
      for (int o = 0; o < a.O; ++o) {
        const float gq = g.g_q[(size_t)o * C + c];
        acc = fmaf(gq, a.seeds[(size_t)o * C + j], acc);
        sq += gq;
      }
      g.d_wq[e] = acc;
      if (j == 0) g.d_bq[c] = sq;
    } else if (which == 1) {
#pragma unroll 8
      for (int d = 0; d < D; ++d) acc = fmaf(g.g_wkf[c * D + d], a.w2[j * D + d], acc);
      g.d_wk[e] = fmaf(g.g_bkf[c], a.b2[j], acc);
      if (j == 0) g.d_bk[c] = g.g_bkf[c];
    } else if (which == 2) {
#pragma unroll 8
      for (int d = 0; d < D; ++d) acc = fmaf(g.gv2e[c * DP + d], a.w2[j * D + d], acc);
      g.d_wv[e] = fmaf(g.gv2e[c * DP + D], a.b2[j], acc);
      if (j == 0) g.d_bv[c] = g.gv2e[c * DP + D];
    } else {
#pragma unroll 8
      for (int d = 0; d < D; ++d) acc = fmaf(g.g_wvf[c * D + d], a.wv2e[j * DP + d], acc);
      g.d_wo[e] = fmaf(g.g_bvf[c], a.wv2e[j * DP + D], acc);
      if (j == 0) g.d_bo[c] = g.g_bvf[c];
    }
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
  return O > 0 && O <= 64 && C >= 64 && C % 64 == 0 && C <= 1024 && (D == 8 || D == 16 || D == 32);
}

extern "C" int scae_seed_fold_fwd_f32(const scae_seed_fold_desc *desc, void *stream) {
  int rc = check(desc);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
#define SCAE_FOLD_FWD(DD)                                                              \
  case DD:                                                                             \
    hipLaunchKernelGGL(fold_fwd1_kernel<DD>, dim3(desc->C), dim3(NT), 0, st, *desc);   \
    hipLaunchKernelGGL(fold_fwd2_kernel<DD>, dim3(desc->C), dim3(NT), 0, st, *desc);   \
    break;
  switch (desc->D) {
    SCAE_FOLD_FWD(8) SCAE_FOLD_FWD(16) SCAE_FOLD_FWD(32)
    default: return SCAE_ERR_UNSUPPORTED;
  }
#undef SCAE_FOLD_FWD
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
  const int DP = desc->D + 1, C = desc->C;
  const int parts = 1024 / C >= 4 ? 4 : (1024 / C >= 2 ? 2 : 1);
  const size_t lds = (size_t)parts * C * sizeof(float);
  hipLaunchKernelGGL(fold_bwdA_kernel, dim3(2 * DP + desc->O), dim3(C, parts), lds,
                     (hipStream_t)stream, *desc, *g);
  hipLaunchKernelGGL(fold_bwdB_kernel, dim3(C + DP), dim3(C, parts), lds, (hipStream_t)stream,
                     *desc, *g);
  return scae_launch_status();
}
