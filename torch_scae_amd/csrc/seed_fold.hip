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
// they cost more than the attention itself.  Here: one launch forward (seed_fold_dev.h; the
// same jobs can ride in the training step's prologue launch, step_prologue.hip), two backward.
//   fwd: row blocks of Wk, Wv (NT/(D+1) rows per workgroup, W2e in LDS, a thread per
//         (row, column)): [wkf|bkf], wv2e = Wv W2e + [0|bv]; row blocks of Wo:
//         [wvf|bvf] = (Wo Wv) W2e + [0 | Wo bv + bo]; column blocks of q
//   bwdA: column jobs, one workgroup per output column, threads over j:
//         gv2e = Wo^T [g_wvf|g_bvf],  t1 = Wk^T [g_wkf|g_bkf],  d_seeds = g_q Wq
//   bwdB: row jobs (outer products, K = O or D+1): d_Wq, d_Wk, d_Wo, d_Wv and
//         the four bias gradients; column jobs: [d_W2|d_b2] = t1 + Wv^T gv2e
#include "common.h"
#include "seed_fold_dev.h"

namespace {
constexpr int NT = 256;

template <int D>
__global__ __launch_bounds__(NT) void fold_fwd_kernel(scae_seed_fold_desc a, scae_fold::Plan pl) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  scae_fold::forward_block<D>(a, pl, blockIdx.x, lds);
}

// Column job: out[j] = sum_c W[c][j] * g(c).  Workgroup (C, parts): thread
// (j, part) walks a quarter of the rows (coalesced W rows, 8 loads in flight),
// the parts are summed through LDS.  Result valid for threadIdx.y == 0.
template <class G>
__device__ __forceinline__ float col_dot(const float *W, int C, float *red, G g) {
  const int j = threadIdx.x, part = threadIdx.y, parts = blockDim.y;
  const int per = (C + parts - 1) / parts, cb = part * per, ce = min(C, cb + per);
  // 16 independent (coalesced) row loads in flight per thread: the dot product is a chain of
  // L2 round trips otherwise
  constexpr int U = 16;
  float s[U];
#pragma unroll
  for (int u = 0; u < U; ++u) s[u] = 0.f;
  int c = cb;
  for (; c + U <= ce; c += U) {
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) w[u] = W[(size_t)(c + u) * C + j];
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] = fmaf(w[u], g(c + u), s[u]);
  }
  for (; c < ce; ++c) s[0] = fmaf(W[(size_t)c * C + j], g(c), s[0]);
  float tot = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) tot += s[u];
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
  if (!(O > 0 && O <= 64 && C >= 64 && C % 64 == 0 && C <= 1024 && (D == 8 || D == 16 || D == 32)))
    return 0;
  return scae_fold::lds_bytes(C, D) <= 160 * 1024;
}

extern "C" int scae_seed_fold_fwd_f32(const scae_seed_fold_desc *desc, void *stream) {
  int rc = check(desc);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const scae_fold::Plan pl = scae_fold::plan(desc->C, desc->D);
  const size_t lds = scae_fold::lds_bytes(desc->C, desc->D);
#define SCAE_FOLD_FWD(DD)                                                                  \
  case DD: {                                                                               \
    if (lds > 48 * 1024) {                                                                 \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fold_fwd_kernel<DD>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return (int)e;                                                  \
    }                                                                                      \
    hipLaunchKernelGGL(fold_fwd_kernel<DD>, dim3(pl.blocks()), dim3(NT), lds, st, *desc, pl); \
  } break;
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
