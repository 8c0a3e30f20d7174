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
// same jobs can ride in the training step's prologue launch, step_prologue.hip), one backward.
//   fwd: row blocks of Wk, Wv (NT/(D+1) rows per workgroup, W2e in LDS, a thread per
//         (row, column)): [wkf|bkf], wv2e = Wv W2e + [0|bv]; row blocks of Wo:
//         [wvf|bvf] = (Wo Wv) W2e + [0 | Wo bv + bo]; column blocks of q
//   bwd: column-block jobs: [d_W2|d_b2] (with (Wo Wv), which the forward keeps),
//         d_seeds = g_q Wq; row jobs (outer products, K = O or D+1): d_Wq, d_Wk, d_Wo, d_Wv
//         and the four bias gradients -- see fold_bwd_kernel
#include "common.h"
#include "seed_fold_dev.h"

namespace {
constexpr int NT = 256;

template <int D>
__global__ __launch_bounds__(NT) void fold_fwd_kernel(scae_seed_fold_desc a, scae_fold::Plan pl) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  scae_fold::forward_block<D>(a, pl, blockIdx.x, lds);
}

// [g_w | g_b](c, d)
__device__ __forceinline__ float ext_at(const float *gw, const float *gb, int D, int c, int d) {
  return d < D ? gw[c * D + d] : gb[c];
}

// ---- backward -------------------------------------------------------------------------
// Column jobs out[j][d] = sum_c W[c][j] G(c, d) (W: C x C, G: C x (D+1) or C x O -- a few
// MFLOP) run on the matrix cores, one 16 x 16 output tile per wave, operands straight from
// global memory: lane (r, q) of v_mfma_f32_16x16x4_f32 supplies W[4 s + q][j0 + r] and
// G(4 s + q, n0 + r) -- 64-byte row segments, 16 k-steps of independent loads in flight.
// (The VALU forms tried first were bound by one CU's path to L2 -- a workgroup per output
// column d reads all of W per dot product, ~5 us each -- or by LDS issue: G broadcast from
// LDS to a column-per-lane layout costs 96 ds_read_b128 per lane.)
typedef float f32x4 __attribute__((ext_vector_type(4)));
// k-steps [0, steps) of the wave's K slice starting at row kb
template <int STEPS, class GP>
__device__ __forceinline__ void mfma_cols(f32x4 &acc, const float *W, int C, int kb, int j0, int n,
                                          int r, int q, GP gp) {
  float av[STEPS], bv[STEPS];
#pragma unroll
  for (int s_ = 0; s_ < STEPS; ++s_) {
    const int c = kb + 4 * s_ + q;
    av[s_] = W[(size_t)c * C + j0 + r];
    bv[s_] = *gp(c, n);
  }
#pragma unroll
  for (int s_ = 0; s_ < STEPS; ++s_)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s_], bv[s_], acc, 0, 0, 0);
}
// &[g_w | g_b](c, d): one load whichever side (a select on the address, no branch)
__device__ __forceinline__ const float *ext_ptr(const float *gw, const float *gb, int D, int c,
                                                int d) {
  return d < D ? gw + c * D + d : gb + c;
}

// ONE launch of (C, parts)-thread workgroups:
//   column workgroups, one per 16 x 16 output tile, the waves splitting K; the partial tiles
//   meet in LDS:
//       [d_W2 | d_b2] = Wk^T [g_wkf|g_bkf] + (Wo Wv)^T [g_wvf|g_bvf]   ((Wo Wv) from the forward)
//       d_seeds = g_q Wq
//   C row workgroups: the outer-product rows d_Wq, d_Wk, d_Wv, d_Wo[c, :] (shared out over the
//       parts) and the four bias gradients; row c of gv2e = Wo^T [g_wvf|g_bvf], which d_Wv
//       needs, is recomputed by the workgroup (D+1 dot products down column c of Wo).
// No job waits for another workgroup.  (Before: a first kernel produced gv2e and Wk^T g, a
// second one consumed them.)
struct BwdPlan {
  int jt;          // 16-column tiles: C / 16
  int nt_w2, nt_seed;   // 16-output tiles: ceil((D+1) / 16), ceil(O / 16)
  int ncol;        // column workgroups (tiles)
};
// Workgroup `blk` as thread (tx, ty) of a (C, parts) block.  STEPS = 4-k steps per K slice;
// the K slices (C / (4 STEPS) of them, one per wave of the (C, 16 / STEPS)-thread launch) are
// walked VW at a time by each wave of a smaller block (conv_mfma.hip runs these workgroups
// as 256-thread riders of a convolution launch: parts = 1, VW = 4 at C = 256): the same
// partial tiles meet in the same order, so the result does not depend on the block shape.
template <int STEPS, int VW = 1>
__device__ __forceinline__ void fold_bwd_body(const scae_seed_fold_desc &a,
                                              const scae_seed_fold_grads &g, const BwdPlan &pl,
                                              float *lds, int blk, int tx, int ty, int parts) {
  const int C = a.C, D = a.D, DP = D + 1, j = tx;
  if (blk < pl.ncol) {
    const int tid = ty * C + tx, nw = VW * ((C * parts) >> 6);
    const int wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
    int tile = blk;
    const bool w2 = tile < pl.jt * pl.nt_w2;   // workgroup-uniform
    if (!w2) tile -= pl.jt * pl.nt_w2;
    const int j0 = (tile % pl.jt) * 16, n0 = (tile / pl.jt) * 16, nout = w2 ? DP : a.O;
    const int n = min(n0 + r, nout - 1);   // lanes past the last output repeat it (dropped below)
#pragma unroll
    for (int v = 0; v < VW; ++v) {
      const int slice = wave * VW + v, kb = slice * 4 * STEPS;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (w2) {
        mfma_cols<STEPS>(acc, a.wk, C, kb, j0, n, r, q,
                         [&](int c, int d) { return ext_ptr(g.g_wkf, g.g_bkf, D, c, d); });
        mfma_cols<STEPS>(acc, a.wowv, C, kb, j0, n, r, q,
                         [&](int c, int d) { return ext_ptr(g.g_wvf, g.g_bvf, D, c, d); });
      } else {
        mfma_cols<STEPS>(acc, a.wq, C, kb, j0, n, r, q,
                         [&](int c, int o) { return g.g_q + (size_t)o * C + c; });
      }
      // partial tiles [slice][row 4 q + e][col r] meet
#pragma unroll
      for (int e = 0; e < 4; ++e) lds[(slice * 16 + 4 * q + e) * 16 + r] = acc[e];
    }
    __syncthreads();
    if (tid < 256) {
      const int row = tid >> 4, col = tid & 15;   // row: column jj of W, col: output
      float sum = 0.f;
      for (int w = 0; w < nw; ++w) sum += lds[(w * 16 + row) * 16 + col];
      const int jj = j0 + row, nn = n0 + col;
      if (nn < nout) {
        if (!w2)
          g.d_seeds[(size_t)nn * C + jj] = sum;
        else if (nn < D)
          g.d_w2[jj * D + nn] = sum;
        else
          g.d_b2[jj] = sum;
      }
    }
    return;
  }
  const int c = blk - pl.ncol;
  float *vec = lds + 64 * 16;   // D + 1 floats behind the wave partials
  {  // gv2e[c, d] = sum_i Wo[i][c] [g_wvf|g_bvf][i, d]: thread (i = j, part) takes d = part, part + parts, ..
    const float wo = a.wo[(size_t)j * C + c];
    const int nw = C >> 6, wave = j >> 6;
    for (int d = ty; d < DP; d += parts) {
      const float v = scae::wave_sum(wo * ext_at(g.g_wvf, g.g_bvf, D, j, d));
      if ((j & 63) == 0) lds[d * nw + wave] = v;
    }
    __syncthreads();
    const int t = ty * C + j;
    if (t < DP) {
      float v = 0.f;
      for (int w = 0; w < nw; ++w) v += lds[t * nw + w];
      vec[t] = v;
    }
    __syncthreads();
  }
  const size_t e = (size_t)c * C + j;
  for (int which = ty; which < 4; which += parts) {
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
      for (int d = 0; d < D; ++d) acc = fmaf(vec[d], a.w2[j * D + d], acc);
      g.d_wv[e] = fmaf(vec[D], a.b2[j], acc);
      if (j == 0) g.d_bv[c] = vec[D];
    } else {
#pragma unroll 8
      for (int d = 0; d < D; ++d) acc = fmaf(g.g_wvf[c * D + d], a.wv2e[j * DP + d], acc);
      g.d_wo[e] = fmaf(g.g_bvf[c], a.wv2e[j * DP + D], acc);
      if (j == 0) g.d_bo[c] = g.g_bvf[c];
    }
  }
}
template <int STEPS>   // 4-k steps per wave: (C / waves) / 4 = 16 / parts
__global__ void fold_bwd_kernel(scae_seed_fold_desc a, scae_seed_fold_grads g, BwdPlan pl) {
  extern __shared__ float lds[];
  fold_bwd_body<STEPS>(a, g, pl, lds, blockIdx.x, threadIdx.x, threadIdx.y, blockDim.y);
}
// (C, parts) block shape, its dynamic LDS and the plan of a backward launch
inline void bwd_shape(const scae_seed_fold_desc *desc, BwdPlan &pl, int &parts, size_t &lds) {
  const int DP = desc->D + 1, C = desc->C;
  parts = 1024 / C >= 4 ? 4 : (1024 / C >= 2 ? 2 : 1);
  pl = BwdPlan{C / 16, (DP + 15) / 16, (desc->O + 15) / 16, 0};
  pl.ncol = pl.jt * (pl.nt_w2 + pl.nt_seed);
  // column jobs: waves x 16 x 16 partial tiles; row jobs: wave partials + a gv2e row
  lds = (size_t)(C * parts / 64) * 256 * sizeof(float) + 8192;
}
inline int check_grads(const scae_seed_fold_grads *g) {
  return g && g->g_q && g->g_wkf && g->g_bkf && g->g_wvf && g->g_bvf && g->d_seeds && g->d_wq &&
                 g->d_bq && g->d_wk && g->d_bk && g->d_wv && g->d_bv && g->d_wo && g->d_bo &&
                 g->d_w2 && g->d_b2
             ? SCAE_OK
             : SCAE_ERR_BAD_ARG;
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

#ifndef SCAE_DEVICE_ONLY   // (conv_mfma.hip includes this file for its device code)
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
    scae::launch(fold_fwd_kernel<DD>, dim3(pl.blocks()), dim3(NT), lds, st, *desc, pl); \
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
  rc = check_grads(g);
  if (rc) return rc;
  SCAE_REQUIRE(desc->wowv);
  BwdPlan pl;
  int parts;
  size_t lds;
  bwd_shape(desc, pl, parts, lds);
  const dim3 grid(pl.ncol + desc->C), block(desc->C, parts);
  hipStream_t st = (hipStream_t)stream;
  if (parts == 4)
    scae::launch(fold_bwd_kernel<4>, grid, block, lds, st, *desc, *g, pl);
  else if (parts == 2)
    scae::launch(fold_bwd_kernel<8>, grid, block, lds, st, *desc, *g, pl);
  else
    scae::launch(fold_bwd_kernel<16>, grid, block, lds, st, *desc, *g, pl);
  return scae_launch_status();
}
#endif  // SCAE_DEVICE_ONLY
