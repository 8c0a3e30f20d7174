// K2d forward device code, shared by seed_fold.hip (scae_seed_fold_fwd_f32) and
// step_prologue.hip (the same jobs riding in the training step's prologue launch).
//
// ONE launch (the stage-2 jobs no longer wait for stage 1: a block of RO rows of Wo first
// forms u = Wo[rows] Wv -- RO x C, C^2 RO MACs, spread over C / RO workgroups -- and then
// [wvf | bvf] = u W2e + [0 | Wo[rows] bv + bo]; wv2e = Wv W2e + [0 | bv] is still produced,
// for the backward pass only).  Blocks of 256 threads:
//   [0, nb)            rows of Wk:  [wkf | bkf] = Wk W2e + [0 | bk]
//   [nb, 2 nb)         rows of Wv:  wv2e
//   [2 nb, 2 nb + no)  rows of Wo (RO per block): [wvf | bvf]
//   then nq blocks     QC columns of q = seeds Wq^T + bq each
#pragma once
#include "common.h"

namespace scae_fold {
constexpr int NT = 256;
constexpr int RO = 4;    // rows of Wo per block
constexpr int QC = 1;    // columns of q per block

// Row-block job: out[r][d] = sum_j W[row0 + r][j] * ext[j][d] for R = NT / (D+1) rows
// and all D+1 columns at once.  ext (C x (D+1)) and the R rows sit in LDS; thread
// (r, d) keeps its own sum -- no cross-lane reduction.
template <int D, class Store>
__device__ __forceinline__ void rows_block(const float *W, int C, const float *mat, int mat_ld,
                                           const float *col, int row0, float *lds, Store store) {
  constexpr int DP = D + 1, R = NT / DP;
  const int t = threadIdx.x, nrows = min(R, C - row0);
  float *ext = lds, *rows = lds + C * DP;
  for (int e = t; e < C * DP; e += NT) {
    const int jj = e / DP, d = e - jj * DP;
    ext[e] = d < D ? mat[(size_t)jj * mat_ld + d] : (col ? col[jj] : mat[(size_t)jj * mat_ld + D]);
  }
  {
    const float4 *src = reinterpret_cast<const float4 *>(W + (size_t)row0 * C);  // C % 64 == 0
    for (int e = t; e < nrows * C / 4; e += NT) reinterpret_cast<float4 *>(rows)[e] = src[e];
  }
  __syncthreads();
  if (t >= nrows * DP) return;
  const int r = t / DP, d = t - r * DP;
  const float *wr = rows + r * C;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int jj = 0; jj < C; jj += 4) {  // C % 64 == 0
    const float4 w = *reinterpret_cast<const float4 *>(wr + jj);
    a0 = fmaf(w.x, ext[(jj + 0) * DP + d], a0);
    a1 = fmaf(w.y, ext[(jj + 1) * DP + d], a1);
    a2 = fmaf(w.z, ext[(jj + 2) * DP + d], a2);
    a3 = fmaf(w.w, ext[(jj + 3) * DP + d], a3);
  }
  store(row0 + r, d, (a0 + a1) + (a2 + a3));
}

// RO rows of Wo: u = Wo[rows] Wv (thread = column j of u, Wv rows read coalesced, the
// Wo rows broadcast from LDS), then [wvf | bvf][row] = u W2e + [0 | Wo[row] bv + bo].
// LDS: ext C x (D+1) | rows RO x C (overwritten by u) | RO bias dots | 4 x RO x C partials.
// (KT: rows of Wv = independent 16-byte loads in flight per lane; 32 on its own, 16 where the
// block rides in a launch whose register budget is smaller)
template <int D, int KT = 32>
__device__ __forceinline__ void wo_block(const scae_seed_fold_desc &a, int row0, float *lds) {
  constexpr int DP = D + 1;
  const int t = threadIdx.x, C = a.C;
  float *ext = lds, *rows = lds + C * DP, *bdot = rows + RO * C;
  for (int e = t; e < C * DP; e += NT) {
    const int jj = e / DP, d = e - jj * DP;
    ext[e] = d < D ? a.w2[(size_t)jj * D + d] : a.b2[jj];
  }
  {
    const float4 *src = reinterpret_cast<const float4 *>(a.wo + (size_t)row0 * C);
    for (int e = t; e < RO * C / 4; e += NT) reinterpret_cast<float4 *>(rows)[e] = src[e];
  }
  __syncthreads();
  {  // Wo[row] . bv: 16 lanes per row
    const int r = t >> 4, l = t & 15;
    float s = 0.f;
    if (r < RO)
      for (int c = l; c < C; c += 16) s = fmaf(rows[r * C + c], a.bv[c], s);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (r < RO && l == 0) bdot[r] = s;
  }
  // u = Wo[rows] Wv: wave w walks the quarter [w C/4, (w+1) C/4) of the contraction for
  // ALL columns (lane l owns 4 consecutive columns), 32 rows of Wv = 32 independent 16-byte
  // coalesced loads in flight per lane -- the product is L2-latency bound, not flop bound;
  // the four partial sums meet in LDS.
  const int w = t >> 6, l = t & 63, cq = C / 4;
  float *part = bdot + RO;   // [4 waves][RO][C]
  for (int p = 0; p * 256 < C; ++p) {   // passes of 256 columns
    // lane l owns columns j .. j + 3 (one 16-byte load per row of Wv); lanes past the
    // last column re-read the last quad and are dropped at the store
    const int j = p * 256 + 4 * l, jc = j < C ? j : C - 4;
    const float *wvp = a.wv + (size_t)(w * cq) * C + jc;
    float acc[4][RO];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < RO; ++r) acc[m][r] = 0.f;
    for (int c = w * cq; c < (w + 1) * cq; c += KT, wvp += (size_t)KT * C) {
      float4 wv[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k)   // (cq is a multiple of 16: the second half may be past it)
        wv[k] = (k < 16 || c + k < (w + 1) * cq) ? *reinterpret_cast<const float4 *>(wvp + (size_t)k * C)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k4 = 0; k4 < KT / 4; ++k4) {
        if (k4 >= 4 && c + 4 * k4 >= (w + 1) * cq) break;   // uniform
#pragma unroll
        for (int r = 0; r < RO; ++r) {
          const float4 rv = *reinterpret_cast<const float4 *>(rows + r * C + c + 4 * k4);
          const float4 w0 = wv[4 * k4], w1 = wv[4 * k4 + 1], w2 = wv[4 * k4 + 2],
                       w3 = wv[4 * k4 + 3];
          acc[0][r] = fmaf(rv.w, w3.x, fmaf(rv.z, w2.x, fmaf(rv.y, w1.x, fmaf(rv.x, w0.x, acc[0][r]))));
          acc[1][r] = fmaf(rv.w, w3.y, fmaf(rv.z, w2.y, fmaf(rv.y, w1.y, fmaf(rv.x, w0.y, acc[1][r]))));
          acc[2][r] = fmaf(rv.w, w3.z, fmaf(rv.z, w2.z, fmaf(rv.y, w1.z, fmaf(rv.x, w0.z, acc[2][r]))));
          acc[3][r] = fmaf(rv.w, w3.w, fmaf(rv.z, w2.w, fmaf(rv.y, w1.w, fmaf(rv.x, w0.w, acc[3][r]))));
        }
      }
    }
    if (j < C) {
#pragma unroll
      for (int r = 0; r < RO; ++r)
        *reinterpret_cast<float4 *>(part + (w * RO + r) * C + j) =
            make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
    }
  }
  __syncthreads();  // partial sums complete; every read of the Wo rows is done: u takes their place
  for (int e = t; e < RO * C; e += NT) {
    const float u = (part[e] + part[RO * C + e]) + (part[2 * RO * C + e] + part[3 * RO * C + e]);
    rows[e] = u;
    if (a.wowv) a.wowv[(size_t)row0 * C + e] = u;   // (Wo Wv), kept for the backward pass
  }
  __syncthreads();
  if (t >= RO * DP) return;
  const int r = t / DP, d = t - r * DP;
  const float *ur = rows + r * C;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int jj = 0; jj < C; jj += 4) {
    const float4 w = *reinterpret_cast<const float4 *>(ur + jj);
    a0 = fmaf(w.x, ext[(jj + 0) * DP + d], a0);
    a1 = fmaf(w.y, ext[(jj + 1) * DP + d], a1);
    a2 = fmaf(w.z, ext[(jj + 2) * DP + d], a2);
    a3 = fmaf(w.w, ext[(jj + 3) * DP + d], a3);
  }
  const float s = (a0 + a1) + (a2 + a3);
  if (d < D)
    a.wvf[(size_t)(row0 + r) * D + d] = s;
  else
    a.bvf[row0 + r] = s + bdot[r] + a.bo[row0 + r];
}

struct Plan {
  int nb, no, nq;   // row blocks of Wk (and of Wv), of Wo, column blocks of q
  __host__ __device__ int blocks() const { return 2 * nb + no + nq; }
};
inline Plan plan(int C, int D) {
  const int R = NT / (D + 1);
  return Plan{(C + R - 1) / R, C / RO, (C + QC - 1) / QC};
}
inline size_t lds_bytes(int C, int D) {
  const size_t rows = (size_t)(NT / (D + 1)) * C, wo = (size_t)5 * RO * C + RO;
  return ((size_t)C * (D + 1) + (rows > wo ? rows : wo)) * sizeof(float);
}

// block `blk` (of plan.blocks()) of the forward folding; 256 threads
template <int D, int KT = 32>
__device__ __forceinline__ void forward_block(const scae_seed_fold_desc &a, const Plan &pl,
                                              int blk, float *lds) {
  constexpr int DP = D + 1, R = NT / DP;
  const int t = threadIdx.x, C = a.C;
  if (blk < pl.nb) {
    rows_block<D>(a.wk, C, a.w2, D, a.b2, blk * R, lds, [&](int c, int d, float s) {
      if (d < D)
        a.wkf[c * D + d] = s;
      else
        a.bkf[c] = s + a.bk[c];
    });
    return;
  }
  if (blk < 2 * pl.nb) {
    rows_block<D>(a.wv, C, a.w2, D, a.b2, (blk - pl.nb) * R, lds, [&](int c, int d, float s) {
      a.wv2e[c * DP + d] = d < D ? s : s + a.bv[c];
    });
    return;
  }
  if (blk < 2 * pl.nb + pl.no) {
    wo_block<D, KT>(a, (blk - 2 * pl.nb) * RO, lds);
    return;
  }
  // q[:, c] = seeds Wq[c, :]^T + bq[c]: 8 lanes per seed o, each an 8-strided
  // slice of j, so that all loads of the dot product are in flight at once
  const int c0 = (blk - 2 * pl.nb - pl.no) * QC, sub = t & 7;
  for (int c = c0; c < min(C, c0 + QC); ++c) {
    const float *wq = a.wq + (size_t)c * C;
    for (int o = t >> 3; o < a.O; o += NT / 8) {
      const float *sd = a.seeds + (size_t)o * C;
      float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
      for (int jj = sub; jj < C; jj += 32) {  // C % 64 == 0
        p0 = fmaf(wq[jj], sd[jj], p0);
        p1 = fmaf(wq[jj + 8], sd[jj + 8], p1);
        p2 = fmaf(wq[jj + 16], sd[jj + 16], p2);
        p3 = fmaf(wq[jj + 24], sd[jj + 24], p3);
      }
      float rr = (p0 + p1) + (p2 + p3);
      rr += __shfl_xor(rr, 1, 64);
      rr += __shfl_xor(rr, 2, 64);
      rr += __shfl_xor(rr, 4, 64);
      if (sub == 0) a.q[(size_t)o * C + c] = rr + a.bq[c];
    }
  }
}

// runtime D -> template
template <int KT = 32>
__device__ __forceinline__ void forward_block_any(const scae_seed_fold_desc &a, const Plan &pl,
                                                  int blk, float *lds) {
  if (a.D == 16)
    forward_block<16, KT>(a, pl, blk, lds);
  else if (a.D == 8)
    forward_block<8, KT>(a, pl, blk, lds);
  else
    forward_block<32, KT>(a, pl, blk, lds);
}
}  // namespace scae_fold
