// K2c on the matrix cores -- the set transformer's output attention
//   MultiHeadQKVAttention(seeds, z, z, presence), z = fc2(h)   (set_transformer.py:218-223)
// with fc2 and the k / v / o projections folded into two (C x 16) maps (seed_attention.hip,
// seed_fold.hip), taken two steps further.  With h the (N x 16) trunk output of a set:
//     K' = h wk^T + bk,  V' = h wv^T + bv                       (N x C, C = 256)
//     S  = q K'^T        = (q wk) h^T + (q bk) 1^T              -- the second term is constant
//                                                                  along the keys: the softmax
//                                                                  over the keys drops it
//     out = P V'         = (P h) wv^T + bv                      -- rows of P sum to one
// so the per-set work contracts over 16, not 256: qk = q wk (O x 16, once per workgroup),
// S = qk h^T, T = P h (O x 16), out = T wv^T + bv.  Backward likewise: dT = gout wv,
// dwv += gout^T T, dP = dT h^T, d(qk) = dS h, dh = P^T dT + dS^T qk; the gradient reaches q
// and wk only through d(qk) summed over the batch (dq = d(qk) wk^T, dwk = q^T d(qk): two
// parameter-sized products done once, in the reduce kernel), and bk gets the exact zero the
// softmax's shift invariance implies (the reference's value is round-off noise around 0).
//
// One workgroup of 4 waves per set.  A wave per 16-row query tile (softmax, P h), then per
// 16-row key tile (dh); all four share the C-wide products.  N, O <= 64.  v_mfma_f32_16x16x4_f32 throughout (exact fp32).
#include <algorithm>

#include "wave_mfma.h"

using namespace scae_wave;

namespace {
constexpr int NTH = 256;

struct SwArgs {
  const float *h;         // (B,N,16)
  const float *q;         // (O,C)
  const float *wk;        // (C,16)
  const float *wv, *bv;   // (C,16), (C)
  const float *presence;  // (B,N) nullable
  float *out;             // fwd (B,O,C)
  const float *gout;      // bwd (B,O,C)
  float *gh;              // bwd (B,N,16)
  float *partial;         // bwd (rows, O*16 + C*16 + C): [d(qk) | dwv | dbv]
  int B, N, O, C;
  float inv_sqrt_c;
  int bf16;               // the attention products (logits, P h, and their backward) in bf16
};

// Tiles.  NT = 16-row tiles per side (2: N, O <= 32; 4: N, O <= 64).  R tiles [16 NT][RS]
// (rows = queries), transposed [16][TSN] and N x N tiles [16 NT][TSN], TSN = 16 NT + 8.
enum { T_QK = 0, T_TS, T_DT, T_QKT, T_TT, T_DTT, T_SMALL };   // R tiles / transposed
enum { T_PS = 0, T_PT, T_DSR, T_DST, T_LARGE };                // N x N tiles
template <int NT>
struct Geo {
  static constexpr int ROWS = 16 * NT, TSN = 16 * NT + 8;
  static constexpr int SMALL = ROWS * RS;   // >= 16 * TSN
  static constexpr int LARGE = ROWS * TSN;
  static constexpr int SCR = 4 * NT * 256;  // the four waves' partial (ROWS x 16) products
  static constexpr int LDS_FLOATS = T_SMALL * SMALL + T_LARGE * LARGE + SCR;
};
template <int NT>
struct Tl {
  float *base;
  __device__ __forceinline__ float *small(int i) const { return base + i * Geo<NT>::SMALL; }
  __device__ __forceinline__ float *large(int i) const {
    return base + T_SMALL * Geo<NT>::SMALL + i * Geo<NT>::LARGE;
  }
  __device__ __forceinline__ float *scr() const {
    return base + T_SMALL * Geo<NT>::SMALL + T_LARGE * Geo<NT>::LARGE;
  }
};

struct Lane {
  int lane, r, q, v;   // v: wave index in the workgroup
};
__device__ __forceinline__ Lane make_lane() {
  Lane l;
  l.lane = threadIdx.x & 63, l.r = l.lane & 15, l.q = l.lane >> 4;
  l.v = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  return l;
}
__device__ __forceinline__ void wr_rows(float *tile, int t, const Lane &l, const f32x4 &o) {
#pragma unroll
  for (int e = 0; e < 4; ++e) tile[(16 * t + 4 * l.q + e) * RS + l.r] = o[e];
}
template <int NT>
__device__ __forceinline__ void wr_cols(float *tile, int t, const Lane &l, const f32x4 &o) {
  *reinterpret_cast<float4 *>(tile + l.r * Geo<NT>::TSN + 16 * t + 4 * l.q) =
      make_float4(o[0], o[1], o[2], o[3]);
}
template <int NT>
__device__ __forceinline__ void wr_nn(float *tile, int t, const Lane &l, const f32x4 (&o)[NT]) {
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      tile[(16 * t + 4 * l.q + e) * Geo<NT>::TSN + 16 * u + l.r] = o[u][e];
}
template <int NT>
__device__ __forceinline__ void wr_nn_t(float *tile, int t, const Lane &l, const f32x4 (&o)[NT]) {
#pragma unroll
  for (int u = 0; u < NT; ++u)
    *reinterpret_cast<float4 *>(tile + (16 * u + l.r) * Geo<NT>::TSN + 16 * t + 4 * l.q) =
        make_float4(o[u][0], o[u][1], o[u][2], o[u][3]);
}
__device__ __forceinline__ float4 rd16(const float *tile, int t, const Lane &l) {
  return ld4(tile + (16 * t + l.r) * RS + 4 * l.q);
}
template <int NT>
__device__ __forceinline__ FK<NT> rdk(const float *tile, int t, const Lane &l) {
  const float *p = tile + (16 * t + l.r) * Geo<NT>::TSN + 4 * NT * l.q;
  FK<NT> f;
#pragma unroll
  for (int j = 0; j < NT; ++j) f.v[j] = ld4(p + 4 * j);
  return f;
}

// qk = q wk (O x 16) into the R tile QK (and transposed into QKT when given): the C-long
// contraction is split over the four waves and met in LDS.  Ends with a barrier.
template <int NT>
__device__ __forceinline__ void fold_qk(const SwArgs &a, const Lane &l, const Tl<NT> &tl,
                                        bool transposed) {
  const int C = a.C, CW = C / 4, run = CW / 4;   // per wave, per lane group
  float *scr = tl.scr();
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = splat(0.f);
  const int k0 = CW * l.v + run * l.q;
  for (int j = 0; j < run; j += 4) {
    const float *wp = a.wk + (size_t)(k0 + j) * D + l.r;
    const float4 b = make_float4(wp[0], wp[D], wp[2 * D], wp[3 * D]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int o = 16 * t + l.r;
      const float4 av = o < a.O ? ld4(a.q + (size_t)o * C + k0 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
      acc[t] = mma16(acc[t], av, b);
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      scr[((l.v * NT + t) * 16 + 4 * l.q + e) * D + l.r] = acc[t][e];
  __syncthreads();
  for (int i = threadIdx.x; i < Geo<NT>::ROWS * D; i += NTH) {
    const float s = (scr[i] + scr[NT * 256 + i]) + (scr[2 * NT * 256 + i] + scr[3 * NT * 256 + i]);
    const int o = i / D, c = i % D;
    tl.small(T_QK)[o * RS + c] = s;
    if (transposed) tl.small(T_QKT)[c * Geo<NT>::TSN + o] = s;
  }
  __syncthreads();
}

// keys of a set as B operands: hB[u] for S / dP (K = 16: row n = 16 u + r, features 4 q ..)
// and hT for P h / dS h (K = 16 NT over n = 4 NT q .., feature r)
template <int NT>
struct Keys {
  float4 hB[NT];
  FK<NT> hT;
  float kmask[NT];
};
template <int NT>
__device__ __forceinline__ Keys<NT> load_keys(const SwArgs &a, int b, const Lane &l) {
  Keys<NT> k;
  const float *hb = a.h + (size_t)b * a.N * D;
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int n = 16 * u + l.r;
    k.hB[u] = n < a.N ? ld4(hb + n * D + 4 * l.q) : make_float4(0.f, 0.f, 0.f, 0.f);
    k.kmask[u] = a.presence && n < a.N ? (1.f - a.presence[(size_t)b * a.N + n]) * 1e32f : 0.f;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    float t[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int n = 4 * NT * l.q + 4 * j + s;
      t[s] = n < a.N ? hb[n * D + l.r] : 0.f;
    }
    k.hT.v[j] = make_float4(t[0], t[1], t[2], t[3]);
  }
  return k;
}

// attention probabilities of the own query tile t (O layout, key tiles u)
template <int NT, bool BF>
__device__ __forceinline__ void probabilities(const SwArgs &a, const Lane &l, const Tl<NT> &tl,
                                              int t, const Keys<NT> &k, f32x4 (&p)[NT]) {
  const float4 qa = rd16(tl.small(T_QK), t, l);
#pragma unroll
  for (int u = 0; u < NT; ++u) p[u] = mma16p<BF>(splat(0.f), qa, k.hB[u]);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v[NT], mx = -INFINITY, sum = 0.f;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      v[u] = 16 * u + l.r < a.N ? (p[u][e] - k.kmask[u]) * a.inv_sqrt_c : -INFINITY;
      mx = fmaxf(mx, v[u]);
    }
    mx = rmax(mx);
#pragma unroll
    for (int u = 0; u < NT; ++u) v[u] = __expf(v[u] - mx), sum += v[u];
    const float inv = __builtin_amdgcn_rcpf(rsum(sum));   // (sum >= 1: the row maximum's term)
#pragma unroll
    for (int u = 0; u < NT; ++u) p[u][e] = v[u] * inv;
  }
}

template <int NT, bool BF>
__global__ __launch_bounds__(NTH) void saw_fwd_kernel(SwArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Tl<NT> tl{smem};
  const Lane l = make_lane();
  const int C = a.C, O = a.O;
  static_assert(Geo<NT>::LDS_FLOATS % 4 == 0, "16-byte zero fill");
  for (int i = 4 * threadIdx.x; i < Geo<NT>::LDS_FLOATS; i += 4 * NTH)
    *reinterpret_cast<float4 *>(smem + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  fold_qk<NT>(a, l, tl, false);
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    if (l.v < NT && 16 * l.v < O) {   // the query tiles
      const int t = l.v;
      const Keys<NT> k = load_keys<NT>(a, b, l);
      f32x4 p[NT];
      probabilities<NT, BF>(a, l, tl, t, k, p);
      wr_nn<NT>(tl.large(T_PS), t, l, p);
      lds_fence();
      const f32x4 T = mmakp<NT, BF>(splat(0.f), rdk<NT>(tl.large(T_PS), t, l), k.hT);
      wr_rows(tl.small(T_TS), t, l, T);
      lds_fence();
    }
    __syncthreads();
    // out = T wv^T + bv: column tiles of 16 shared out over the waves
    float *ob = a.out + (size_t)b * O * C;
    for (int ct = l.v; ct * 16 < C; ct += 4) {
      const float4 wb = ld4(a.wv + (size_t)(16 * ct + l.r) * D + 4 * l.q);
      const float bias = a.bv[16 * ct + l.r];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (16 * t >= O) break;
        const f32x4 o = mma16(splat(bias), rd16(tl.small(T_TS), t, l), wb);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * t + 4 * l.q + e;
          if (row < O) ob[(size_t)row * C + 16 * ct + l.r] = o[e];
        }
      }
    }
    __syncthreads();   // T is rewritten by the next set
  }
}

// workgroup `blk` of `nblk` (seed_bwd_gemm.hip runs these as the head of a shared launch)
template <int NT, bool BF>
__device__ __forceinline__ void saw_bwd_body(const SwArgs &a, float *smem, int blk, int nblk) {
  const Tl<NT> tl{smem};
  const Lane l = make_lane();
  const int C = a.C, O = a.O, N = a.N, CW = C / 4, run = CW / 4;
  constexpr int TSN = Geo<NT>::TSN;
  const int npar = O * D + C * D + C;
  float *part = a.partial + (size_t)blk * npar;
  float *scr = tl.scr();
  static_assert(Geo<NT>::LDS_FLOATS % 4 == 0, "16-byte zero fill");
  for (int i = 4 * threadIdx.x; i < Geo<NT>::LDS_FLOATS; i += 4 * NTH)
    *reinterpret_cast<float4 *>(smem + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  fold_qk<NT>(a, l, tl, true);
  bool first = true;
  auto put = [&](int idx, float v) { part[idx] = first ? v : part[idx] + v; };
  const int k0 = CW * l.v + run * l.q;   // this lane group's run of C in the dT product
  for (int b = blk; b < a.B; b += nblk) {
    const float *gb = a.gout + (size_t)b * O * C;
    // dT = gout wv: partial over this wave's quarter of C
    {
      f32x4 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = splat(0.f);
      for (int j = 0; j < run; j += 4) {
        const float *wp = a.wv + (size_t)(k0 + j) * D + l.r;
        const float4 wb = make_float4(wp[0], wp[D], wp[2 * D], wp[3 * D]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int o = 16 * t + l.r;
          const float4 av =
              o < O ? ld4(gb + (size_t)o * C + k0 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
          acc[t] = mma16(acc[t], av, wb);
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          scr[((l.v * NT + t) * 16 + 4 * l.q + e) * D + l.r] = acc[t][e];
    }
    // query waves: probabilities and T = P h of their tile
    Keys<NT> k;
    f32x4 p[NT];
    const bool qwave = l.v < NT && 16 * l.v < O;
    if (qwave) {
      const int t = l.v;
      k = load_keys<NT>(a, b, l);
      probabilities<NT, BF>(a, l, tl, t, k, p);
      wr_nn<NT>(tl.large(T_PS), t, l, p);
      wr_nn_t<NT>(tl.large(T_PT), t, l, p);
      lds_fence();
      const f32x4 T = mmakp<NT, BF>(splat(0.f), rdk<NT>(tl.large(T_PS), t, l), k.hT);
      wr_cols<NT>(tl.small(T_TT), t, l, T);
    }
    lds_fence();
    __syncthreads();
    for (int i = threadIdx.x; i < Geo<NT>::ROWS * D; i += NTH) {   // dT: the four partials meet
      const float s =
          (scr[i] + scr[NT * 256 + i]) + (scr[2 * NT * 256 + i] + scr[3 * NT * 256 + i]);
      const int o = i / D, c = i % D;
      tl.small(T_DT)[o * RS + c] = s;
      tl.small(T_DTT)[c * TSN + o] = s;
    }
    __syncthreads();
    // dwv[c][i] = sum_o gout[o][c] T[o][i], dbv[c] = sum_o gout[o][c]: column tiles of 16
    {
      const FK<NT> tb = rdk<NT>(tl.small(T_TT), 0, l);   // B[k = o][col = i]: row i of T^T
      for (int ct = l.v; ct * 16 < C; ct += 4) {
        FK<NT> ga;
        float cs = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          float g[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int o = 4 * NT * l.q + 4 * j + s;
            g[s] = o < O ? gb[(size_t)o * C + 16 * ct + l.r] : 0.f;
          }
          ga.v[j] = make_float4(g[0], g[1], g[2], g[3]);
          cs += (g[0] + g[1]) + (g[2] + g[3]);
        }
        const f32x4 dw = mmak<NT>(splat(0.f), ga, tb);
#pragma unroll
        for (int e = 0; e < 4; ++e) put(O * D + (16 * ct + 4 * l.q + e) * D + l.r, dw[e]);
        cs += __shfl_xor(cs, 16, 64);
        cs += __shfl_xor(cs, 32, 64);
        if (l.q == 0) put(O * D + C * D + 16 * ct + l.r, cs);
      }
    }
    // softmax backward, d(qk) = dS h
    if (qwave) {
      const int t = l.v;
      const float4 dta = rd16(tl.small(T_DT), t, l);
      f32x4 ds[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) ds[u] = mma16p<BF>(splat(0.f), dta, k.hB[u]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float dot = 0.f;
#pragma unroll
        for (int u = 0; u < NT; ++u) dot = fmaf(p[u][e], ds[u][e], dot);
        dot = rsum(dot);
#pragma unroll
        for (int u = 0; u < NT; ++u) ds[u][e] = p[u][e] * (ds[u][e] - dot) * a.inv_sqrt_c;
      }
      wr_nn<NT>(tl.large(T_DSR), t, l, ds);
      wr_nn_t<NT>(tl.large(T_DST), t, l, ds);
      lds_fence();
      const f32x4 dqk = mmakp<NT, BF>(splat(0.f), rdk<NT>(tl.large(T_DSR), t, l), k.hT);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int o = 16 * t + 4 * l.q + e;
        if (o < O) put(o * D + l.r, dqk[e]);
      }
    }
    lds_fence();
    __syncthreads();
    // dh[n][i] = sum_o P[o][n] dT[o][i] + dS[o][n] qk[o][i]: one key tile per wave
    if (l.v < NT && 16 * l.v < N) {
      const int u = l.v;
      f32x4 gh = mmakp<NT, BF>(splat(0.f), rdk<NT>(tl.large(T_PT), u, l), rdk<NT>(tl.small(T_DTT), 0, l));
      gh = mmakp<NT, BF>(gh, rdk<NT>(tl.large(T_DST), u, l), rdk<NT>(tl.small(T_QKT), 0, l));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = 16 * u + 4 * l.q + e;
        if (n < N) a.gh[((size_t)b * N + n) * D + l.r] = gh[e];
      }
    }
    first = false;
    __syncthreads();
  }
}
template <int NT, bool BF>
__global__ __launch_bounds__(NTH) void saw_bwd_kernel(SwArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  saw_bwd_body<NT, BF>(a, smem, blockIdx.x, gridDim.x);
}

// partial (rows, O*16 + C*16 + C) -> gq (O,C), gwk (C,16), gbk (C) = 0, gwv (C,16), gbv (C).
// The canonical form is 1024 threads = 64 columns x 16 row parts: a column sum is 8-16
// independent loads per thread and one LDS meeting.  Blocks [0, nsum): 64 columns of
// [dwv | dbv] each; the others first sum the O*16 columns of d(qk) (every one of them: 200 KB
// from L2) and then produce RT entries of [gq | gwk].  A block of RT < 1024 threads walks
// 1024 / RT row parts per thread (conv_mfma.hip runs these workgroups as 256-thread riders
// of a convolution launch): the same partial sums meet in the same order, so the result does
// not depend on the block shape.  lds: 16 * 65 + 64 * 16 floats.
constexpr int RTH = 1024;
struct ReduceArgs {
  const float *partial;
  int rows;
  const float *q, *wk;
  float *gq, *gwk, *gbk, *gwv, *gbv;
  int O, C, nsum;
};
inline int reduce_blocks(const ReduceArgs &r, int RT) {
  return r.nsum + (r.O * r.C + r.C * D + RT - 1) / RT;
}
template <int RT>
__device__ __forceinline__ void saw_reduce_body(const ReduceArgs &a, float *lds, int blk) {
  constexpr int VR = RTH / RT;   // row parts per thread
  float (*red)[65] = reinterpret_cast<float (*)[65]>(lds);
  float *dqk = lds + 16 * 65;
  const float *__restrict__ partial = a.partial;
  const int rows = a.rows, O = a.O, C = a.C, nsum = a.nsum;
  const int npar = O * D + C * D + C;
  const int cx = threadIdx.x & 63, ry0 = (threadIdx.x >> 6) * VR;
  // sum of column `col` over the rows ry, ry + 16, ..; valid in the threads with ry0 == 0
  auto colsum = [&](int col, bool ok) {
#pragma unroll
    for (int vr = 0; vr < VR; ++vr) {
      const int ry = ry0 + vr;
      float acc = 0.f;
      if (ok) {
        float v[8];
        int r = ry;
        for (; r + 7 * 16 < rows; r += 8 * 16) {
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(r + 16 * u) * npar + col];
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; r < rows; r += 16) acc += partial[(size_t)r * npar + col];
      }
      red[ry][cx] = acc;
    }
    __syncthreads();
    float tot = 0.f;
    if (ry0 == 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k) tot += red[k][cx];
    }
    __syncthreads();
    return tot;
  };
  if (blk < nsum) {   // gwv, gbv (and the zero bk gradient)
    const int i = blk * 64 + cx;
    const float tot = colsum(O * D + i, i < C * D + C);
    if (ry0 == 0) {
      if (i < C * D)
        a.gwv[i] = tot;
      else if (i < C * D + C)
        a.gbv[i - C * D] = tot;
      if (i < C) a.gbk[i] = 0.f;
    }
    return;
  }
  for (int c0 = 0; c0 < O * D; c0 += 64) {
    const float tot = colsum(c0 + cx, c0 + cx < O * D);
    if (ry0 == 0 && c0 + cx < O * D) dqk[c0 + cx] = tot;
  }
  __syncthreads();
  const int i = (blk - nsum) * RT + threadIdx.x;
  if (i < O * C) {   // gq[o][c] = sum_i d(qk)[o][i] wk[c][i]
    const int o = i / C, c = i - o * C;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) acc = fmaf(dqk[o * D + j], a.wk[(size_t)c * D + j], acc);
    a.gq[i] = acc;
  } else if (i < O * C + C * D) {   // gwk[c][i] = sum_o q[o][c] d(qk)[o][i]
    const int e = i - O * C, c = e / D, j = e - c * D;
    float acc = 0.f;
    for (int o = 0; o < O; ++o) acc = fmaf(a.q[(size_t)o * C + c], dqk[o * D + j], acc);
    a.gwk[e] = acc;
  }
}
inline int reduce_args(ReduceArgs &r, const float *partial, int rows, const float *q,
                       const float *wk, float *gq, float *gwk, float *gbk, float *gwv, float *gbv,
                       int O, int C) {
  SCAE_REQUIRE(partial && q && wk && gq && gwk && gbk && gwv && gbv && rows > 0 && O > 0 &&
               O <= 64 && C > 0);
  r = ReduceArgs{partial, rows, q, wk, gq, gwk, gbk, gwv, gbv, O, C, (C * D + C + 63) / 64};
  return SCAE_OK;
}
#ifndef SCAE_DEVICE_ONLY   // (seed_bwd_gemm.hip, conv_mfma.hip include this file for its device code)
__global__ __launch_bounds__(RTH) void saw_reduce_kernel(ReduceArgs a) {
  __shared__ float lds[16 * 65 + 64 * D];
  saw_reduce_body<RTH>(a, lds, blockIdx.x);
}
#endif
int check(const SwArgs &a) {
  if (a.B <= 0 || a.N <= 0 || a.O <= 0 || a.C <= 0) return SCAE_ERR_BAD_ARG;
  if (a.N > 64 || a.O > 64 || (a.C & 63)) return SCAE_ERR_UNSUPPORTED;
  return SCAE_OK;
}

#ifndef SCAE_DEVICE_ONLY
template <int NT, bool BF>
int launch_nt(const SwArgs &a, bool bwd, hipStream_t st) {
  const size_t lds = Geo<NT>::LDS_FLOATS * sizeof(float);
  const void *fn = bwd ? reinterpret_cast<const void *>(saw_bwd_kernel<NT, BF>)
                       : reinterpret_cast<const void *>(saw_fwd_kernel<NT, BF>);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const dim3 grid(a.B < 512 ? a.B : 512);
  if (bwd)
    scae::launch((saw_bwd_kernel<NT, BF>), grid, dim3(NTH), lds, st, a);
  else
    scae::launch((saw_fwd_kernel<NT, BF>), grid, dim3(NTH), lds, st, a);
  return scae_launch_status();
}
int launch(const SwArgs &a, bool bwd, hipStream_t st) {
  if (a.N <= 32 && a.O <= 32)
    return a.bf16 ? launch_nt<2, true>(a, bwd, st) : launch_nt<2, false>(a, bwd, st);
  return a.bf16 ? launch_nt<4, true>(a, bwd, st) : launch_nt<4, false>(a, bwd, st);
}
#endif
}  // namespace

#ifndef SCAE_DEVICE_ONLY

extern "C" int scae_seed_attention_mfma_supported(int N, int O, int D_, int C) {
  return D_ == D && N > 0 && N <= 64 && O > 0 && O <= 64 && C > 0 && (C & 63) == 0 ? 1 : 0;
}
extern "C" int scae_seed_attention_mfma_rows(int B) { return B <= 0 ? 0 : (B < 512 ? B : 512); }

#define SCAE_SAW_ENTRY(SUFFIX, BF)                                                             \
  extern "C" int scae_seed_attention_mfma_fwd_##SUFFIX(                                        \
      const float *h, const float *q, const float *wk, const float *wv, const float *bv,       \
      const float *presence, float *out, int B, int N, int O, int C, void *stream) {           \
    SCAE_REQUIRE(h && q && wk && wv && bv && out);                                             \
    SwArgs a{h,       q,       wk, wv, bv, presence, out, nullptr,                             \
             nullptr, nullptr, B,  N,  O,  C,        1.f / sqrtf((float)C), BF};               \
    int rc = check(a);                                                                         \
    if (rc) return rc;                                                                         \
    return launch(a, false, (hipStream_t)stream);                                              \
  }                                                                                            \
  extern "C" int scae_seed_attention_mfma_bwd_##SUFFIX(                                        \
      const float *h, const float *q, const float *wk, const float *wv, const float *presence, \
      const float *gout, float *gh, float *partial, int B, int N, int O, int C,                \
      void *stream) {                                                                          \
    SCAE_REQUIRE(h && q && wk && wv && gout && gh && partial);                                 \
    SwArgs a{h,  q,       wk, wv, nullptr, presence, nullptr, gout,                            \
             gh, partial, B,  N,  O,       C,        1.f / sqrtf((float)C), BF};               \
    int rc = check(a);                                                                         \
    if (rc) return rc;                                                                         \
    return launch(a, true, (hipStream_t)stream);                                               \
  }
SCAE_SAW_ENTRY(f32, 0)
// configs[2]: logits, P h and the three products of their backward with bf16 operands
SCAE_SAW_ENTRY(bf16, 1)
#undef SCAE_SAW_ENTRY

extern "C" int scae_seed_attention_mfma_reduce_f32(const float *partial, int rows, const float *q,
                                                   const float *wk, float *gq, float *gwk,
                                                   float *gbk, float *gwv, float *gbv, int O,
                                                   int C, void *stream) {
  ReduceArgs r;
  int rc = reduce_args(r, partial, rows, q, wk, gq, gwk, gbk, gwv, gbv, O, C);
  if (rc) return rc;
  scae::launch(saw_reduce_kernel, dim3(reduce_blocks(r, RTH)), dim3(RTH), 0,
                     (hipStream_t)stream, r);
  return scae_launch_status();
}
#endif  // SCAE_DEVICE_ONLY
