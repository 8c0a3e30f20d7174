// K2b -- fused set-transformer encoder trunk for gfx950:
//     fc1 -> L x SAB (single head, optional LayerNorm) -> fc2
// i.e. SetTransformer.forward up to (not including) the output attention,
// set_transformer.py:212-219 with SAB = MAB(x, x, presence) (:107-142).
//
// Why one kernel: at the reference's sizes (24 capsules x 16 hidden dims) a
// SAB is ~15 launch-bound ATen ops forward and ~40 backward; three of them
// plus fc1/fc2 were ~190 of the ~500 launches of a training step while doing
// < 0.1 GFLOP.  One workgroup (4 wavefronts) owns one set.  Every (N x D)
// activation is an LDS tile and each stage is ELEMENT-parallel: lane (n, i)
// produces one output element from a 16-byte-vectorised dot product of an
// activation row and a weight row, so dependent chains are D long instead of
// D*D (a first, row-per-lane version of this kernel was latency bound:
// 132 us forward / 335 us backward).  LayerNorm and the softmax use D-lane /
// 16-lane xor-shuffle reductions.  All trunk weights are staged in LDS once
// per workgroup, rows padded so that 16 lanes reading 16 different rows hit
// 16 different 16-byte bank groups.  The D = 16 contractions are far below an
// MFMA tile, so this trunk uses VALU FMAs; the matrix cores serve the wide
// (d = 256) output attention (set_attention.hip).  The backward kernel
// recomputes each block from its saved input, needs no atomics (every
// weight-gradient entry has one owning lane; per-workgroup partial sums are
// reduced by the caller) and is bit-reproducible.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "set_encoder_args.h"

namespace {

constexpr int NT = 512;  // 8 waves: every (N x D) stage of a 24 x 16 set is one pass
using scae_st::kLnEps;
using scae_st::MAXSEG;
using scae_st::NMAX;
using scae_st::Seg;
using scae_st::StArgs;

// packed GLOBAL parameter layout (floats), D = hidden width:
//   W1 [D][Din], b1 [D]
//   per layer: Wq,bq, Wk,bk, Wv,bv, Wo,bo, (ln0w, ln0b), Wf,bf, (ln1w, ln1b)
//   W2 [Dout][D], b2 [Dout]
// matrices in a layer are numbered q=0 k=1 v=2 o=3 f=4
template <int D>
struct Layout {
  int Din, Dout, L, ln;
  static constexpr int TS = D + 4;  // padded row stride of LDS tiles / matrices
  __host__ __device__ int layer_size() const { return 5 * (D * D + D) + (ln ? 4 * D : 0); }
  __host__ __device__ int off_b1() const { return D * Din; }
  __host__ __device__ int off_layer(int l) const { return D * Din + D + l * layer_size(); }
  __host__ __device__ int g_w(int m) const {  // matrix m inside a layer (global layout)
    return m < 4 ? m * (D * D + D) : 4 * (D * D + D) + (ln ? 2 * D : 0);
  }
  __host__ __device__ int g_b(int m) const { return g_w(m) + D * D; }
  __host__ __device__ int g_ln0() const { return 4 * (D * D + D); }
  __host__ __device__ int g_ln1() const { return g_b(4) + D; }
  __host__ __device__ int off_w2() const { return off_layer(L); }
  __host__ __device__ int off_b2() const { return off_w2() + Dout * D; }
  __host__ __device__ int total() const { return off_b2() + Dout; }
  // LDS copies: W1 rows padded to DinS, layer matrices padded to TS
  __host__ __device__ int DinS() const {
    int s = (Din + 3) & ~3;
    if (((s >> 2) & 1) == 0) s += 4;  // odd number of 16-byte units per row
    return s;
  }
  __host__ __device__ int l_w(int m) const { return m * D * TS; }
  __host__ __device__ int l_b(int m) const { return 5 * D * TS + m * D; }
  __host__ __device__ int l_ln0() const { return 5 * D * TS + 5 * D; }
  __host__ __device__ int l_ln1() const { return l_ln0() + 2 * D; }
  __host__ __device__ int lds_layer_size() const { return 5 * D * TS + 9 * D; }
};

// ---- lane-group reductions (groups of G consecutive lanes, G | 64) ---------
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// dot product of two 16-byte aligned rows of n4 float4
__device__ __forceinline__ float dot4(const float *a, const float *b, int n4) {
  const float4 *pa = reinterpret_cast<const float4 *>(a);
  const float4 *pb = reinterpret_cast<const float4 *>(b);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // 4 independent chains
#pragma unroll 4
  for (int j = 0; j < n4; ++j) {
    const float4 x = pa[j], y = pb[j];
    a0 = fmaf(x.x, y.x, a0);
    a1 = fmaf(x.y, y.y, a1);
    a2 = fmaf(x.z, y.z, a2);
    a3 = fmaf(x.w, y.w, a3);
  }
  return (a0 + a1) + (a2 + a3);
}
template <int D>
__device__ __forceinline__ float dotD(const float *a, const float *b) {
  const float4 *pa = reinterpret_cast<const float4 *>(a);
  const float4 *pb = reinterpret_cast<const float4 *>(b);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // 4 independent chains
#pragma unroll
  for (int j = 0; j < D / 4; ++j) {
    const float4 x = pa[j], y = pb[j];
    a0 = fmaf(x.x, y.x, a0);
    a1 = fmaf(x.y, y.y, a1);
    a2 = fmaf(x.z, y.z, a2);
    a3 = fmaf(x.w, y.w, a3);
  }
  return (a0 + a1) + (a2 + a3);
}
// sum_k g[k] * W[k][i]  (column i of a TS-strided matrix; lanes i consecutive)
template <int D>
__device__ __forceinline__ float dot_col(const float *grow, const float *W, int i) {
  constexpr int TS = D + 4;
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int k = 0; k < D; k += 2) {
    a0 = fmaf(grow[k], W[k * TS + i], a0);
    a1 = fmaf(grow[k + 1], W[(k + 1) * TS + i], a1);
  }
  return a0 + a1;
}

template <int D>
struct Tiles {  // LDS carve.  The (N x TS) activation tiles are addressed as
                // base + index * tile so that only one pointer stays live
                // (25 separate tile pointers overflowed the SGPR file and the
                // compiler spilled the whole struct to scratch memory).
  float *w1, *b1, *lw, *x, *tiles, *S, *GS, *rstd0, *rstd1, *pg, *scr;
  int tile;
#define SCAE_TILE(name, k) \
  __device__ __forceinline__ float *name() const { return tiles + (k) * tile; }
  SCAE_TILE(H, 0) SCAE_TILE(Q, 1) SCAE_TILE(K, 2) SCAE_TILE(V, 3) SCAE_TILE(A, 4)
  SCAE_TILE(H1, 5) SCAE_TILE(T, 6) SCAE_TILE(XH0, 7) SCAE_TILE(XH1, 8) SCAE_TILE(G, 9)
  SCAE_TILE(GN, 10) SCAE_TILE(GH1, 11) SCAE_TILE(GO, 12) SCAE_TILE(GA, 13)
  SCAE_TILE(GQ, 14) SCAE_TILE(GK, 15) SCAE_TILE(GV, 16) SCAE_TILE(HIN, 17)
#undef SCAE_TILE
};

template <int D>
__host__ __device__ size_t carve(const Layout<D> &lay, int N, bool bwd, float *base,
                                 Tiles<D> *out) {
  constexpr int TS = D + 4;
  size_t o = 0;
  auto take = [&](size_t n) {
    float *p = base ? base + o : nullptr;
    o += (n + 3) & ~(size_t)3;
    return p;
  };
  Tiles<D> r{};
  r.w1 = take((size_t)D * lay.DinS());
  r.b1 = take(D);
  r.lw = take((size_t)lay.L * lay.lds_layer_size());
  r.x = bwd ? nullptr : take((size_t)N * lay.DinS());
  r.tile = N * TS;  // multiple of 4 floats
  r.tiles = take((size_t)r.tile * (bwd ? 18 : 6));
  r.S = take((size_t)N * (N + 1));
  if (bwd) {
    r.GS = take((size_t)N * (N + 1));
    r.rstd0 = take(N);
    r.rstd1 = take(N);
    r.pg = take((size_t)lay.L * lay.layer_size());
    // staging scratch: max(gz chunk [N][65] + W2 chunk [64][TS], x chunk [N][129])
    const size_t s1 = (size_t)N * 65 + 64 * TS, s2 = (size_t)N * 129;
    r.scr = take(s1 > s2 ? s1 : s2);
  }
  if (out) *out = r;
  return o;
}

template <int D>
__device__ __forceinline__ void stage_weights(const Layout<D> &lay, const float *params, const Tiles<D> &t) {
  constexpr int TS = D + 4;
  const int Din = lay.Din, DinS = lay.DinS();
  // 16-byte copies when the packed layout allows (every row start a multiple of 4
  // floats from a 16-byte aligned base): a quarter of the loads and index divisions
  const bool vec = (Din & 3) == 0 && ((size_t)params & 15) == 0 &&
                   (lay.off_layer(0) & 3) == 0 && (lay.layer_size() & 3) == 0;
  if (vec) {
    const int q = Din / 4, qs = DinS / 4;
    for (int i = threadIdx.x; i < D * qs; i += NT) {
      const int r = i / qs, c4 = i - r * qs;
      reinterpret_cast<float4 *>(t.w1)[i] =
          c4 < q ? reinterpret_cast<const float4 *>(params)[r * q + c4]
                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
#pragma unroll 4
    for (int i = threadIdx.x; i < D * DinS; i += NT) {
      const int r = i / DinS, c = i - r * DinS;
      t.w1[i] = c < Din ? params[r * Din + c] : 0.f;
    }
  }
  for (int i = threadIdx.x; i < D; i += NT) t.b1[i] = params[lay.off_b1() + i];
  for (int l = 0; l < lay.L; ++l) {
    const float *g = params + lay.off_layer(l);
    float *w = t.lw + l * lay.lds_layer_size();
    if (vec) {
      constexpr int Q = D / 4;
      for (int i = threadIdx.x; i < 5 * D * Q; i += NT) {
        const int m = i / (D * Q), rq = i - m * D * Q, r = rq / Q, c4 = rq - r * Q;
        *reinterpret_cast<float4 *>(w + lay.l_w(m) + r * TS + 4 * c4) =
            *reinterpret_cast<const float4 *>(g + lay.g_w(m) + r * D + 4 * c4);
      }
    } else {
#pragma unroll 4
      for (int i = threadIdx.x; i < 5 * D * D; i += NT) {
        const int m = i / (D * D), rc = i - m * D * D, r = rc / D, c = rc - r * D;
        w[lay.l_w(m) + r * TS + c] = g[lay.g_w(m) + rc];
      }
    }
    for (int i = threadIdx.x; i < 5 * D; i += NT) {
      const int m = i / D, c = i - m * D;
      w[lay.l_b(m) + c] = g[lay.g_b(m) + c];
    }
    if (lay.ln)
      for (int i = threadIdx.x; i < 2 * D; i += NT) {
        w[lay.l_ln0() + i] = g[lay.g_ln0() + i];
        w[lay.l_ln1() + i] = g[lay.g_ln1() + i];
      }
  }
}

// One SAB forward on LDS tiles: reads t.H(), leaves the block output in t.H().
// KEEP: also store what the backward needs (T, XH0, XH1, rstd0/1).
// Ends with a __syncthreads().
template <int D, bool KEEP>
__device__ __forceinline__ void sab_forward(const Layout<D> &lay, const float *W, const float *presence_b,
                            int N, float sqrt_d, const Tiles<D> &t) {
  constexpr int TS = D + 4;
  const int NS = N + 1;
  const int tid = threadIdx.x;
  // s1: Q, K, V projections
  for (int e = tid; e < N * D; e += NT) {
    const int n = e / D, i = e - n * D;
    const float *h = t.H() + n * TS;
    t.Q()[n * TS + i] = W[lay.l_b(0) + i] + dotD<D>(h, W + lay.l_w(0) + i * TS);
    t.K()[n * TS + i] = W[lay.l_b(1) + i] + dotD<D>(h, W + lay.l_w(1) + i * TS);
    t.V()[n * TS + i] = W[lay.l_b(2) + i] + dotD<D>(h, W + lay.l_w(2) + i * TS);
  }
  __syncthreads();
  // s2: routing = (q k^T - (1 - presence) 1e32) / sqrt(d)   (set_transformer.py:40-43)
  for (int e = tid; e < N * N; e += NT) {
    const int n = e / N, m = e - n * N;
    float s = dotD<D>(t.Q() + n * TS, t.K() + m * TS);
    if (presence_b) s = s - (1.f - presence_b[m]) * 1e32f;
    t.S[n * NS + m] = s / sqrt_d;
  }
  __syncthreads();
  // s3: row softmax, 16 lanes per row
  for (int e = tid; e < ((N * 16 + NT - 1) / NT) * NT; e += NT) {
    const int n = e >> 4, l = e & 15;
    float v[NMAX / 16];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NMAX / 16; ++k) {
      const int m = l + 16 * k;
      v[k] = (n < N && m < N) ? t.S[n * NS + m] : -INFINITY;
      mx = fmaxf(mx, v[k]);
    }
    mx = group_max<16>(mx);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NMAX / 16; ++k) {
      v[k] = v[k] == -INFINITY ? 0.f : expf(v[k] - mx);
      sum += v[k];
    }
    sum = group_sum<16>(sum);
#pragma unroll
    for (int k = 0; k < NMAX / 16; ++k) {
      const int m = l + 16 * k;
      if (n < N && m < N) t.S[n * NS + m] = v[k] / sum;
    }
  }
  __syncthreads();
  // s4: A = P V
  for (int e = tid; e < N * D; e += NT) {
    const int n = e / D, i = e - n * D;
    float acc = 0.f;
#pragma unroll 4
    for (int m = 0; m < N; ++m) acc = fmaf(t.S[n * NS + m], t.V()[m * TS + i], acc);
    t.A()[n * TS + i] = acc;
  }
  __syncthreads();
  // s5: r = (Wo a + bo + h) * presence_n ; LN0 -> H1
  for (int e = tid; e < ((N * D + NT - 1) / NT) * NT; e += NT) {
    const bool ok = e < N * D;
    const int n = ok ? e / D : 0, i = ok ? e - n * D : 0;
    const float pn = presence_b ? presence_b[n] : 1.f;
    const float r = (W[lay.l_b(3) + i] + dotD<D>(t.A() + n * TS, W + lay.l_w(3) + i * TS) +
                     t.H()[n * TS + i]) * pn;
    float y = r;
    if (lay.ln) {
      const float mean = group_sum<D>(r) * (1.f / D);
      const float d = r - mean;
      const float rstd = 1.f / sqrtf(group_sum<D>(d * d) * (1.f / D) + kLnEps);
      const float xh = d * rstd;
      y = fmaf(xh, W[lay.l_ln0() + i], W[lay.l_ln0() + D + i]);
      if (KEEP && ok) {
        t.XH0()[n * TS + i] = xh;
        if (i == 0) t.rstd0[n] = rstd;
      }
    }
    if (ok) t.H1()[n * TS + i] = y;
  }
  __syncthreads();
  // s6: h2 = h1n + relu(Wf h1n + bf) ; LN1 -> H
  for (int e = tid; e < ((N * D + NT - 1) / NT) * NT; e += NT) {
    const bool ok = e < N * D;
    const int n = ok ? e / D : 0, i = ok ? e - n * D : 0;
    const float tv = W[lay.l_b(4) + i] + dotD<D>(t.H1() + n * TS, W + lay.l_w(4) + i * TS);
    const float h2 = t.H1()[n * TS + i] + fmaxf(tv, 0.f);
    float y = h2;
    if (lay.ln) {
      const float mean = group_sum<D>(h2) * (1.f / D);
      const float d = h2 - mean;
      const float rstd = 1.f / sqrtf(group_sum<D>(d * d) * (1.f / D) + kLnEps);
      const float xh = d * rstd;
      y = fmaf(xh, W[lay.l_ln1() + i], W[lay.l_ln1() + D + i]);
      if (KEEP && ok) {
        t.XH1()[n * TS + i] = xh;
        if (i == 0) t.rstd1[n] = rstd;
      }
    }
    if (KEEP && ok) t.T()[n * TS + i] = tv;
    if (ok) t.H()[n * TS + i] = y;
  }
  __syncthreads();
}

template <int D>
__global__ __launch_bounds__(NT) void st_fwd_kernel(StArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TS = D + 4;
  const Layout<D> lay{a.Din, a.Dout, a.L, a.layer_norm};
  const int N = a.N, Din = a.Din, DinS = lay.DinS(), tid = threadIdx.x;
  Tiles<D> t;
  carve<D>(lay, N, false, smem, &t);
  stage_weights<D>(lay, a.params, t);

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    __syncthreads();
    {  // stage the input rows (zero padded to DinS)
      int col = 0;
#pragma unroll
      for (int s = 0; s < MAXSEG; ++s) {
        if (s < a.nseg) {
          const int w = a.seg[s].width;
#pragma unroll 4
          for (int i = tid; i < N * w; i += NT) {
            const int n = i / w, j = i - n * w;
            t.x[n * DinS + col + j] =
                a.seg[s].ptr[(size_t)b * a.seg[s].bs + (size_t)n * a.seg[s].rs + j];
          }
          col += w;
        }
      }
      const int pad = DinS - Din;
      for (int i = tid; i < N * pad; i += NT) t.x[(i / pad) * DinS + Din + (i % pad)] = 0.f;
    }
    __syncthreads();
    const float *presence_b = a.presence ? a.presence + (size_t)b * N : nullptr;
    for (int e = tid; e < N * D; e += NT) {  // fc1
      const int n = e / D, i = e - n * D;
      t.H()[n * TS + i] = t.b1[i] + dot4(t.x + n * DinS, t.w1 + i * DinS, DinS / 4);
    }
    __syncthreads();
    float *hs = a.hsave + (size_t)b * (a.L + 1) * N * D;
    for (int l = 0; l <= a.L; ++l) {
      for (int e = tid; e < N * D; e += NT)
        hs[(size_t)l * N * D + e] = t.H()[(e / D) * TS + (e % D)];
      if (l < a.L)
        sab_forward<D, false>(lay, t.lw + l * lay.lds_layer_size(), presence_b, N, a.sqrt_d, t);
    }
    if (a.Dout == 0) {  // trunk only: the caller folds fc2 into what follows
      for (int e = tid; e < N * D; e += NT)
        a.z[(size_t)b * N * D + e] = t.H()[(e / D) * TS + (e % D)];
    }
    // fc2: z[n][c] = b2[c] + h[n] . W2[c]   (W2 rows straight from L2)
    const float *W2 = a.params + lay.off_w2();
    for (int e = tid; e < N * a.Dout; e += NT) {
      const int n = e / a.Dout, c = e - n * a.Dout;
      a.z[((size_t)b * N + n) * a.Dout + c] =
          a.params[lay.off_b2() + c] + dotD<D>(t.H() + n * TS, W2 + (size_t)c * D);
    }
  }
}

template <int D>
__global__ __launch_bounds__(NT) void st_bwd_kernel(StArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TS = D + 4;
  const Layout<D> lay{a.Din, a.Dout, a.L, a.layer_norm};
  const int N = a.N, Din = a.Din, DinS = lay.DinS(), tid = threadIdx.x, NS = N + 1;
  const int P = lay.total();
  Tiles<D> t;
  carve<D>(lay, N, true, smem, &t);
  stage_weights<D>(lay, a.params, t);
  for (int i = tid; i < a.L * lay.layer_size(); i += NT) t.pg[i] = 0.f;
  float *part = a.pg_partial + (size_t)blockIdx.x * P;  // this workgroup's partial grads
  const float *W2 = a.params + lay.off_w2();
  bool first = true;

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    __syncthreads();
    float *G = t.G(), *GN = t.GN();  // ping-pong: gradient w.r.t. the current block output
    const float *presence_b = a.presence ? a.presence + (size_t)b * N : nullptr;
    const float *hs = a.hsave + (size_t)b * (a.L + 1) * N * D;
    const float *gzb = a.gz + (size_t)b * N * a.Dout;

    // ---- fc2 backward -----------------------------------------------------
    for (int e = tid; e < N * D; e += NT)
      t.H()[(e / D) * TS + (e % D)] = hs[(size_t)a.L * N * D + e];
    __syncthreads();
    if (a.Dout == 0) {  // no fc2: gz is already the gradient of the trunk output
      for (int e = tid; e < N * D; e += NT)
        G[(e / D) * TS + (e % D)] = a.gz[(size_t)b * N * D + e];
      __syncthreads();
    }
    for (int c0 = 0; c0 < a.Dout; c0 += 64) {  // Dout walked in 64-column chunks via LDS
      const int cn = min(64, a.Dout - c0);
      float *sg = t.scr, *sw = t.scr + N * 65;
#pragma unroll 4
      for (int e = tid; e < N * cn; e += NT) {
        const int n = e / cn, cc = e - n * cn;
        sg[n * 65 + cc] = gzb[(size_t)n * a.Dout + c0 + cc];
      }
      for (int e = tid; e < cn * D; e += NT) sw[(e / D) * TS + (e % D)] = W2[(size_t)c0 * D + e];
      __syncthreads();
      for (int e = tid; e < N * D; e += NT) {  // G += gz W2
        const int n = e / D, j = e - n * D;
        float acc = c0 == 0 ? 0.f : G[n * TS + j];
#pragma unroll 8
        for (int cc = 0; cc < cn; ++cc) acc = fmaf(sg[n * 65 + cc], sw[cc * TS + j], acc);
        G[n * TS + j] = acc;
      }
      for (int e = tid; e < cn * D; e += NT) {  // dW2[c][j] = sum_n gz[n][c] h[n][j]
        const int cc = e / D, j = e - cc * D;
        float acc = 0.f, bacc = 0.f;
        for (int n = 0; n < N; ++n) {
          const float g = sg[n * 65 + cc];
          acc = fmaf(g, t.H()[n * TS + j], acc);
          bacc += g;
        }
        float *pw = part + lay.off_w2() + (size_t)c0 * D + e;
        *pw = first ? acc : *pw + acc;
        if (j == 0) {
          float *pb = part + lay.off_b2() + c0 + cc;
          *pb = first ? bacc : *pb + bacc;
        }
      }
      __syncthreads();
    }

    // ---- SAB blocks, last to first ----------------------------------------
    // (a block's saved input sits in L2: when one element per thread covers it, the
    // load for block l - 1 is issued while block l is being processed)
    const bool one_pass = N * D <= NT;
    float h_ahead = (one_pass && a.L > 0 && tid < N * D)
                        ? hs[(size_t)(a.L - 1) * N * D + tid] : 0.f;
    for (int l = a.L - 1; l >= 0; --l) {
      const float *W = t.lw + l * lay.lds_layer_size();
      float *PG = t.pg + l * lay.layer_size();
      const float *hin = hs + (size_t)l * N * D;  // the block's input (global, L2)
      if (one_pass) {
        if (tid < N * D) {
          t.H()[(tid / D) * TS + (tid % D)] = h_ahead;
          t.HIN()[(tid / D) * TS + (tid % D)] = h_ahead;
          if (l > 0) h_ahead = hs[(size_t)(l - 1) * N * D + tid];
        }
      } else {
        for (int e = tid; e < N * D; e += NT) {
          const float h = hin[e];
          t.H()[(e / D) * TS + (e % D)] = h;
          t.HIN()[(e / D) * TS + (e % D)] = h;
        }
      }
      __syncthreads();
      sab_forward<D, true>(lay, W, presence_b, N, a.sqrt_d, t);

      // b1: LN1 backward, ReLU gate
      for (int e = tid; e < ((N * D + NT - 1) / NT) * NT; e += NT) {
        const bool ok = e < N * D;
        const int n = ok ? e / D : 0, i = ok ? e - n * D : 0;
        const float g = ok ? G[n * TS + i] : 0.f;
        float g_h2 = g;
        if (lay.ln) {
          const float xh = t.XH1()[n * TS + i];
          const float gh = g * W[lay.l_ln1() + i];
          const float s1 = group_sum<D>(gh) * (1.f / D);
          const float s2 = group_sum<D>(gh * xh) * (1.f / D);
          g_h2 = (gh - s1 - xh * s2) * t.rstd1[n];
        }
        if (ok) {
          t.GH1()[n * TS + i] = g_h2;
          t.T()[n * TS + i] = t.T()[n * TS + i] > 0.f ? g_h2 : 0.f;  // g_t, in place
        }
      }
      __syncthreads();
      // b2: through Wf, LN0 backward, presence gate
      for (int e = tid; e < ((N * D + NT - 1) / NT) * NT; e += NT) {
        const bool ok = e < N * D;
        const int n = ok ? e / D : 0, i = ok ? e - n * D : 0;
        const float g_h1n = t.GH1()[n * TS + i] + dot_col<D>(t.T() + n * TS, W + lay.l_w(4), i);
        float g_r = g_h1n;
        if (lay.ln) {
          const float xh = t.XH0()[n * TS + i];
          const float gh = g_h1n * W[lay.l_ln0() + i];
          const float s1 = group_sum<D>(gh) * (1.f / D);
          const float s2 = group_sum<D>(gh * xh) * (1.f / D);
          g_r = (gh - s1 - xh * s2) * t.rstd0[n];
        }
        if (ok) {
          t.GO()[n * TS + i] = g_r * (presence_b ? presence_b[n] : 1.f);
          t.GH1()[n * TS + i] = g_h1n;  // own element: kept for the LN0 parameter grads
        }
      }
      __syncthreads();
      // b3: through Wo
      for (int e = tid; e < N * D; e += NT) {
        const int n = e / D, i = e - n * D;
        t.GA()[n * TS + i] = dot_col<D>(t.GO() + n * TS, W + lay.l_w(3), i);
      }
      __syncthreads();
      // b4: dL/dP
      for (int e = tid; e < N * N; e += NT) {
        const int n = e / N, m = e - n * N;
        t.GS[n * NS + m] = dotD<D>(t.GA() + n * TS, t.V() + m * TS);
      }
      __syncthreads();
      // b5: softmax backward, 16 lanes per row
      for (int e = tid; e < ((N * 16 + NT - 1) / NT) * NT; e += NT) {
        const int n = e >> 4, l16 = e & 15;
        float p[NMAX / 16], gp[NMAX / 16];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < NMAX / 16; ++k) {
          const int m = l16 + 16 * k;
          const bool in = n < N && m < N;
          p[k] = in ? t.S[n * NS + m] : 0.f;
          gp[k] = in ? t.GS[n * NS + m] : 0.f;
          dot = fmaf(p[k], gp[k], dot);
        }
        dot = group_sum<16>(dot);
#pragma unroll
        for (int k = 0; k < NMAX / 16; ++k) {
          const int m = l16 + 16 * k;
          if (n < N && m < N) t.GS[n * NS + m] = p[k] * (gp[k] - dot) / a.sqrt_d;
        }
      }
      __syncthreads();
      // b6: dQ, dK, dV
      for (int e = tid; e < N * D; e += NT) {
        const int n = e / D, i = e - n * D;
        float gq = 0.f, gk = 0.f, gv = 0.f;
#pragma unroll 4
        for (int m = 0; m < N; ++m) {
          gq = fmaf(t.GS[n * NS + m], t.K()[m * TS + i], gq);
          gk = fmaf(t.GS[m * NS + n], t.Q()[m * TS + i], gk);
          gv = fmaf(t.S[m * NS + n], t.GA()[m * TS + i], gv);
        }
        t.GQ()[n * TS + i] = gq;
        t.GK()[n * TS + i] = gk;
        t.GV()[n * TS + i] = gv;
      }
      __syncthreads();
      // b7: gradient w.r.t. the block input -> GN ; weight / bias / LN grads
      for (int e = tid; e < N * D; e += NT) {
        const int n = e / D, i = e - n * D;
        GN[n * TS + i] = t.GO()[n * TS + i] + dot_col<D>(t.GQ() + n * TS, W + lay.l_w(0), i) +
                           dot_col<D>(t.GK() + n * TS, W + lay.l_w(1), i) +
                           dot_col<D>(t.GV() + n * TS, W + lay.l_w(2), i);
      }
      // dW[i][j] += sum_n gy[n][i] x[n][j], db[i] += sum_n gy[n][i]; one call per
      // matrix with compile-time tile pointers (a runtime-selected pointer table
      // would push the whole tile struct to scratch memory)
      for (int e = tid; e < 5 * D * D; e += NT) {  // all five matrices in one pass
        const int m = e / (D * D), rc = e - m * D * D, i = rc / D, j = rc - i * D;
        // tile indices (see Tiles): gy = GQ,GK,GV,GO,T ; x = HIN,HIN,HIN,A,H1
        const int gi = m < 3 ? 14 + m : (m == 3 ? 12 : 6);
        const int xi = m < 3 ? 17 : (m == 3 ? 4 : 5);
        const float *gy = t.tiles + gi * t.tile, *xin = t.tiles + xi * t.tile;
        float a0 = 0.f, a1 = 0.f;
        int n = 0;
        for (; n + 1 < N; n += 2) {
          a0 = fmaf(gy[n * TS + i], xin[n * TS + j], a0);
          a1 = fmaf(gy[(n + 1) * TS + i], xin[(n + 1) * TS + j], a1);
        }
        if (n < N) a0 = fmaf(gy[n * TS + i], xin[n * TS + j], a0);
        PG[lay.g_w(m) + rc] += a0 + a1;
      }
      // (the column-sum loops below go to the threads with one weight-gradient element
      // less -- the upper half of the workgroup when 5 D D is not a multiple of NT)
      for (int e = NT >= 5 * D + 256 ? tid - (NT - 5 * D) : tid; e >= 0 && e < 5 * D;
           e += NT) {  // bias grads: column sums
        const int m = e / D, i = e - m * D;
        const int gi = m < 3 ? 14 + m : (m == 3 ? 12 : 6);
        const float *gy = t.tiles + gi * t.tile;
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc += gy[n * TS + i];
        PG[lay.g_b(m) + i] += acc;
      }
      if (lay.ln) {  // LN gamma / beta: column sums of gy * xhat and gy
        auto lngrad = [&](const float *gy, const float *xh, int off, int t0) {
          for (int e = tid - t0; e >= 0 && e < 2 * D; e += NT) {
            const int beta = e / D, i = e - beta * D;
            float acc = 0.f;
            for (int n = 0; n < N; ++n)
              acc += beta ? gy[n * TS + i] : gy[n * TS + i] * xh[n * TS + i];
            PG[off + e] += acc;
          }
        };
        const int t0 = NT >= 5 * D + 256 ? 256 : 0;  // idle-ish threads, disjoint ranges
        lngrad(t.GH1(), t.XH0(), lay.g_ln0(), t0);
        lngrad(G, t.XH1(), lay.g_ln1(), t0 + (t0 ? 2 * D : 0));
      }
      __syncthreads();
      {  // next block's output gradient
        float *tmp = G;
        G = GN;
        GN = tmp;
      }
    }

    // ---- fc1 backward -------------------------------------------------------
    for (int j0 = 0; j0 < Din; j0 += 128) {  // Din walked in 128-column chunks via LDS
      const int jn = min(128, Din - j0);
      __syncthreads();
      {  // the chunk's columns segment by segment (no per-element table walk)
        int c0 = 0;
#pragma unroll
        for (int sgi = 0; sgi < MAXSEG; ++sgi) {
          if (sgi < a.nseg) {
            const int w = a.seg[sgi].width;
            const int lo = max(c0, j0), hi = min(c0 + w, j0 + jn), cw = hi - lo;
            if (cw > 0) {  // workgroup-uniform
              const float *src = a.seg[sgi].ptr + (size_t)b * a.seg[sgi].bs + (lo - c0);
              const int rs = a.seg[sgi].rs;
#pragma unroll 4
              for (int e = tid; e < N * cw; e += NT) {
                const int n = e / cw, jj = e - n * cw;
                t.scr[n * 129 + (lo - j0) + jj] = src[(size_t)n * rs + jj];
              }
            }
            c0 += w;
          }
        }
      }
      __syncthreads();
      for (int e = tid; e < D * jn; e += NT) {  // dW1[i][j] = sum_n g[n][i] x[n][j]
        const int i = e / jn, jj = e - i * jn;
        float acc = 0.f;
#pragma unroll 4
        for (int n = 0; n < N; ++n) acc = fmaf(G[n * TS + i], t.scr[n * 129 + jj], acc);
        float *pw = part + (size_t)i * Din + j0 + jj;
        *pw = first ? acc : *pw + acc;
      }
    }
    for (int i = tid; i < D; i += NT) {
      float acc = 0.f;
      for (int n = 0; n < N; ++n) acc += G[n * TS + i];
      float *pb = part + lay.off_b1() + i;
      *pb = first ? acc : *pb + acc;
    }
    {  // input gradients for the segments that want one
      int col = 0;
#pragma unroll
      for (int s = 0; s < MAXSEG; ++s) {
        if (s < a.nseg) {
          const int w = a.seg[s].width;
          float *gdst = a.seg[s].grad;
          if (gdst) {
            for (int e = tid; e < N * w; e += NT) {
              const int n = e / w, j = e - n * w;
              float acc = 0.f;
#pragma unroll
              for (int i = 0; i < D; ++i)
                acc = fmaf(G[n * TS + i], t.w1[i * DinS + col + j], acc);
              gdst[((size_t)b * N + n) * w + j] = acc;
            }
          }
          col += w;
        }
      }
    }
    first = false;
  }
  __syncthreads();
  for (int i = tid; i < a.L * lay.layer_size(); i += NT) part[lay.off_layer(0) + i] = t.pg[i];
}

template <int D>
size_t lds_bytes(const StArgs &a, bool bwd) {
  const Layout<D> lay{a.Din, a.Dout, a.L, a.layer_norm};
  return carve<D>(lay, a.N, bwd, nullptr, nullptr) * sizeof(float);
}

template <int D>
int launch(const StArgs &a, bool bwd, int grid, hipStream_t st) {
  const size_t lds = lds_bytes<D>(a, bwd);
  if (lds > 160 * 1024) return SCAE_ERR_UNSUPPORTED;
  const void *fn = bwd ? reinterpret_cast<const void *>(st_bwd_kernel<D>)
                       : reinterpret_cast<const void *>(st_fwd_kernel<D>);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  if (bwd)
    scae::launch(st_bwd_kernel<D>, dim3(grid), dim3(NT), lds, st, a);
  else
    scae::launch(st_fwd_kernel<D>, dim3(grid), dim3(NT), lds, st, a);
  return scae_launch_status();
}

int fill_args(StArgs &a, int nseg, const float *const *seg_ptr, const int *seg_width,
              const int *seg_row_stride, const int64_t *seg_batch_stride,
              float *const *seg_grad, const float *presence, const float *params, int B, int N,
              int D, int Din, int Dout, int L, int layer_norm) {
  if (nseg < 1 || nseg > MAXSEG || !seg_ptr || !seg_width || !seg_row_stride ||
      !seg_batch_stride || !params)
    return SCAE_ERR_BAD_ARG;
  if (B <= 0 || N <= 0 || Din <= 0 || Dout < 0 || L < 0) return SCAE_ERR_BAD_ARG;
  if (N > NMAX || (D != 8 && D != 16 && D != 32)) return SCAE_ERR_UNSUPPORTED;
  int tot = 0;
  for (int s = 0; s < nseg; ++s) {
    if (!seg_ptr[s] || seg_width[s] <= 0) return SCAE_ERR_BAD_ARG;
    a.seg[s] = Seg{seg_ptr[s], seg_grad ? seg_grad[s] : nullptr, seg_width[s],
                   seg_row_stride[s], (long)seg_batch_stride[s]};
    tot += seg_width[s];
  }
  if (tot != Din) return SCAE_ERR_BAD_ARG;
  a.nseg = nseg;
  a.presence = presence;
  a.params = params;
  a.B = B;
  a.N = N;
  a.Din = Din;
  a.Dout = Dout;
  a.L = L;
  a.layer_norm = layer_norm;
  a.sqrt_d = sqrtf((float)D);
  return SCAE_OK;
}


// the wave-per-set MFMA kernels (set_encoder_wave.hip) where they apply; the
// environment variable SCAE_ST_WAVE=0 keeps the workgroup-per-set kernels (A/B timing)
bool use_wave(const StArgs &a, int D) {
  const char *e = getenv("SCAE_ST_WAVE");
  return !(e && *e == '0') && scae_st::wave_supported(a, D);
}
}  // namespace

extern "C" int scae_set_encoder_param_count(int D, int Din, int Dout, int L, int layer_norm) {
  return D * Din + D + L * (5 * (D * D + D) + (layer_norm ? 4 * D : 0)) + Dout * D + Dout;
}

extern "C" int scae_set_encoder_grid(int B) { return B < 512 ? B : 512; }

extern "C" int scae_set_encoder_supported(int N, int D, int Din, int Dout, int L,
                                          int layer_norm) {
  if (N <= 0 || N > NMAX || Din <= 0 || Dout < 0 || L < 0) return 0;
  StArgs a{};
  a.N = N;
  a.Din = Din;
  a.Dout = Dout;
  a.L = L;
  a.layer_norm = layer_norm;
  size_t need;
  switch (D) {
    case 8: need = std::max(lds_bytes<8>(a, true), lds_bytes<8>(a, false)); break;
    case 16: need = std::max(lds_bytes<16>(a, true), lds_bytes<16>(a, false)); break;
    case 32: need = std::max(lds_bytes<32>(a, true), lds_bytes<32>(a, false)); break;
    default: return 0;
  }
  return need <= 160 * 1024 ? 1 : 0;
}

extern "C" int scae_set_encoder_bf16_supported(int N, int D, int Din, int Dout, int L,
                                               int layer_norm) {
  if (N <= 0 || N > NMAX || Din <= 0 || Dout < 0 || L < 0) return 0;
  StArgs a{};
  a.N = N, a.Din = Din, a.Dout = Dout, a.L = L, a.layer_norm = layer_norm;
  return use_wave(a, D) ? 1 : 0;
}

namespace scae_fused {   // trunk_logprob.hip
bool trunk_logprob_supported(const scae_st::StArgs &a, int Dh, const scae_decoder_desc *d);
int trunk_logprob_launch(const scae_st::StArgs &a, int n_trunk, const scae_decoder_desc *d,
                         const float *x, float *tile_sums, float *lse_post, float *lse_prior,
                         hipStream_t st);
}  // namespace scae_fused

// The trunk forward with the part decoder's likelihood (tile sums) riding in the same launch
// where the shapes allow; two launches otherwise.  Same results either way.
static int encoder_fwd_logprob(bool bf16, int nseg, const float *const *seg_ptr,
                               const int *seg_width, const int *seg_row_stride,
                               const int64_t *seg_batch_stride, const float *presence,
                               const float *params, float *z, float *hsave, int B, int N, int D,
                               int Din, int Dout, int L, int layer_norm,
                               const scae_decoder_desc *d, const float *x, float *tile_sums,
                               float *lse_post, float *lse_prior, void *stream) {
  SCAE_REQUIRE(d && x && tile_sums && lse_post && lse_prior && z && hsave);
  StArgs a{};
  int rc = fill_args(a, nseg, seg_ptr, seg_width, seg_row_stride, seg_batch_stride, nullptr,
                     presence, params, B, N, D, Din, Dout, L, layer_norm);
  if (rc) return rc;
  a.z = z;
  a.hsave = hsave;
  a.bf16_attention = bf16;
  const char *e = getenv("SCAE_FUSE_TRUNK_LOGPROB");
  if (!(e && *e == '0') && use_wave(a, D) && scae_fused::trunk_logprob_supported(a, D, d) &&
      scae_render_gmm_logprob_tiles(d) > 0)
    return scae_fused::trunk_logprob_launch(a, scae_set_encoder_grid(B), d, x, tile_sums,
                                            lse_post, lse_prior, (hipStream_t)stream);
  rc = (bf16 ? scae_set_encoder_fwd_bf16 : scae_set_encoder_fwd_f32)(
      nseg, seg_ptr, seg_width, seg_row_stride, seg_batch_stride, presence, params, z, hsave, B,
      N, D, Din, Dout, L, layer_norm, stream);
  if (rc) return rc;
  return scae_render_gmm_logprob_sums_fwd_f32(d, x, tile_sums, lse_post, lse_prior, stream);
}
extern "C" int scae_set_encoder_fwd_logprob_f32(
    int nseg, const float *const *seg_ptr, const int *seg_width, const int *seg_row_stride,
    const int64_t *seg_batch_stride, const float *presence, const float *params, float *z,
    float *hsave, int B, int N, int D, int Din, int Dout, int L, int layer_norm,
    const scae_decoder_desc *d, const float *x, float *tile_sums, float *lse_post,
    float *lse_prior, void *stream) {
  return encoder_fwd_logprob(false, nseg, seg_ptr, seg_width, seg_row_stride, seg_batch_stride,
                             presence, params, z, hsave, B, N, D, Din, Dout, L, layer_norm, d, x,
                             tile_sums, lse_post, lse_prior, stream);
}
// ... with the bf16 attention products of scae_set_encoder_fwd_bf16 (BASELINE configs[2])
extern "C" int scae_set_encoder_fwd_logprob_bf16(
    int nseg, const float *const *seg_ptr, const int *seg_width, const int *seg_row_stride,
    const int64_t *seg_batch_stride, const float *presence, const float *params, float *z,
    float *hsave, int B, int N, int D, int Din, int Dout, int L, int layer_norm,
    const scae_decoder_desc *d, const float *x, float *tile_sums, float *lse_post,
    float *lse_prior, void *stream) {
  return encoder_fwd_logprob(true, nseg, seg_ptr, seg_width, seg_row_stride, seg_batch_stride,
                             presence, params, z, hsave, B, N, D, Din, Dout, L, layer_norm, d, x,
                             tile_sums, lse_post, lse_prior, stream);
}

namespace {
int encoder_fwd(bool bf16, int nseg, const float *const *seg_ptr, const int *seg_width,
                const int *seg_row_stride, const int64_t *seg_batch_stride,
                const float *presence, const float *params, float *z, float *hsave, int B, int N,
                int D, int Din, int Dout, int L, int layer_norm, void *stream) {
  StArgs a{};
  int rc = fill_args(a, nseg, seg_ptr, seg_width, seg_row_stride, seg_batch_stride, nullptr,
                     presence, params, B, N, D, Din, Dout, L, layer_norm);
  if (rc) return rc;
  SCAE_REQUIRE(z && hsave);
  a.z = z;
  a.hsave = hsave;
  a.bf16_attention = bf16;
  const int grid = scae_set_encoder_grid(B);
  if (use_wave(a, D)) return scae_st::wave_launch(a, false, grid, (hipStream_t)stream);
  if (bf16) return SCAE_ERR_UNSUPPORTED;   // only the matrix-core kernels have the bf16 form
  switch (D) {
    case 8: return launch<8>(a, false, grid, (hipStream_t)stream);
    case 16: return launch<16>(a, false, grid, (hipStream_t)stream);
    default: return launch<32>(a, false, grid, (hipStream_t)stream);
  }
}

int encoder_bwd(bool bf16, int nseg, const float *const *seg_ptr, const int *seg_width,
                const int *seg_row_stride, const int64_t *seg_batch_stride,
                float *const *seg_grad, const float *presence, const float *params,
                const float *hsave, const float *gz, float *pg_partial, int B, int N, int D,
                int Din, int Dout, int L, int layer_norm, void *stream) {
  StArgs a{};
  int rc = fill_args(a, nseg, seg_ptr, seg_width, seg_row_stride, seg_batch_stride, seg_grad,
                     presence, params, B, N, D, Din, Dout, L, layer_norm);
  if (rc) return rc;
  SCAE_REQUIRE(hsave && gz && pg_partial);
  a.hsave = const_cast<float *>(hsave);
  a.gz = gz;
  a.pg_partial = pg_partial;
  a.bf16_attention = bf16;
  const int grid = scae_set_encoder_grid(B);
  if (use_wave(a, D)) return scae_st::wave_launch(a, true, grid, (hipStream_t)stream);
  if (bf16) return SCAE_ERR_UNSUPPORTED;
  switch (D) {
    case 8: return launch<8>(a, true, grid, (hipStream_t)stream);
    case 16: return launch<16>(a, true, grid, (hipStream_t)stream);
    default: return launch<32>(a, true, grid, (hipStream_t)stream);
  }
}
}  // namespace

#define SCAE_ENCODER_ENTRY(SUFFIX, BF)                                                          \
  extern "C" int scae_set_encoder_fwd_##SUFFIX(                                                 \
      int nseg, const float *const *seg_ptr, const int *seg_width, const int *seg_row_stride,   \
      const int64_t *seg_batch_stride, const float *presence, const float *params, float *z,    \
      float *hsave, int B, int N, int D, int Din, int Dout, int L, int layer_norm,              \
      void *stream) {                                                                           \
    return encoder_fwd(BF, nseg, seg_ptr, seg_width, seg_row_stride, seg_batch_stride,          \
                       presence, params, z, hsave, B, N, D, Din, Dout, L, layer_norm, stream);  \
  }                                                                                             \
  extern "C" int scae_set_encoder_bwd_##SUFFIX(                                                 \
      int nseg, const float *const *seg_ptr, const int *seg_width, const int *seg_row_stride,   \
      const int64_t *seg_batch_stride, float *const *seg_grad, const float *presence,           \
      const float *params, const float *hsave, const float *gz, float *pg_partial, int B,       \
      int N, int D, int Din, int Dout, int L, int layer_norm, void *stream) {                    \
    return encoder_bwd(BF, nseg, seg_ptr, seg_width, seg_row_stride, seg_batch_stride,          \
                       seg_grad, presence, params, hsave, gz, pg_partial, B, N, D, Din, Dout,   \
                       L, layer_norm, stream);                                                  \
  }
SCAE_ENCODER_ENTRY(f32, false)
// bf16 operands / fp32 accumulation for the attention products of every block
// (BASELINE.json configs[2]); projections, LayerNorm, softmax and the saved activations stay
// fp32.  SCAE_ERR_UNSUPPORTED where only the workgroup-per-set kernels apply.
SCAE_ENCODER_ENTRY(bf16, true)
#undef SCAE_ENCODER_ENTRY
