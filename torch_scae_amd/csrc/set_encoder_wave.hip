// K2b on the matrix cores: ONE WAVEFRONT PER SET.
//
// The fused set-transformer trunk (fc1 -> L x SAB, set_transformer.py:107-142,:212-219)
// at the reference's sizes is a chain of ~25 tiny dependent products per set (24 x 16
// activations, 16 x 16 weights, 24 x 24 attention).  The workgroup-per-set kernels of
// set_encoder.hip spread each product over 512 threads and pay a workgroup barrier
// between every two stages: ~1 us per stage, 25 / 58 us per pass.  Here a single wave
// owns a set: every product is a handful of v_mfma_f32_16x16x4_f32 (exact fp32), the
// activations stay in registers in the MFMA output layout, and a stage boundary is an
// LDS round trip of the same wave (no barrier) whenever the next product needs the
// values as an operand.
//
// Layouts (lane l: r = l & 15, q = l >> 4; rows are padded to 32 = two 16-row tiles t):
//   O layout   : what MFMA leaves: o[t][reg] = M[16 t + 4 q + reg][r]   (all N x 16 /
//                N x N activations live like this; LayerNorm / softmax reduce over r
//                with 16-lane xor shuffles, column sums over n are register sums plus
//                xor 16 / 32);
//   operand    : v_mfma_f32_16x16x4_f32 takes A[row r][k], B[k][col r] with k picked by q;
//                which four k an instruction contracts is free as long as A and B
//                agree, so lane q owns K/4 CONSECUTIVE k and reads them as float4s:
//                  x W^T (B[k][c] = W[c][k])   : a float4 of W's row c (global, L2);
//                  g W   (B[k][c] = W[k][c])   : four floats of W's column c;
//                  products over n or m (K = 32): both operands from LDS tiles written
//                                                 transposed ([col][row], 16-byte stores).
//   LDS tiles  : "R" [16 NT rows][24] row-major (A operands with K = 16; stride 24 floats =
//                6 x 16 B makes the row-per-lane ds_read_b128 conflict free), "T"
//                [16][16 NT + 8] transposed / [16 NT][16 NT + 8] N x N (K = 16 NT); NT = number
//                of 16-row tiles of the set = waves of its workgroup (N <= 64).
// Rows / columns >= N carry finite padding that never reaches a valid entry: padded
// keys get probability 0, padded rows get presence 0 and a zero output gradient.
#include "set_encoder_args.h"
#include "wave_mfma.h"

// This file's device code is compiled twice -- here and inside trunk_logprob.hip's shared
// launch -- and both must produce the same bits (tests/test_timed_path.py): no implicit
// multiply-add contraction, whose choice depends on the surrounding code; the fused
// operations below are written as fmaf.
#pragma clang fp contract(off)

namespace scae_st {
namespace {
using namespace scae_wave;

// -DSCAE_STW_PROF: s_memtime stamps of workgroup 5's first lane at the stage boundaries of
// the backward kernel (tools/stw_prof.py reads them through scae_debug_stw_prof)
#ifdef SCAE_STW_PROF
__device__ unsigned long long g_stw_prof[160];
__device__ int g_stw_n;
#define STW_T()                                                                  \
  do {                                                                           \
    if (blockIdx.x == 5 && threadIdx.x == 0) {                                   \
      const int i_ = g_stw_n;                                                    \
      if (i_ < 160) g_stw_prof[i_] = __builtin_readcyclecounter(), g_stw_n = i_ + 1; \
    }                                                                            \
  } while (0)
#else
#define STW_T()
#endif

struct Lay {   // packed parameter offsets (floats), see set_encoder.hip
  int Din, L, ln;
  __device__ int b1() const { return D * Din; }
  __device__ int layer(int l) const { return D * Din + D + l * (5 * 272 + (ln ? 64 : 0)); }
  __device__ int w(int m) const { return m < 4 ? m * 272 : 4 * 272 + (ln ? 32 : 0); }
  __device__ int b(int m) const { return w(m) + 256; }
  __device__ int ln0() const { return 4 * 272; }
  __device__ int ln1() const { return b(4) + 16; }
  __device__ int total() const { return layer(L); }
};

struct Wave {
  int lane, r, q, t;   // t: the 16-row tile this wave owns (= its index in the workgroup)
  int ts;              // row stride of T tiles: 16 NT + 8
  // O layout (own tile) -> R tile (row-major, stride RS)
  __device__ __forceinline__ void wr_rows(float *tile, const f32x4 &o) const {
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[(16 * t + 4 * q + e) * RS + r] = o[e];
  }
  // O layout -> T tile, transposed: tile[col r][row]
  __device__ __forceinline__ void wr_cols(float *tile, const f32x4 &o) const {
    *reinterpret_cast<float4 *>(tile + r * ts + 16 * t + 4 * q) =
        make_float4(o[0], o[1], o[2], o[3]);
  }
  // own rows of an N x N matrix (column tiles u) -> [16 NT][ts] row-major
  template <int NT>
  __device__ __forceinline__ void wr_nn(float *tile, const f32x4 (&o)[NT]) const {
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[(16 * t + 4 * q + e) * ts + 16 * u + r] = o[u][e];
  }
  // ... and transposed: tile[col][row]
  template <int NT>
  __device__ __forceinline__ void wr_nn_t(float *tile, const f32x4 (&o)[NT]) const {
#pragma unroll
    for (int u = 0; u < NT; ++u)
      *reinterpret_cast<float4 *>(tile + (16 * u + r) * ts + 16 * t + 4 * q) =
          make_float4(o[u][0], o[u][1], o[u][2], o[u][3]);
  }
  // operand with K = 16 from an R tile: row 16 u + r, k = 4 q ..
  __device__ __forceinline__ float4 rd16(const float *tile, int u) const {
    return ld4(tile + (16 * u + r) * RS + 4 * q);
  }
  // operand with K = 16 NT from a T tile: row 16 u + r, k = 4 NT q ..
  template <int NT>
  __device__ __forceinline__ FK<NT> rdk(const float *tile, int u) const {
    const float *p = tile + (16 * u + r) * ts + 4 * NT * q;
    FK<NT> f;
#pragma unroll
    for (int j = 0; j < NT; ++j) f.v[j] = ld4(p + 4 * j);
    return f;
  }
  // B operand of x W^T from a 16 x 16 row-major matrix in global memory
  __device__ __forceinline__ float4 w_rows(const float *W) const { return ld4(W + r * D + 4 * q); }
  // B operand of g W (column r of W)
  __device__ __forceinline__ float4 w_cols(const float *W) const {
    return make_float4(W[(4 * q) * D + r], W[(4 * q + 1) * D + r], W[(4 * q + 2) * D + r],
                       W[(4 * q + 3) * D + r]);
  }
};
// products over the keys / queries: K = 16 NT
template <int NT>
__device__ __forceinline__ f32x4 mma_n(f32x4 acc, const Wave &w, const float *A, int ua,
                                       const float *B, int ub) {
  return mmak<NT>(acc, w.template rdk<NT>(A, ua), w.template rdk<NT>(B, ub));
}
// ... in the attention's precision (BF: bf16 operands, fp32 accumulate)
template <int NT, bool BF>
__device__ __forceinline__ f32x4 mma_np(f32x4 acc, const Wave &w, const float *A, int ua,
                                        const float *B, int ub) {
  return mmakp<NT, BF>(acc, w.template rdk<NT>(A, ua), w.template rdk<NT>(B, ub));
}

// LayerNorm over the 16 features of the rows of an O-layout tile
template <bool KEEP>
__device__ __forceinline__ void layer_norm(f32x4 &v, float gamma, float beta, f32x4 &xh,
                                           f32x4 &rstd) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x = v[e];
    const float mean = rsum(x) * (1.f / D);
    const float d = x - mean;
#ifdef SCAE_NO_RSQ
    const float rs = 1.f / sqrtf(rsum(d * d) * (1.f / D) + kLnEps);
#else
    const float rs = __builtin_amdgcn_rsqf(rsum(d * d) * (1.f / D) + kLnEps);   // (1 ulp; the
    // argument is >= 1e-5: the IEEE sqrt + division were ~40 instructions per row)
#endif
    const float h = d * rs;
    if (KEEP) xh[e] = h, rstd[e] = rs;
    v[e] = fmaf(h, gamma, beta);
  }
}

// everything a SAB forward leaves behind for the backward pass (own tile)
template <int NT>
struct SabState {
  f32x4 hin;          // block input
  f32x4 p[NT];        // attention probabilities of the own queries
  f32x4 ao;           // attention output P V
  f32x4 h1;           // after LN0 (input of the feed-forward)
  f32x4 tv;           // feed-forward pre-activation
  f32x4 xh0, rstd0, xh1, rstd1;
};

// LDS tile slots.  Small slots (an R tile [16 NT][24] or a transposed [16][16 NT + 8]
// tile), then large ones (N x N tiles [16 NT][16 NT + 8]).  Rows 16 t .. of a row-major
// slot / columns 16 t .. of a transposed one belong to wave t; "shared" slots are read
// across waves after a workgroup barrier.  Slots whose lifetimes do not overlap share
// storage (Hs / As / the backward's generic R tile; Qs / H1s; Ps / dS); the backward, which
// has three barriers per block, needs no double buffering of K / V.
template <int NT>
struct Geo {
  static constexpr int TSN = 16 * NT + 8;
  static constexpr int SMALL = 16 * NT * RS > 16 * TSN ? 16 * NT * RS : 16 * TSN;
  static constexpr int LARGE = 16 * NT * TSN;
  static constexpr int SCR = NT * (5 * 256 + 10 * D);   // hand-over area: a slot per wave
};
enum {
  S_HS = 0, S_QS,                      // private rows: Hs = As = GR, Qs = H1s
  S_KS0, S_VT0,                        // shared
  S_BWD_FIRST,
  S_KS1 = S_BWD_FIRST, S_VT1, S_FWD_SMALL,   // forward only: second K / V buffers
  S_VS = S_BWD_FIRST, S_KT, S_QT, S_GT,      // backward only: shared
  S_XT1, S_XT2,                              //                private transposed scratch
  S_BWD_SMALL
};
constexpr int S_AS = S_HS, S_GR = S_HS, S_H1S = S_QS;
enum { L_PS = 0, L_FWD_LARGE, L_PT = L_FWD_LARGE, L_DST, L_BWD_LARGE };
constexpr int L_DSR = L_PS;
template <int NT>
struct Tiles {
  float *base;    // small slots
  float *lbase;   // large slots
  float *scr;     // hand-over area (backward)
  __device__ __forceinline__ float *small(int i) const { return base + i * Geo<NT>::SMALL; }
  __device__ __forceinline__ float *large(int i) const { return lbase + i * Geo<NT>::LARGE; }
};

// One SAB: h (O layout, own tile) -> h.  NT = number of 16-row tiles (= waves) of the set.
// `par`: parity of the block counter (the forward's K / V buffers).  KEEP: fill `st` and
// leave V row-major / K transposed / Q transposed behind for the backward pass.
template <int NT, bool KEEP, bool BF>
__device__ __forceinline__ void sab_forward(const Wave &w, const Lay &lay, const float *Wl,
                                            const Tiles<NT> &tl, f32x4 &h, const f32x4 &pres,
                                            const float (&kmask)[NT], int N, float sqrt_d,
                                            int par, SabState<NT> *st) {
  // (the forward kernel alternates two K / V buffers: it has one barrier per block)
  float *Hs = tl.small(S_HS), *Qs = tl.small(S_QS), *As = tl.small(S_AS),
        *H1s = tl.small(S_H1S), *Ps = tl.large(L_PS),
        *Ks = tl.small(!KEEP && par ? S_KS1 : S_KS0),
        *Vt = tl.small(!KEEP && par ? S_VT1 : S_VT0);
  const int r = w.r, t = w.t;
  const float inv_sqrt_d = 1.f / sqrt_d;
  // weights of the block as B operands (issued up front: they come from L2)
  const float4 wq = w.w_rows(Wl + lay.w(0)), wk = w.w_rows(Wl + lay.w(1)),
               wv = w.w_rows(Wl + lay.w(2)), wo = w.w_rows(Wl + lay.w(3)),
               wf = w.w_rows(Wl + lay.w(4));
  const float bq = Wl[lay.b(0) + r], bk = Wl[lay.b(1) + r], bv = Wl[lay.b(2) + r],
              bo = Wl[lay.b(3) + r], bf = Wl[lay.b(4) + r];
  float g0 = 1.f, be0 = 0.f, g1 = 1.f, be1 = 0.f;
  if (lay.ln) {
    g0 = Wl[lay.ln0() + r], be0 = Wl[lay.ln0() + D + r];
    g1 = Wl[lay.ln1() + r], be1 = Wl[lay.ln1() + D + r];
  }
  if (KEEP) st->hin = h;
  STW_T();
  // q, k, v projections of the own rows
  w.wr_rows(Hs, h);
  lds_fence();
  f32x4 qo, ko, vo;
  {
    const float4 a = w.rd16(Hs, t);
    qo = mma16(splat(bq), a, wq);
    ko = mma16(splat(bk), a, wk);
    vo = mma16(splat(bv), a, wv);
  }
  w.wr_rows(Qs, qo);
  w.wr_rows(Ks, ko);
  w.wr_cols(Vt, vo);
  if (KEEP) {
    w.wr_rows(tl.small(S_VS), vo);
    w.wr_cols(tl.small(S_KT), ko);
    w.wr_cols(tl.small(S_QT), qo);
  }
  lds_fence();
  if (NT > 1) __syncthreads();   // the other wave's keys / values
  STW_T();
  // routing logits (q k^T - (1 - presence_m) 1e32) / sqrt(d), softmax over the keys m
  f32x4 s[NT];
  {
    const float4 qa = w.rd16(Qs, t);
#pragma unroll
    for (int u = 0; u < NT; ++u) s[u] = mma16p<BF>(splat(0.f), qa, w.rd16(Ks, u));
  }
  STW_T();
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    // (sqrt(16) = 4: the division is an exact multiplication)
    float v[NT], mx = -INFINITY, sum = 0.f;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      v[u] = 16 * u + r < N ? (s[u][e] - kmask[u]) * inv_sqrt_d : -INFINITY;
      mx = fmaxf(mx, v[u]);
    }
    mx = rmax(mx);
#pragma unroll
    for (int u = 0; u < NT; ++u) v[u] = __expf(v[u] - mx), sum += v[u];   // exp(-inf) = 0
#ifdef SCAE_NO_RCP
    const float inv = 1.f / rsum(sum);
#else
    const float inv = __builtin_amdgcn_rcpf(rsum(sum));   // (sum >= 1: the row maximum's term)
#endif
#pragma unroll
    for (int u = 0; u < NT; ++u) s[u][e] = v[u] * inv;
  }
  if (KEEP) {
#pragma unroll
    for (int u = 0; u < NT; ++u) st->p[u] = s[u];
  }
  STW_T();
  // a = P V
  w.template wr_nn<NT>(Ps, s);
  lds_fence();
  const f32x4 ao = mma_np<NT, BF>(splat(0.f), w, Ps, t, Vt, 0);   // B[k = m][c]: row c of Vt
  if (KEEP) st->ao = ao;
  STW_T();
  // r = (Wo a + bo + h) presence_n; LN0
  w.wr_rows(As, ao);
  lds_fence();
  f32x4 h1 = mma16(splat(bo), w.rd16(As, t), wo);
#pragma unroll
  for (int e = 0; e < 4; ++e) h1[e] = (h1[e] + h[e]) * pres[e];
  f32x4 dummy;
  if (lay.ln) layer_norm<KEEP>(h1, g0, be0, KEEP ? st->xh0 : dummy, KEEP ? st->rstd0 : dummy);
  STW_T();
  // h2 = h1 + relu(Wf h1 + bf); LN1
  w.wr_rows(H1s, h1);
  lds_fence();
  const f32x4 tv = mma16(splat(bf), w.rd16(H1s, t), wf);
  if (KEEP) st->tv = tv, st->h1 = h1;
#pragma unroll
  for (int e = 0; e < 4; ++e) h[e] = h1[e] + fmaxf(tv[e], 0.f);
  if (lay.ln) layer_norm<KEEP>(h, g1, be1, KEEP ? st->xh1 : dummy, KEEP ? st->rstd1 : dummy);
  STW_T();
}

// LDS of a workgroup (floats).  Forward: X [16 NT][XS] | W1s [16][XS] | small | large.
// Backward: small | hand-over area | large (fc1's backward reads its operands from global
// memory).
__host__ __device__ inline int xs_of(int Din) { return (Din + 15) / 16 * 16 + 4; }
template <int NT>
__host__ __device__ inline size_t lds_floats_fwd_aliased(int Din) {
  const size_t xw = (size_t)(16 * NT + 16) * xs_of(Din),
               tl = (size_t)S_FWD_SMALL * Geo<NT>::SMALL + (size_t)L_FWD_LARGE * Geo<NT>::LARGE;
  return xw > tl ? xw : tl;
}
template <int NT>
__host__ __device__ inline size_t lds_floats(int Din, bool bwd) {
  const size_t xs = xs_of(Din);
  if (!bwd)
    return (16 * NT + 16) * xs + (size_t)S_FWD_SMALL * Geo<NT>::SMALL +
           (size_t)L_FWD_LARGE * Geo<NT>::LARGE;
  return (size_t)S_BWD_SMALL * Geo<NT>::SMALL + Geo<NT>::SCR +
         (size_t)L_BWD_LARGE * Geo<NT>::LARGE;
}

// 4 bytes per lane, global -> LDS (lane l lands at lds + 4 l bytes): a row of up to 64
// floats per instruction, nothing held in registers, one wait for everything
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7ffffffc, 0x00020000);
}
__device__ __forceinline__ void dma4(rsrc_t r, float *lds, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 4,
                                           voff, soff, 0, 0);
}
// rows [n0, n1) x width floats (row stride sstride) -> LDS rows of stride XS, asynchronously
__device__ __forceinline__ void dma_rows(const float *src, int sstride, int n0, int n1, int width,
                                         float *dst, int XS, int lane, int nstep = 1) {
  const rsrc_t rs = make_rsrc(src);
  for (int c = 0; c < width; c += 64)
    if (c + lane < width)
      for (int n = n0; n < n1; n += nstep)
        dma4(rs, dst + n * XS + c, 4 * lane, 4 * (n * sstride + c));
}
__device__ __forceinline__ void dma_wait() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// [p, p + n) floats of LDS (p 16-byte aligned, n a multiple of 4) <- 0, 16 bytes per lane
__device__ __forceinline__ void lds_zero(float *p, int n, int tid, int nthreads) {
  for (int i = 4 * tid; i < n; i += 4 * nthreads)
    *reinterpret_cast<float4 *>(p + i) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// The own rows of the set's input (segments side by side) into X; X's padding stays zero.
// Wide segments travel by LDS-DMA (one instruction per 64 floats of a row, asynchronous; an
// LDS-DMA instruction costs the wave 60-100 cycles of issue whatever it moves).  A narrow
// segment would cost an instruction per ROW -- the 16 x (6 | 1 | 16) rows of cfg-2 were 48 of
// the wave's 80 -- so the narrow segments' columns of the 16 rows are walked as ONE index
// space, a lane per float, through registers: NF loads per lane.  Two halves: every transfer
// is issued before the caller does anything else, `commit` writes the registers to LDS.
constexpr int NARROW = 16;   // widest segment that goes through registers ...
constexpr int NF = 6;        // ... while rows x (their widths) fits 64 NF floats (else: all by DMA)
struct NarrowRows {
  float v[NF];
  int off[NF];   // LDS float index, -1: nothing
};
__device__ __forceinline__ void stage_input_issue(const StArgs &a, int b, const Wave &w, float *X,
                                                  int XS, NarrowRows &nr) {
  const int n0 = 16 * w.t, n1 = min(a.N, n0 + 16);
  int wn = 0;
#pragma unroll
  for (int s = 0; s < MAXSEG; ++s)
    if (s < a.nseg && a.seg[s].width <= NARROW) wn += a.seg[s].width;
  const int total = (n1 - n0) * wn;
  const bool regs = total > 0 && total <= 64 * NF;
  int col = 0;
#pragma unroll 1
  for (int s = 0; s < a.nseg; ++s) {
    const Seg &sg = a.seg[s];
    if (!(regs && sg.width <= NARROW))
      dma_rows(sg.ptr + (size_t)b * sg.bs, sg.rs, n0, n1, sg.width, X + col, XS, w.lane);
    col += sg.width;
  }
  const float inv_wn = 1.f / (float)(wn > 0 ? wn : 1);   // ((i + 0.5) / wn is exact enough here)
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    const int i = 64 * j + w.lane;
    nr.off[j] = -1, nr.v[j] = 0.f;
    if (regs && i < total) {
      const int n = (int)(((float)i + 0.5f) * inv_wn), cn = i - n * wn;
      const float *src = nullptr;
      int rs = 0, dc = 0, c0 = 0, c0n = 0;
#pragma unroll
      for (int s = 0; s < MAXSEG; ++s) {
        if (s < a.nseg) {
          const Seg &sg = a.seg[s];
          if (sg.width <= NARROW) {
            if (cn >= c0n) src = sg.ptr + (size_t)b * sg.bs + (cn - c0n), rs = sg.rs, dc = c0 + cn - c0n;
            c0n += sg.width;
          }
          c0 += sg.width;
        }
      }
      nr.off[j] = (n0 + n) * XS + dc;
      nr.v[j] = src[(size_t)(n0 + n) * rs];
    }
  }
}
__device__ __forceinline__ void stage_input_commit(float *X, const NarrowRows &nr) {
#pragma unroll
  for (int j = 0; j < NF; ++j)
    if (nr.off[j] >= 0) X[nr.off[j]] = nr.v[j];
}

// fc1 of the own tile: x W1^T + b1, two independent accumulation chains
__device__ __forceinline__ f32x4 fc1_forward(const Wave &w, const float *X, const float *W1s,
                                             int XS, float bias) {
  const int kq = (XS - 4) / 4;   // k per lane group (a multiple of 4)
  const float *xa = X + (16 * w.t + w.r) * XS + kq * w.q, *wb = W1s + w.r * XS + kq * w.q;
  f32x4 h0 = splat(bias), h1 = splat(0.f);
  int j = 0;
  for (; j + 8 <= kq; j += 8) {
    h0 = mma16(h0, ld4(xa + j), ld4(wb + j));
    h1 = mma16(h1, ld4(xa + j + 4), ld4(wb + j + 4));
  }
  if (j < kq) h0 = mma16(h0, ld4(xa + j), ld4(wb + j));
  return h0 + h1;
}

// ---- backward --------------------------------------------------------------------
// operand with K = 16 from the wave's own columns of a transposed tile: row r, k = 16 t + 4 q ..
__device__ __forceinline__ float4 rdT_own(const Wave &w, const float *tile) {
  return ld4(tile + w.r * w.ts + 16 * w.t + 4 * w.q);
}
// gradient through a LayerNorm (g: w.r.t. the output -> w.r.t. the input) and the
// own-tile column sums for gamma / beta
__device__ __forceinline__ void layer_norm_bwd(f32x4 &g, float gamma, const f32x4 &xh,
                                               const f32x4 &rstd, float &dgamma, float &dbeta) {
  f32x4 gx;
#pragma unroll
  for (int e = 0; e < 4; ++e) gx[e] = g[e] * xh[e];
  dgamma = csum(gx);
  dbeta = csum(g);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float gh = g[e] * gamma;
    const float s1 = rsum(gh) * (1.f / D), s2 = rsum(gh * xh[e]) * (1.f / D);
    g[e] = (gh - s1 - xh[e] * s2) * rstd[e];
  }
}
// parameter gradients of one block from the wave's own rows
struct LayerGrads {
  f32x4 w[5];    // dWq dWk dWv dWo dWf, O layout: w[m][e] = dW[4 q + e][r]
  float v[9];    // dbq dbk dbv dbo dbf dgamma0 dbeta0 dgamma1 dbeta1 (column r)
};

// Backward of one SAB given everything sab_forward<KEEP> left behind; G: gradient w.r.t.
// the block output on entry, w.r.t. its input on return (own tile).
template <int NT, bool BF>
__device__ __forceinline__ void sab_backward(const Wave &w, const Lay &lay, const float *Wl,
                                             const Tiles<NT> &tl, f32x4 &G, const f32x4 &pres,
                                             int N, float sqrt_d, const SabState<NT> &st,
                                             LayerGrads &lg) {
  float *XT1 = tl.small(S_XT1), *XT2 = tl.small(S_XT2), *GR = tl.small(S_GR),
        *GT = tl.small(S_GT), *Vs = tl.small(S_VS), *Kt = tl.small(S_KT),
        *Qt = tl.small(S_QT), *PT = tl.large(L_PT), *DSR = tl.large(L_DSR),
        *DST = tl.large(L_DST);
  const int r = w.r, t = w.t;
  const float inv_sqrt_d = 1.f / sqrt_d;
  const float4 wqT = w.w_cols(Wl + lay.w(0)), wkT = w.w_cols(Wl + lay.w(1)),
               wvT = w.w_cols(Wl + lay.w(2)), woT = w.w_cols(Wl + lay.w(3)),
               wfT = w.w_cols(Wl + lay.w(4));
#pragma unroll
  for (int i = 5; i < 9; ++i) lg.v[i] = 0.f;
  // LN1, ReLU, feed-forward
  if (lay.ln) layer_norm_bwd(G, Wl[lay.ln1() + r], st.xh1, st.rstd1, lg.v[7], lg.v[8]);
  f32x4 gt;
#pragma unroll
  for (int e = 0; e < 4; ++e) gt[e] = st.tv[e] > 0.f ? G[e] : 0.f;
  w.wr_cols(XT1, gt);
  w.wr_cols(XT2, st.h1);
  w.wr_rows(GR, gt);
  lds_fence();
  lg.w[4] = mma16(splat(0.f), rdT_own(w, XT1), rdT_own(w, XT2));
  lg.v[4] = csum(gt);
  f32x4 g1 = mma16(G, w.rd16(GR, t), wfT);   // g_h2 + g_t Wf
  STW_T();
  // LN0, presence gate, output projection
  if (lay.ln) layer_norm_bwd(g1, Wl[lay.ln0() + r], st.xh0, st.rstd0, lg.v[5], lg.v[6]);
  f32x4 go;
#pragma unroll
  for (int e = 0; e < 4; ++e) go[e] = g1[e] * pres[e];
  w.wr_cols(XT1, go);
  w.wr_cols(XT2, st.ao);
  w.wr_rows(GR, go);
  lds_fence();
  lg.w[3] = mma16(splat(0.f), rdT_own(w, XT1), rdT_own(w, XT2));
  lg.v[3] = csum(go);
  const f32x4 ga = mma16(splat(0.f), w.rd16(GR, t), woT);
  STW_T();
  // attention: dP = GA V^T, softmax backward
  w.wr_rows(GR, ga);
  w.wr_cols(GT, ga);
  lds_fence();
  f32x4 ds[NT];
  {
    const float4 gaa = w.rd16(GR, t);
#pragma unroll
    for (int u = 0; u < NT; ++u) ds[u] = mma16p<BF>(splat(0.f), gaa, w.rd16(Vs, u));
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float dot = 0.f;
#pragma unroll
    for (int u = 0; u < NT; ++u) dot = fmaf(st.p[u][e], ds[u][e], dot);
    dot = rsum(dot);
#pragma unroll
    for (int u = 0; u < NT; ++u) ds[u][e] = st.p[u][e] * (ds[u][e] - dot) * inv_sqrt_d;
  }
  STW_T();
  w.template wr_nn<NT>(DSR, ds);
  w.template wr_nn_t<NT>(DST, ds);
  w.template wr_nn_t<NT>(PT, st.p);
  lds_fence();
  if (NT > 1) __syncthreads();   // the other wave's dS^T, P^T, GA^T columns
  STW_T();
  const f32x4 dq = mma_np<NT, BF>(splat(0.f), w, DSR, t, Kt, 0);   // sum_m dS[n][m] K[m][i]
  const f32x4 dk = mma_np<NT, BF>(splat(0.f), w, DST, t, Qt, 0);   // sum_n dS[n][m] Q[n][i]
  const f32x4 dv = mma_np<NT, BF>(splat(0.f), w, PT, t, GT, 0);    // sum_n P[n][m] GA[n][i]
  STW_T();
  // projections: weight gradients and the gradient w.r.t. the block input
  w.wr_cols(XT2, st.hin);
  f32x4 gin = go;
  const f32x4 *dd[3] = {&dq, &dk, &dv};
  const float4 wT[3] = {wqT, wkT, wvT};
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    w.wr_cols(XT1, *dd[m]);
    w.wr_rows(GR, *dd[m]);
    lds_fence();
    lg.w[m] = mma16(splat(0.f), rdT_own(w, XT1), rdT_own(w, XT2));
    lg.v[m] = csum(*dd[m]);
    gin = mma16(gin, w.rd16(GR, t), wT[m]);
  }
  G = gin;
  STW_T();
}

// own-rows parameter gradients -> the workgroup's row of the partial matrix.  One wave
// stores its registers; several waves meet in LDS (a slot per wave) and the whole workgroup
// sums and stores, a thread per value: coalesced, and nobody reads 29 LDS values in a row
// (the hand-over to wave 0 was 1.2 us per block).
template <int NT>
__device__ __forceinline__ void flush_layer(const Wave &w, const Lay &lay, const Tiles<NT> &tl,
                                            const LayerGrads &lg, float *part, bool first) {
  constexpr int PW = 5 * 256 + 10 * D;   // floats per wave: 5 matrices, 9 (+1) vectors
  auto put = [&](int idx, float v) { part[idx] = first ? v : part[idx] + v; };
  if (NT == 1) {
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) put(lay.w(m) + (4 * w.q + e) * D + w.r, lg.w[m][e]);
    if (w.q == 0) {
      const int voff[9] = {lay.b(0), lay.b(1), lay.b(2), lay.b(3), lay.b(4), lay.ln0(),
                           lay.ln0() + D, lay.ln1(), lay.ln1() + D};
#pragma unroll
      for (int i = 0; i < 9; ++i)
        if (i < 5 || lay.ln) put(voff[i] + w.r, lg.v[i]);
    }
    return;
  }
  float *mine = tl.scr + w.t * PW;
#pragma unroll
  for (int m = 0; m < 5; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e) mine[m * 256 + (4 * w.q + e) * D + w.r] = lg.w[m][e];
  if (w.q == 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) mine[5 * 256 + i * D + w.r] = lg.v[i];
  }
  lds_fence();
  __syncthreads();
  static_assert(D == 16, "vector index split");
  // 356 items of four values: 320 of the five [16][16] matrices, 36 of the nine vectors; every
  // offset of the packed layout is a multiple of four floats (16 bytes with the row's base)
  constexpr int NTH = 64 * NT, NI = 5 * 64 + 9 * 4, ITS = (NI + NTH - 1) / NTH;
  float4 v[ITS];
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const int k = 4 * min((int)threadIdx.x + NTH * it, NI - 1);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 1; t < NT; ++t) {
      const float4 x = ld4(tl.scr + t * PW + k);
      o.x += x.x, o.y += x.y, o.z += x.z, o.w += x.w;
    }
    const float4 x = ld4(tl.scr + k);   // (wave 0's + the others' in order: the sum the
    v[it] = make_float4(x.x + o.x, x.y + o.y, x.z + o.z, x.w + o.w);   // hand-over made)
  }
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const int item = (int)threadIdx.x + NTH * it, k = 4 * item;
    const int m = k >> 8, iv = (k - 5 * 256) >> 4;
    const int wm = m * 272 + (m == 4 && lay.ln ? 32 : 0);   // lay.w(m)
    const int vo = iv < 5 ? iv * 272 + (iv == 4 && lay.ln ? 32 : 0) + 256   // lay.b(iv)
                          : iv == 5 ? lay.ln0()
                                    : iv == 6 ? lay.ln0() + D : iv == 7 ? lay.ln1() : lay.ln1() + D;
    const bool mat = k < 5 * 256;
    if (item < NI && (mat || iv < 5 || lay.ln)) {
      float4 *dst = reinterpret_cast<float4 *>(part + (mat ? wm + (k & 255) : vo + (k & 15)));
      float4 r = v[it];
      if (!first) {
        const float4 p = *dst;
        r.x += p.x, r.y += p.y, r.z += p.z, r.w += p.w;
      }
      *dst = r;
    }
  }
}

template <int NT, bool BF>
__global__ __launch_bounds__(64 * NT) void stw_bwd_kernel(StArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Lay lay{a.Din, a.L, a.layer_norm};
  const int N = a.N, Din = a.Din;
  Wave w;
  w.lane = threadIdx.x & 63, w.r = w.lane & 15, w.q = w.lane >> 4;
  w.t = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  w.ts = Geo<NT>::TSN;
  // small slots | hand-over area | large tiles
  float *sm = smem, *scr = sm + S_BWD_SMALL * Geo<NT>::SMALL, *arena = scr + Geo<NT>::SCR;
  const Tiles<NT> tiles{sm, arena, scr};
  const int lds_total = (int)lds_floats<NT>(Din, true);
#ifdef SCAE_STW_PROF
  if (blockIdx.x == 5 && threadIdx.x == 0) g_stw_n = 0;
#endif
  STW_T();
  lds_zero(smem, lds_total, threadIdx.x, 64 * NT);
  lds_fence();
  if (NT > 1) __syncthreads();
  float *part = a.pg_partial + (size_t)blockIdx.x * lay.total();
  STW_T();
  bool first = true;
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    f32x4 pres, G;
    float kmask[NT];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = 16 * w.t + 4 * w.q + e;
      pres[e] = n < N ? (a.presence ? a.presence[(size_t)b * N + n] : 1.f) : 0.f;
      G[e] = n < N ? a.gz[((size_t)b * N + n) * D + w.r] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int m = 16 * u + w.r;
      kmask[u] = a.presence && m < N ? (1.f - a.presence[(size_t)b * N + m]) * 1e32f : 0.f;
    }
    const float *hs = a.hsave + (size_t)b * (a.L + 1) * N * D;
    for (int l = a.L - 1; l >= 0; --l) {
      f32x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = 16 * w.t + 4 * w.q + e;
        h[e] = n < N ? hs[((size_t)l * N + n) * D + w.r] : 0.f;
      }
      SabState<NT> st;
      const float *Wl = a.params + lay.layer(l);
      sab_forward<NT, true, BF>(w, lay, Wl, tiles, h, pres, kmask, N, a.sqrt_d, 0, &st);
      LayerGrads lg;
      sab_backward<NT, BF>(w, lay, Wl, tiles, G, pres, N, a.sqrt_d, st, lg);
      flush_layer<NT>(w, lay, tiles, lg, part + lay.layer(l), first);
  STW_T();
    }
    // fc1: db1, dW1 = G^T x (column tiles split between the waves, K over all rows), input
    // gradients g W1 of the segments that want one.  Both B operands -- a column of the
    // set's input rows, a column of W1 -- come straight from global memory, 16 consecutive
    // columns per row group: every load of the first chunks is in flight before the first
    // product.  (Staging the input rows in LDS cost the wave one LDS-DMA instruction per
    // (row, segment), 60-100 cycles of issue each: 10 us^-1 of this kernel at cfg-2.)
    float *GT = tiles.small(S_GT), *GR = tiles.small(S_GR);
    constexpr int XT = NT <= 2 ? 5 : (NT == 3 ? 3 : 2);   // dW1 tiles per chunk (4 NT floats each)
    constexpr int WT = 10;                                  // input-gradient tiles per chunk
    const int ntile = (Din + 15) / 16;
    // column c of the concatenated input: its segment, as the lane sees it
    struct Col {
      const float *x;   // element (b, row 0, c)
      float *g;         // gradient of element (b, row 0, c), or null
      int rs, wd;       // row strides of the two
    };
    auto column = [&](int c) {
      Col k{nullptr, nullptr, 0, 0};
      int c0 = 0;
#pragma unroll
      for (int sgi = 0; sgi < MAXSEG; ++sgi) {
        if (sgi < a.nseg) {
          const Seg &sg = a.seg[sgi];
          if (c >= c0) {
            k.x = sg.ptr + (size_t)b * sg.bs + (c - c0);
            k.g = sg.grad ? sg.grad + (size_t)b * N * sg.width + (c - c0) : nullptr;
            k.rs = sg.rs, k.wd = sg.width;
          }
          c0 += sg.width;
        }
      }
      return k;
    };
    // column tiles that hold a column of a segment that wants its gradient (uniform; tiles
    // from 64 on count as wanted)
    unsigned long long want = 0;
    {
      int c0 = 0;
#pragma unroll
      for (int sgi = 0; sgi < MAXSEG; ++sgi) {
        if (sgi < a.nseg) {
          const int lo = c0 >> 4, hi = (c0 + a.seg[sgi].width - 1) >> 4;
          if (a.seg[sgi].grad && a.seg[sgi].width > 0 && lo < 64)
            want |= (hi >= 63 ? ~0ull : (2ull << hi) - 1) & ~((1ull << lo) - 1);
          c0 += a.seg[sgi].width;
        }
      }
    }
    float xv[XT][4 * NT];
    // (no branches around the loads -- tiles past the end re-read the last one -- or the
    // compiler moves each load down to the branch that uses it)
    auto load_x = [&](int jt0) {   // tiles jt0 + NT i of this wave
#pragma unroll
      for (int i = 0; i < XT; ++i) {
        const Col k = column(min(16 * min(jt0 + NT * i, ntile - 1) + w.r, Din - 1));
#pragma unroll
        for (int j = 0; j < 4 * NT; ++j)   // (rows >= N meet zero rows of G: any finite value)
          xv[i][j] = k.x[(size_t)min(4 * NT * w.q + j, N - 1) * k.rs];
      }
    };
    float4 wv[WT];
    auto load_w = [&](int jt0) {
#pragma unroll
      for (int i = 0; i < WT; ++i) {
        const float *wp = a.params + (size_t)(4 * w.q) * Din +
                          min(16 * min(jt0 + i, ntile - 1) + w.r, Din - 1);
        wv[i] = make_float4(wp[0], wp[Din], wp[2 * Din], wp[3 * Din]);
      }
    };
    load_x(w.t);
    load_w(0);
    w.wr_cols(GT, G);
    w.wr_rows(GR, G);
    const float db1 = csum(G);
    constexpr int PW = 5 * 256 + 10 * D;
    if (NT > 1 && w.t > 0 && w.q == 0) scr[(w.t - 1) * PW + 5 * 256 + 9 * D + w.r] = db1;
    lds_fence();
    if (NT > 1) __syncthreads();
  STW_T();
    if (w.t == 0 && w.q == 0) {
      float v = db1;
#pragma unroll
      for (int t = 1; t < NT; ++t) v += scr[(t - 1) * PW + 5 * 256 + 9 * D + w.r];
      part[lay.b1() + w.r] = first ? v : part[lay.b1() + w.r] + v;
    }
    // Every product of a chunk first, unconditionally (tiles past the end repeat the last
    // one), then the stores: a load whose only use sits in a guarded block is moved down
    // into that block by the compiler, and each tile then waits for its own latency.
    {
      const FK<NT> ga = w.template rdk<NT>(GT, 0);   // A[row i][k = n]: row i of G^T
      const float4 gr = w.rd16(GR, w.t);
      int jx = w.t, jw = 0;
#pragma unroll 1
      do {
        f32x4 gx[WT], dw[XT];
#pragma unroll
        for (int i = 0; i < WT; ++i) gx[i] = mma16(splat(0.f), gr, wv[i]);
#pragma unroll
        for (int i = 0; i < XT; ++i) {
          FK<NT> xb;   // B[k = n][col]: column 16 jt + r of the input
#pragma unroll
          for (int j = 0; j < NT; ++j)
            xb.v[j] = make_float4(xv[i][4 * j], xv[i][4 * j + 1], xv[i][4 * j + 2],
                                  xv[i][4 * j + 3]);
          dw[i] = mmak<NT>(splat(0.f), ga, xb);
        }
  STW_T();
#pragma unroll
        for (int i = 0; i < XT; ++i) {
          const int jt = jx + NT * i;
          if (jt < ntile && 16 * jt + w.r < Din) {
            float *dst = part + (4 * w.q) * Din + 16 * jt + w.r;
            if (!first) {
#pragma unroll
              for (int e = 0; e < 4; ++e) dw[i][e] += dst[e * Din];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[e * Din] = dw[i][e];
          }
        }
#pragma unroll
        for (int i = 0; i < WT; ++i) {
          const int jt = jw + i, c = 16 * jt + w.r;
          if (jt < ntile && (jt >= 64 || (want >> jt & 1))) {
            const Col k = column(min(c, Din - 1));
            if (c < Din && k.g) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int n = 16 * w.t + 4 * w.q + e;
                if (n < N) k.g[(size_t)n * k.wd] = gx[i][e];
              }
            }
          }
        }
        // (more than one chunk: wide inputs only)
        jx += NT * XT, jw += WT;
        if (jx < ntile) load_x(jx);
        if (jw < ntile) load_w(jw);
      } while (jx < ntile || jw < ntile);
    }
    first = false;
  STW_T();
    if (NT > 1) __syncthreads();   // GT / GR / the hand-over area are reused
  }
}

// The forward pass as a device function of workgroup `blk` of `nblk` (threads 0 .. 64 NT - 1 of
// the workgroup; `smem`: its dynamic LDS) -- so that it can also run as a block range of a
// launch it shares with an independent kernel (trunk_logprob.hip).
// ALIAS: the tiles of the blocks lie over X | W1s (dead after fc1) -- (16 NT + 16) XS floats
// instead of that plus the tiles, 28 KB instead of 51 at cfg-2 -- for the launch this body
// shares with the likelihood, where every workgroup is given the larger body's LDS; costs a
// re-staging of W1 and three more barriers per set.
template <int NT, bool BF, bool ALIAS = false>
__device__ __forceinline__ void stw_fwd_body(const StArgs &a, float *smem, int blk_id, int nblk) {
  const Lay lay{a.Din, a.L, a.layer_norm};
  const int N = a.N, XS = xs_of(a.Din);
  Wave w;
  w.lane = threadIdx.x & 63, w.r = w.lane & 15, w.q = w.lane >> 4;
  w.t = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  w.ts = Geo<NT>::TSN;
  // X | W1s | small slots | large slots  (ALIAS: the slots start at X)
  float *X = smem, *W1s = X + 16 * NT * XS, *sm = ALIAS ? smem : W1s + 16 * XS;
  const Tiles<NT> tiles{sm, sm + S_FWD_SMALL * Geo<NT>::SMALL, nullptr};
  // zero padding of X / W1s (rows >= N, columns >= Din), then W1: once per workgroup, or
  // (ALIAS) per set, after the previous set's tiles
  lds_zero(X, (16 * NT + 16) * XS, threadIdx.x, 64 * NT);
  lds_fence();
  if (NT > 1) __syncthreads();
  NarrowRows nr;
  if (blk_id < a.B) stage_input_issue(a, blk_id, w, X, XS, nr);   // (in flight with W1)
  dma_rows(a.params, a.Din, w.t, D, a.Din, W1s, XS, w.lane, NT);
  int blk = 0;   // running SAB counter: parity picks the K / V buffers
  for (int b = blk_id; b < a.B; b += nblk) {
    if (b != blk_id) {
      if (ALIAS) {
        if (NT > 1) __syncthreads();   // every wave is past its last tile read
        lds_zero(X, (16 * NT + 16) * XS, threadIdx.x, 64 * NT);
        lds_fence();
        if (NT > 1) __syncthreads();
        dma_rows(a.params, a.Din, w.t, D, a.Din, W1s, XS, w.lane, NT);
      }
      stage_input_issue(a, b, w, X, XS, nr);
    }
    f32x4 pres;
    float kmask[NT];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = 16 * w.t + 4 * w.q + e;
      pres[e] = n < N ? (a.presence ? a.presence[(size_t)b * N + n] : 1.f) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int m = 16 * u + w.r;
      kmask[u] = a.presence && m < N ? (1.f - a.presence[(size_t)b * N + m]) * 1e32f : 0.f;
    }
    const float bias1 = a.params[lay.b1() + w.r];
    stage_input_commit(X, nr);
    dma_wait();
    lds_fence();
    if (NT > 1 && (ALIAS || b == blk_id)) __syncthreads();   // W1s: rows of every wave
    f32x4 h = fc1_forward(w, X, W1s, XS, bias1);
    if (ALIAS) {   // the tiles take over: every wave has read W1s
      lds_fence();
      if (NT > 1) __syncthreads();
    }
    float *hs = a.hsave + (size_t)b * (a.L + 1) * N * D;
    const bool quad_ok = 16 * w.t + 4 * w.q + 3 < N;   // the lane's four rows all valid
    auto save = [&](float *dst) {
      if (quad_ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[(16 * w.t + 4 * w.q + e) * D + w.r] = h[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (16 * w.t + 4 * w.q + e < N) dst[(16 * w.t + 4 * w.q + e) * D + w.r] = h[e];
      }
    };
    for (int l = 0; l <= a.L; ++l) {
      save(hs + (size_t)l * N * D);
      if (l < a.L) {
        sab_forward<NT, false, BF>(w, lay, a.params + lay.layer(l), tiles, h, pres, kmask, N,
                               a.sqrt_d, blk & 1, nullptr);
        ++blk;
      }
    }
    save(a.z + (size_t)b * N * D);
  }
}

template <int NT, bool BF>
__global__ __launch_bounds__(64 * NT) void stw_fwd_kernel(StArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  stw_fwd_body<NT, BF>(a, smem, blockIdx.x, gridDim.x);
}
}  // namespace

#pragma clang fp contract(fast)   // (the toolchain's default, for what an includer adds)

#ifndef SCAE_DEVICE_ONLY   // (trunk_logprob.hip includes this file for its device code)
#ifdef SCAE_STW_PROF
}  // namespace scae_st
extern "C" int scae_debug_stw_prof(unsigned long long *out, int n) {
  int cnt = 0;
  (void)hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(&cnt, HIP_SYMBOL(scae_st::g_stw_n), sizeof(int)) != hipSuccess) return -1;
  if (cnt > n) cnt = n;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(scae_st::g_stw_prof), cnt * 8) != hipSuccess) return -1;
  return cnt;
}
namespace scae_st {
#endif
static int tiles_of(int N) { return (N + 15) / 16; }
static size_t lds_need(int N, int Din, bool bwd) {
  switch (tiles_of(N)) {
    case 1: return lds_floats<1>(Din, bwd) * sizeof(float);
    case 2: return lds_floats<2>(Din, bwd) * sizeof(float);
    case 3: return lds_floats<3>(Din, bwd) * sizeof(float);
    default: return lds_floats<4>(Din, bwd) * sizeof(float);
  }
}

bool wave_supported(const StArgs &a, int Dh) {
  return Dh == D && a.N <= 64 && a.Dout == 0 && a.L >= 0 && a.Din >= 1 &&
         lds_need(a.N, a.Din, true) <= 160 * 1024 && lds_need(a.N, a.Din, false) <= 160 * 1024;
}

template <int NT, bool BF>
static int launch_nt(const StArgs &a, bool bwd, int grid, hipStream_t st) {
  const size_t lds = lds_floats<NT>(a.Din, bwd) * sizeof(float);
  const void *fn = bwd ? reinterpret_cast<const void *>(stw_bwd_kernel<NT, BF>)
                       : reinterpret_cast<const void *>(stw_fwd_kernel<NT, BF>);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  if (bwd)
    scae::launch((stw_bwd_kernel<NT, BF>), dim3(grid), dim3(64 * NT), lds, st, a);
  else
    scae::launch((stw_fwd_kernel<NT, BF>), dim3(grid), dim3(64 * NT), lds, st, a);
  return scae_launch_status();
}

// a.bf16_attention: the attention products of every block (Q K^T, P V; dO V^T, dS K, dS^T Q,
// P^T dO in the backward) take bf16 operands on v_mfma_f32_16x16x16_bf16
int wave_launch(const StArgs &a, bool bwd, int grid, hipStream_t st) {
#define SCAE_STW(NTV) \
  return a.bf16_attention ? launch_nt<NTV, true>(a, bwd, grid, st) : launch_nt<NTV, false>(a, bwd, grid, st)
  switch (tiles_of(a.N)) {
    case 1: SCAE_STW(1);
    case 2: SCAE_STW(2);
    case 3: SCAE_STW(3);
    default: SCAE_STW(4);
  }
#undef SCAE_STW
}
#else
// the LDS floats a forward workgroup needs (for the shared launch)
template <int NT>
static size_t stw_fwd_lds_floats(int Din) { return lds_floats_fwd_aliased<NT>(Din); }
#endif
}  // namespace scae_st
