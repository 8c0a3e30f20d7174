// K8 -- the 3x3 "valid" convolutions of the part-capsule CNN encoder
// (part_encoder.py:26-44, nn_ext.py:34-59: Conv2d(k=3, stride s, no padding) +
// ReLU stacks) as implicit GEMMs on the CDNA4 fp32 matrix cores.
//
// Activations are kept NHWC so that one (output pixel, filter tap) row of the
// implicit A matrix is C_in contiguous floats; weights are re-laid once per
// step as Wf[co][tap][ci] (forward) and Wd[ci][tap][co] (data gradient).
//   forward : out[m][co]      = relu(sum_{tap,ci} in[pix(m,tap)][ci] Wf[co][tap][ci] + b)
//   dgrad   : din[m'][ci]     = gate(sum_{tap,co} dpre[opix(m',tap)][co] Wd[ci][tap][co])
//             tiles hold input pixels that are reached by the same set of taps
//             (border / stride-parity classes), so that no tile multiplies
//             structurally-zero taps
//   wgrad   : dW[tap][co][ci] = sum_m dpre[m][co] in[pix(m,tap)][ci]   (split over m),
//             db[co] = sum_m dpre[m][co] falls out of the staged A operand
// All passes share one MFMA tile loop (v_mfma_f32_16x16x4_f32, exact fp32
// products, fp32 accumulate) over 32-wide K chunks with register-prefetched
// staging, in two shapes:
//   * 64 x 64 tiles, 2 x 2 waves each owning a 32 x 32 sub-tile (large layers);
//   * 32 x 32 tiles whose 4 waves split every K chunk four ways and are summed
//     through LDS at the end (the encoder's layers have only 3k-10k output
//     pixels x 128 channels: 64 x 64 tiles would leave most of the 256 CUs
//     without a workgroup, 32 x 32 split-K tiles give 4x as many).
// Either way the accumulators leave through LDS as float4 rows (coalesced NHWC
// stores).  The first layer (C_in <= 4: nine-tap dot products) is a direct kernel.
#include <cstdlib>

#include "mfma_tile.h"
#include "mfma_pipe.h"
#include "conv_first_dev.h"
#include "seed_fold_dev.h"

#include <algorithm>

#include "wave_mfma.h"

// (device code of the riders of conv_bwd_pair_mixed_rider_kernel: the folding products'
// backward and the output attention's partial-row reduction)
#define SCAE_DEVICE_ONLY
namespace scae_sf {
#include "seed_fold.hip"
}
namespace scae_saw {
#include "seed_attention_wave.hip"
}
#undef SCAE_DEVICE_ONLY

namespace {
using namespace scae_tile;

using scae_first::ConvGeom;

// MODE: workgroup shape of mfma_tile.h (a `bool SK` argument selects 0 / 1);
// `smem` (Tile<MODE>::SMEM floats of LDS) is supplied by the enclosing kernel
#define SCAE_TILE_PROLOGUE_M(MODE)                                              \
  using TL = Tile<MODE>;                                                        \
  constexpr int T = TL::T, NQ = TL::NQ;                                         \
  float *As = smem, *Bs = smem + TL::OPER;                                      \
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63, r = lane & 15,  \
            q = lane >> 4;                                                      \
  typename TL::Acc acc[2][TL::NJ];                                              \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < TL::NJ; ++j) \
      acc[i][j] = typename TL::Acc{};
#define SCAE_TILE_PROLOGUE SCAE_TILE_PROLOGUE_M(SK)

// ---- forward: grid (Cout/TB, ceil(M/TA)) -----------------------------------------
template <int MODE>
__global__ __launch_bounds__(NT) void conv_fwd_kernel(const float *__restrict__ in,
                                                      const float *__restrict__ wf,
                                                      const float *__restrict__ bias,
                                                      float *__restrict__ out,
                                                      const float *__restrict__ post_bias,
                                                      float *__restrict__ out_post, ConvGeom g) {
  __shared__ __attribute__((aligned(16))) float smem[Tile<MODE>::SMEM];
  SCAE_TILE_PROLOGUE_M(MODE)
  constexpr int TB = TL::TB, NQB = TL::NQB;
  const int M = g.B * g.OH * g.OW, K = 9 * g.Cin;
  const int m0 = blockIdx.y * T, n0 = blockIdx.x * TB;
  // each thread stages the same rows every chunk: resolve their pixels once
  long abase[NQ];
  const float *bptr[NQB];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int id = tid + NT * i, m = m0 + id / QPR, kq = 4 * (id % QPR);
    abase[i] = -1;
    if (m < M) {
      const int n = m / (g.OH * g.OW), rem = m - n * g.OH * g.OW, oh = rem / g.OW,
                ow = rem - oh * g.OW;
      abase[i] = (((long)n * g.IH + oh * g.stride) * g.IW + ow * g.stride) * g.Cin + kq;
    }
  }
#pragma unroll
  for (int i = 0; i < NQB; ++i) {
    const int id = tid + NT * i;
    bptr[i] = wf + (size_t)(n0 + id / QPR) * K + 4 * (id % QPR);
  }
  auto fetch = [&](int c, Quads<NQ> &ra, Quads<NQB> &rb) {
    const int k0 = c * BK;
    const int tap = k0 / g.Cin, ci0 = k0 - tap * g.Cin, kh = tap / 3, kw = tap - kh * 3;
    const int off = (kh * g.IW + kw) * g.Cin + ci0;
#pragma unroll
    for (int i = 0; i < NQ; ++i) ra.v[i] = abase[i] >= 0 ? ld4(in + abase[i] + off) : zero4();
#pragma unroll
    for (int i = 0; i < NQB; ++i) rb.v[i] = ld4(bptr[i] + k0);
  };
  tile_mainloop<STAGES, MODE, true, true>(K / BK, As, Bs, acc, wid, r, q, fetch,
                                          [](const Quads<NQ> &) {});
  tile_epilogue<MODE>(smem, acc, wid, r, q, [&](int row, int col, float4 v) {
    const int m = m0 + row, n = n0 + col;
    if (m >= M) return;
    const float4 b = ld4(bias + n);
    const float4 o = make_float4(fmaxf(v.x + b.x, 0.f), fmaxf(v.y + b.y, 0.f),
                                 fmaxf(v.z + b.z, 0.f), fmaxf(v.w + b.w, 0.f));
    *reinterpret_cast<float4 *>(out + (size_t)m * g.Cout + n) = o;
    if (out_post) {  // + the per-(channel, pixel) embedding bias, (Cout, OH, OW)
      const int hw = g.OH * g.OW;
      const float *pb = post_bias + (size_t)n * hw + m % hw;
      *reinterpret_cast<float4 *>(out_post + (size_t)m * g.Cout + n) =
          make_float4(o.x + pb[0], o.y + pb[hw], o.z + pb[2 * hw], o.w + pb[3 * hw]);
    }
  });
}

// ---- data gradient ------------------------------------------------------------------
// din[pixel][ci] = gate(sum over the taps that reach the pixel of dpre[.][co] Wd[ci][tap][co]).
// An input row ih receives tap kh iff (ih - kh) is a multiple of the stride and
// (ih - kh)/s < OH: near the border (and for one stride parity) most of the nine
// taps miss.  Rows are therefore grouped into classes of equal valid-kh set,
// columns likewise; a workgroup tile holds pixels of ONE (row class, column
// class) pair and walks exactly that pair's taps -- no MFMA multiplies a
// structural zero (on a 7x7 input 51 % of the gather formulation would), and
// the operand fetch needs no bounds checks.  gate: the ReLU output of the
// producing layer at the same pixels, or nullptr.
constexpr int DG_MAXDIM = 128;  // input height / width covered by the class tables
struct DgradPlan {
  int nrc, ncc;                      // row / column classes (<= 8 each)
  unsigned char rmask[8], cmask[8];  // bit kh (kw) set: the tap reaches the class
  short rcount[8], ccount[8], rstart[8], cstart[8];
  unsigned char rlist[DG_MAXDIM], clist[DG_MAXDIM];  // ih (iw) of each class, ascending
  int tile_start[65];                // first tile (along grid.y) of class pair z
};

// classes of one axis: returns the number of classes.  Stride 1: one class per
// distinct valid-tap set (interior + the border rows).  Stride 2: one class per
// parity (tap sets differ mainly by parity there; splitting the single border
// row off each parity fragments the tiles for no gain -- measured), the class
// mask is the union and the fetch bounds-checks.
inline int dgrad_axis(int I, int O, int stride, unsigned char *mask, short *count, short *start,
                      unsigned char *list) {
  auto taps = [&](int i) {
    int m = 0;
    for (int k = 0; k < 3; ++k) {
      const int d = i - k;
      if (d >= 0 && d % stride == 0 && d / stride < O) m |= 1 << k;
    }
    return m;
  };
  int n = 0, pos = 0;
  // classes in order of decreasing tap count: the tiles with the longest K loops (the
  // interior, 3 x 3 taps) are dispatched first, the one-tap corners fill the tail
  static const int by_taps[8] = {7, 3, 5, 6, 1, 2, 4, 0};
  for (int wi = 0; wi < (stride == 1 ? 8 : stride); ++wi) {
    const int want = stride == 1 ? by_taps[wi] : wi;
    int cnt = 0, m = 0;
    for (int i = 0; i < I; ++i)
      if ((stride == 1 ? taps(i) : i % stride) == want) {
        list[pos + cnt++] = (unsigned char)i;
        m |= taps(i);
      }
    if (cnt) {
      mask[n] = (unsigned char)m, count[n] = (short)cnt, start[n] = (short)pos;
      pos += cnt, ++n;
    }
  }
  return n;
}

// (bx, by): the tile's position in the (C_in tiles, pixel tiles) grid
template <int MODE>
__device__ __forceinline__ void dgrad_tile(float *smem, int bx, int by,
                                           const float *__restrict__ dpre,
                                           const float *__restrict__ wd,
                                           const float *__restrict__ gate,
                                           float *__restrict__ din, const ConvGeom &g,
                                           const DgradPlan &pl) {
  SCAE_TILE_PROLOGUE_M(MODE)
  constexpr int TB = TL::TB, NQB = TL::NQB;
  // class pair of this tile: the number of class starts at or before it
  const int nz = pl.nrc * pl.ncc;
  const int z = __popcll(__ballot(lane + 1 < nz && by >= pl.tile_start[min(lane + 1, 64)]));
  const int rc = z / pl.ncc, cc = z - rc * pl.ncc;
  const int AH = pl.rcount[rc], AW = pl.ccount[cc], M = g.B * AH * AW, KT = 9 * g.Cout;
  const int m0 = (by - pl.tile_start[z]) * T, n0 = bx * TB;
  const int sh = g.stride - 1;  // stride 1 or 2
  // the class pair's taps: set bits of rmask x cmask
  const int rm = pl.rmask[rc], cm = pl.cmask[cc];
  const int nkh = __popc(rm), nkw = __popc(cm);
  auto nth_bit = [](int mask, int n) {  // n-th set bit of a 3-bit mask
    const int k0 = (mask & 1) ? 0 : ((mask & 2) ? 1 : 2);
    if (n == 0) return k0;
    const int rest = mask & ~(1 << k0);
    return (n == 1 && (rest & 2)) ? 1 : 2;
  };
  const int cpt = g.Cout / BK, nchunk = nkh * nkw * cpt;
  int pn[NQ], pih[NQ], piw[NQ];
  const float *bptr[NQB];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int id = tid + NT * i, m = m0 + id / QPR;
    pn[i] = -1, pih[i] = 0, piw[i] = 0;
    if (m < M) {
      const int n = m / (AH * AW), rem = m - n * AH * AW, a = rem / AW, b = rem - a * AW;
      pn[i] = n * g.OH * g.OW;
      pih[i] = pl.rlist[pl.rstart[rc] + a], piw[i] = pl.clist[pl.cstart[cc] + b];
    }
  }
#pragma unroll
  for (int i = 0; i < NQB; ++i) {
    const int id = tid + NT * i;
    bptr[i] = wd + (size_t)(n0 + id / QPR) * KT + 4 * (id % QPR);
  }
  auto fetch = [&](int c, Quads<NQ> &ra, Quads<NQB> &rb) {
    const int t = c / cpt, co0 = (c - t * cpt) * BK;
    const int ti = t / nkw, tj = t - ti * nkw;
    const int kh = nth_bit(rm, ti), kw = nth_bit(cm, tj);
    const int koff = (kh * 3 + kw) * g.Cout + co0;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      // (ih - kh) is a multiple of the stride for every tap of the class; the
      // range check only ever fails on the merged classes of stride 2
      const int dh = pih[i] - kh, dw = piw[i] - kw, oh = dh >> sh, ow = dw >> sh;
      const bool ok = pn[i] >= 0 && dh >= 0 && dw >= 0 && oh < g.OH && ow < g.OW;
      ra.v[i] = ok ? ld4(dpre + (size_t)(pn[i] + oh * g.OW + ow) * g.Cout + co0 +
                         4 * ((tid + NT * i) % QPR))
                   : zero4();
    }
#pragma unroll
    for (int i = 0; i < NQB; ++i) rb.v[i] = ld4(bptr[i] + koff);
  };
  tile_mainloop<STAGES, MODE, true, true>(nchunk, As, Bs, acc, wid, r, q, fetch,
                                          [](const Quads<NQ> &) {});
  tile_epilogue<MODE>(smem, acc, wid, r, q, [&](int row, int col, float4 v) {
    const int m = m0 + row;
    if (m >= M) return;
    const int nb = m / (AH * AW), rem = m - nb * AH * AW, a = rem / AW, b = rem - a * AW;
    const int ih = pl.rlist[pl.rstart[rc] + a], iw = pl.clist[pl.cstart[cc] + b];
    const size_t o = (((size_t)nb * g.IH + ih) * g.IW + iw) * g.Cin + n0 + col;
    if (gate) {
      const float4 gt = ld4(gate + o);
      v.x = gt.x > 0.f ? v.x : 0.f, v.y = gt.y > 0.f ? v.y : 0.f;
      v.z = gt.z > 0.f ? v.z : 0.f, v.w = gt.w > 0.f ? v.w : 0.f;
    }
    *reinterpret_cast<float4 *>(din + o) = v;
  });
}

template <int MODE>
__global__ __launch_bounds__(NT) void conv_dgrad_kernel(const float *__restrict__ dpre,
                                                        const float *__restrict__ wd,
                                                        const float *__restrict__ gate,
                                                        float *__restrict__ din, ConvGeom g,
                                                        DgradPlan pl) {
  __shared__ __attribute__((aligned(16))) float smem[Tile<MODE>::SMEM];
  dgrad_tile<MODE>(smem, blockIdx.x, blockIdx.y, dpre, wd, gate, din, g, pl);
}

// ---- weight gradient: grid (Cin/T, Cout/T, 9 taps * S splits) --------------------
// partial[(split*9 + tap)][co][ci], then bias partials [split][co] after 9*S slabs
template <int SK>   // workgroup shape (mfma_tile.h MODE 0, 1 or 3)
__device__ __forceinline__ void wgrad_tile(float *smem, int bx, int by, int bz,
                                           const float *__restrict__ dpre,
                                           const float *__restrict__ in,
                                           float *__restrict__ partial, const ConvGeom &g,
                                           int splits) {
  SCAE_TILE_PROLOGUE
  const int M = g.B * g.OH * g.OW;
  const int tap = bz % 9, split = bz / 9, kh = tap / 3, kw = tap - kh * 3;
  const int per = ((M + splits - 1) / splits + BK - 1) / BK * BK;
  const int kbeg = split * per, kend = min(M, kbeg + per);
  const int co0 = by * T, ci0 = bx * T;
  const bool want_bias = tap == 0 && bx == 0;  // workgroup-uniform
  float4 bsum = zero4();
  // each thread stages pixel m = kbeg + c*BK + (its k slot) of chunk c; the
  // mainloop fetches chunks in ascending order exactly once, so (n, oh, ow) of
  // that pixel is carried from chunk to chunk instead of re-divided
  int pn[NQ], poh[NQ], pow_[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    int kq, rq;
    kstr_pos<SK, T>(tid, i, kq, rq);
    const int m = kbeg + kq;
    pn[i] = m / (g.OH * g.OW);
    const int rem = m - pn[i] * g.OH * g.OW;
    poh[i] = rem / g.OW, pow_[i] = rem - poh[i] * g.OW;
  }
  const int dq = BK / g.OW, dr = BK - dq * g.OW;
  auto fetch = [&](int c, Quads<NQ> &ra, Quads<NQ> &rb) {
    const int k0 = kbeg + c * BK;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      int kq, rq;
      kstr_pos<SK, T>(tid, i, kq, rq);
      const int m = k0 + kq;
      ra.v[i] = rb.v[i] = zero4();
      if (m < kend) {
        const size_t pix = ((size_t)pn[i] * g.IH + poh[i] * g.stride + kh) * g.IW +
                           pow_[i] * g.stride + kw;
        ra.v[i] = ld4(dpre + (size_t)m * g.Cout + co0 + rq);
        rb.v[i] = ld4(in + pix * g.Cin + ci0 + rq);
      }
      pow_[i] += dr;  // advance by BK pixels
      const int carry = pow_[i] >= g.OW;
      pow_[i] -= carry ? g.OW : 0;
      poh[i] += dq + carry;
      while (poh[i] >= g.OH) poh[i] -= g.OH, ++pn[i];
    }
  };
  const int nchunk = kbeg < kend ? (kend - kbeg + BK - 1) / BK : 0;
  tile_mainloop<STAGES, SK, false, false>(nchunk, As, Bs, acc, wid, r, q, fetch,
                                          [&](const Quads<NQ> &ra) {
    if (want_bias) {
#pragma unroll
      for (int i = 0; i < NQ; ++i)
        bsum.x += ra.v[i].x, bsum.y += ra.v[i].y, bsum.z += ra.v[i].z, bsum.w += ra.v[i].w;
    }
  });
  float *dst = partial + (size_t)(split * 9 + tap) * g.Cout * g.Cin;
  tile_epilogue<SK>(smem, acc, wid, r, q, [&](int row, int col, float4 v) {
    *reinterpret_cast<float4 *>(dst + (size_t)(co0 + row) * g.Cin + ci0 + col) = v;
  });
  if (want_bias) {  // column sums of the staged dpre rows: threads tid % (T/4) share a quad
    __syncthreads();
    reinterpret_cast<float4 *>(smem)[tid] = bsum;
    __syncthreads();
    if (tid < T) {
      float sum = 0.f;
      for (int j = 0; j < NT / (T / 4); ++j) sum += smem[4 * (tid / 4 + (T / 4) * j) + (tid & 3)];
      partial[(size_t)splits * 9 * g.Cout * g.Cin + (size_t)split * g.Cout + co0 + tid] = sum;
    }
  }
}

template <int SK>
__global__ __launch_bounds__(NT) void conv_wgrad_kernel(const float *__restrict__ dpre,
                                                        const float *__restrict__ in,
                                                        float *__restrict__ partial, ConvGeom g,
                                                        int splits) {
  __shared__ __attribute__((aligned(16))) float smem[Tile<SK>::SMEM];
  wgrad_tile<SK>(smem, blockIdx.x, blockIdx.y, blockIdx.z, dpre, in, partial, g, splits);
}

// Data and weight gradient of one layer in ONE launch: both only wait for dpre, and
// at this model's sizes neither fills the 256 CUs on its own.  1-D grid: the first
// nd workgroups are data-gradient tiles (gx per pixel-tile row), the rest walk the
// weight-gradient grid (wx, wy, 9 * splits).
struct PairGrid {
  int nd, gx, wx, wy;
};
template <int DMODE, int WSK>
__global__ __launch_bounds__(NT) void conv_bwd_pair_kernel(
    const float *__restrict__ dpre, const float *__restrict__ wd, const float *__restrict__ gate,
    float *__restrict__ din, const float *__restrict__ in, float *__restrict__ partial,
    ConvGeom g, DgradPlan pl, int splits, PairGrid pg) {
  constexpr int SM = Tile<DMODE>::SMEM > Tile<WSK>::SMEM ? Tile<DMODE>::SMEM : Tile<WSK>::SMEM;
  __shared__ __attribute__((aligned(16))) float smem[SM];
  const int bid = blockIdx.x;
  if (bid < pg.nd) {  // workgroup-uniform
    dgrad_tile<DMODE>(smem, bid % pg.gx, bid / pg.gx, dpre, wd, gate, din, g, pl);
  } else {
    const int w = bid - pg.nd, bx = w % pg.wx, t = w / pg.wx;
    wgrad_tile<WSK>(smem, bx, t % pg.wy, t / pg.wy, dpre, in, partial, g, splits);
  }
}

// =====================================================================================
// Second-generation tiles (mfma_pipe.h): v_mfma_f32_32x32x2_f32, operands DMA'd global ->
// LDS into a 3-stage ring, one barrier per 64-wide K chunk.  Same math, same operand
// layouts (NHWC activations, Wf / Wd filters, per-split weight-gradient partials) and
// the same tap-class decomposition of the data gradient as above.
// =====================================================================================
namespace pipe = scae_pipe;

#define SCAE_PIPE_IDS                                                                  \
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6),        \
            lane = tid & 63, li = lane & 31, lk = lane >> 5;                           \
  const int wn = wid % T::WN, ks = wid / T::WN;

#ifdef SCAE_FWD_PROF   // phase stamps (s_memrealtime, 100 MHz) of every forward workgroup, a slot
// per layer (by input height): entry, set-up done, main loop done, end (tools/fwd_prof.py)
__device__ unsigned long long g_fwd_prof[3][2048][4];
#define FW_STAMP(g, blk, i)                                                               \
  do {                                                                                    \
    if (threadIdx.x == 0 && (blk) < 2048)                                                 \
      g_fwd_prof[(g).IH >= 15 ? 0 : ((g).IH >= 9 ? 1 : 2)][blk][i] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define FW_STAMP(g, blk, i)
#endif

// ---- forward ------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ void fwd_pipe_tile(float *smem, int bx, int by,
                                              const float *__restrict__ in,
                                              const float *__restrict__ wf,
                                              const float *__restrict__ bias,
                                              float *__restrict__ out,
                                              const float *__restrict__ post_bias,
                                              float *__restrict__ out_post, const ConvGeom &g) {
  SCAE_PIPE_IDS
  FW_STAMP(g, by * (g.Cout / T::TB) + bx, 0);
  const int M = g.B * g.OH * g.OW, K = 9 * g.Cin;
  const int m0 = by * T::TA, n0 = bx * T::TB;
  const pipe::DmaLane da = pipe::dma_lane<T::TA>(wid, lane), db = pipe::dma_lane<T::TB>(wid, lane);
  const pipe::rsrc_t ra = pipe::make_rsrc(in, (unsigned)((size_t)g.B * g.IH * g.IW * g.Cin * 4));
  const pipe::rsrc_t rb = pipe::make_rsrc(wf, (unsigned)((size_t)g.Cout * K * 4));
  int avo[T::RA], bvo[T::RB];   // per-lane byte offsets of the rows this lane moves
#pragma unroll
  for (int j = 0; j < T::RA; ++j) {
    const int m = min(m0 + da.row0 + pipe::DMA_ROWS * j, M - 1);   // rows past M: any valid line
    const int n = m / (g.OH * g.OW), rem = m - n * g.OH * g.OW, oh = rem / g.OW,
              ow = rem - oh * g.OW;
    avo[j] = ((((n * g.IH + oh * g.stride) * g.IW + ow * g.stride) * g.Cin) + da.koff) * 4;
  }
#pragma unroll
  for (int j = 0; j < T::RB; ++j)
    bvo[j] = ((n0 + db.row0 + pipe::DMA_ROWS * j) * K + db.koff) * 4;
  struct Ctx {
    int soa, sob;
  };
  // chunk c covers k = 32 c .. 32 c + 31 = channels ci0 .. of tap (kh, kw): carried
  int c_ci = 0, c_kw = 0, c_kh = 0, c_k = 0;
  auto chunk = [&](int) {
    const Ctx x{((c_kh * g.IW + c_kw) * g.Cin + c_ci) * 4, c_k * 4};
    c_k += pipe::BK, c_ci += pipe::BK;
    if (c_ci == g.Cin) {
      c_ci = 0;
      if (++c_kw == 3) c_kw = 0, ++c_kh;
    }
    return x;
  };
  auto issue = [&](const Ctx &x, float *st, int j) {   // j: compile-time after unrolling
    if (j < T::RA)
      pipe::dma16(ra, st + da.loff + j * pipe::DMA_ROWS * pipe::BKH, avo[j], x.soa);
    else
      pipe::dma16(rb, st + T::TA * pipe::BK + db.loff + (j - T::RA) * pipe::DMA_ROWS * pipe::BKH,
                  bvo[j - T::RA], x.sob);
  };
  pipe::f32x16 acc[T::MI][T::NI];
  pipe::kk_zero<T>(acc);
  FW_STAMP(g, by * (g.Cout / T::TB) + bx, 1);
  pipe::kk_mainloop<T>(K / pipe::BK, smem, acc, wn, ks, li, lk, chunk, issue);
  FW_STAMP(g, by * (g.Cout / T::TB) + bx, 2);
  const int hw = g.OH * g.OW;
  // every load of the epilogue is in flight before the first store (a load behind the row
  // guard costs its own L2 round trip per element: 16 x 2 of them were 2.4 us per tile)
  float bn[T::NI];
#pragma unroll
  for (int ni = 0; ni < T::NI; ++ni) bn[ni] = bias[n0 + (wn * T::NI + ni) * 32 + li];
  float pb[T::MI][T::NI][16];
  if (out_post) {   // the per-(channel, pixel) embedding bias, (Cout, OH, OW)
#pragma unroll
    for (int mi = 0; mi < T::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < T::NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = min(m0 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk, M - 1);
          pb[mi][ni][e] = post_bias[(size_t)(n0 + (wn * T::NI + ni) * 32 + li) * hw + m % hw];
        }
  }
  // Stores through range-checked descriptors: a row past M is dropped by the hardware, so the
  // epilogue has no divergent branch -- behind one the compiler cannot count the outstanding
  // stores and waits for each of them (vmcnt(0)) before the next one's operands.
  const pipe::rsrc_t ro = pipe::make_rsrc(out, (unsigned)((size_t)M * g.Cout * 4));
  const pipe::rsrc_t rp = pipe::make_rsrc(out_post ? out_post : out, (unsigned)((size_t)M * g.Cout * 4));
  pipe::kk_epilogue_idx<T>(smem, acc, wid, wn, ks, li, lk,
                           [&](int mi, int ni, int e, int row, int col, float v) {
    const int off = ((m0 + row) * g.Cout + n0 + col) * 4;
    const float o = fmaxf(v + bn[ni], 0.f);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, o), ro, off, 0, 0);
    if (out_post)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, o + pb[mi][ni][e]), rp, off, 0, 0);
  });
  FW_STAMP(g, by * (g.Cout / T::TB) + bx, 3);
}

template <class T>
__global__ __launch_bounds__(pipe::NT) void conv_fwd_pipe_kernel(
    const float *__restrict__ in, const float *__restrict__ wf, const float *__restrict__ bias,
    float *__restrict__ out, const float *__restrict__ post_bias, float *__restrict__ out_post,
    ConvGeom g) {
  __shared__ __attribute__((aligned(1024))) float smem[T::SMEM];
  fwd_pipe_tile<T>(smem, blockIdx.x, blockIdx.y, in, wf, bias, out, post_bias, out_post, g);
}

// The forward of a layer with the parameter-only folding products of the output attention
// (seed_fold_dev.h, K2d) as the tail of its grid.  In a training step those products used to
// ride in the prologue launch, where they were the longest part (15.5 us; the image layer
// next to them needs 11.5): nothing needs them before the object encoder, much later.  The
// second layer's launch -- 648 workgroups at B = 128, four per CU by LDS, 60 VGPRs -- has
// the room: 356 more workgroups of 256 threads fit beside its tiles (38 KB of LDS each,
// <= 128 VGPRs), and their dependent L2 round trips hide behind its MFMAs.
constexpr int FOLD_SMEM = 38 * 256;   // floats: scae_fold::lds_bytes(256, 16) = 37.0 KiB
template <class T>
__global__ __launch_bounds__(pipe::NT, 4) void conv_fwd_pipe_fold_kernel(
    const float *__restrict__ in, const float *__restrict__ wf, const float *__restrict__ bias,
    float *__restrict__ out, const float *__restrict__ post_bias, float *__restrict__ out_post,
    ConvGeom g, int gx, int n_conv, scae_seed_fold_desc fold, scae_fold::Plan plan) {
  __shared__ __attribute__((aligned(1024))) float smem[T::SMEM > FOLD_SMEM ? T::SMEM : FOLD_SMEM];
  const int blk = blockIdx.x;
  if (blk < n_conv)   // workgroup-uniform
    fwd_pipe_tile<T>(smem, blk % gx, blk / gx, in, wf, bias, out, post_bias, out_post, g);
  else
    scae_fold::forward_block_any<16>(fold, plan, blk - n_conv, smem);
}

// ---- data gradient (tap classes as in dgrad_tile) -----------------------------------
template <class T>
__device__ __forceinline__ void dgrad_pipe_tile(float *smem, int bx, int by,
                                                const float *__restrict__ dpre,
                                                const float *__restrict__ wd,
                                                const float *__restrict__ gate,
                                                float *__restrict__ din, const ConvGeom &g,
                                                const DgradPlan &pl) {
  SCAE_PIPE_IDS
  const int nz = pl.nrc * pl.ncc;
  const int z = __popcll(__ballot(lane + 1 < nz && by >= pl.tile_start[min(lane + 1, 64)]));
  const int rc = z / pl.ncc, cc = z - rc * pl.ncc;
  const int AH = pl.rcount[rc], AW = pl.ccount[cc], M = g.B * AH * AW, KT = 9 * g.Cout;
  const int m0 = (by - pl.tile_start[z]) * T::TA, n0 = bx * T::TB;
  const int sh = g.stride - 1;
  const int rm = pl.rmask[rc], cm = pl.cmask[cc];
  const int nkh = __popc(rm), nkw = __popc(cm);
  auto nth_bit = [](int mask, int n) {
    const int k0 = (mask & 1) ? 0 : ((mask & 2) ? 1 : 2);
    if (n == 0) return k0;
    const int rest = mask & ~(1 << k0);
    return (n == 1 && (rest & 2)) ? 1 : 2;
  };
  const int cpt = g.Cout / pipe::BK, nchunk = nkh * nkw * cpt;
  const pipe::DmaLane da = pipe::dma_lane<T::TA>(wid, lane), db = pipe::dma_lane<T::TB>(wid, lane);
  const pipe::rsrc_t ra =
      pipe::make_rsrc(dpre, (unsigned)((size_t)g.B * g.OH * g.OW * g.Cout * 4));
  const pipe::rsrc_t rb = pipe::make_rsrc(wd, (unsigned)((size_t)g.Cin * KT * 4));
  int pn[T::RA], pih[T::RA], piw[T::RA], bvo[T::RB];
#pragma unroll
  for (int j = 0; j < T::RA; ++j) {
    const int m = min(m0 + da.row0 + pipe::DMA_ROWS * j, M - 1);
    const int n = m / (AH * AW), rem = m - n * AH * AW, a = rem / AW, b = rem - a * AW;
    pn[j] = n * g.OH * g.OW;
    pih[j] = pl.rlist[pl.rstart[rc] + a], piw[j] = pl.clist[pl.cstart[cc] + b];
  }
#pragma unroll
  for (int j = 0; j < T::RB; ++j)
    bvo[j] = ((n0 + db.row0 + pipe::DMA_ROWS * j) * KT + db.koff) * 4;
  struct Ctx {
    int kh, kw, soa, sob;
  };
  int c_co = 0, c_ti = 0, c_tj = 0;   // chunk -> (tap (ti, tj) of the class, channel block)
  auto chunk = [&](int) {
    const int kh = nth_bit(rm, c_ti), kw = nth_bit(cm, c_tj);
    const Ctx x{kh, kw, c_co * 4, ((kh * 3 + kw) * g.Cout + c_co) * 4};
    c_co += pipe::BK;
    if (c_co == g.Cout) {
      c_co = 0;
      if (++c_tj == nkw) c_tj = 0, ++c_ti;
    }
    return x;
  };
  auto issue = [&](const Ctx &x, float *st, int j) {
    if (j < T::RA) {
      // the range check only ever fails on the merged classes of stride 2: such a
      // lane reads zeros (an offset outside the descriptor)
      const int dh = pih[j] - x.kh, dw = piw[j] - x.kw, oh = dh >> sh, ow = dw >> sh;
      const bool ok = dh >= 0 && dw >= 0 && oh < g.OH && ow < g.OW;
      const int vo = ok ? ((pn[j] + oh * g.OW + ow) * g.Cout + da.koff) * 4 : pipe::DMA_ZERO;
      pipe::dma16(ra, st + da.loff + j * pipe::DMA_ROWS * pipe::BKH, vo, x.soa);
    } else {
      pipe::dma16(rb, st + T::TA * pipe::BK + db.loff + (j - T::RA) * pipe::DMA_ROWS * pipe::BKH,
                  bvo[j - T::RA], x.sob);
    }
  };
  pipe::f32x16 acc[T::MI][T::NI];
  pipe::kk_zero<T>(acc);
  pipe::kk_mainloop<T>(nchunk, smem, acc, wn, ks, li, lk, chunk, issue);
  // the rows this lane finishes (16 per 32-row block), their pixels and gates: every load in
  // flight before the first store, stores through a range-checked descriptor (no branches)
  int ooff[T::MI][16];
  float gt[T::MI][T::NI][16];
#pragma unroll
  for (int mi = 0; mi < T::MI; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk, mc = min(m, M - 1);
      const int nb = mc / (AH * AW), rem = mc - nb * AH * AW, a = rem / AW, b = rem - a * AW;
      const int ih = pl.rlist[pl.rstart[rc] + a], iw = pl.clist[pl.cstart[cc] + b];
      const int o = (((nb * g.IH + ih) * g.IW + iw) * g.Cin + n0) * 4;
      ooff[mi][e] = m < M ? o : pipe::DMA_ZERO;
#pragma unroll
      for (int ni = 0; ni < T::NI; ++ni)
        gt[mi][ni][e] = gate ? gate[o / 4 + (wn * T::NI + ni) * 32 + li] : 1.f;
    }
  const pipe::rsrc_t rdin = pipe::make_rsrc(din, (unsigned)((size_t)g.B * g.IH * g.IW * g.Cin * 4));
  pipe::kk_epilogue_idx<T>(smem, acc, wid, wn, ks, li, lk,
                           [&](int mi, int ni, int e, int, int col, float v) {
    v = gt[mi][ni][e] > 0.f ? v : 0.f;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rdin, ooff[mi][e] + col * 4, 0, 0);
  });
}

// ---- weight gradient: tile (bx, by) of tap / split bz ------------------------------------
template <class T>
__device__ __forceinline__ void wgrad_pipe_tile(float *smem, int bx, int by, int bz,
                                                const float *__restrict__ dpre,
                                                const float *__restrict__ in,
                                                float *__restrict__ partial, const ConvGeom &g,
                                                int splits) {
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63,
            li = lane & 31, lk = lane >> 5, wm = wid >> 1, wn = wid & 1;
  const int M = g.B * g.OH * g.OW;
  const int tap = bz % 9, split = bz / 9, kh = tap / 3, kw = tap - kh * 3;
  const int per = ((M + splits - 1) / splits + T::BKW - 1) / T::BKW * T::BKW;
  const int kbeg = split * per, kend = min(M, kbeg + per);
  const int co0 = by * T::TA, ci0 = bx * T::TB;
  const bool want_bias = tap == 0 && bx == 0;   // workgroup-uniform
  // DMA pieces: 256 consecutive floats of the [k][rows] tile; piece p = wid + 4 j of the
  // A tile, then of the B tile.  Lane l moves floats 4 l .. 4 l + 3 of its piece.
  constexpr int KA = 256 / T::TA, KB = 256 / T::TB;   // k rows per piece
  const int ka = (4 * lane) / T::TA, ca = (4 * lane) % T::TA;
  const int kb = (4 * lane) / T::TB, cb = (4 * lane) % T::TB;
  // descriptors end at pixel kend: rows past the split's end read zeros
  const pipe::rsrc_t ra = pipe::make_rsrc(dpre, (unsigned)((size_t)kend * g.Cout * 4));
  const pipe::rsrc_t rb = pipe::make_rsrc(in, (unsigned)((size_t)g.B * g.IH * g.IW * g.Cin * 4));
  // the pixel (n, oh, ow) of each B piece is carried from chunk to chunk (every piece is
  // issued exactly once per chunk, in ascending chunk order) instead of re-divided
  constexpr int NA = T::PA / 4, NB = T::PB / 4;
  int avo[NA], pm[NB], pn[NB], poh[NB], pow_[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j)
    avo[j] = ((kbeg + (wid + 4 * j) * KA + ka) * g.Cout + co0 + ca) * 4;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    pm[j] = kbeg + (wid + 4 * j) * KB + kb;
    pn[j] = pm[j] / (g.OH * g.OW);
    const int rem = pm[j] - pn[j] * g.OH * g.OW;
    poh[j] = rem / g.OW, pow_[j] = rem - poh[j] * g.OW;
  }
  const int dq = T::BKW / g.OW, dr = T::BKW - dq * g.OW;
  const int cstep = T::BKW * g.Cout * 4;
  auto issue = [&](int c, float *st, int j) {
    if (j < NA) {
      // (the chunk offset rides in the per-lane offset: the descriptor's range check,
      // which makes the rows past kend read zeros, does not see the scalar offset)
      pipe::dma16(ra, st + (wid + 4 * j) * 256, avo[j] + c * cstep, 0);
    } else {
      const int jb = j - NA, p = wid + 4 * jb;
      const int vo = pm[jb] < kend
                         ? ((((pn[jb] * g.IH + poh[jb] * g.stride + kh) * g.IW +
                              pow_[jb] * g.stride + kw) * g.Cin) + ci0 + cb) * 4
                         : pipe::DMA_ZERO;
      pipe::dma16(rb, st + T::TA * T::BKW + p * 256, vo, 0);
      pm[jb] += T::BKW;   // advance by one chunk of pixels
      pow_[jb] += dr;
      const int carry = pow_[jb] >= g.OW;
      pow_[jb] -= carry ? g.OW : 0;
      poh[jb] += dq + carry;
      while (poh[jb] >= g.OH) poh[jb] -= g.OH, ++pn[jb];
    }
  };
  const int nchunk = kbeg < kend ? (kend - kbeg + T::BKW - 1) / T::BKW : 0;
  pipe::f32x16 acc[T::MI][T::NI];
#pragma unroll
  for (int a = 0; a < T::MI; ++a)
#pragma unroll
    for (int b = 0; b < T::NI; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  float bsum = 0.f;   // column sum of dpre over this split (thread = channel, tid < TA)
  pipe::ss_mainloop<T>(nchunk, smem, acc, wm, wn, li, lk, issue, [&](const float *st) {
    if (want_bias && tid < T::TA) {
#pragma unroll 16
      for (int k = 0; k < T::BKW; ++k) bsum += st[k * T::TA + tid];
    }
  });
  float *dst = partial + (size_t)(split * 9 + tap) * g.Cout * g.Cin;
#pragma unroll
  for (int a = 0; a < T::MI; ++a)
#pragma unroll
    for (int b = 0; b < T::NI; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (wm * T::MI + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        const int col = (wn * T::NI + b) * 32 + li;
        dst[(size_t)(co0 + row) * g.Cin + ci0 + col] = acc[a][b][e];
      }
  if (want_bias && tid < T::TA)
    partial[(size_t)splits * 9 * g.Cout * g.Cin + (size_t)split * g.Cout + co0 + tid] = bsum;
}

template <class T>
__global__ __launch_bounds__(pipe::NT) void conv_dgrad_pipe_kernel(
    const float *__restrict__ dpre, const float *__restrict__ wd, const float *__restrict__ gate,
    float *__restrict__ din, ConvGeom g, DgradPlan pl) {
  __shared__ __attribute__((aligned(1024))) float smem[T::SMEM];
  dgrad_pipe_tile<T>(smem, blockIdx.x, blockIdx.y, dpre, wd, gate, din, g, pl);
}
template <class T>
__global__ __launch_bounds__(pipe::NT) void conv_wgrad_pipe_kernel(
    const float *__restrict__ dpre, const float *__restrict__ in, float *__restrict__ partial,
    ConvGeom g, int splits) {
  __shared__ __attribute__((aligned(1024))) float smem[T::SMEM];
  wgrad_pipe_tile<T>(smem, blockIdx.x, blockIdx.y, blockIdx.z, dpre, in, partial, g, splits);
}
// data- and weight-gradient tiles of a layer in one launch (see conv_bwd_pair_kernel)
template <class TD, class TW>
__global__ __launch_bounds__(pipe::NT) void conv_bwd_pair_pipe_kernel(
    const float *__restrict__ dpre, const float *__restrict__ wd, const float *__restrict__ gate,
    float *__restrict__ din, const float *__restrict__ in, float *__restrict__ partial,
    ConvGeom g, DgradPlan pl, int splits, PairGrid pg) {
  constexpr int SM = TD::SMEM > TW::SMEM ? TD::SMEM : TW::SMEM;
  __shared__ __attribute__((aligned(1024))) float smem[SM];
  const int bid = blockIdx.x;
  if (bid < pg.nd) {   // workgroup-uniform
    dgrad_pipe_tile<TD>(smem, bid % pg.gx, bid / pg.gx, dpre, wd, gate, din, g, pl);
  } else {
    const int w = bid - pg.nd, bx = w % pg.wx, t = w / pg.wx;
    wgrad_pipe_tile<TW>(smem, bx, t % pg.wy, t / pg.wy, dpre, in, partial, g, splits);
  }
}

// ---- data-gradient tile, third form (DMODE 4) ------------------------------------------------
// 64 input pixels of ONE tap-class pair x 128 input channels, 2 x 2 waves of 32 x 64.  Both
// operands reach LDS by DMA as the fp32 rows they are in memory -- 32-float chunks = 128-byte
// rows, 16-byte quads XOR-swizzled on the SOURCE side so that the 16-lane groups of a
// ds_read_b128 cover all 64 banks --, a lane's fragment (8 consecutive k) is two quads, split in
// registers into its three bf16 parts and multiplied as six exact products (bf16x6.h).  The
// gradient rows of a tap that misses a pixel (stride 2: the merged parity classes) and the rows
// past the class's end are DMA zeros.
namespace dgx {
constexpr int TM = 64, TN = 128, BKF = 32, ROWB = 4 * BKF;
constexpr int A_B = TM * ROWB, B_B = TN * ROWB, STAGE_B = A_B + B_B;   // 8 + 16 KiB
constexpr int SLAB = 32 * 36;                                          // epilogue: floats per wave
constexpr int SMEM = STAGE_B / 4;                                      // floats of one stage
constexpr int PPW = 6;                                                 // DMA pieces per wave, chunk
static_assert(4 * SLAB <= SMEM, "the epilogue slabs alias a stage");
__device__ __forceinline__ int swz(int row, int q) { return q ^ ((row >> 1) & 7); }
__device__ __forceinline__ int nth_bit(int mask, int n) {   // n-th set bit of a 3-bit mask
  const int k0 = (mask & 1) ? 0 : ((mask & 2) ? 1 : 2);
  if (n == 0) return k0;
  const int rest = mask & ~(1 << k0);
  return (n == 1 && (rest & 2)) ? 1 : 2;
}
}  // namespace dgx

template <int NS>
__device__ __forceinline__ void dgrad_x6_tile(float *smemf, int bx, int by,
                                              const float *__restrict__ dpre,
                                              const float *__restrict__ wd,
                                              const float *__restrict__ gate,
                                              float *__restrict__ din, const ConvGeom &g,
                                              const DgradPlan &pl) {
  using namespace dgx;
  using scae_x6::Split3;
  unsigned char *smem = reinterpret_cast<unsigned char *>(smemf);
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int i = lane & 31, kk = lane >> 5, wm = wid >> 1, wn = wid & 1;
  const int nz = pl.nrc * pl.ncc;
  const int z = __popcll(__ballot(lane + 1 < nz && by >= pl.tile_start[min(lane + 1, 64)]));
  const int rc = z / pl.ncc, cc = z - rc * pl.ncc;
  const int AH = pl.rcount[rc], AW = pl.ccount[cc], M = g.B * AH * AW;
  const int m0 = (by - pl.tile_start[z]) * TM, n0 = bx * TN;
  const int rm = pl.rmask[rc], cm = pl.cmask[cc], nkh = __popc(rm), nkw = __popc(cm);
  const bool s2 = g.stride == 2;
  const pipe::rsrc_t ra = pipe::make_rsrc(dpre, (unsigned)((size_t)g.B * g.OH * g.OW * g.Cout * 4));
  const pipe::rsrc_t rb = pipe::make_rsrc(wd, (unsigned)((size_t)g.Cin * 9 * g.Cout * 4));
  // the class pair's LAST tap (largest kh, kw) is the lanes' reference position at stride 1,
  // where every tap of a class reaches every pixel of it
  const int khl = nth_bit(rm, nkh - 1), kwl = nth_bit(cm, nkw - 1);
  // DMA piece j of the A tile: rows 16 wid + 8 j .. + 7; of the B tile: rows 32 wid + 8 j .. + 7;
  // lane l moves the quad that lands in slot l & 7 of row .. + (l >> 3)
  int pn[2], pih[2], piw[2], va[2], vb[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 16 * wid + 8 * j + (lane >> 3), m = m0 + row;
    pn[j] = -1, pih[j] = 0, piw[j] = 0, va[j] = pipe::DMA_ZERO;
    if (m < M) {
      const int n = m / (AH * AW), rem = m - n * AH * AW, a = rem / AW, b = rem - a * AW;
      pn[j] = n * g.OH * g.OW;
      pih[j] = pl.rlist[pl.rstart[rc] + a], piw[j] = pl.clist[pl.cstart[cc] + b];
      if (!s2)
        va[j] = ((pn[j] + (pih[j] - khl) * g.OW + piw[j] - kwl) * g.Cout) * 4 +
                swz(row, lane & 7) * 16;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 32 * wid + 8 * j + (lane >> 3);
    vb[j] = ((n0 + row) * 9 * g.Cout) * 4 + swz(row, lane & 7) * 16;
  }
  const int cpt = g.Cout / BKF, nchunk = nkh * nkw * cpt;
  auto issue = [&](int c, unsigned char *stage) {
    const int t = c / cpt, h = c - t * cpt, ti = t / nkw, tj = t - ti * nkw;
    const int kh = nth_bit(rm, ti), kw = nth_bit(cm, tj);
    const int sb = ((kh * 3 + kw) * g.Cout + h * BKF) * 4;
    if (!s2) {
      const int sa = (((khl - kh) * g.OW + kwl - kw) * g.Cout + h * BKF) * 4;
#pragma unroll
      for (int j = 0; j < 2; ++j)
        pipe::dma16(ra, reinterpret_cast<float *>(stage + (16 * wid + 8 * j) * ROWB), va[j], sa);
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = 16 * wid + 8 * j + (lane >> 3);
        const int dh = pih[j] - kh, dw = piw[j] - kw, oh = dh >> 1, ow = dw >> 1;
        const bool ok = pn[j] >= 0 && dh >= 0 && dw >= 0 && oh < g.OH && ow < g.OW;
        const int v = ok ? ((pn[j] + oh * g.OW + ow) * g.Cout) * 4 + swz(row, lane & 7) * 16
                         : pipe::DMA_ZERO;
        pipe::dma16(ra, reinterpret_cast<float *>(stage + (16 * wid + 8 * j) * ROWB), v,
                    h * BKF * 4);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      pipe::dma16(rb, reinterpret_cast<float *>(stage + A_B + (32 * wid + 8 * j) * ROWB), vb[j],
                  sb);
  };
  pipe::f32x16 acc[2], accl[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[u][e] = 0.f, accl[u][e] = 0.f;
  const int rowa = wm * 32 + i, aoff = rowa * ROWB, asw = (rowa >> 1) & 7;
  int boff[2], bsw[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int rowb = wn * 64 + u * 32 + i;
    boff[u] = A_B + rowb * ROWB, bsw[u] = (rowb >> 1) & 7;
  }
  auto frag = [&](const unsigned char *stage, int off, int sw, int q) {   // quads q, q + 1 of a row
    return scae_x6::split3(
        pipe::lds4(reinterpret_cast<const float *>(stage + off + ((q ^ sw) << 4))),
        pipe::lds4(reinterpret_cast<const float *>(stage + off + (((q + 1) ^ sw) << 4))));
  };
  auto mma = [&](const unsigned char *stage) {
#pragma unroll
    for (int s = 0; s < BKF / 16; ++s) {
      const int q = 4 * s + 2 * kk;
      const Split3 a = frag(stage, aoff, asw, q);
      Split3 b[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) b[u] = frag(stage, boff[u], bsw[u], q);
#define SCAE_DGX_MMA(AP, BP, ACC)                                                            \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) ACC[u] =                                     \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.AP, b[u].BP, ACC[u], 0, 0, 0)
      SCAE_DGX_MMA(hi, lo, accl);
      SCAE_DGX_MMA(lo, hi, accl);
      SCAE_DGX_MMA(mid, mid, accl);
      SCAE_DGX_MMA(hi, mid, accl);
      SCAE_DGX_MMA(mid, hi, accl);
      SCAE_DGX_MMA(hi, hi, acc);
#undef SCAE_DGX_MMA
    }
  };
  // the K loop: NS stages, NS - 1 chunks in flight under the MFMAs of the current one
  if (NS == 1) {
    for (int c = 0; c < nchunk; ++c) {
      issue(c, smem);
      pipe::wait_vm<0>();
      pipe::wg_barrier();
      mma(smem);
      pipe::wg_barrier();   // everyone has read the stage: the next chunk may land
    }
  } else {
#pragma unroll
    for (int c = 0; c < NS - 1; ++c)
      if (c < nchunk) issue(c, smem + c * STAGE_B);
    int s = 0;   // stage of chunk c
    for (int c = 0; c < nchunk; ++c) {
      pipe::wait_chunk<PPW, NS>(min(NS - 2, nchunk - 1 - c));
      pipe::wg_barrier();   // chunk c has landed; everyone is done reading chunk c - 1's stage
      const int sp = s == 0 ? NS - 1 : s - 1;   // = the stage of chunk c + NS - 1
      if (c + NS - 1 < nchunk) issue(c + NS - 1, smem + sp * STAGE_B);
      mma(smem + s * STAGE_B);
      s = s + 1 == NS ? 0 : s + 1;
    }
  }
  // accumulators out: each wave's two 32 x 32 tiles pass through its [32][36] slab and leave as
  // rows of 8 consecutive channels per lane
  pipe::wg_barrier();   // the stages are dead: the slabs alias the first
  float *slab = smemf + wid * SLAB;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
#pragma unroll
    for (int e = 0; e < 16; ++e)
      slab[((e & 3) + 8 * (e >> 2) + 4 * kk) * 36 + i] = acc[u][e] + accl[u][e];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the wave's own slab)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int row = (lane >> 2) + 16 * pass, c8 = 8 * (lane & 3);
      float4 v0 = pipe::lds4(slab + row * 36 + c8), v1 = pipe::lds4(slab + row * 36 + c8 + 4);
      const int m = m0 + wm * 32 + row;
      if (m < M) {
        const int nb = m / (AH * AW), rem = m - nb * AH * AW, a = rem / AW, b = rem - a * AW;
        const int ih = pl.rlist[pl.rstart[rc] + a], iw = pl.clist[pl.cstart[cc] + b];
        const size_t o = (((size_t)nb * g.IH + ih) * g.IW + iw) * g.Cin + n0 + wn * 64 + u * 32 + c8;
        if (gate) {
          const float4 g0 = ld4(gate + o), g1 = ld4(gate + o + 4);
          v0.x = g0.x > 0.f ? v0.x : 0.f, v0.y = g0.y > 0.f ? v0.y : 0.f;
          v0.z = g0.z > 0.f ? v0.z : 0.f, v0.w = g0.w > 0.f ? v0.w : 0.f;
          v1.x = g1.x > 0.f ? v1.x : 0.f, v1.y = g1.y > 0.f ? v1.y : 0.f;
          v1.z = g1.z > 0.f ? v1.z : 0.f, v1.w = g1.w > 0.f ? v1.w : 0.f;
        }
        *reinterpret_cast<float4 *>(din + o) = v0;
        *reinterpret_cast<float4 *>(din + o + 4) = v1;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// ---- data-gradient tile, fourth form (DMODE 5): the K loop split over the waves ---------------
// For the layers with FEW tiles and LONG K loops (cfg-2's layers 3 / 4 at B = 128: 100 - 160 of the
// 64 x 128 tiles above, up to 36 chunks each -- a serial chain per workgroup).  32 input pixels of one
// tap-class pair x 64 channels per workgroup; every wave computes the WHOLE tile for a quarter of
// the chunks (wave w: chunks w, w + 4, ..), from a stage of its own (4 + 8 KiB, DMA as above), with
// no workgroup barrier in the loop: four independent DMA -> split -> MFMA pipelines per workgroup,
// chains a quarter as long.  The four partial tiles meet in LDS (each wave's slab aliases its own
// stage) and are summed in a fixed order, gated and stored by all 256 threads.
namespace dgk {
constexpr int TM = 32, TN = 64, BKF = 32, ROWB = 4 * BKF;
constexpr int A_B = TM * ROWB, B_B = TN * ROWB, WAVE_B = A_B + B_B;   // 4 + 8 KiB per wave
constexpr int LDS = 68;                                                // slab row stride (floats)
constexpr int SMEM = 4 * WAVE_B / 4;                                   // floats per workgroup
static_assert(TM * LDS * 4 <= WAVE_B, "a wave's slab aliases its stage");
static_assert(BKF == 32, "two 16-deep steps per chunk");

// One wave's pipeline over its chunks wid, wid + 4, ..: issue(c) = its 12 DMA pieces of chunk c
// into ITS stage (A rows 0 .. 31 then B rows 0 .. 63, 128-byte rows, quads swizzled dgx::swz),
// then the wave's partial 32 x 64 tile into its slab, a workgroup barrier, and the sum of the four
// slabs in a fixed order: thread tid returns row tid / 8, columns 8 (tid % 8) .. + 7.
struct Out8 {
  float4 lo, hi;
};
template <class Issue>
__device__ __forceinline__ Out8 pipeline(float *smemf, int nchunk, Issue issue) {
  using scae_x6::Split3;
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int i = lane & 31, kk = lane >> 5;
  unsigned char *stage = reinterpret_cast<unsigned char *>(smemf) + wid * WAVE_B;
  pipe::f32x16 acc[2], accl[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[u][e] = 0.f, accl[u][e] = 0.f;
  const int aoff = i * ROWB, asw = (i >> 1) & 7;
  int boff[2], bsw[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int rowb = u * 32 + i;
    boff[u] = A_B + rowb * ROWB, bsw[u] = (rowb >> 1) & 7;
  }
  struct Raw {
    float4 lo, hi;   // quads q, q + 1 of a row: the lane's 8 k
  };
  auto raw = [&](int off, int sw, int q) {
    Raw r;
    r.lo = pipe::lds4(reinterpret_cast<const float *>(stage + off + ((q ^ sw) << 4)));
    r.hi = pipe::lds4(reinterpret_cast<const float *>(stage + off + (((q + 1) ^ sw) << 4)));
    return r;
  };
  auto mma = [&](const Raw &ar, const Raw (&br)[2]) {   // one 16-deep step: split, six products
    const Split3 a = scae_x6::split3(ar.lo, ar.hi);
    Split3 b[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) b[u] = scae_x6::split3(br[u].lo, br[u].hi);
#define SCAE_DGK_MMA(AP, BP, ACC)                                                            \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) ACC[u] =                                     \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.AP, b[u].BP, ACC[u], 0, 0, 0)
    SCAE_DGK_MMA(hi, lo, accl);
    SCAE_DGK_MMA(lo, hi, accl);
    SCAE_DGK_MMA(mid, mid, accl);
    SCAE_DGK_MMA(hi, mid, accl);
    SCAE_DGK_MMA(mid, hi, accl);
    SCAE_DGK_MMA(hi, hi, acc);
#undef SCAE_DGK_MMA
  };
  if (wid < nchunk) issue(wid, stage);
  for (int c = wid; c < nchunk; c += 4) {   // (wave-uniform)
    pipe::wait_vm<0>();   // this wave's own pieces: nobody else writes or reads its stage
    Raw ar = raw(aoff, asw, 2 * kk), br[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) br[u] = raw(boff[u], bsw[u], 2 * kk);
    mma(ar, br);
    // the second step's fragments into registers: then the stage is free, and the next chunk's
    // DMA flies under this step's splits and MFMAs
    ar = raw(aoff, asw, 4 + 2 * kk);
#pragma unroll
    for (int u = 0; u < 2; ++u) br[u] = raw(boff[u], bsw[u], 4 + 2 * kk);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (c + 4 < nchunk) issue(c + 4, stage);
    mma(ar, br);
  }
  // the wave's partial tile into its slab [32][LDS] (over its own stage: its reads are done)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float *slab = reinterpret_cast<float *>(stage);
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int e = 0; e < 16; ++e)
      slab[((e & 3) + 8 * (e >> 2) + 4 * kk) * LDS + u * 32 + i] = acc[u][e] + accl[u][e];
  pipe::wg_barrier();
  const float *p0 = smemf + (tid >> 3) * LDS + 8 * (tid & 7);
  constexpr int WS = WAVE_B / 4;
  const float4 a0 = pipe::lds4(p0), a1 = pipe::lds4(p0 + 4);
  const float4 b0 = pipe::lds4(p0 + WS), b1 = pipe::lds4(p0 + WS + 4);
  const float4 c0 = pipe::lds4(p0 + 2 * WS), c1 = pipe::lds4(p0 + 2 * WS + 4);
  const float4 d0 = pipe::lds4(p0 + 3 * WS), d1 = pipe::lds4(p0 + 3 * WS + 4);
  Out8 o;
  o.lo = make_float4((a0.x + b0.x) + (c0.x + d0.x), (a0.y + b0.y) + (c0.y + d0.y),
                     (a0.z + b0.z) + (c0.z + d0.z), (a0.w + b0.w) + (c0.w + d0.w));
  o.hi = make_float4((a1.x + b1.x) + (c1.x + d1.x), (a1.y + b1.y) + (c1.y + d1.y),
                     (a1.z + b1.z) + (c1.z + d1.z), (a1.w + b1.w) + (c1.w + d1.w));
  return o;
}
}  // namespace dgk

__device__ __forceinline__ void dgrad_x6k_tile(float *smemf, int bx, int by,
                                               const float *__restrict__ dpre,
                                               const float *__restrict__ wd,
                                               const float *__restrict__ gate,
                                               float *__restrict__ din, const ConvGeom &g,
                                               const DgradPlan &pl) {
  using namespace dgk;
  using dgx::nth_bit;
  using dgx::swz;
  const int tid = threadIdx.x, lane = tid & 63;
  const int nz = pl.nrc * pl.ncc;
  const int z = __popcll(__ballot(lane + 1 < nz && by >= pl.tile_start[min(lane + 1, 64)]));
  const int rc = z / pl.ncc, cc = z - rc * pl.ncc;
  const int AH = pl.rcount[rc], AW = pl.ccount[cc], M = g.B * AH * AW;
  const int m0 = (by - pl.tile_start[z]) * TM, n0 = bx * TN;
  const int rm = pl.rmask[rc], cm = pl.cmask[cc], nkh = __popc(rm), nkw = __popc(cm);
  const bool s2 = g.stride == 2;
  const pipe::rsrc_t ra = pipe::make_rsrc(dpre, (unsigned)((size_t)g.B * g.OH * g.OW * g.Cout * 4));
  const pipe::rsrc_t rb = pipe::make_rsrc(wd, (unsigned)((size_t)g.Cin * 9 * g.Cout * 4));
  const int khl = nth_bit(rm, nkh - 1), kwl = nth_bit(cm, nkw - 1);
  // every wave moves the whole A tile (4 pieces of 8 rows) and the whole B tile (8 pieces)
  int pn[4], pih[4], piw[4], va[4], vb[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 8 * j + (lane >> 3), m = m0 + row;
    pn[j] = -1, pih[j] = 0, piw[j] = 0, va[j] = pipe::DMA_ZERO;
    if (m < M) {
      const int n = m / (AH * AW), rem = m - n * AH * AW, a = rem / AW, b = rem - a * AW;
      pn[j] = n * g.OH * g.OW;
      pih[j] = pl.rlist[pl.rstart[rc] + a], piw[j] = pl.clist[pl.cstart[cc] + b];
      if (!s2)
        va[j] = ((pn[j] + (pih[j] - khl) * g.OW + piw[j] - kwl) * g.Cout) * 4 +
                swz(row, lane & 7) * 16;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = 8 * j + (lane >> 3);
    vb[j] = ((n0 + row) * 9 * g.Cout) * 4 + swz(row, lane & 7) * 16;
  }
  const int cpt = g.Cout / BKF;
  Out8 v = pipeline(smemf, nkh * nkw * cpt, [&](int c, unsigned char *stage) {
    const int t = c / cpt, h = c - t * cpt, ti = t / nkw, tj = t - ti * nkw;
    const int kh = nth_bit(rm, ti), kw = nth_bit(cm, tj);
    const int sb = ((kh * 3 + kw) * g.Cout + h * BKF) * 4;
    if (!s2) {
      const int sa = (((khl - kh) * g.OW + kwl - kw) * g.Cout + h * BKF) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        pipe::dma16(ra, reinterpret_cast<float *>(stage + 8 * j * ROWB), va[j], sa);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 8 * j + (lane >> 3);
        const int dh = pih[j] - kh, dw = piw[j] - kw, oh = dh >> 1, ow = dw >> 1;
        const bool ok = pn[j] >= 0 && dh >= 0 && dw >= 0 && oh < g.OH && ow < g.OW;
        const int vv = ok ? ((pn[j] + oh * g.OW + ow) * g.Cout) * 4 + swz(row, lane & 7) * 16
                          : pipe::DMA_ZERO;
        pipe::dma16(ra, reinterpret_cast<float *>(stage + 8 * j * ROWB), vv, h * BKF * 4);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      pipe::dma16(rb, reinterpret_cast<float *>(stage + A_B + 8 * j * ROWB), vb[j], sb);
  });
  const int row = tid >> 3, c8 = 8 * (tid & 7), m = m0 + row;
  if (m >= M) return;
  const int nb = m / (AH * AW), rem = m - nb * AH * AW, a = rem / AW, b = rem - a * AW;
  const int ih = pl.rlist[pl.rstart[rc] + a], iw = pl.clist[pl.cstart[cc] + b];
  const size_t o = (((size_t)nb * g.IH + ih) * g.IW + iw) * g.Cin + n0 + c8;
  if (gate) {
    const float4 g0 = ld4(gate + o), g1 = ld4(gate + o + 4);
    v.lo.x = g0.x > 0.f ? v.lo.x : 0.f, v.lo.y = g0.y > 0.f ? v.lo.y : 0.f;
    v.lo.z = g0.z > 0.f ? v.lo.z : 0.f, v.lo.w = g0.w > 0.f ? v.lo.w : 0.f;
    v.hi.x = g1.x > 0.f ? v.hi.x : 0.f, v.hi.y = g1.y > 0.f ? v.hi.y : 0.f;
    v.hi.z = g1.z > 0.f ? v.hi.z : 0.f, v.hi.w = g1.w > 0.f ? v.hi.w : 0.f;
  }
  *reinterpret_cast<float4 *>(din + o) = v.lo;
  *reinterpret_cast<float4 *>(din + o + 4) = v.hi;
}

// The forward of a layer in the same form: 32 output pixels x 64 output channels, the 9 x C_in / 32
// chunks dealt to the four waves.  A = input rows (a lane's source row is fixed, the tap moves the
// wave-uniform offset), B = wf (C_out, 9, C_in).
__device__ __forceinline__ void fwd_x6k_tile(float *smemf, int bx, int by,
                                             const float *__restrict__ in,
                                             const float *__restrict__ wf,
                                             const float *__restrict__ bias,
                                             float *__restrict__ out,
                                             const float *__restrict__ post_bias,
                                             float *__restrict__ out_post, const ConvGeom &g) {
  using namespace dgk;
  using dgx::swz;
  const int tid = threadIdx.x, lane = tid & 63;
  const int M = g.B * g.OH * g.OW, hw = g.OH * g.OW;
  const int m0 = by * TM, n0 = bx * TN;
  const pipe::rsrc_t ra = pipe::make_rsrc(in, (unsigned)((size_t)g.B * g.IH * g.IW * g.Cin * 4));
  const pipe::rsrc_t rb = pipe::make_rsrc(wf, (unsigned)((size_t)g.Cout * 9 * g.Cin * 4));
  int va[4], vb[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 8 * j + (lane >> 3), m = m0 + row;
    va[j] = pipe::DMA_ZERO;
    if (m < M) {
      const int n = m / hw, rem = m - n * hw, oh = rem / g.OW, ow = rem - oh * g.OW;
      va[j] = (((n * g.IH + oh * g.stride) * g.IW + ow * g.stride) * g.Cin) * 4 +
              swz(row, lane & 7) * 16;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = 8 * j + (lane >> 3);
    vb[j] = ((n0 + row) * 9 * g.Cin) * 4 + swz(row, lane & 7) * 16;
  }
  const int cpt = g.Cin / BKF;
  const Out8 v = pipeline(smemf, 9 * cpt, [&](int c, unsigned char *stage) {
    const int tap = c / cpt, h = c - tap * cpt, kh = tap / 3, kw = tap - kh * 3;
    const int sa = ((kh * g.IW + kw) * g.Cin + h * BKF) * 4, sb = (tap * g.Cin + h * BKF) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      pipe::dma16(ra, reinterpret_cast<float *>(stage + 8 * j * ROWB), va[j], sa);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      pipe::dma16(rb, reinterpret_cast<float *>(stage + A_B + 8 * j * ROWB), vb[j], sb);
  });
  const int row = tid >> 3, c8 = 8 * (tid & 7), m = m0 + row, n = n0 + c8;
  if (m >= M) return;
  const float4 b0 = ld4(bias + n), b1 = ld4(bias + n + 4);
  const float o[8] = {fmaxf(v.lo.x + b0.x, 0.f), fmaxf(v.lo.y + b0.y, 0.f), fmaxf(v.lo.z + b0.z, 0.f),
                      fmaxf(v.lo.w + b0.w, 0.f), fmaxf(v.hi.x + b1.x, 0.f), fmaxf(v.hi.y + b1.y, 0.f),
                      fmaxf(v.hi.z + b1.z, 0.f), fmaxf(v.hi.w + b1.w, 0.f)};
  const size_t at = (size_t)m * g.Cout + n;
  *reinterpret_cast<float4 *>(out + at) = make_float4(o[0], o[1], o[2], o[3]);
  *reinterpret_cast<float4 *>(out + at + 4) = make_float4(o[4], o[5], o[6], o[7]);
  if (out_post) {   // + the per-(channel, pixel) embedding bias, (Cout, OH, OW)
    const float *pb = post_bias + (size_t)n * hw + m % hw;
    float q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) q[e] = o[e] + pb[(size_t)e * hw];
    *reinterpret_cast<float4 *>(out_post + at) = make_float4(q[0], q[1], q[2], q[3]);
    *reinterpret_cast<float4 *>(out_post + at + 4) = make_float4(q[4], q[5], q[6], q[7]);
  }
}
__global__ __launch_bounds__(NT) void conv_fwd_x6k_kernel(
    const float *__restrict__ in, const float *__restrict__ wf, const float *__restrict__ bias,
    float *__restrict__ out, const float *__restrict__ post_bias, float *__restrict__ out_post,
    ConvGeom g) {
  __shared__ __attribute__((aligned(1024))) float smem[dgk::SMEM];
  fwd_x6k_tile(smem, blockIdx.x, blockIdx.y, in, wf, bias, out, post_bias, out_post, g);
}
// ... with the folding products of the output attention as the tail of its grid
// (conv_fwd_pipe_fold_kernel's riders).  Beside the ring tiles (four workgroups per CU: tiles
// and riders all resident at once) the riders cost 0.8 us, beside these (three per CU) 8 - 10 us:
// worth it for the large layers only (fwd_ksplit).
__global__ __launch_bounds__(NT) void conv_fwd_x6k_fold_kernel(
    const float *__restrict__ in, const float *__restrict__ wf, const float *__restrict__ bias,
    float *__restrict__ out, const float *__restrict__ post_bias, float *__restrict__ out_post,
    ConvGeom g, int gx, int n_conv, scae_seed_fold_desc fold, scae_fold::Plan plan) {
  static_assert(dgk::SMEM >= FOLD_SMEM, "the riders' LDS");
  __shared__ __attribute__((aligned(1024))) float smem[dgk::SMEM];
  const int blk = blockIdx.x;
  if (blk < n_conv)   // workgroup-uniform
    fwd_x6k_tile(smem, blk % gx, blk / gx, in, wf, bias, out, post_bias, out_post, g);
  else
    scae_fold::forward_block_any<16>(fold, plan, blk - n_conv, smem);
}

// data-gradient tile by DMODE: 0 - 2 the first-generation shapes of mfma_tile.h, 4 / 5 the forms above
template <int DMODE>
struct DgradShape {
  static constexpr int SMEM = Tile<DMODE>::SMEM;
};
template <>
struct DgradShape<4> {
  static constexpr int SMEM = dgx::SMEM;
};
template <>
struct DgradShape<5> {
  static constexpr int SMEM = dgk::SMEM;
};
template <int DMODE, int SM>
__device__ __forceinline__ void dgrad_any_tile(float *smem, int bx, int by,
                                               const float *__restrict__ dpre,
                                               const float *__restrict__ wd,
                                               const float *__restrict__ gate,
                                               float *__restrict__ din, const ConvGeom &g,
                                               const DgradPlan &pl) {
  if constexpr (DMODE == 4)
    dgrad_x6_tile<(SM >= 2 * dgx::SMEM ? 2 : 1)>(smem, bx, by, dpre, wd, gate, din, g, pl);
  else if constexpr (DMODE == 5)
    dgrad_x6k_tile(smem, bx, by, dpre, wd, gate, din, g, pl);
  else
    dgrad_tile<DMODE>(smem, bx, by, dpre, wd, gate, din, g, pl);
}

template <int NS>
__global__ __launch_bounds__(NT) void conv_dgrad_x6_kernel(const float *__restrict__ dpre,
                                                           const float *__restrict__ wd,
                                                           const float *__restrict__ gate,
                                                           float *__restrict__ din, ConvGeom g,
                                                           DgradPlan pl) {
  __shared__ __attribute__((aligned(1024))) float smem[NS * dgx::SMEM];
  dgrad_x6_tile<NS>(smem, blockIdx.x, blockIdx.y, dpre, wd, gate, din, g, pl);
}

__global__ __launch_bounds__(NT) void conv_dgrad_x6k_kernel(const float *__restrict__ dpre,
                                                            const float *__restrict__ wd,
                                                            const float *__restrict__ gate,
                                                            float *__restrict__ din, ConvGeom g,
                                                            DgradPlan pl) {
  __shared__ __attribute__((aligned(1024))) float smem[dgk::SMEM];
  dgrad_x6k_tile(smem, blockIdx.x, blockIdx.y, dpre, wd, gate, din, g, pl);
}

#ifdef SCAE_CONV_PROF   // start / end stamp (s_memrealtime, 100 MHz) of every workgroup of the
// mixed backward pairs, a slot per DMODE (tools/conv_prof.py)
__device__ unsigned long long g_conv_prof[6][4096][2];
#define CV_STAMP(mode, i)                                                       \
  do {                                                                          \
    if (threadIdx.x == 0 && blockIdx.x < 4096)                                  \
      g_conv_prof[mode][blockIdx.x][i] = __builtin_amdgcn_s_memrealtime();      \
  } while (0)
#else
#define CV_STAMP(mode, i)
#endif

// first-generation data-gradient tiles (many small workgroups per CU suit the short K
// loops of the tap classes) beside second-generation weight-gradient tiles
template <int DMODE, class TW>
__global__ __launch_bounds__(NT) void conv_bwd_pair_mixed_kernel(
    const float *__restrict__ dpre, const float *__restrict__ wd, const float *__restrict__ gate,
    float *__restrict__ din, const float *__restrict__ in, float *__restrict__ partial,
    ConvGeom g, DgradPlan pl, int splits, PairGrid pg) {
  constexpr int SM = DgradShape<DMODE>::SMEM > TW::SMEM ? DgradShape<DMODE>::SMEM : TW::SMEM;
  __shared__ __attribute__((aligned(1024))) float smem[SM];
  const int bid = blockIdx.x;
  CV_STAMP(DMODE, 0);
  if (bid < pg.nd) {   // workgroup-uniform
    dgrad_any_tile<DMODE, SM>(smem, bid % pg.gx, bid / pg.gx, dpre, wd, gate, din, g, pl);
  } else {
    const int w = bid - pg.nd, bx = w % pg.wx, t = w / pg.wx;
    wgrad_pipe_tile<TW>(smem, bx, t % pg.wy, t / pg.wy, dpre, in, partial, g, splits);
  }
  CV_STAMP(DMODE, 1);
}

// The same launch with a rider as the HEAD of its grid: a kernel of the object encoder's
// backward that writes parameter gradients only -- the output attention's partial-row
// reduction (seed_attention_wave.hip, saw_reduce_body) or the backward of its folding
// products (seed_fold.hip, fold_bwd_body; it reads what the reduction wrote, so the two ride
// in consecutive launches).  In a training step those launches wait (ops._PendingReduce /
// _PendingFoldBackward) for the part encoder's convolution backward, the next launches with
// 256-thread workgroups and time to spare: ~100 / 320 short latency-bound workgroups
// (canonical row parts / K slices walked four per thread / wave: the same bits as their
// 1024-thread launches) beside 48 / 75 us of MFMA tiles, instead of 8.5 + 9 us of their own on
// the step's dependent chain.
struct PairRider {
  int n;      // rider workgroups (blocks [0, n))
  int kind;   // 1: reduce, 2: fold
  scae_saw::ReduceArgs red;
  scae_seed_fold_desc a;
  scae_seed_fold_grads g;
  scae_sf::BwdPlan pl;
};
template <int DMODE, class TW>
__global__ __launch_bounds__(NT) void conv_bwd_pair_mixed_rider_kernel(
    const float *__restrict__ dpre, const float *__restrict__ wd, const float *__restrict__ gate,
    float *__restrict__ din, const float *__restrict__ in, float *__restrict__ partial,
    ConvGeom g, DgradPlan pl, int splits, PairGrid pg, PairRider r) {
  constexpr int SM = DgradShape<DMODE>::SMEM > TW::SMEM ? DgradShape<DMODE>::SMEM : TW::SMEM;
  static_assert(SM * sizeof(float) >= 24 * 1024, "the riders' LDS");
  __shared__ __attribute__((aligned(1024))) float smem[SM];
  CV_STAMP(DMODE, 0);
  if ((int)blockIdx.x < r.n) {   // workgroup-uniform
    if (r.kind == 1)
      scae_saw::saw_reduce_body<NT>(r.red, smem, blockIdx.x);
    else   // C = 256 threads, one part
      scae_sf::fold_bwd_body<4, 4>(r.a, r.g, r.pl, smem, blockIdx.x, threadIdx.x, 0, 1);
    CV_STAMP(DMODE, 1);
    return;
  }
  const int bid = (int)blockIdx.x - r.n;
  if (bid < pg.nd) {
    dgrad_any_tile<DMODE, SM>(smem, bid % pg.gx, bid / pg.gx, dpre, wd, gate, din, g, pl);
  } else {
    const int w = bid - pg.nd, bx = w % pg.wx, t = w / pg.wx;
    wgrad_pipe_tile<TW>(smem, bx, t % pg.wy, t / pg.wy, dpre, in, partial, g, splits);
  }
  CV_STAMP(DMODE, 1);
#ifdef SCAE_CONV_PROF
  if (blockIdx.x == 0 && threadIdx.x == 1) {   // (grid layout for the reader)
    g_conv_prof[DMODE][4095][0] = ((unsigned long long)r.n << 32) | (unsigned)pg.nd;
    g_conv_prof[DMODE][4095][1] = gridDim.x;
  }
#endif
}
#ifdef SCAE_CONV_PROF
}  // namespace
extern "C" int scae_debug_conv_prof(unsigned long long *out) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_prof), sizeof(g_conv_prof));
}
namespace {
#endif

// ---- small helpers ---------------------------------------------------------------
// W[co][ci][3][3] -> Wf[co][tap][ci] (+ its fragment-major copy), Wd[ci][tap][co]
__global__ void relayout_weights_kernel(const float *__restrict__ w, float *__restrict__ wf,
                                        float *__restrict__ wd, int Cout, int Cin) {
  scae_first::relayout_one(w, wf, wd, Cout, Cin, blockIdx.x * blockDim.x + threadIdx.x);
}

// (RelayoutBatch / relayout_batch: conv_first_dev.h)
using scae_first::RelayoutBatch;
using scae_first::relayout_batch;
__global__ void relayout_batch_kernel(RelayoutBatch r) {
  relayout_batch(r, blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x);
}

// dW[co][ci][tap] = sum_split partial[(split*9+tap)][co][ci]; db[co] = sum_split bias partials
__device__ __forceinline__ void reduce_wgrad(const float *__restrict__ partial,
                                             float *__restrict__ dw, float *__restrict__ db,
                                             int Cout, int Cin, int splits, int e) {
  const int n = 9 * Cout * Cin;  // e over (tap, co, ci): coalesced reads
  if (e < n) {
    const int tap = e / (Cout * Cin), rem = e - tap * Cout * Cin, co = rem / Cin,
              ci = rem - co * Cin;
    // (the splits' loads are independent: eight in flight instead of a chain of `splits`
    // L2 / MALL round trips; summed in split order all the same)
    const float *src = partial + ((size_t)tap * Cout + co) * Cin + ci;
    const size_t step = (size_t)9 * Cout * Cin;
    float acc = 0.f;
    int s = 0;
    for (; s + 8 <= splits; s += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(s + u) * step];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; s < splits; ++s) acc += src[(size_t)s * step];
    dw[((size_t)co * Cin + ci) * 9 + tap] = acc;
  } else if (e < n + Cout && db) {
    float acc = 0.f;
    for (int s = 0; s < splits; ++s) acc += partial[(size_t)splits * n + (size_t)s * Cout + e - n];
    db[e - n] = acc;
  }
}
__global__ void reduce_wgrad_kernel(const float *__restrict__ partial, float *__restrict__ dw,
                                    float *__restrict__ db, int Cout, int Cin, int splits) {
  reduce_wgrad(partial, dw, db, Cout, Cin, splits, blockIdx.x * blockDim.x + threadIdx.x);
}
// ... for up to 8 layers in one launch (blockIdx.y = layer)
struct ReduceBatch {
  const float *partial[8];
  float *dw[8], *db[8];
  int Cout[8], Cin[8], splits[8];
};
__global__ void reduce_wgrad_batch_kernel(ReduceBatch r) {
  const int l = blockIdx.y;
  reduce_wgrad(r.partial[l], r.dw[l], r.db[l], r.Cout[l], r.Cin[l], r.splits[l],
               blockIdx.x * blockDim.x + threadIdx.x);
}

// ---- first layer (image, C_in <= 4): direct kernels ----------------------------
// One workgroup per (image, pixel slice): the image is staged in LDS, each wave
// owns 64 output channels (lane = channel: NHWC stores / loads are 256-byte
// coalesced rows) and, when C_out < 256, a share of the slice's pixels.
using scae_first::FirstSplit;
using scae_first::first_split;
using scae_first::first_vec;
using scae_first::stage_image;

// image NCHW (B,Cin,IH,IW), w [Cout][Cin][3][3] -> out NHWC, ReLU
// Riders: the filter re-layouts of the following layers (RelayoutBatch, parameter-only,
// independent of this layer) run as extra workgroups of the same launch: blocks
// [n_first, n_first + n_layers * rb) with rb workgroups of 256 elements per layer.
template <int CIN>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(const float *__restrict__ img,
                                                             const float *__restrict__ w,
                                                             const float *__restrict__ bias,
                                                             float *__restrict__ out,
                                                             ConvGeom g, RelayoutBatch rl,
                                                             int n_first, int rb,
                                                             unsigned short *__restrict__ out_h) {
  extern __shared__ float s_img[];
  if ((int)blockIdx.x >= n_first) {  // workgroup-uniform
    const int w_ = (int)blockIdx.x - n_first;
    relayout_batch(rl, w_ / rb, (w_ % rb) * 256 + threadIdx.x);
    return;
  }
  scae_first::fwd_block<CIN>(img, w, bias, out, g, blockIdx.x, s_img, out_h);
}

// weight/bias gradient partials: partial[(n*slices + slice)*parts + part] = one row
// [dW (Cout, CIN*9) | db (Cout)]; the caller sums the rows.
// Riders: the split reductions of the other layers' weight-gradient partials
// (ReduceBatch; they wait for the same kernels as this one) as extra workgroups.
template <int CIN>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(const float *__restrict__ dpre,
                                                               const float *__restrict__ img,
                                                               float *__restrict__ partial,
                                                               ConvGeom g, ReduceBatch rd,
                                                               int n_first, int rb) {
  extern __shared__ float s_img[];
  if ((int)blockIdx.x >= n_first) {  // workgroup-uniform
    const int w_ = (int)blockIdx.x - n_first, l = w_ / rb;
    reduce_wgrad(rd.partial[l], rd.dw[l], rd.db[l], rd.Cout[l], rd.Cin[l], rd.splits[l],
                 (w_ % rb) * 256 + threadIdx.x);
    return;
  }
  constexpr int K1 = CIN * 9 + 1, NLD = 4;
  const FirstSplit f = first_split(g.B, g.Cout);
  const int n = blockIdx.x / f.slices, slice = blockIdx.x % f.slices;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int img_floats = CIN * g.IH * g.IW;
  float *s_red = s_img + img_floats;  // [parts][Cout][K1] cross-wave sums (parts > 1)
  stage_image(s_img, img, n, img_floats);
  const int P = g.OH * g.OW, per = (P + f.slices - 1) / f.slices;
  const int pbeg = slice * per, pend = min(P, pbeg + per);
  // row layout [dW (Cout x CIN*9) | db (Cout)]
  float *row = partial + (size_t)blockIdx.x * g.Cout * K1;
  for (int wi = wave; wi < f.nchunk * f.parts; wi += 4) {
    const int co = (wi % f.nchunk) * 64 + lane, part = wi / f.nchunk;
    float acc[K1];
#pragma unroll
    for (int k = 0; k < K1; ++k) acc[k] = 0.f;
    int oh = (pbeg + part) / g.OW, ow = (pbeg + part) - oh * g.OW;
    const size_t dstep = (size_t)f.parts * g.Cout;
    const float *dp = dpre + ((size_t)n * P + pbeg + part) * g.Cout + co;
    for (int p = pbeg + part; p < pend; p += NLD * f.parts) {
      float d[NLD];  // gradient loads in flight (the loop is a chain of L2 round trips)
#pragma unroll
      for (int u = 0; u < NLD; ++u) d[u] = p + u * f.parts < pend ? dp[u * dstep] : 0.f;
      dp += NLD * dstep;
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        const float *src = s_img + oh * g.stride * g.IW + ow * g.stride;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
          for (int t = 0; t < 9; ++t)
            acc[ci * 9 + t] =
                fmaf(d[u], src[(ci * g.IH + t / 3) * g.IW + t % 3], acc[ci * 9 + t]);
        acc[K1 - 1] += d[u];
        ow += f.parts;  // past the slice the products are with d = 0; keep src inside the image
        while (ow >= g.OW) ow -= g.OW, oh = oh + 1 < g.OH ? oh + 1 : oh;
      }
    }
    if (f.parts == 1) {
#pragma unroll
      for (int k = 0; k < K1 - 1; ++k) row[(size_t)co * (K1 - 1) + k] = acc[k];
      row[(size_t)g.Cout * (K1 - 1) + co] = acc[K1 - 1];
    } else {
#pragma unroll
      for (int k = 0; k < K1; ++k) s_red[((size_t)part * g.Cout + co) * K1 + k] = acc[k];
    }
  }
  if (f.parts > 1) {  // the waves that shared a channel chunk meet here
    __syncthreads();
    for (int e = threadIdx.x; e < g.Cout * K1; e += 256) {
      float t = 0.f;
      for (int part = 0; part < f.parts; ++part) t += s_red[(size_t)part * g.Cout * K1 + e];
      const int co = e / K1, k = e - co * K1;
      row[k < K1 - 1 ? (size_t)co * (K1 - 1) + k : (size_t)g.Cout * (K1 - 1) + co] = t;
    }
  }
}

int check_geom(const ConvGeom &g, bool gemm) {
  if (g.B <= 0 || g.IH <= 0 || g.IW <= 0 || g.Cin <= 0 || g.Cout <= 0 || g.stride <= 0)
    return SCAE_ERR_BAD_ARG;
  if (g.OH != (g.IH - 3) / g.stride + 1 || g.OW != (g.IW - 3) / g.stride + 1 || g.OH <= 0 ||
      g.OW <= 0)
    return SCAE_ERR_BAD_ARG;
  if (gemm && (g.Cin % 64 || g.Cout % 64 || g.stride > 2)) return SCAE_ERR_UNSUPPORTED;
  return SCAE_OK;
}

// 64x64 tiles once they fill the chip a few times over; below that split-K tiles:
// 32x64 while those still give enough workgroups (measured on the 128-channel
// encoder layers at B=128: the forward gains 12-14 % down to ~390 wide tiles, the
// data gradient -- shorter K per class, costlier prologue -- only well above 512),
// else 32x32
#ifndef SCAE_SMALL_TILES
#define SCAE_SMALL_TILES 1024
#endif
inline bool small_tiles(long tiles64) { return tiles64 < SCAE_SMALL_TILES; }
inline int tile_mode(long tiles64, long tiles_wide, long wide_min) {
  return !small_tiles(tiles64) ? 0 : (tiles_wide >= wide_min ? 2 : 1);
}

struct WgradPlan {
  bool small;
  int splits;
};
WgradPlan wgrad_plan(int M, int Cin, int Cout) {
  // 64x64 tiles (half the L2 traffic per flop of the 32x32 shape); the grid is
  // filled by splitting the pixel (K) dimension instead: >= 768 workgroups of
  // >= 8 K chunks each, measured best on the encoder's 128-channel layers
  WgradPlan p;
  const long tiles64 = (long)(Cin / 64) * (Cout / 64) * 9;
  p.small = false;
#ifndef SCAE_WGRAD_BLOCKS
#define SCAE_WGRAD_BLOCKS 768
#endif
  long s = (SCAE_WGRAD_BLOCKS + tiles64 - 1) / tiles64;
  const long cap = (M / BK) / 8;
  s = s > cap ? cap : s;
  p.splits = (int)(s < 1 ? 1 : (s > 32 ? 32 : s));
  return p;
}

// second-generation tile shapes (mfma_pipe.h).  Ring depth: 3 stages; 4 / 5 / 6 measured at
// B = 128 (-DSCAE_PIPE_NS=..): forward 82.0 -> 84.3 / 90.8 / 92.1 us, backward pairs 156.6 ->
// 175.6 / 176.7 / 238 -- the small layers are bound by the dependent MFMA / LDS-read chain of
// one wave per SIMD, not by DMA latency, and a deeper ring only costs co-resident workgroups.
#ifndef SCAE_PIPE_NS
#define SCAE_PIPE_NS 3
#endif
#ifndef SCAE_PIPE_WBK
#define SCAE_PIPE_WBK 32
#endif
using PipeC0 = pipe::KK<64, 128, 4, SCAE_PIPE_NS>;   // 4 waves x (64 x 32); 24 KiB / stage
using PipeC1 = pipe::KK<32, 128, 4, SCAE_PIPE_NS>;   // 4 waves x (32 x 32); 20 KiB / stage
using PipeC2 = pipe::KK<32, 64, 2, SCAE_PIPE_NS>;    // 2 x 2 (columns x k halves); 12 KiB
using PipeC3 = pipe::KK<64, 64, 2, SCAE_PIPE_NS>;    // 2 x 2, 64 rows; 16 KiB
// weight gradient: 2 x 2 waves x (32 x 32)
using PipeW = pipe::SS<64, 64, SCAE_PIPE_WBK, SCAE_PIPE_NS>;
// ... with 16-pixel chunks: a 24 KiB ring instead of 48, so that the pair launches of the small
// layers hold more than three workgroups per CU (their data-gradient tiles need 18 KiB and
// 60-114 registers).  B = 128: layer 4 31.7 -> 28.9 us, layer 3 47.4 -> 46.1, layer 2 +0.4;
// at B = 1024 the 32-pixel chunks win by 7 % (tools/conv_multi_probe.py, round 5).
using PipeW16 = pipe::SS<64, 64, 16, SCAE_PIPE_NS>;
#ifndef SCAE_WGRAD_SHORT_CHUNK_PIXELS
#define SCAE_WGRAD_SHORT_CHUNK_PIXELS 8192   // layers with fewer output pixels take PipeW16
#endif

// shape for an (M rows) x (N columns) k-contiguous problem; an environment variable
// (read per call; tuning aid) overrides: -1 = first-generation kernels
inline int pipe_cfg(const char *env, long M, int N) {
  const char *e = getenv(env);
  if (e && *e) {
    const int v = atoi(e);
    if (v < 0) return -1;
    if (N % 128 == 0 || v >= 2) return v > 3 ? 3 : v;
  }
  // measured on the encoder's 128-channel layers at B = 128 and B = 1024: the 32 x 64
  // shape (4 workgroups per CU, in-workgroup k split) wins or ties everywhere -- with
  // few tiles because it quantises best over 256 CUs, with many because four
  // workgroups per CU cover each other's barrier and DMA waits
  (void)M;
  return 2;
}
}  // namespace

extern "C" int64_t scae_conv3x3_wf_floats(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0) return 0;
  const int64_t n = (int64_t)Cout * 9 * Cin;
  return scae_first::packed_copy(Cout, Cin) ? n + (3 * n + 1) / 2 : n;   // + three bf16 planes
}

extern "C" int scae_conv3x3_relayout_f32(const float *w, float *wf, float *wd, int Cout,
                                         int Cin, void *stream) {
  SCAE_REQUIRE(w && wf && wd && Cout > 0 && Cin > 0);
  const int n = Cout * Cin * 9;
  scae::launch(relayout_weights_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, w, wf, wd, Cout, Cin);
  return scae_launch_status();
}

using scae_first::fill_relayout;

extern "C" int scae_conv3x3_relayout_batch_f32(int n_layers, const float *const *w,
                                               float *const *wf, float *const *wd,
                                               const int *Cout, const int *Cin, void *stream) {
  RelayoutBatch r{};
  const int rb = fill_relayout(r, n_layers, w, wf, wd, Cout, Cin);
  SCAE_REQUIRE(rb > 0);
  scae::launch(relayout_batch_kernel, dim3(rb, n_layers), dim3(256), 0,
                     (hipStream_t)stream, r);
  return scae_launch_status();
}

static int first_fwd_relayout(
    const float *img, const float *w, const float *bias, float *out, unsigned short *out_h, int B,
    int Cin, int IH, int IW, int Cout, int stride, int n_layers, const float *const *rw,
    float *const *rwf, float *const *rwd, unsigned short *const *rwfh,
    unsigned short *const *rwdh, const int *rCout, const int *rCin, void *stream) {
  ConvGeom g{B, IH, IW, (IH - 3) / stride + 1, (IW - 3) / stride + 1, Cin, Cout, stride};
  int rc = check_geom(g, false);
  if (rc) return rc;
  SCAE_REQUIRE(img && w && bias && (out || out_h) && n_layers >= 0);
  if (!out) out = reinterpret_cast<float *>(out_h);   // (only its address arithmetic is used)
  if (Cout % 64) return SCAE_ERR_UNSUPPORTED;
  const FirstSplit f = first_split(B, Cout);
  const size_t lds = (size_t)Cin * IH * IW * sizeof(float);
  if (lds > 64 * 1024) return SCAE_ERR_UNSUPPORTED;
  RelayoutBatch r{};
  int rb = 0;
  if (n_layers > 0) {
    rb = fill_relayout(r, n_layers, rw, rwf, rwd, rCout, rCin, rwfh, rwdh);
    SCAE_REQUIRE(rb > 0);
  }
  const int n_first = B * f.slices;
#define SCAE_FIRST_FWD(CI)                                                                    \
  case CI:                                                                                    \
    scae::launch(conv_first_fwd_kernel<CI>, dim3(n_first + n_layers * rb), dim3(256),   \
                       lds, (hipStream_t)stream, img, w, bias, out, g, r, n_first, rb, out_h); \
    break;
  switch (Cin) {
    SCAE_FIRST_FWD(1) SCAE_FIRST_FWD(2) SCAE_FIRST_FWD(3) SCAE_FIRST_FWD(4)
    default: return SCAE_ERR_UNSUPPORTED;
  }
#undef SCAE_FIRST_FWD
  return scae_launch_status();
}

extern "C" int scae_conv3x3_first_fwd_relayout_f32(
    const float *img, const float *w, const float *bias, float *out, int B, int Cin, int IH,
    int IW, int Cout, int stride, int n_layers, const float *const *rw, float *const *rwf,
    float *const *rwd, const int *rCout, const int *rCin, void *stream) {
  SCAE_REQUIRE(out);
  return first_fwd_relayout(img, w, bias, out, nullptr, B, Cin, IH, IW, Cout, stride, n_layers, rw,
                            rwf, rwd, nullptr, nullptr, rCout, rCin, stream);
}
extern "C" int scae_conv3x3_first_fwd_relayout_bf16(
    const float *img, const float *w, const float *bias, uint16_t *out_h, int B, int Cin, int IH,
    int IW, int Cout, int stride, int n_layers, const float *const *rw, float *const *rwf,
    float *const *rwd, uint16_t *const *rwfh, uint16_t *const *rwdh, const int *rCout,
    const int *rCin, void *stream) {
  SCAE_REQUIRE(out_h && (n_layers == 0 || (rwfh && rwdh)));
  return first_fwd_relayout(img, w, bias, nullptr, out_h, B, Cin, IH, IW, Cout, stride, n_layers,
                            rw, rwf, rwd, rwfh, rwdh, rCout, rCin, stream);
}

extern "C" int scae_conv3x3_first_fwd_f32(const float *img, const float *w, const float *bias,
                                          float *out, int B, int Cin, int IH, int IW, int Cout,
                                          int stride, void *stream) {
  return scae_conv3x3_first_fwd_relayout_f32(img, w, bias, out, B, Cin, IH, IW, Cout, stride, 0,
                                             nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int scae_conv3x3_first_wgrad_rows(int B, int Cout) {
  if (B <= 0 || Cout <= 0 || Cout % 64) return 0;
  const FirstSplit f = first_split(B, Cout);
  return B * f.slices;
}

// fills a ReduceBatch; returns the workgroups (of 256 elements) per layer or < 0
static int fill_reduce(ReduceBatch &r, int n_layers, const float *const *partial,
                       float *const *dw, float *const *db, const int *Cout, const int *Cin,
                       const int *splits) {
  if (!(n_layers > 0 && n_layers <= 8 && partial && dw && db && Cout && Cin && splits)) return -1;
  int nmax = 0;
  for (int l = 0; l < n_layers; ++l) {
    if (!(partial[l] && dw[l] && Cout[l] > 0 && Cin[l] > 0 && splits[l] > 0)) return -1;
    r.partial[l] = partial[l], r.dw[l] = dw[l], r.db[l] = db[l];
    r.Cout[l] = Cout[l], r.Cin[l] = Cin[l], r.splits[l] = splits[l];
    const int n = 9 * Cout[l] * Cin[l] + Cout[l];
    nmax = nmax > n ? nmax : n;
  }
  return (nmax + 255) / 256;
}

extern "C" int scae_conv3x3_first_wgrad_reduce_f32(
    const float *dpre, const float *img, float *partial, int B, int Cin, int IH, int IW,
    int Cout, int stride, int n_layers, const float *const *rpartial, float *const *rdw,
    float *const *rdb, const int *rCout, const int *rCin, const int *rsplits, void *stream) {
  ConvGeom g{B, IH, IW, (IH - 3) / stride + 1, (IW - 3) / stride + 1, Cin, Cout, stride};
  int rc = check_geom(g, false);
  if (rc) return rc;
  SCAE_REQUIRE(dpre && img && partial && n_layers >= 0);
  if (Cout % 64) return SCAE_ERR_UNSUPPORTED;
  const FirstSplit f = first_split(B, Cout);
  // (a lane = (pixel, channel quad) form like the forward's, with 16-byte gradient loads,
  // measured slower here: 40 accumulators per lane and a cross-lane meeting per entry)
  const size_t lds = ((size_t)Cin * IH * IW +
                      (f.parts > 1 ? (size_t)f.parts * Cout * (Cin * 9 + 1) : 0)) * sizeof(float);
  if (lds > 64 * 1024) return SCAE_ERR_UNSUPPORTED;
  ReduceBatch r{};
  int rb = 0;
  if (n_layers > 0) {
    rb = fill_reduce(r, n_layers, rpartial, rdw, rdb, rCout, rCin, rsplits);
    SCAE_REQUIRE(rb > 0);
  }
  const int n_first = B * f.slices;
#define SCAE_FIRST_WGRAD(CI)                                                                  \
  case CI:                                                                                    \
    scae::launch(conv_first_wgrad_kernel<CI>, dim3(n_first + n_layers * rb), dim3(256), \
                       lds, (hipStream_t)stream, dpre, img, partial, g, r, n_first, rb);      \
    break;
  switch (Cin) {
    SCAE_FIRST_WGRAD(1) SCAE_FIRST_WGRAD(2) SCAE_FIRST_WGRAD(3) SCAE_FIRST_WGRAD(4)
    default: return SCAE_ERR_UNSUPPORTED;
  }
#undef SCAE_FIRST_WGRAD
  return scae_launch_status();
}

extern "C" int scae_conv3x3_first_wgrad_f32(const float *dpre, const float *img, float *partial,
                                            int B, int Cin, int IH, int IW, int Cout, int stride,
                                            void *stream) {
  return scae_conv3x3_first_wgrad_reduce_f32(dpre, img, partial, B, Cin, IH, IW, Cout, stride, 0,
                                             nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                             stream);
}

// bf16 operands (mfma_tile.h MODE 3: 128 x 128 tiles) need 128-channel multiples and
// enough output rows for a few tiles per CU; otherwise the fp32 kernels run
static bool conv_bf16_shape(long rows, int Cin, int Cout) {
  return Cin % 128 == 0 && Cout % 128 == 0 && rows * (Cout / 128) >= 128 * 128;
}

// the forward form with the K loop dealt to the waves (fwd_x6k_tile) against the ring-pipelined
// tiles, each ALONE: 34 -> 27 us at cfg-2's second layer, 42 -> 35 at CIFAR's, 210 / 121 / 68 ->
// 184 / 102 / 57 at B = 1024.  In a training step the second layer's launch carries the folding
// products, and beside these tiles (three workgroups per CU instead of four: tiles and riders no
// longer fit one round) the riders cost 8 - 10 us instead of 0.8: cfg-2 35 -> 42 - 44 us, the step
// 0.4966 -> 0.505 ms (0.498 with the riders back in the prologue); the B = 1024 step 4.61 -> 4.66
// ms.  So it is an OPTION (SCAE_K8_FWDK = 1, or a tile-count threshold at build time), and ONE rule
// serves the plain and the carrying launch: an eager step and a replayed one stay bit-identical.
#ifndef SCAE_FWDK_MIN_TILES
#define SCAE_FWDK_MIN_TILES 0   // 0: never by default
#endif
static bool fwd_ksplit(const ConvGeom &g) {
  if (g.Cin % dgk::BKF || g.Cout % dgk::TN ||
      (size_t)g.B * g.IH * g.IW * g.Cin * 4 >= (1u << 31))
    return false;
  const char *e = getenv("SCAE_K8_FWDK"), *f = getenv("SCAE_K8_FWD");
  if (e && *e) return atoi(e) != 0;
  if (f && *f) return false;
  const long tiles = (long)(g.Cout / dgk::TN) * (((long)g.B * g.OH * g.OW + dgk::TM - 1) / dgk::TM);
  return SCAE_FWDK_MIN_TILES > 0 && tiles >= SCAE_FWDK_MIN_TILES;
}

static int conv_fwd_impl(const float *in, const float *wf, const float *bias, float *out,
                         const float *post_bias, float *out_post, int B, int IH, int IW, int Cin,
                         int Cout, int stride, bool bf16, void *stream) {
  ConvGeom g{B, IH, IW, (IH - 3) / stride + 1, (IW - 3) / stride + 1, Cin, Cout, stride};
  int rc = check_geom(g, true);
  if (rc) return rc;
  SCAE_REQUIRE(in && wf && bias && out && (!out_post || post_bias));
  const int M = B * g.OH * g.OW;
  hipStream_t st = (hipStream_t)stream;
  if (bf16 && conv_bf16_shape(M, Cin, Cout)) {
    scae::launch(conv_fwd_kernel<3>, dim3(Cout / 128, (M + 127) / 128), dim3(NT), 0, st, in,
                       wf, bias, out, post_bias, out_post, g);
    return scae_launch_status();
  }
  if (fwd_ksplit(g)) {   // the K loop dealt to the waves (fwd_x6k_tile)
    scae::launch(conv_fwd_x6k_kernel, dim3(Cout / dgk::TN, (M + dgk::TM - 1) / dgk::TM), dim3(NT),
                 0, st, in, wf, bias, out, post_bias, out_post, g);
    return scae_launch_status();
  }
  const int cfg = pipe_cfg("SCAE_K8_FWD", M, Cout);
  if (cfg >= 0) {
#define SCAE_FWD_PIPE(TT)                                                                    \
  scae::launch(conv_fwd_pipe_kernel<TT>, dim3(Cout / TT::TB, (M + TT::TA - 1) / TT::TA), \
                     dim3(pipe::NT), 0, st, in, wf, bias, out, post_bias, out_post, g)
    switch (cfg) {
      case 0: SCAE_FWD_PIPE(PipeC0); break;
      case 1: SCAE_FWD_PIPE(PipeC1); break;
      case 2: SCAE_FWD_PIPE(PipeC2); break;
      default: SCAE_FWD_PIPE(PipeC3);
    }
#undef SCAE_FWD_PIPE
    return scae_launch_status();
  }
#ifndef SCAE_FWD_WIDE_MIN
#define SCAE_FWD_WIDE_MIN 300
#endif
#ifndef SCAE_FWD_SMALL_TILES
#define SCAE_FWD_SMALL_TILES SCAE_SMALL_TILES
#endif
  const long f64 = (long)(Cout / 64) * ((M + 63) / 64), f32 = (long)(Cout / 64) * ((M + 31) / 32);
  switch (f64 >= SCAE_FWD_SMALL_TILES ? 0 : (f32 >= SCAE_FWD_WIDE_MIN ? 2 : 1)) {
    case 0:
      scae::launch(conv_fwd_kernel<0>, dim3(Cout / 64, (M + 63) / 64), dim3(NT), 0, st, in,
                         wf, bias, out, post_bias, out_post, g);
      break;
    case 2:
      scae::launch(conv_fwd_kernel<2>, dim3(Cout / 64, (M + 31) / 32), dim3(NT), 0, st, in,
                         wf, bias, out, post_bias, out_post, g);
      break;
    default:
      scae::launch(conv_fwd_kernel<1>, dim3(Cout / 32, (M + 31) / 32), dim3(NT), 0, st, in,
                         wf, bias, out, post_bias, out_post, g);
  }
  return scae_launch_status();
}

extern "C" int scae_conv3x3_fwd_f32(const float *in, const float *wf, const float *bias,
                                    float *out, const float *post_bias, float *out_post, int B,
                                    int IH, int IW, int Cin, int Cout, int stride,
                                    void *stream) {
  return conv_fwd_impl(in, wf, bias, out, post_bias, out_post, B, IH, IW, Cin, Cout, stride,
                       false, stream);
}
// scae_conv3x3_fwd_f32 carrying scae_seed_fold_fwd_f32(fold) in the same launch
extern "C" int scae_conv3x3_fwd_fold_f32(const float *in, const float *wf, const float *bias,
                                         float *out, const float *post_bias, float *out_post,
                                         int B, int IH, int IW, int Cin, int Cout, int stride,
                                         const scae_seed_fold_desc *fold, void *stream) {
  SCAE_REQUIRE(in && wf && bias && out && fold && B > 0 && stride > 0 && IH >= 3 && IW >= 3);
  if (Cin % pipe::BK || Cout % PipeC2::TB) return SCAE_ERR_UNSUPPORTED;
  const scae_seed_fold_desc &a = *fold;
  if (!(a.seeds && a.wq && a.bq && a.wk && a.bk && a.wv && a.bv && a.wo && a.bo && a.w2 && a.b2 &&
        a.q && a.wkf && a.bkf && a.wvf && a.bvf && a.wv2e))
    return SCAE_ERR_BAD_ARG;
  if (!scae_seed_fold_supported(a.O, a.C, a.D) ||
      scae_fold::lds_bytes(a.C, a.D) > FOLD_SMEM * sizeof(float))
    return SCAE_ERR_UNSUPPORTED;
  ConvGeom g{B, IH, IW, (IH - 3) / stride + 1, (IW - 3) / stride + 1, Cin, Cout, stride};
  const scae_fold::Plan plan = scae_fold::plan(a.C, a.D);
  if (fwd_ksplit(g)) {
    const int M = B * g.OH * g.OW, gx = Cout / dgk::TN, n_conv = gx * ((M + dgk::TM - 1) / dgk::TM);
    scae::launch(conv_fwd_x6k_fold_kernel, dim3(n_conv + plan.blocks()), dim3(NT), 0,
                 (hipStream_t)stream, in, wf, bias, out, post_bias, out_post, g, gx, n_conv, a, plan);
    return scae_launch_status();
  }
  const int M = B * g.OH * g.OW, gx = Cout / PipeC2::TB,
            n_conv = gx * ((M + PipeC2::TA - 1) / PipeC2::TA);
  scae::launch(conv_fwd_pipe_fold_kernel<PipeC2>, dim3(n_conv + plan.blocks()),
                     dim3(pipe::NT), 0, (hipStream_t)stream, in, wf, bias, out, post_bias, out_post,
                     g, gx, n_conv, a, plan);
  return scae_launch_status();
}

extern "C" int scae_conv3x3_fwd_bf16(const float *in, const float *wf, const float *bias,
                                     float *out, const float *post_bias, float *out_post, int B,
                                     int IH, int IW, int Cin, int Cout, int stride,
                                     void *stream) {
  return conv_fwd_impl(in, wf, bias, out, post_bias, out_post, B, IH, IW, Cin, Cout, stride,
                       true, stream);
}

// the data-gradient tiling of a layer: class tables, tile shape, grid
struct DgradLaunch {
  DgradPlan pl;
  int mode, gx, ny;
  int cfg;   // >= 0: second-generation tile shape (mode / gx / ny then refer to it)
};
// (pair = true: the launch also carries the weight-gradient tiles, so the data
// gradient does not have to fill the chip on its own)
#ifndef SCAE_DGK_DEFAULT
#define SCAE_DGK_DEFAULT 1
#endif
#ifndef SCAE_DGX_MIN_TILES
#define SCAE_DGX_MIN_TILES 500
#endif
#ifndef SCAE_PAIR_SMALL_TILES
#define SCAE_PAIR_SMALL_TILES SCAE_SMALL_TILES
#endif
#ifndef SCAE_PAIR_WIDE_MIN
#define SCAE_PAIR_WIDE_MIN 600
#endif
static DgradLaunch plan_dgrad(const ConvGeom &g, bool pair = false, bool bf16 = false) {
  DgradLaunch d;
  DgradPlan &pl = d.pl;
  pl.nrc = dgrad_axis(g.IH, g.OH, g.stride, pl.rmask, pl.rcount, pl.rstart, pl.rlist);
  pl.ncc = dgrad_axis(g.IW, g.OW, g.stride, pl.cmask, pl.ccount, pl.cstart, pl.clist);
  auto tiles = [&](int T) {  // fills tile_start for tile size T, returns the total
    int tot = 0;
    for (int z = 0; z < pl.nrc * pl.ncc; ++z) {
      pl.tile_start[z] = tot;
      tot += (g.B * pl.rcount[z / pl.ncc] * pl.ccount[z % pl.ncc] + T - 1) / T;
    }
    pl.tile_start[pl.nrc * pl.ncc] = tot;
    return tot;
  };
  // rows of the problem: input pixels (class tiles are ragged; close enough to choose)
  if (bf16) {   // mfma_tile.h MODE 3: 128 input pixels x 128 channels
    d.cfg = -1;
    d.mode = 3;
    d.ny = tiles(128);
    d.gx = g.Cin / 128;
    return d;
  }
  // (default: first-generation data-gradient tiles, see conv_bwd_pair_mixed_kernel)
  const char *env = getenv(pair ? "SCAE_K8_PAIR" : "SCAE_K8_DG");
  d.cfg = env && *env ? pipe_cfg(pair ? "SCAE_K8_PAIR" : "SCAE_K8_DG", (long)g.B * g.IH * g.IW,
                                 g.Cin)
                      : -1;
  if (d.cfg >= 0) {
    const int ta = (d.cfg == 0 || d.cfg == 3) ? 64 : 32, tb = d.cfg <= 1 ? 128 : 64;
    d.mode = 0;
    d.ny = tiles(ta);
    d.gx = g.Cin / tb;
    return d;
  }
  // the DMA-fed exact-split tile (DMODE 4): SCAE_K8_DGX = minimal number of its 64 x 128 tiles
  // for a layer to take it (0 = never)
  if (!(env && *env) && g.Cin % dgx::TN == 0 && g.Cout % dgx::BKF == 0 &&
      (size_t)g.B * g.IH * g.IW * g.Cin * 4 < (1u << 31)) {
    const char *xe = getenv("SCAE_K8_DGX");
    const long min_tiles = xe && *xe ? atol(xe) : SCAE_DGX_MIN_TILES;
    const long tx = (long)(g.Cin / dgx::TN) * tiles(dgx::TM);
    if (min_tiles > 0 && tx >= min_tiles) {
      d.mode = 4, d.ny = tiles(dgx::TM), d.gx = g.Cin / dgx::TN;
      return d;
    }
  }
  // ... and for the layers below that, the form with the K loop split over the waves (DMODE 5):
  // SCAE_K8_DGK = 1 / 0
  if (!(env && *env) && g.Cin % dgk::TN == 0 && g.Cout % dgk::BKF == 0 &&
      (size_t)g.B * g.IH * g.IW * g.Cin * 4 < (1u << 31)) {
    const char *ke = getenv("SCAE_K8_DGK");
    if (ke && *ke ? atoi(ke) != 0 : SCAE_DGK_DEFAULT != 0) {
      d.mode = 5, d.ny = tiles(dgk::TM), d.gx = g.Cin / dgk::TN;
      return d;
    }
  }
  const long t64 = (long)(g.Cin / 64) * tiles(64), t32 = (long)(g.Cin / 64) * tiles(32);
  d.mode = pair ? (t64 >= SCAE_PAIR_SMALL_TILES ? 0 : (t32 >= SCAE_PAIR_WIDE_MIN ? 2 : 1))
                : tile_mode(t64, t32, 600);
  d.ny = tiles(d.mode == 0 ? 64 : 32);
  d.gx = d.mode == 1 ? g.Cin / 32 : g.Cin / 64;
  return d;
}

extern "C" int scae_conv3x3_dgrad_f32(const float *dpre, const float *wd, const float *gate,
                                      float *din, int B, int IH, int IW, int Cin, int Cout,
                                      int stride, void *stream) {
  ConvGeom g{B, IH, IW, (IH - 3) / stride + 1, (IW - 3) / stride + 1, Cin, Cout, stride};
  int rc = check_geom(g, true);
  if (rc) return rc;
  SCAE_REQUIRE(dpre && wd && din);
  if (IH > DG_MAXDIM || IW > DG_MAXDIM) return SCAE_ERR_UNSUPPORTED;
  const DgradLaunch d = plan_dgrad(g);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(d.gx, d.ny);
  if (d.cfg >= 0) {
#define SCAE_DG_PIPE(TT)                                                                   \
  scae::launch(conv_dgrad_pipe_kernel<TT>, grid, dim3(pipe::NT), 0, st, dpre, wd, gate, \
                     din, g, d.pl)
    switch (d.cfg) {
      case 0: SCAE_DG_PIPE(PipeC0); break;
      case 1: SCAE_DG_PIPE(PipeC1); break;
      case 2: SCAE_DG_PIPE(PipeC2); break;
      default: SCAE_DG_PIPE(PipeC3);
    }
#undef SCAE_DG_PIPE
    return scae_launch_status();
  }
  if (d.mode == 4) {
    const char *ne = getenv("SCAE_K8_DGX_NS");   // (tuning aid: ring depth)
    const int ns = ne && *ne ? atoi(ne) : 2;
    if (ns == 1)
      scae::launch(conv_dgrad_x6_kernel<1>, grid, dim3(NT), 0, st, dpre, wd, gate, din, g, d.pl);
    else if (ns == 3)
      scae::launch(conv_dgrad_x6_kernel<3>, grid, dim3(NT), 0, st, dpre, wd, gate, din, g, d.pl);
    else
      scae::launch(conv_dgrad_x6_kernel<2>, grid, dim3(NT), 0, st, dpre, wd, gate, din, g, d.pl);
  } else if (d.mode == 5)
    scae::launch(conv_dgrad_x6k_kernel, grid, dim3(NT), 0, st, dpre, wd, gate, din, g, d.pl);
  else if (d.mode == 0)
    scae::launch(conv_dgrad_kernel<0>, grid, dim3(NT), 0, st, dpre, wd, gate, din, g, d.pl);
  else if (d.mode == 2)
    scae::launch(conv_dgrad_kernel<2>, grid, dim3(NT), 0, st, dpre, wd, gate, din, g, d.pl);
  else
    scae::launch(conv_dgrad_kernel<1>, grid, dim3(NT), 0, st, dpre, wd, gate, din, g, d.pl);
  return scae_launch_status();
}

static int conv_bwd_pair_impl(const float *dpre, const float *wd, const float *in, float *din,
                              float *partial, int B, int IH, int IW, int Cin, int Cout,
                              int stride, bool bf16, void *stream,
                              const PairRider *rider = nullptr) {
  ConvGeom g{B, IH, IW, (IH - 3) / stride + 1, (IW - 3) / stride + 1, Cin, Cout, stride};
  int rc = check_geom(g, true);
  if (rc) return rc;
  SCAE_REQUIRE(dpre && wd && in && din && partial);
  if (IH > DG_MAXDIM || IW > DG_MAXDIM) return SCAE_ERR_UNSUPPORTED;
  bf16 = bf16 && conv_bf16_shape((long)B * IH * IW, Cin, Cout);
  const DgradLaunch d = plan_dgrad(g, true, bf16);
  const WgradPlan p = wgrad_plan(B * g.OH * g.OW, Cin, Cout);
  if (rider) {   // only the mixed form below carries one
    const char *pe = getenv("SCAE_K8_PAIR");
    if (bf16 || d.cfg >= 0 || (pe && atoi(pe) < 0)) return SCAE_ERR_UNSUPPORTED;
  }
  if (bf16) {   // both gradients on bf16 operands, 128 x 128 tiles
    const PairGrid bg{d.gx * d.ny, d.gx, Cin / 128, Cout / 128};
    scae::launch((conv_bwd_pair_kernel<3, 3>), dim3(bg.nd + bg.wx * bg.wy * 9 * p.splits),
                       dim3(NT), 0, (hipStream_t)stream, dpre, wd, in, din, in, partial, g, d.pl,
                       p.splits, bg);
    return scae_launch_status();
  }
  const int wt = p.small && d.cfg < 0 ? 32 : 64;
  const PairGrid pg{d.gx * d.ny, d.gx, Cin / wt, Cout / wt};
  const dim3 grid(pg.nd + pg.wx * pg.wy * 9 * p.splits);
  hipStream_t st = (hipStream_t)stream;
  if (d.cfg >= 0) {
#define SCAE_PAIR_PIPE(TT)                                                                  \
  scae::launch((conv_bwd_pair_pipe_kernel<TT, PipeW>), grid, dim3(pipe::NT), 0, st,   \
                     dpre, wd, in, din, in, partial, g, d.pl, p.splits, pg)
    switch (d.cfg) {
      case 0: SCAE_PAIR_PIPE(PipeC0); break;
      case 1: SCAE_PAIR_PIPE(PipeC1); break;
      case 2: SCAE_PAIR_PIPE(PipeC2); break;
      default: SCAE_PAIR_PIPE(PipeC3);
    }
#undef SCAE_PAIR_PIPE
    return scae_launch_status();
  }
  const char *pe = getenv("SCAE_K8_PAIR");
  if (!(pe && atoi(pe) < 0)) {   // second-generation weight-gradient tiles (64 x 64)
    const PairGrid mg{d.gx * d.ny, d.gx, Cin / 64, Cout / 64};
    const dim3 mgrid(mg.nd + mg.wx * mg.wy * 9 * p.splits);
    const char *we = getenv("SCAE_K8_W16");   // (tuning aid: 0 / 1 forces the ring)
    // (the K-split data-gradient tile brings 48 KiB of LDS to the launch anyway: the short ring
    // would buy no workgroup)
    const bool w16 = we && *we ? atoi(we) != 0
                               : d.mode != 5 &&
                                     (long)B * g.OH * g.OW < SCAE_WGRAD_SHORT_CHUNK_PIXELS;
#define SCAE_PAIR_MIXED(DM, TW)                                                              \
  scae::launch((conv_bwd_pair_mixed_kernel<DM, TW>), mgrid, dim3(NT), 0, st, dpre, wd, in, \
                     din, in, partial, g, d.pl, p.splits, mg)
#define SCAE_PAIR_MIXED_FOLD(DM, TW)                                                          \
  scae::launch((conv_bwd_pair_mixed_rider_kernel<DM, TW>), rgrid, dim3(NT), 0, st, dpre, wd, \
                     in, din, in, partial, g, d.pl, p.splits, mg, *rider)
#define SCAE_PAIR_BY_MODE(LAUNCH, TW) \
  if (d.mode == 0) LAUNCH(0, TW);     \
  else if (d.mode == 2) LAUNCH(2, TW); \
  else if (d.mode == 4) LAUNCH(4, TW); \
  else if (d.mode == 5) LAUNCH(5, TW); \
  else LAUNCH(1, TW)
    if (rider) {
      const dim3 rgrid(mgrid.x + rider->n);
      if (w16) { SCAE_PAIR_BY_MODE(SCAE_PAIR_MIXED_FOLD, PipeW16); }
      else { SCAE_PAIR_BY_MODE(SCAE_PAIR_MIXED_FOLD, PipeW); }
      return scae_launch_status();
    }
    if (w16) { SCAE_PAIR_BY_MODE(SCAE_PAIR_MIXED, PipeW16); }
    else { SCAE_PAIR_BY_MODE(SCAE_PAIR_MIXED, PipeW); }
#undef SCAE_PAIR_BY_MODE
#undef SCAE_PAIR_MIXED_FOLD
#undef SCAE_PAIR_MIXED
    return scae_launch_status();
  }
#define SCAE_PAIR(DM, WS)                                                                     \
  scae::launch((conv_bwd_pair_kernel<DM, WS>), grid, dim3(NT), 0, st, dpre, wd, in, din, \
                     in, partial, g, d.pl, p.splits, pg)
  if (d.mode == 0) {
    if (p.small) SCAE_PAIR(0, true); else SCAE_PAIR(0, false);
  } else if (d.mode == 2) {
    if (p.small) SCAE_PAIR(2, true); else SCAE_PAIR(2, false);
  } else {
    if (p.small) SCAE_PAIR(1, true); else SCAE_PAIR(1, false);
  }
#undef SCAE_PAIR
  return scae_launch_status();
}

extern "C" int scae_conv3x3_bwd_pair_f32(const float *dpre, const float *wd, const float *in,
                                         float *din, float *partial, int B, int IH, int IW,
                                         int Cin, int Cout, int stride, void *stream) {
  return conv_bwd_pair_impl(dpre, wd, in, din, partial, B, IH, IW, Cin, Cout, stride, false,
                            stream);
}
// scae_conv3x3_bwd_pair_f32 carrying scae_seed_fold_bwd_f32(fold, fold_grads) in the same
// launch (C = 256 folding width, the mixed tile form of the pair): else SCAE_ERR_UNSUPPORTED
extern "C" int scae_conv3x3_bwd_pair_fold_f32(const float *dpre, const float *wd, const float *in,
                                              float *din, float *partial, int B, int IH, int IW,
                                              int Cin, int Cout, int stride,
                                              const scae_seed_fold_desc *fold,
                                              const scae_seed_fold_grads *fold_grads,
                                              void *stream) {
  int rc = scae_sf::check(fold);
  if (rc) return rc;
  rc = scae_sf::check_grads(fold_grads);
  if (rc) return rc;
  SCAE_REQUIRE(fold->wowv);
  if (fold->C != NT) return SCAE_ERR_UNSUPPORTED;   // a row workgroup = one thread per column
  PairRider r{};
  r.kind = 2, r.a = *fold, r.g = *fold_grads;
  int parts;
  size_t lds;
  scae_sf::bwd_shape(fold, r.pl, parts, lds);
  if (parts != 4 || lds > 24 * 1024) return SCAE_ERR_UNSUPPORTED;
  r.n = r.pl.ncol + fold->C;
  return conv_bwd_pair_impl(dpre, wd, in, din, partial, B, IH, IW, Cin, Cout, stride, false,
                            stream, &r);
}
// ... carrying scae_seed_attention_mfma_reduce_f32(rpartial .. C) instead
extern "C" int scae_conv3x3_bwd_pair_reduce_f32(const float *dpre, const float *wd,
                                                const float *in, float *din, float *partial,
                                                int B, int IH, int IW, int Cin, int Cout,
                                                int stride, const float *rpartial, int rows,
                                                const float *q, const float *wk, float *gq,
                                                float *gwk, float *gbk, float *gwv, float *gbv,
                                                int O, int C, void *stream) {
  PairRider r{};
  r.kind = 1;
  int rc = scae_saw::reduce_args(r.red, rpartial, rows, q, wk, gq, gwk, gbk, gwv, gbv, O, C);
  if (rc) return rc;
  r.n = scae_saw::reduce_blocks(r.red, NT);
  return conv_bwd_pair_impl(dpre, wd, in, din, partial, B, IH, IW, Cin, Cout, stride, false,
                            stream, &r);
}
extern "C" int scae_conv3x3_bwd_pair_bf16(const float *dpre, const float *wd, const float *in,
                                          float *din, float *partial, int B, int IH, int IW,
                                          int Cin, int Cout, int stride, void *stream) {
  return conv_bwd_pair_impl(dpre, wd, in, din, partial, B, IH, IW, Cin, Cout, stride, true,
                            stream);
}

#ifdef SCAE_FWD_PROF
extern "C" int scae_debug_fwd_prof(unsigned long long *out) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwd_prof), sizeof(g_fwd_prof));
}
#endif
#ifdef SCAE_CONV_MULTI_PROBE
// Upper-bound probe (tools/conv_multi_probe.py): the tiles of up to 3 layers in ONE launch
// WITHOUT dependencies between them (every layer reads buffers that already exist) -- what a
// dependency-tracking multi-layer launch could gain at most over one launch per layer.
namespace {
struct MultiFwd {
  const float *in[3], *wf[3], *bias[3];
  float *out[3];
  ConvGeom g[3];
  int start[4], gx[3];
};
template <class T>
__global__ __launch_bounds__(pipe::NT) void conv_fwd_multi_probe_kernel(MultiFwd a) {
  __shared__ __attribute__((aligned(1024))) float smem[T::SMEM];
  const int b = blockIdx.x;
  const int l = b >= a.start[2] ? 2 : (b >= a.start[1] ? 1 : 0);
  const int t = b - a.start[l];
  fwd_pipe_tile<T>(smem, t % a.gx[l], t / a.gx[l], a.in[l], a.wf[l], a.bias[l], a.out[l], nullptr,
                   nullptr, a.g[l]);
}
struct MultiBwd {
  const float *dpre[3], *wd[3], *gate[3], *in[3];
  float *din[3], *partial[3];
  ConvGeom g[3];
  DgradPlan pl[3];
  int splits[3], mode[3];
  PairGrid pg[3];
  int start[4];
};
__global__ __launch_bounds__(NT) void conv_bwd_multi_probe_kernel(const MultiBwd *ap) {
  constexpr int SM = Tile<0>::SMEM > PipeW::SMEM ? Tile<0>::SMEM : PipeW::SMEM;
  __shared__ __attribute__((aligned(1024))) float smem[SM];
  const MultiBwd &a = *ap;
  const int b = blockIdx.x;
  const int l = b >= a.start[2] ? 2 : (b >= a.start[1] ? 1 : 0);
  const int bid = b - a.start[l];
  const PairGrid pg = a.pg[l];
  if (bid < pg.nd) {
    if (a.mode[l] == 0)
      dgrad_tile<0>(smem, bid % pg.gx, bid / pg.gx, a.dpre[l], a.wd[l], a.gate[l], a.din[l], a.g[l], a.pl[l]);
    else if (a.mode[l] == 2)
      dgrad_tile<2>(smem, bid % pg.gx, bid / pg.gx, a.dpre[l], a.wd[l], a.gate[l], a.din[l], a.g[l], a.pl[l]);
    else if (a.mode[l] == 5)
      dgrad_any_tile<5, SM>(smem, bid % pg.gx, bid / pg.gx, a.dpre[l], a.wd[l], a.gate[l], a.din[l], a.g[l], a.pl[l]);
    else if (a.mode[l] == 4)
      dgrad_any_tile<4, SM>(smem, bid % pg.gx, bid / pg.gx, a.dpre[l], a.wd[l], a.gate[l], a.din[l], a.g[l], a.pl[l]);
    else
      dgrad_tile<1>(smem, bid % pg.gx, bid / pg.gx, a.dpre[l], a.wd[l], a.gate[l], a.din[l], a.g[l], a.pl[l]);
  } else {
    const int w = bid - pg.nd, bx = w % pg.wx, t = w / pg.wx;
    wgrad_pipe_tile<PipeW>(smem, bx, t % pg.wy, t / pg.wy, a.dpre[l], a.in[l], a.partial[l], a.g[l],
                           a.splits[l]);
  }
}
}  // namespace
extern "C" int scae_debug_conv_fwd_multi(int n, const float *const *in, const float *const *wf,
                                         const float *const *bias, float *const *out, const int *B,
                                         const int *IH, const int *Cin, const int *Cout,
                                         const int *stride, void *stream) {
  MultiFwd a{};
  int tot = 0;
  for (int l = 0; l < 3; ++l) {
    const int k = l < n ? l : n - 1;
    a.in[l] = in[k], a.wf[l] = wf[k], a.bias[l] = bias[k], a.out[l] = out[k];
    a.g[l] = ConvGeom{B[k], IH[k], IH[k], (IH[k] - 3) / stride[k] + 1, (IH[k] - 3) / stride[k] + 1,
                      Cin[k], Cout[k], stride[k]};
    a.gx[l] = Cout[k] / PipeC2::TB;
    a.start[l] = tot;
    if (l < n) tot += a.gx[l] * ((B[k] * a.g[l].OH * a.g[l].OW + PipeC2::TA - 1) / PipeC2::TA);
  }
  a.start[3] = tot;
  for (int l = n; l < 3; ++l) a.start[l] = tot;
  scae::launch(conv_fwd_multi_probe_kernel<PipeC2>, dim3(tot), dim3(pipe::NT), 0,
                     (hipStream_t)stream, a);
  return scae_launch_status();
}
// `scratch`: device memory for the argument block (sizeof(MultiBwd) bytes, >= 8 KiB given)
extern "C" int scae_debug_conv_bwd_multi(int n, const float *const *dpre, const float *const *wd,
                                         const float *const *in, float *const *din,
                                         float *const *partial, const int *B, const int *IH,
                                         const int *Cin, const int *Cout, const int *stride,
                                         void *scratch, void *stream) {
  static MultiBwd a;
  a = MultiBwd{};
  int tot = 0;
  for (int l = 0; l < 3; ++l) {
    const int k = l < n ? l : n - 1;
    a.dpre[l] = dpre[k], a.wd[l] = wd[k], a.gate[l] = in[k], a.in[l] = in[k], a.din[l] = din[k],
    a.partial[l] = partial[k];
    ConvGeom g{B[k], IH[k], IH[k], (IH[k] - 3) / stride[k] + 1, (IH[k] - 3) / stride[k] + 1,
               Cin[k], Cout[k], stride[k]};
    a.g[l] = g;
    const DgradLaunch d = plan_dgrad(g, true, false);
    const WgradPlan p = wgrad_plan(B[k] * g.OH * g.OW, Cin[k], Cout[k]);
    a.pl[l] = d.pl, a.mode[l] = d.mode, a.splits[l] = p.splits;
    a.pg[l] = PairGrid{d.gx * d.ny, d.gx, Cin[k] / 64, Cout[k] / 64};
    a.start[l] = tot;
    if (l < n) tot += a.pg[l].nd + a.pg[l].wx * a.pg[l].wy * 9 * p.splits;
  }
  a.start[3] = tot;
  for (int l = n; l < 3; ++l) a.start[l] = tot;
  hipError_t e = hipMemcpyAsync(scratch, &a, sizeof(a), hipMemcpyHostToDevice, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  scae::launch(conv_bwd_multi_probe_kernel, dim3(tot), dim3(NT), 0, (hipStream_t)stream,
                     (const MultiBwd *)scratch);
  return scae_launch_status();
}
extern "C" int scae_debug_conv_bwd_multi_bytes(void) { return (int)sizeof(MultiBwd); }
#endif

extern "C" int scae_conv3x3_wgrad_splits(int B, int OH, int OW, int Cin, int Cout) {
  if (B <= 0 || OH <= 0 || OW <= 0 || Cin <= 0 || Cout <= 0) return 0;
  return wgrad_plan(B * OH * OW, Cin, Cout).splits;
}

extern "C" int scae_conv3x3_wgrad_f32(const float *dpre, const float *in, float *partial,
                                      float *dw, float *db, int B, int IH, int IW, int Cin,
                                      int Cout, int stride, void *stream) {
  ConvGeom g{B, IH, IW, (IH - 3) / stride + 1, (IW - 3) / stride + 1, Cin, Cout, stride};
  int rc = check_geom(g, true);
  if (rc) return rc;
  SCAE_REQUIRE(dpre && in && partial);
  const WgradPlan p = wgrad_plan(B * g.OH * g.OW, Cin, Cout);
  const char *e = getenv("SCAE_K8_WG");
  if (!(e && atoi(e) < 0))
    scae::launch(conv_wgrad_pipe_kernel<PipeW>, dim3(Cin / 64, Cout / 64, 9 * p.splits),
                       dim3(pipe::NT), 0, (hipStream_t)stream, dpre, in, partial, g, p.splits);
  else if (p.small)
    scae::launch(conv_wgrad_kernel<true>, dim3(Cin / 32, Cout / 32, 9 * p.splits), dim3(NT),
                       0, (hipStream_t)stream, dpre, in, partial, g, p.splits);
  else
    scae::launch(conv_wgrad_kernel<false>, dim3(Cin / 64, Cout / 64, 9 * p.splits),
                       dim3(NT), 0, (hipStream_t)stream, dpre, in, partial, g, p.splits);
  if (dw) {  // else: the caller reduces later (scae_conv3x3_wgrad_reduce_batch_f32)
    const int n = 9 * Cout * Cin + Cout;
    scae::launch(reduce_wgrad_kernel, dim3((n + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, partial, dw, db, Cout, Cin, p.splits);
  }
  return scae_launch_status();
}

extern "C" int scae_conv3x3_wgrad_reduce_batch_f32(int n_layers, const float *const *partial,
                                                   float *const *dw, float *const *db,
                                                   const int *Cout, const int *Cin,
                                                   const int *splits, void *stream) {
  ReduceBatch r{};
  const int rb = fill_reduce(r, n_layers, partial, dw, db, Cout, Cin, splits);
  SCAE_REQUIRE(rb > 0);
  scae::launch(reduce_wgrad_batch_kernel, dim3(rb, n_layers), dim3(256), 0,
                     (hipStream_t)stream, r);
  return scae_launch_status();
}
