// K8 with bf16-RESIDENT operands (BASELINE.json configs[2]: "bs=1024 bf16"): the 3x3 "valid"
// convolutions of the part-capsule CNN encoder (part_encoder.py:26-44, nn_ext.py:34-59) and
// their backward as implicit GEMMs on v_mfma_f32_32x32x16_bf16 (fp32 accumulate), with the
// activations, the pre-activation gradients and the re-laid-out filters kept as bf16 in HBM.
//
// The first bf16 form (mfma_tile.h MODE 3) reads fp32 tensors and rounds them on their way
// into LDS: twice the bytes per operand, a v_cvt per pair in the tile loop, 32-deep K chunks
// staged through registers with a barrier every 8 MFMAs -- 0.07 of the bf16 matrix peak on the
// backward pair.  Here every GEMM operand already IS bf16:
//   * NHWC activations (B, H, W, C) as bf16: one (pixel, tap) row of the implicit A matrix is
//     C contiguous bf16 -- 256 bytes at C = 128;
//   * 128 x 128 output tiles, 2 x 2 waves of 64 x 64 (2 x 2 MFMA tiles: four 32x32x16 products
//     per pair of 16-byte fragment reads), K walked in chunks of 64 (= one tap x 64 channels:
//     128-byte LDS rows, 16 MFMAs per wave between barriers);
//   * k-contiguous operands (forward, data gradient) go global -> LDS by DMA
//     (buffer_load ... lds, 16 bytes per lane, 1 KiB per wave instruction) into one or two stages
//     of 32 KiB (one: five workgroups per CU cover each other -- the faster form once a layer
//     has more than ~500 tiles; two: the DMA of chunk c + 1 under the MFMAs of chunk c); the
//     eight 16-byte quads of a 128-byte row are XOR-swizzled with (row >> 1) & 7 on the SOURCE
//     address (the LDS image of a DMA piece is lane-linear), which makes the row-per-lane
//     ds_read_b128 fragment reads conflict free;
//   * k-strided operands (weight gradient: K = pixels) go global -> LDS by DMA as they lie in
//     memory ([pixel][channel] rows of 256 bytes) and the MFMA fragments come out of that image
//     through gfx950's transposing read ds_read_b64_tr_b16;
//   * the data gradient walks input pixels grouped by stride parity, each class only over the
//     taps that reach it (no MFMA multiplies a structural zero of the stride-2 layer); rows whose
//     tap falls outside the output read zeros through the buffer descriptor's range check.
// Accumulation order over K is (tap, channel) ascending in 16-wide MFMA steps -- the order of
// the first form, whose results these kernels reproduce to round-off of the fp32 accumulators.
#include <cstdlib>

#include "mfma_pipe.h"
#include "conv_first_dev.h"

namespace {
namespace pipe = scae_pipe;
using scae_first::ConvGeom;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;   // storage type

constexpr int NT = 256, TM = 128, TN = 128, BKE = 64;   // BKE: K chunk in elements
constexpr int ROWB = 2 * BKE;                            // bytes per LDS row (128)
constexpr int TILE_B = TM * ROWB;                        // 16 KiB per operand tile
constexpr int STAGE_B = 2 * TILE_B;                      // A + B
constexpr int SLAB = 32 * 36;                            // epilogue: floats per wave

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  const b2 r = __builtin_convertvector((f2){lo, hi}, b2);   // v_cvt_pk_bf16_f32 (RNE)
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// slot of 16-byte quad g of LDS row `row`
__device__ __forceinline__ int swz(int row, int g) { return g ^ ((row >> 1) & 7); }

// ---- the MFMAs of one 64-deep chunk: stage = [A tile | B tile], rows of 128 bytes --------
__device__ __forceinline__ void mma_chunk(const unsigned char *stage, f32x16 (&acc)[2][2], int wm,
                                          int wn, int i, int kk) {
  const unsigned char *As = stage, *Bs = stage + TILE_B;
#pragma unroll
  for (int s = 0; s < BKE / 16; ++s) {
    bf16x8 a[2], b[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ra = wm * 64 + t * 32 + i, rb = wn * 64 + t * 32 + i;
      a[t] = *reinterpret_cast<const bf16x8 *>(As + ra * ROWB + swz(ra, 2 * s + kk) * 16);
      b[t] = *reinterpret_cast<const bf16x8 *>(Bs + rb * ROWB + swz(rb, 2 * s + kk) * 16);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[u], acc[t][u], 0, 0, 0);
  }
}

// ---- accumulators out: epi(tile row, tile col (multiple of 8), 8 floats) ------------------
// each wave's four 32 x 32 accumulator tiles pass one at a time through its private
// [32][36] LDS slab and leave as rows of 8 consecutive columns per lane
template <class Epi>
__device__ __forceinline__ void tile_epilogue(float *smem, const f32x16 (&acc)[2][2], int wid,
                                              int lane, Epi epi) {
  const int i = lane & 31, kk = lane >> 5, wm = wid >> 1, wn = wid & 1;
  pipe::wg_barrier();   // the operand stages are dead: the slabs alias them
  float *slab = smem + wid * SLAB;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int e = 0; e < 16; ++e) slab[((e & 3) + 8 * (e >> 2) + 4 * kk) * 36 + i] = acc[t][u][e];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the wave's own slab)
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int row = (lane >> 2) + 16 * pass, c8 = 8 * (lane & 3);
        const float4 v0 = pipe::lds4(slab + row * 36 + c8), v1 = pipe::lds4(slab + row * 36 + c8 + 4);
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        epi(wm * 64 + t * 32 + row, wn * 64 + u * 32 + c8, v);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// ---- the K loop of the DMA-staged passes --------------------------------------------------
// issue(c, stage): this wave's 8 DMA pieces of chunk c (4 of the A tile, 4 of the B tile).
// NS stages of 32 KiB: NS - 1 chunks in flight under the MFMAs of the current one, ONE barrier
// per chunk (NS = 1: no overlap inside the workgroup, two barriers per chunk, but five
// workgroups share a CU's LDS and cover each other).
template <int NS, class Issue, class Mma>
__device__ __forceinline__ void dma_loop(int nchunk, unsigned char *smem, Issue issue, Mma mma) {
  if (nchunk <= 0) return;
  if (NS == 1) {
    for (int c = 0; c < nchunk; ++c) {
      issue(c, smem);
      pipe::wait_vm<0>();
      pipe::wg_barrier();
      mma(smem);
      pipe::wg_barrier();   // everyone has read the stage: the next chunk may land
    }
    return;
  }
#pragma unroll
  for (int c = 0; c < NS - 1; ++c)
    if (c < nchunk) issue(c, smem + c * STAGE_B);
  int s = 0;   // stage of chunk c
  for (int c = 0; c < nchunk; ++c) {
    // this wave's pieces of chunk c have landed: what may still be in flight are the (up to
    // NS - 2) chunks issued after it
    pipe::wait_chunk<8, NS>(min(NS - 2, nchunk - 1 - c));
    pipe::wg_barrier();   // ... everyone's; and everyone is done reading chunk c - 1's stage
    const int sp = s == 0 ? NS - 1 : s - 1;   // = stage of chunk c - 1 = of chunk c + NS - 1
    if (c + NS - 1 < nchunk) issue(c + NS - 1, smem + sp * STAGE_B);
    mma(smem + s * STAGE_B);
    s = s + 1 == NS ? 0 : s + 1;
  }
}
template <int NS, class Issue>
__device__ __forceinline__ void dma_mainloop(int nchunk, unsigned char *smem, f32x16 (&acc)[2][2],
                                             int wid, int lane, Issue issue) {
  const int i = lane & 31, kk = lane >> 5, wm = wid >> 1, wn = wid & 1;
  dma_loop<NS>(nchunk, smem, issue,
               [&](const unsigned char *stage) { mma_chunk(stage, acc, wm, wn, i, kk); });
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;
}

// the rows a lane moves by DMA: piece j of wave w covers tile rows 32 w + 8 j .. + 7, lane l
// the row 32 w + 8 j + (l >> 3) and the LDS slot l & 7 of it
struct DmaRows {
  int row[4];    // tile row of piece j
  int lds[4];    // byte offset of the piece in an operand tile (wave-uniform)
};
__device__ __forceinline__ DmaRows dma_rows(int wid, int lane) {
  DmaRows d;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    d.row[j] = 32 * wid + 8 * j + (lane >> 3);
    d.lds[j] = (32 * wid + 8 * j) * ROWB;
  }
  return d;
}
// byte offset within a 128-byte source row of the quad that lands in this lane's LDS slot
__device__ __forceinline__ int src_quad(int row, int lane) { return swz(row, lane & 7) * 16; }

// ---- forward -------------------------------------------------------------------------------
// out[m][co] = relu(sum_{tap, ci} in[pix(m, tap)][ci] wf[co][tap][ci] + bias[co])
//   grid (Cout / 128, ceil(M / 128)); out_h bf16 NHWC; out_f / out_post (nullable) fp32
template <int NS>
__global__ __launch_bounds__(NT, 2) void conv_fwd_bf16r_kernel(
    const bf16_t *__restrict__ in, const bf16_t *__restrict__ wf, const float *__restrict__ bias,
    bf16_t *__restrict__ out_h, float *__restrict__ out_f, const float *__restrict__ post_bias,
    float *__restrict__ out_post, ConvGeom g) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int M = g.B * g.OH * g.OW;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const pipe::rsrc_t ra = pipe::make_rsrc(in, (unsigned)((size_t)g.B * g.IH * g.IW * g.Cin * 2));
  const pipe::rsrc_t rb = pipe::make_rsrc(wf, (unsigned)((size_t)g.Cout * 9 * g.Cin * 2));
  const DmaRows dr = dma_rows(wid, lane);
  int va[4], vb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + dr.row[j];
    va[j] = pipe::DMA_ZERO;
    if (m < M) {
      const int n = m / (g.OH * g.OW), rem = m - n * g.OH * g.OW, oh = rem / g.OW,
                ow = rem - oh * g.OW;
      va[j] = (((n * g.IH + oh * g.stride) * g.IW + ow * g.stride) * g.Cin) * 2 +
              src_quad(dr.row[j], lane);
    }
    vb[j] = ((n0 + dr.row[j]) * 9 * g.Cin) * 2 + src_quad(dr.row[j], lane);
  }
  const int cpt = g.Cin / BKE;   // chunks per tap
  f32x16 acc[2][2];
  zero_acc(acc);
  dma_mainloop<NS>(9 * cpt, smem, acc, wid, lane, [&](int c, unsigned char *stage) {
    const int tap = c / cpt, h = c - tap * cpt, kh = tap / 3, kw = tap - kh * 3;
    const int sa = ((kh * g.IW + kw) * g.Cin + h * BKE) * 2, sb = (tap * g.Cin + h * BKE) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      pipe::dma16(ra, reinterpret_cast<float *>(stage + dr.lds[j]), va[j], sa);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      pipe::dma16(rb, reinterpret_cast<float *>(stage + TILE_B + dr.lds[j]), vb[j], sb);
  });
  const int hw = g.OH * g.OW;
  tile_epilogue(reinterpret_cast<float *>(smem), acc, wid, lane,
                [&](int row, int col, const float (&v)[8]) {
    const int m = m0 + row, n = n0 + col;
    if (m >= M) return;
    const float4 b0 = pipe::lds4(bias + n), b1 = pipe::lds4(bias + n + 4);
    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = fmaxf(v[e] + bb[e], 0.f);
    const size_t at = (size_t)m * g.Cout + n;
    *reinterpret_cast<uint4 *>(out_h + at) =
        make_uint4(pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7]));
    if (out_f) {
      *reinterpret_cast<float4 *>(out_f + at) = make_float4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<float4 *>(out_f + at + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
    if (out_post) {   // + the per-(channel, pixel) embedding bias, (Cout, OH, OW)
      const float *pb = post_bias + (size_t)n * hw + m % hw;
      float p[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) p[e] = o[e] + pb[(size_t)e * hw];
      *reinterpret_cast<float4 *>(out_post + at) = make_float4(p[0], p[1], p[2], p[3]);
      *reinterpret_cast<float4 *>(out_post + at + 4) = make_float4(p[4], p[5], p[6], p[7]);
    }
  });
}

// ---- data gradient -------------------------------------------------------------------------
// din[pixel][ci] = gate(sum over the taps that reach the pixel of dpre[.][co] wd[ci][tap][co]).
// An input row ih receives tap kh iff (ih - kh) is a multiple of the stride with (ih - kh) / s
// < OH: near the border -- and for one stride parity -- most of the nine taps miss (on a 7 x 7
// input of a stride-1 layer 49 % of the (pixel, tap) pairs of the gather form are structural
// zeros).  Rows are grouped into classes of equal valid-kh set, columns likewise; a tile holds
// pixels of ONE (row class, column class) pair and walks exactly that pair's taps: no MFMA
// multiplies a structural zero and the operand addresses need no per-tap range checks.
constexpr int DG_MAXCLS = 6;     // classes per axis (5 at stride 1, 2 at stride 2)
// every class is an arithmetic progression: i = first + step a, a < count (stride 1: the runs
// {0}, {1}, {2 .. O - 1}, {O}, {O + 1}, step 1; stride 2: the two parities, step 2)
struct DgradPlan {
  int nrc, ncc, step;
  unsigned char rmask[DG_MAXCLS], cmask[DG_MAXCLS];   // bit kh (kw): the tap reaches the class
  short rcount[DG_MAXCLS], ccount[DG_MAXCLS], rfirst[DG_MAXCLS], cfirst[DG_MAXCLS];
  int tile_start[DG_MAXCLS * DG_MAXCLS + 1];          // first tile (grid.y) of class pair z
};
// classes of one axis, largest tap sets first (the tiles with the longest K loops are
// dispatched first); returns their number or -1
inline int dgrad_axis(int I, int O, int stride, unsigned char *mask, short *count, short *first) {
  auto taps = [&](int i) {
    int m = 0;
    for (int k = 0; k < 3; ++k) {
      const int d = i - k;
      if (d >= 0 && d % stride == 0 && d / stride < O) m |= 1 << k;
    }
    return m;
  };
  static const int order[8] = {7, 3, 5, 6, 1, 2, 4, 0};
  int n = 0;
  // stride 1: one class per distinct tap set (interior + the border rows).  Stride 2: one
  // class per parity -- the tap sets differ mainly by parity there, and splitting the single
  // border row off each parity only fragments the tiles (measured at B = 1024, 19 x 19: 73.7 us
  // against 69) --, the class mask is the union and the operand fetch range-checks
  for (int wi = 0; wi < (stride == 1 ? 8 : stride); ++wi) {
    int cnt = 0, m = 0, f = -1, last = -1;
    for (int i = 0; i < I; ++i)
      if ((stride == 1 ? taps(i) == order[wi] : i % stride == wi)) {
        if (f < 0) f = i;
        if (last >= 0 && i != last + stride) return -1;   // (not a progression: never)
        last = i, ++cnt, m |= taps(i);
      }
    if (!cnt) continue;
    if (!m) return -1;   // a row no tap reaches (stride 1: impossible)
    if (n == DG_MAXCLS) return -1;
    mask[n] = (unsigned char)m, count[n] = (short)cnt, first[n] = (short)f;
    ++n;
  }
  return n;
}
__device__ __forceinline__ int nth_bit(int mask, int n) {   // n-th set bit of a 3-bit mask
  const int k0 = (mask & 1) ? 0 : ((mask & 2) ? 1 : 2);
  if (n == 0) return k0;
  const int rest = mask & ~(1 << k0);
  return (n == 1 && (rest & 2)) ? 1 : 2;
}
// gate: the producing layer's ReLU output (bf16, nullable); din_h (bf16) and / or din_f (fp32)
// S2: stride 2 (merged parity classes: per-lane range checks); stride 1: every tap of a class
// reaches every pixel of it, so a lane's source offset is fixed and the tap moves the
// wave-uniform offset only
template <int NS, bool S2>
__global__ __launch_bounds__(NT, 2) void conv_dgrad_bf16r_kernel(
    const bf16_t *__restrict__ dpre, const bf16_t *__restrict__ wd, const bf16_t *__restrict__ gate,
    bf16_t *__restrict__ din_h, float *__restrict__ din_f, ConvGeom g, DgradPlan pl) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int nz = pl.nrc * pl.ncc;
  int z = 0;
  for (int c = 1; c < nz; ++c)   // (workgroup-uniform)
    if ((int)blockIdx.y >= pl.tile_start[c]) z = c;
  const int rc = z / pl.ncc, cc = z - rc * pl.ncc;
  const int AH = pl.rcount[rc], AW = pl.ccount[cc], M = g.B * AH * AW;
  const int rm = pl.rmask[rc], cm = pl.cmask[cc], nkh = __popc(rm), nkw = __popc(cm);
  const int m0 = ((int)blockIdx.y - pl.tile_start[z]) * TM, n0 = blockIdx.x * TN;
  const pipe::rsrc_t ra = pipe::make_rsrc(dpre, (unsigned)((size_t)g.B * g.OH * g.OW * g.Cout * 2));
  const pipe::rsrc_t rb = pipe::make_rsrc(wd, (unsigned)((size_t)g.Cin * 9 * g.Cout * 2));
  const DmaRows dr = dma_rows(wid, lane);
  // the class pair's taps; the LAST one (largest kh, kw) is the lanes' reference position
  const int khl = nth_bit(rm, nkh - 1), kwl = nth_bit(cm, nkw - 1);
  int pn[4], pih[4], piw[4], va[4], vb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + dr.row[j];
    pn[j] = -1, pih[j] = 0, piw[j] = 0, va[j] = pipe::DMA_ZERO;
    if (m < M) {
      const int n = m / (AH * AW), rem = m - n * AH * AW, a = rem / AW, b = rem - a * AW;
      pn[j] = n * g.OH * g.OW;
      pih[j] = pl.rfirst[rc] + pl.step * a, piw[j] = pl.cfirst[cc] + pl.step * b;
      if (!S2)   // output pixel of the reference tap: inside for every pixel of the class
        va[j] = ((pn[j] + (pih[j] - khl) * g.OW + piw[j] - kwl) * g.Cout) * 2 +
                src_quad(dr.row[j], lane);
    }
    vb[j] = ((n0 + dr.row[j]) * 9 * g.Cout) * 2 + src_quad(dr.row[j], lane);
  }
  const int cpt = g.Cout / BKE;
  f32x16 acc[2][2];
  zero_acc(acc);
  dma_mainloop<NS>(nkh * nkw * cpt, smem, acc, wid, lane, [&](int c, unsigned char *stage) {
    const int t = c / cpt, h = c - t * cpt, ti = t / nkw, tj = t - ti * nkw;
    const int kh = nth_bit(rm, ti), kw = nth_bit(cm, tj);
    const int sb = ((kh * 3 + kw) * g.Cout + h * BKE) * 2;
    if (!S2) {
      // tap (kh, kw) reads (khl - kh) rows and (kwl - kw) columns behind the reference
      const int sa = (((khl - kh) * g.OW + kwl - kw) * g.Cout + h * BKE) * 2;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        pipe::dma16(ra, reinterpret_cast<float *>(stage + dr.lds[j]), va[j], sa);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // (ih - kh) is even for every tap of the parity class; the range check fails on the
        // class's border row / column only
        const int dh = pih[j] - kh, dw = piw[j] - kw, oh = dh >> 1, ow = dw >> 1;
        const bool ok = pn[j] >= 0 && dh >= 0 && dw >= 0 && oh < g.OH && ow < g.OW;
        const int v = ok ? ((pn[j] + oh * g.OW + ow) * g.Cout) * 2 + src_quad(dr.row[j], lane)
                         : pipe::DMA_ZERO;
        pipe::dma16(ra, reinterpret_cast<float *>(stage + dr.lds[j]), v, h * BKE * 2);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      pipe::dma16(rb, reinterpret_cast<float *>(stage + TILE_B + dr.lds[j]), vb[j], sb);
  });
  tile_epilogue(reinterpret_cast<float *>(smem), acc, wid, lane,
                [&](int row, int col, const float (&v)[8]) {
    const int m = m0 + row;
    if (m >= M) return;
    const int n = m / (AH * AW), rem = m - n * AH * AW, a = rem / AW, b = rem - a * AW;
    const int ih = pl.rfirst[rc] + pl.step * a, iw = pl.cfirst[cc] + pl.step * b;
    const size_t o = (((size_t)n * g.IH + ih) * g.IW + iw) * g.Cin + n0 + col;
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = v[e];
    if (gate) {
      const uint4 gt = *reinterpret_cast<const uint4 *>(gate + o);
      const unsigned gw[4] = {gt.x, gt.y, gt.z, gt.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r[2 * e] = bf_lo(gw[e]) > 0.f ? r[2 * e] : 0.f;
        r[2 * e + 1] = bf_hi(gw[e]) > 0.f ? r[2 * e + 1] : 0.f;
      }
    }
    if (din_h)
      *reinterpret_cast<uint4 *>(din_h + o) =
          make_uint4(pack2(r[0], r[1]), pack2(r[2], r[3]), pack2(r[4], r[5]), pack2(r[6], r[7]));
    if (din_f) {
      *reinterpret_cast<float4 *>(din_f + o) = make_float4(r[0], r[1], r[2], r[3]);
      *reinterpret_cast<float4 *>(din_f + o + 4) = make_float4(r[4], r[5], r[6], r[7]);
    }
  });
}

// ---- weight gradient -----------------------------------------------------------------------
// partial[(split * 9 + tap)][co][ci] = sum over the split's pixels m of dpre[m][co] x[pix(m, tap)][ci];
// bias partials [split][co] behind the 9 * splits slabs (the layout of the fp32 kernels: the
// same reduction launch sums them).  grid (Cin / 128, Cout / 128, 9 * splits).
// Both operands are k-strided in memory ([pixel][channel]).
struct PixelWalk {   // (n, oh, ow) of a pixel index, advanced without divisions
  int n, oh, ow;
};
__device__ __forceinline__ void walk(PixelWalk &p, int dq, int drm, const ConvGeom &g) {
  p.ow += drm;
  const int carry = p.ow >= g.OW;
  p.ow -= carry ? g.OW : 0;
  p.oh += dq + carry;
  while (p.oh >= g.OH) p.oh -= g.OH, ++p.n;
}
// A first form loaded (4 pixels) x (4 channels) blocks, transposed them in registers (v_perm_b32)
// and wrote 8-byte runs into a k-contiguous image: 16 8-byte loads, 32 v_perm and 16
// ds_write_b64 per thread and chunk -- 150 us for the three layers at B = 1024 against 121 for
// what follows.  gfx950's ds_read_b64_tr_b16 hands a lane
// COLUMN t of a 4 x 16 block of 16-bit elements whose rows the 16 lanes of its group address
// (tools/probes/tr_read.cpp: any row stride) -- the MFMA fragment of a k-strided operand
// straight from a [pixel][channel] image.  So both operands go global -> LDS by DMA exactly as
// they lie in memory (a pixel's 128 channels = 256 bytes = 16 lanes x 16 bytes; a 1 KiB piece
// = 4 pixels), and the fragments are two transposing reads each:
//   lane l = 16 g + t of a wave: channel 16 (g & 1) + t of the 32-row block, k half g >> 1;
//   it supplies the address of pixel k0 + (t >> 2), channels 16 (g & 1) + 4 (t & 3) .. + 3 and
//   receives its channel at pixels k0 .. k0 + 3 (k0 = 8 (g >> 1), then + 4).
// The sixteen 16-byte units of a pixel row are XOR-swizzled with a bit permutation of the row
// index (on the DMA's source address), so that the four rows a lane group reads -- and the rows
// of the other groups -- lie in different banks.  The bias gradient (column sums of dpre) is
// one more MFMA per fragment against a tile of ones in the tap-0 workgroups.
__device__ __forceinline__ int swz_px(int row) {   // 4-bit XOR mask of pixel row `row`
  return ((row & 3) << 1) | ((row >> 2) & 1) | (row & 8);
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char *tile, int c0, int k0, int t) {
  // channels c0 + 4 (t & 3) .. + 3 of pixel rows k0 + (t >> 2) and k0 + 4 + (t >> 2)
  const int ch = c0 + 4 * (t & 3), u = ch >> 3, half = (ch >> 2) & 1;
  const int r0 = k0 + (t >> 2), r1 = r0 + 4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4 *)(tile + r0 * 256 + ((u ^ swz_px(r0)) << 4) + half * 8));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4 *)(tile + r1 * 256 + ((u ^ swz_px(r1)) << 4) + half * 8));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}
template <int NS>
__global__ __launch_bounds__(NT, 2) void conv_wgrad_bf16r_tr_kernel(
    const bf16_t *__restrict__ dpre, const bf16_t *__restrict__ x, float *__restrict__ partial,
    ConvGeom g, int splits) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wid >> 1, wn = wid & 1, t = lane & 15, grp = lane >> 4;
  const int M = g.B * g.OH * g.OW;
  const int tap = blockIdx.z % 9, split = blockIdx.z / 9, kh = tap / 3, kw = tap - kh * 3;
  const int per = ((M + splits - 1) / splits + BKE - 1) / BKE * BKE;
  const int kbeg = split * per, kend = min(M, kbeg + per);
  const int co0 = blockIdx.y * TM, ci0 = blockIdx.x * TN;
  const bool want_bias = tap == 0 && blockIdx.x == 0;   // (workgroup-uniform)
  const pipe::rsrc_t ra = pipe::make_rsrc(dpre, (unsigned)((size_t)M * g.Cout * 2));
  const pipe::rsrc_t rb = pipe::make_rsrc(x, (unsigned)((size_t)g.B * g.IH * g.IW * g.Cin * 2));
  // DMA: piece j of wave w = pixel rows 4 (4 w + j) .. + 3 of the chunk; lane l the row
  // 4 (4 w + j) + (l >> 4) and the LDS unit l & 15 of it
  int prow[4], usrc[4];
  PixelWalk pwk[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    prow[j] = 4 * (4 * wid + j) + (lane >> 4);
    usrc[j] = ((lane & 15) ^ swz_px(prow[j])) << 4;   // byte offset of the source unit
    const int m = kbeg + prow[j];
    pwk[j].n = m / (g.OH * g.OW);
    const int rem = m - pwk[j].n * g.OH * g.OW;
    pwk[j].oh = rem / g.OW, pwk[j].ow = rem - pwk[j].oh * g.OW;
  }
  const int dq = BKE / g.OW, drm = BKE - dq * g.OW;
  f32x16 acc[2][2], accb[2];
  zero_acc(acc);
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int e = 0; e < 16; ++e) accb[tt][e] = 0.f;
  const int nchunk = kbeg < kend ? (kend - kbeg + BKE - 1) / BKE : 0;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  dma_loop<NS>(nchunk, smem,
               [&](int c, unsigned char *stage) {   // (called once per chunk, c ascending)
    const int k0 = kbeg + c * BKE;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = k0 + prow[j];
      const bool ok = m < kend;
      const int va = ok ? (m * g.Cout + co0) * 2 + usrc[j] : pipe::DMA_ZERO;
      const int pix = (pwk[j].n * g.IH + pwk[j].oh * g.stride + kh) * g.IW + pwk[j].ow * g.stride + kw;
      const int vb = ok ? (pix * g.Cin + ci0) * 2 + usrc[j] : pipe::DMA_ZERO;
      pipe::dma16(ra, reinterpret_cast<float *>(stage + (4 * wid + j) * 1024), va, 0);
      pipe::dma16(rb, reinterpret_cast<float *>(stage + TILE_B + (4 * wid + j) * 1024), vb, 0);
      walk(pwk[j], dq, drm, g);
    }
  },
               [&](const unsigned char *stage) {
#pragma unroll
    for (int s16 = 0; s16 < BKE / 16; ++s16) {
      const int k0 = 16 * s16 + 8 * (grp >> 1);
      bf16x8 a[2], b[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        a[tt] = tr_frag(stage, wm * 64 + tt * 32 + 16 * (grp & 1), k0, t);
        b[tt] = tr_frag(stage + TILE_B, wn * 64 + tt * 32 + 16 * (grp & 1), k0, t);
      }
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          acc[tt][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tt], b[u], acc[tt][u], 0, 0, 0);
      if (want_bias && wn == 0) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
          accb[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tt], ones, accb[tt], 0, 0, 0);
      }
    }
  });
  float *dst = partial + (size_t)(split * 9 + tap) * g.Cout * g.Cin;
  tile_epilogue(reinterpret_cast<float *>(smem), acc, wid, lane,
                [&](int row, int col, const float (&v)[8]) {
    float *o = dst + (size_t)(co0 + row) * g.Cin + ci0 + col;
    *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4 *>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
  });
  if (want_bias && wn == 0 && (lane & 31) == 0) {   // column 0 of the ones product
    const int kk = lane >> 5;
    float *db = partial + (size_t)splits * 9 * g.Cout * g.Cin + (size_t)split * g.Cout + co0;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        db[wm * 64 + tt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk] = accb[tt][e];
  }
}

// ---- helpers: fp32 -> bf16 copies -----------------------------------------------------------
// up to 8 arrays in one launch (grid.y = array): activations of the image layer, the top
// pre-activation gradient, the re-laid-out filters
struct CvtBatch {
  const float *src[8];
  bf16_t *dst[8];
  long n[8];   // multiples of 8
};
__global__ __launch_bounds__(NT) void cvt_bf16_kernel(CvtBatch c) {
  const int a = blockIdx.y;
  const long n8 = c.n[a] >> 3;
  for (long e = (long)blockIdx.x * NT + threadIdx.x; e < n8; e += (long)gridDim.x * NT) {
    const float4 v0 = pipe::lds4(c.src[a] + 8 * e), v1 = pipe::lds4(c.src[a] + 8 * e + 4);
    *reinterpret_cast<uint4 *>(c.dst[a] + 8 * e) =
        make_uint4(pack2(v0.x, v0.y), pack2(v0.z, v0.w), pack2(v1.x, v1.y), pack2(v1.z, v1.w));
  }
}

int check(const ConvGeom &g) {
  if (g.B <= 0 || g.IH < 3 || g.IW < 3 || g.Cin <= 0 || g.Cout <= 0 || g.stride < 1 || g.stride > 2)
    return SCAE_ERR_BAD_ARG;
  if (g.Cin % 128 || g.Cout % 128) return SCAE_ERR_UNSUPPORTED;
  // (the DMA offsets are 32-bit byte offsets)
  const size_t big = (size_t)g.B * g.IH * g.IW * (g.Cin > g.Cout ? g.Cin : g.Cout) * 2;
  if (big >= (1ull << 31) - 4096) return SCAE_ERR_UNSUPPORTED;
  return SCAE_OK;
}
template <class K>
int raise_lds(K kernel, int bytes) {
  if (bytes <= 48 * 1024) return SCAE_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  return e == hipSuccess ? SCAE_OK : (int)e;
}
// ring depth of the DMA-staged passes (tuning aid: SCAE_BF16R_NS=1..4)
int ring_depth(const char *env, int dflt) {
  const char *e = getenv(env);
  const int v = e && *e ? atoi(e) : dflt;
  return v < 1 ? 1 : (v > 4 ? 4 : v);
}
}  // namespace

extern "C" int scae_conv3x3_bf16r_supported(int B, int IH, int IW, int Cin, int Cout, int stride) {
  if (IH < 3 || IW < 3 || stride < 1) return 0;
  ConvGeom g{B, IH, IW, (IH - 3) / stride + 1, (IW - 3) / stride + 1, Cin, Cout, stride};
  return check(g) == SCAE_OK ? 1 : 0;
}

extern "C" int scae_cvt_bf16_batch(int n_arrays, const float *const *src, uint16_t *const *dst,
                                   const int64_t *n, void *stream) {
  SCAE_REQUIRE(n_arrays > 0 && n_arrays <= 8 && src && dst && n);
  CvtBatch c{};
  long nmax = 0;
  for (int a = 0; a < n_arrays; ++a) {
    SCAE_REQUIRE(src[a] && dst[a] && n[a] > 0 && n[a] % 8 == 0);
    c.src[a] = src[a], c.dst[a] = dst[a], c.n[a] = n[a];
    nmax = nmax > n[a] ? nmax : n[a];
  }
  long blocks = (nmax / 8 + NT - 1) / NT;
  blocks = blocks > 4096 ? 4096 : blocks;
  scae::launch(cvt_bf16_kernel, dim3((unsigned)blocks, n_arrays), dim3(NT), 0, (hipStream_t)stream, c);
  return scae_launch_status();
}

extern "C" int scae_conv3x3_fwd_bf16r(const uint16_t *in, const uint16_t *wf, const float *bias,
                                      uint16_t *out_h, float *out_f, const float *post_bias,
                                      float *out_post, int B, int IH, int IW, int Cin, int Cout,
                                      int stride, void *stream) {
  SCAE_REQUIRE(in && wf && bias && out_h && (!out_post || post_bias) && IH >= 3 && IW >= 3 && stride >= 1);
  ConvGeom g{B, IH, IW, (IH - 3) / stride + 1, (IW - 3) / stride + 1, Cin, Cout, stride};
  int rc = check(g);
  if (rc) return rc;
  const int M = B * g.OH * g.OW;
  const dim3 grid(Cout / TN, (M + TM - 1) / TM);
#define SCAE_FWD_NS(N)                                                                          \
  case N:                                                                                       \
    if ((rc = raise_lds(conv_fwd_bf16r_kernel<N>, N * STAGE_B))) return rc;                     \
    scae::launch(conv_fwd_bf16r_kernel<N>, grid, dim3(NT), N * STAGE_B, (hipStream_t)stream, in, \
                 wf, bias, out_h, out_f, post_bias, out_post, g);                                \
    break;
  // many tiles: one stage, five workgroups per CU covering each other; few: two stages
  // (measured at B = 1024: 648 tiles 37.5 us against 41.6, 392 tiles 27.1 against 23.3)
  switch (ring_depth("SCAE_BF16R_FWD_NS", (long)grid.x * grid.y > 512 ? 1 : 2)) {
    SCAE_FWD_NS(1) SCAE_FWD_NS(2) SCAE_FWD_NS(3) SCAE_FWD_NS(4)
  }
#undef SCAE_FWD_NS
  return scae_launch_status();
}

extern "C" int scae_conv3x3_dgrad_bf16r(const uint16_t *dpre, const uint16_t *wd,
                                        const uint16_t *gate, uint16_t *din_h, float *din_f, int B,
                                        int IH, int IW, int Cin, int Cout, int stride,
                                        void *stream) {
  SCAE_REQUIRE(dpre && wd && (din_h || din_f) && IH >= 3 && IW >= 3 && stride >= 1);
  ConvGeom g{B, IH, IW, (IH - 3) / stride + 1, (IW - 3) / stride + 1, Cin, Cout, stride};
  int rc = check(g);
  if (rc) return rc;
  DgradPlan pl{};
  pl.step = stride;
  pl.nrc = dgrad_axis(IH, g.OH, stride, pl.rmask, pl.rcount, pl.rfirst);
  pl.ncc = dgrad_axis(IW, g.OW, stride, pl.cmask, pl.ccount, pl.cfirst);
  if (pl.nrc <= 0 || pl.ncc <= 0) return SCAE_ERR_UNSUPPORTED;
  int tiles = 0, covered = 0;
  for (int rc_ = 0; rc_ < pl.nrc; ++rc_)
    for (int cc_ = 0; cc_ < pl.ncc; ++cc_) {
      pl.tile_start[rc_ * pl.ncc + cc_] = tiles;
      tiles += (B * pl.rcount[rc_] * pl.ccount[cc_] + TM - 1) / TM;
      covered += pl.rcount[rc_] * pl.ccount[cc_];
    }
  pl.tile_start[pl.nrc * pl.ncc] = tiles;
  if (covered != IH * IW) return SCAE_ERR_UNSUPPORTED;   // (every pixel is in one class pair)
  const dim3 grid(Cin / TN, tiles);
#define SCAE_DG_NS(N)                                                                           \
  case N:                                                                                       \
    if (stride == 2) {                                                                          \
      if ((rc = raise_lds(conv_dgrad_bf16r_kernel<N, true>, N * STAGE_B))) return rc;           \
      scae::launch((conv_dgrad_bf16r_kernel<N, true>), grid, dim3(NT), N * STAGE_B,             \
                   (hipStream_t)stream, dpre, wd, gate, din_h, din_f, g, pl);                    \
    } else {                                                                                    \
      if ((rc = raise_lds(conv_dgrad_bf16r_kernel<N, false>, N * STAGE_B))) return rc;          \
      scae::launch((conv_dgrad_bf16r_kernel<N, false>), grid, dim3(NT), N * STAGE_B,            \
                   (hipStream_t)stream, dpre, wd, gate, din_h, din_f, g, pl);                    \
    }                                                                                           \
    break;
  switch (ring_depth("SCAE_BF16R_DGRAD_NS", (long)grid.x * grid.y > 512 ? 1 : 2)) {
    SCAE_DG_NS(1) SCAE_DG_NS(2) SCAE_DG_NS(3) SCAE_DG_NS(4)
  }
#undef SCAE_DG_NS
  return scae_launch_status();
}

extern "C" int scae_conv3x3_wgrad_bf16r_splits(int B, int OH, int OW, int Cin, int Cout) {
  if (B <= 0 || OH <= 0 || OW <= 0 || Cin <= 0 || Cout <= 0 || Cin % 128 || Cout % 128) return 0;
  // two workgroups per CU over the 9 taps x channel tiles (measured at B = 1024, the three
  // layers: 256 workgroups 166 us, 384 132, 512 121, 768 126, 1024 145; fewer splits also
  // mean fewer partial slabs to write and reduce); at least 4 chunks of K each
  const long tiles = 9L * (Cin / 128) * (Cout / 128), M = (long)B * OH * OW;
  long s = 512 / tiles;
  const long cap = M / (4 * BKE);
  s = s > cap ? cap : s;
  return (int)(s < 1 ? 1 : (s > 128 ? 128 : s));
}

extern "C" int scae_conv3x3_wgrad_bf16r(const uint16_t *dpre, const uint16_t *x, float *partial,
                                        int B, int IH, int IW, int Cin, int Cout, int stride,
                                        void *stream) {
  SCAE_REQUIRE(dpre && x && partial && IH >= 3 && IW >= 3 && stride >= 1);
  ConvGeom g{B, IH, IW, (IH - 3) / stride + 1, (IW - 3) / stride + 1, Cin, Cout, stride};
  int rc = check(g);
  if (rc) return rc;
  const int splits = scae_conv3x3_wgrad_bf16r_splits(B, g.OH, g.OW, Cin, Cout);
  const dim3 grid(Cin / TN, Cout / TM, 9 * splits);
#define SCAE_WG_NS(N)                                                                           \
  case N:                                                                                       \
    if ((rc = raise_lds(conv_wgrad_bf16r_tr_kernel<N>, N * STAGE_B))) return rc;                \
    scae::launch(conv_wgrad_bf16r_tr_kernel<N>, grid, dim3(NT), N * STAGE_B,                    \
                 (hipStream_t)stream, dpre, x, partial, g, splits);                             \
    break;
  switch (ring_depth("SCAE_BF16R_WGRAD_NS", 1)) {
    SCAE_WG_NS(1) SCAE_WG_NS(2) SCAE_WG_NS(3) SCAE_WG_NS(4)
  }
#undef SCAE_WG_NS
  return scae_launch_status();
}
