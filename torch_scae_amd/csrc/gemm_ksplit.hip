// Small fp32 GEMMs with LONG K and few tiles -- the backward of the part encoder's 1 x 1 attention
// convolution (part_encoder.py:71-73: dx = dy W gated, dW = dy^T x per group of images + its bias
// sums) -- on wave-private pipelines: the form conv_mfma.hip's dgk::pipeline gave the small
// convolution layers.  A 32 x 64 output tile per workgroup; EVERY wave computes the whole tile for
// a quarter of the 32-deep K chunks (wave w: chunks w, w + 4, ..) from a 12 KiB stage of its own
// (its own DMA pieces, its own vmcnt, no workgroup barrier in the loop; the next chunk's DMA flies
// under the second 16-deep step's splits and MFMAs), products as the six exact bf16 partial
// products of bf16x6.h; the four partial tiles meet in LDS and are summed in a fixed order.
//     C[z](M x N) = epilogue( A[z](M x K) B[z](N x K)^T ),  B k-strided (B[k * ldb + n]),
//     A k-contiguous (A[m * lda + k]; gate mask + ungated copy in the epilogue) or k-strided
//     (A[k * lda + m]; asum[m] = sum_k A(m, k) from the staged operand)
// -- the two layouts of that backward; M % 32 == 0, N % 64 == 0, K % 32 == 0 for a k-contiguous A (the k
// rows past K of k-strided operands are DMA zeros), 16-byte aligned rows,
// no bias / ReLU.  Anything else: SCAE_ERR_UNSUPPORTED, and the caller (gemm_mfma.hip) takes the
// register-staged tiles of mfma_tile.h.
#include <cstdlib>

#include "mfma_pipe.h"

#ifndef SCAE_GEMM_KSPLIT_DEFAULT
#define SCAE_GEMM_KSPLIT_DEFAULT 1   // SCAE_GEMM_KSPLIT = 1 / 0 at run time
#endif

namespace {
namespace pipe = scae_pipe;
using scae_x6::Split3;
constexpr int NT = 256, TM = 32, TN = 64, BKF = 32;
constexpr int A_B = TM * BKF * 4, B_B = TN * BKF * 4, WAVE_B = A_B + B_B;   // 4 + 8 KiB per wave
constexpr int LDS = 68;                                                     // slab row stride
constexpr int SMEM = 4 * WAVE_B / 4;
static_assert(TM * LDS * 4 <= WAVE_B, "a wave's slab aliases its stage");

struct Prob {
  const float *A, *B, *mask;
  float *C, *craw, *asum;
  long a_batch, b_batch, c_batch, mask_batch, asum_batch;
  int lda, ldb, ldc, ldmask, asum_ld;
  int M, N, K, akc;   // akc: A k-contiguous
  unsigned a_bytes, b_bytes;   // extents of one batch entry of A / B
  int tx, ty, first;           // tiles along N / M, first block
};
struct Probs {
  Prob p[4];
  int n;
};

__device__ __forceinline__ int swz(int row, int q) { return q ^ ((row >> 1) & 7); }

__global__ __launch_bounds__(NT) void gemm_x6k_kernel(Probs ps) {
  __shared__ __attribute__((aligned(1024))) float smemf[SMEM];
  int which = 0;
  while (which + 1 < ps.n && (int)blockIdx.x >= ps.p[which + 1].first) ++which;
  const Prob &g = ps.p[which];
  const int id = (int)blockIdx.x - g.first, per = g.tx * g.ty;
  const int z = id / per, rem = id - z * per, by = rem / g.tx, bx = rem - by * g.tx;
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int i = lane & 31, kk = lane >> 5;
  const int m0 = by * TM, n0 = bx * TN;
  unsigned char *stage = reinterpret_cast<unsigned char *>(smemf) + wid * WAVE_B;
  const float *As = reinterpret_cast<const float *>(stage);
  const float *Bs = reinterpret_cast<const float *>(stage + A_B);
  const pipe::rsrc_t ra = pipe::make_rsrc(g.A + z * g.a_batch, g.a_bytes);
  const pipe::rsrc_t rb = pipe::make_rsrc(g.B + z * g.b_batch, g.b_bytes);
  const bool akc = g.akc != 0;   // workgroup-uniform
  // A pieces (4 of 1 KiB): 8 rows of 128 bytes.  k-contiguous: rows = m, quads swizzled on the
  // source side; k-strided: rows = k, four m per quad, as in memory.
  // B pieces (8): 4 k-rows of 256 bytes (64 n), as in memory.
  const int ar = lane >> 3, aq = lane & 7, br = lane >> 4, bq = lane & 15;
  int va[4], vb[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 8 * j + ar;
    va[j] = akc ? ((m0 + row) * g.lda) * 4 + swz(row, aq) * 16 : (row * g.lda + m0) * 4 + aq * 16;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) vb[j] = ((4 * j + br) * g.ldb + n0) * 4 + bq * 16;
  const int sa_step = akc ? BKF * 4 : BKF * g.lda * 4, sb_step = BKF * g.ldb * 4;
  // (the chunk offset rides in the per-lane offset: the descriptor's range check, which makes the
  // k rows past K of a k-strided operand read zeros, does not see the scalar offset)
  auto issue = [&](int c) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      pipe::dma16(ra, reinterpret_cast<float *>(stage + j * 1024), va[j] + c * sa_step, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      pipe::dma16(rb, reinterpret_cast<float *>(stage + A_B + j * 1024), vb[j] + c * sb_step, 0);
  };
  pipe::f32x16 acc[2], accl[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[u][e] = 0.f, accl[u][e] = 0.f;
  struct Raw {
    float v[8];
  };
  auto raw_a = [&](int s) {   // the lane's 8 k of row m = i
    Raw r;
    if (akc) {
      const int q = 4 * s + 2 * kk, sw = (i >> 1) & 7;
      const float4 lo = pipe::lds4(As + i * BKF + ((q ^ sw) << 2));
      const float4 hi = pipe::lds4(As + i * BKF + (((q + 1) ^ sw) << 2));
      r.v[0] = lo.x, r.v[1] = lo.y, r.v[2] = lo.z, r.v[3] = lo.w;
      r.v[4] = hi.x, r.v[5] = hi.y, r.v[6] = hi.z, r.v[7] = hi.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) r.v[e] = As[(16 * s + 8 * kk + e) * TM + i];
    }
    return r;
  };
  auto raw_b = [&](int s, int u) {   // the lane's 8 k of column n = 32 u + i
    Raw r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r.v[e] = Bs[(16 * s + 8 * kk + e) * TN + u * 32 + i];
    return r;
  };
  auto mma = [&](const Raw &ar_, const Raw (&br_)[2]) {
    const Split3 a = scae_x6::split3(ar_.v);
    Split3 b[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) b[u] = scae_x6::split3(br_[u].v);
#define SCAE_GK_MMA(AP, BP, ACC)                                                             \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) ACC[u] =                                     \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.AP, b[u].BP, ACC[u], 0, 0, 0)
    SCAE_GK_MMA(hi, lo, accl);
    SCAE_GK_MMA(lo, hi, accl);
    SCAE_GK_MMA(mid, mid, accl);
    SCAE_GK_MMA(hi, mid, accl);
    SCAE_GK_MMA(mid, hi, accl);
    SCAE_GK_MMA(hi, hi, acc);
#undef SCAE_GK_MMA
  };
  const bool want_asum = !akc && g.asum && bx == 0;   // workgroup-uniform
  float asum = 0.f;   // lane < 32: sum over this wave's chunks of A(m0 + lane, k)
  const int nchunk = (g.K + BKF - 1) / BKF;   // (a ragged last chunk: k-strided operands only)
  if (wid < nchunk) issue(wid);
  for (int c = wid; c < nchunk; c += 4) {   // (wave-uniform)
    pipe::wait_vm<0>();   // this wave's own pieces: nobody else writes or reads its stage
    if (want_asum) {
#pragma unroll 8
      for (int k = 0; k < BKF; ++k) asum += As[k * TM + i];
    }
    Raw ar_ = raw_a(0), br_[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) br_[u] = raw_b(0, u);
    mma(ar_, br_);
    ar_ = raw_a(1);
#pragma unroll
    for (int u = 0; u < 2; ++u) br_[u] = raw_b(1, u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the stage is free
    if (c + 4 < nchunk) issue(c + 4);
    mma(ar_, br_);
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
  {
    const int row = tid >> 3, c8 = 8 * (tid & 7);
    const float *p0 = smemf + row * LDS + c8;
    constexpr int WS = WAVE_B / 4;
    const float4 a0 = pipe::lds4(p0), a1 = pipe::lds4(p0 + 4);
    const float4 b0 = pipe::lds4(p0 + WS), b1 = pipe::lds4(p0 + WS + 4);
    const float4 c0 = pipe::lds4(p0 + 2 * WS), c1 = pipe::lds4(p0 + 2 * WS + 4);
    const float4 d0 = pipe::lds4(p0 + 3 * WS), d1 = pipe::lds4(p0 + 3 * WS + 4);
    float4 v0 = make_float4((a0.x + b0.x) + (c0.x + d0.x), (a0.y + b0.y) + (c0.y + d0.y),
                            (a0.z + b0.z) + (c0.z + d0.z), (a0.w + b0.w) + (c0.w + d0.w));
    float4 v1 = make_float4((a1.x + b1.x) + (c1.x + d1.x), (a1.y + b1.y) + (c1.y + d1.y),
                            (a1.z + b1.z) + (c1.z + d1.z), (a1.w + b1.w) + (c1.w + d1.w));
    const size_t o = (size_t)z * g.c_batch + (size_t)(m0 + row) * g.ldc + n0 + c8;
    if (g.craw) {   // the values before the gate (layout of C)
      *reinterpret_cast<float4 *>(g.craw + o) = v0;
      *reinterpret_cast<float4 *>(g.craw + o + 4) = v1;
    }
    if (g.mask) {
      const float *mp = g.mask + (size_t)z * g.mask_batch + (size_t)(m0 + row) * g.ldmask + n0 + c8;
      const float4 k0 = *reinterpret_cast<const float4 *>(mp),
                   k1 = *reinterpret_cast<const float4 *>(mp + 4);
      v0 = make_float4(k0.x > 0.f ? v0.x : 0.f, k0.y > 0.f ? v0.y : 0.f, k0.z > 0.f ? v0.z : 0.f,
                       k0.w > 0.f ? v0.w : 0.f);
      v1 = make_float4(k1.x > 0.f ? v1.x : 0.f, k1.y > 0.f ? v1.y : 0.f, k1.z > 0.f ? v1.z : 0.f,
                       k1.w > 0.f ? v1.w : 0.f);
    }
    *reinterpret_cast<float4 *>(g.C + o) = v0;
    *reinterpret_cast<float4 *>(g.C + o + 4) = v1;
  }
  if (want_asum) {   // (workgroup-uniform) the four waves' sums, in a fixed order
    pipe::wg_barrier();   // the slabs have been read
    if (lane < TM) smemf[wid * TM + lane] = asum;
    pipe::wg_barrier();
    if (tid < TM)
      g.asum[z * g.asum_batch + (size_t)(m0 + tid) * g.asum_ld] =
          (smemf[tid] + smemf[TM + tid]) + (smemf[2 * TM + tid] + smemf[3 * TM + tid]);
  }
}

bool aligned16(const void *p) { return ((size_t)p & 15) == 0; }
}  // namespace

// (called by gemm_mfma.hip's problem-list launcher before it plans its own tiles)
int scae_gemm_x6k_try(const scae_gemm_desc *descs, int n, void *stream) {
  if (!descs || n < 1 || n > 4) return SCAE_ERR_BAD_ARG;
  const char *e = getenv("SCAE_GEMM_KSPLIT");
  if (e && *e ? atoi(e) == 0 : SCAE_GEMM_KSPLIT_DEFAULT == 0) return SCAE_ERR_UNSUPPORTED;
  Probs ps{};
  ps.n = n;
  int first = 0;
  for (int k = 0; k < n; ++k) {
    const scae_gemm_desc &d = descs[k];
    if (!d.A || !d.B || !d.C || d.batch <= 0 || d.M <= 0 || d.N <= 0 || d.K <= 0)
      return SCAE_ERR_BAD_ARG;
    // the shapes this form exists for: few tiles, a long K
    if (d.b_kcontig || d.bias || d.relu || d.M % TM || d.N % TN || d.K < 4 * BKF ||
        (d.a_kcontig && d.K % BKF) || (d.asum && d.a_kcontig))
      return SCAE_ERR_UNSUPPORTED;
    if ((d.lda | d.ldb | d.ldc) & 3 || (d.a_batch | d.b_batch | d.c_batch) & 3 || !aligned16(d.A) ||
        !aligned16(d.B) || !aligned16(d.C) || (d.c_nomask && !aligned16(d.c_nomask)) ||
        (d.mask && ((d.ldmask & 3) || (d.mask_batch & 3) || !aligned16(d.mask))))
      return SCAE_ERR_UNSUPPORTED;
    const size_t a_ext = d.a_kcontig ? (size_t)(d.M - 1) * d.lda + d.K : (size_t)(d.K - 1) * d.lda + d.M;
    const size_t b_ext = (size_t)(d.K - 1) * d.ldb + d.N;
    if (a_ext * 4 >= (1u << 31) || b_ext * 4 >= (1u << 31)) return SCAE_ERR_UNSUPPORTED;
    Prob &p = ps.p[k];
    p = Prob{d.A, d.B, d.mask, d.C, d.c_nomask, d.asum, (long)d.a_batch, (long)d.b_batch,
             (long)d.c_batch, (long)d.mask_batch, (long)d.asum_batch, d.lda, d.ldb, d.ldc, d.ldmask,
             d.asum_ld > 0 ? d.asum_ld : 1, d.M, d.N, d.K, d.a_kcontig != 0, (unsigned)(a_ext * 4),
             (unsigned)(b_ext * 4), d.N / TN, d.M / TM, first};
    first += p.tx * p.ty * d.batch;
  }
  if (first > 1024) return SCAE_ERR_UNSUPPORTED;   // (larger lists: the tiles of mfma_tile.h)
  for (int k = n; k < 4; ++k) ps.p[k].first = first;
  scae::launch(gemm_x6k_kernel, dim3(first), dim3(NT), 0, (hipStream_t)stream, ps);
  return scae_launch_status();
}
