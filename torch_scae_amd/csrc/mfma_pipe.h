// fp32 MFMA tile pipeline, second generation (K8 implicit-GEMM convolutions):
//   * v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate; half the LDS
//     operand bytes per flop of the 16x16x4 form);
//   * operands go global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave
//     instruction) into a ring of NS stages: the loads of chunks c+1 .. c+NS-1 are in
//     flight while chunk c is multiplied; ONE s_barrier per 32-wide K chunk, counted
//     vmcnt waits (no __syncthreads(): it would drain the ring); the ring is small
//     (24-48 KiB) so that 3+ workgroups share a CU and cover each other's prologues,
//     epilogues and barrier waits -- the tiles of this model's layers are short-lived;
//   * k-contiguous operand tiles are stored [k plane (2)][row][16 floats]: plane kk
//     holds k = 16 kk .. 16 kk + 15 of the chunk, i.e. the k half that MFMA lane
//     group kk = lane >> 5 contracts (a DMA piece is 16 rows x 64 B); the four
//     16-byte quads of a row are XOR-swizzled with (row >> 2) & 3 -- applied on the
//     SOURCE address, the LDS image of a DMA piece is lane-linear -- which makes the
//     row-per-lane ds_read_b128 fragment reads bank-conflict free;
//   * k-strided operand tiles (weight gradient: K = pixels, rows = channels) are
//     stored [k][rows]: a DMA piece is 1 KiB of consecutive (k, row) floats and the
//     fragment reads are 32 consecutive floats per k (conflict free as they are).
// Workgroup = 256 threads = 4 waves.
#pragma once
#include <type_traits>

#include "bf16x6.h"
#include "common.h"

// The products of the second-generation loops: exact three-way bf16 split, six bf16 MFMAs per
// 16 k (bf16x6.h) -- fp32 results at 6 / 16 of the fp32 MFMA time; -DSCAE_PIPE_X6=0: the
// fp32 MFMA chain (v_mfma_f32_32x32x2_f32), for A/B builds.
#ifndef SCAE_PIPE_X6
#define SCAE_PIPE_X6 1
#endif

namespace scae_pipe {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef scae_x6::bf16x8 bf16x8;
constexpr int NT = 256;
constexpr int BK = 32;   // K chunk per ring stage
constexpr int BKH = 16;  // per k plane

// 16 bytes per lane, global -> LDS, through a raw buffer descriptor: lane address =
// base + voff (per lane, bytes) + soff (wave-uniform, bytes); a lane whose voff is
// >= the descriptor's size reads zeros (structural zeros, rows past the end, for
// free).  `lds` is the wave-uniform base of a 1 KiB piece (lane l lands at + 16 l).
// One VMEM instruction, no per-lane 64-bit address arithmetic.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void dma16(rsrc_t r, float *lds, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16,
                                           voff, soff, 0, 0);
}
constexpr int DMA_ZERO = 0x7ffffff0;   // a voff no descriptor covers
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// chunk c has landed when at most `younger` x PPW of this wave's DMA pieces -- those of the
// chunks issued after it -- are still outstanding (vector memory returns in order)
template <int PPW, int LA>
__device__ __forceinline__ void wait_chunk(int younger) {   // wave-uniform
  static_assert(5 * PPW <= 63, "vmcnt range");
  if (LA >= 6 && younger >= 5)
    wait_vm<5 * PPW>();
  else if (LA >= 5 && younger >= 4)
    wait_vm<4 * PPW>();
  else if (LA >= 4 && younger >= 3)
    wait_vm<3 * PPW>();
  else if (LA >= 3 && younger >= 2)
    wait_vm<2 * PPW>();
  else if (LA >= 2 && younger >= 1)
    wait_vm<PPW>();
  else
    wait_vm<0>();
}
__device__ __forceinline__ void wg_barrier() {
  // LDS reads of the previous chunk are complete (their results were consumed by
  // MFMAs already issued); only the barrier itself is needed
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ float4 lds4(const float *p) {
  return *reinterpret_cast<const float4 *>(p);
}

// ---------------------------------------------------------------------------------
// k-contiguous x k-contiguous tile: C[TA x TB] += A[TA x K] B[TB x K]^T
//   WN waves side by side along the TB columns (each owns TB / WN = 32 NI columns and
//   all TA = 32 MI rows); with WN < 4 the remaining factor KS = 4 / WN splits the
//   eight 4-k groups of every chunk between wave groups (summed through LDS at the
//   end) -- more, smaller workgroups for the small layers.
template <int TA_, int TB_, int WN_, int NS_>
struct KK {
  static constexpr int TA = TA_, TB = TB_, WN = WN_, KS = 4 / WN_, NS = NS_;
  static constexpr int MI = TA / 32, NI = TB / (32 * WN);
  static constexpr int STAGE = (TA + TB) * BK;        // floats per ring stage
  static constexpr int RA = TA / 32, RB = TB / 32;    // DMA rows per lane and chunk
  static constexpr int PPW = RA + RB;                 // DMA pieces per wave and chunk
  static constexpr int RED = (KS - 1) * WN * MI * NI * 1024;   // k-split meeting slabs
  static constexpr int SMEM = NS * STAGE > RED ? NS * STAGE : RED;   // floats
  static_assert(NI >= 1 && MI >= 1 && TA % 32 == 0 && TB % 32 == 0 && (NS >= 2 && NS <= 7),
                "tile shape");
};

// which rows / which k quad a lane moves: wave w owns plane w & 1 and the 16-row
// blocks (w >> 1) + 2 j; lane l the row 16 (w >> 1) + (l >> 2) + 32 j of it
struct DmaLane {
  int row0;   // + 32 j
  int koff;   // floats into the row's 32-k chunk: plane * 16 + swizzled quad * 4
  int loff;   // floats from the operand tile's LDS base to this wave's piece j = 0
};
constexpr int DMA_ROWS = 32;   // row step between a lane's pieces
template <int ROWS>
__device__ __forceinline__ DmaLane dma_lane(int wid, int lane) {
  DmaLane d;
  d.row0 = 16 * (wid >> 1) + (lane >> 2);
  const int plane = wid & 1;
  d.koff = plane * BKH + (((lane & 3) ^ ((d.row0 >> 2) & 3)) << 2);
  d.loff = (plane * ROWS + 16 * (wid >> 1)) * BKH;   // piece j adds 32 rows
  return d;
}

template <class T>
__device__ __forceinline__ void kk_zero(f32x16 (&acc)[T::MI][T::NI]) {
#pragma unroll
  for (int i = 0; i < T::MI; ++i)
#pragma unroll
    for (int j = 0; j < T::NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
}

// The K loop (third form, round 5).  Measured with phase stamps (tools/fwd_prof.py): a 32 x 64
// tile alone on its CU (layer 4 at B = 128: 200 workgroups on 256 CUs) spent ~1100 cycles per
// 32-wide chunk on 512 cycles of MFMAs -- after the chunk's barrier the wave waited for its
// first fragment reads, and the three DMA instructions of the chunk (60-180 cycles of issue
// each) sat behind the last MFMA where nothing covers them.  Now
//   * the wait + barrier for chunk c + 1 and the read of its first fragments happen in the
//     MIDDLE of chunk c, ahead of chunk c's last MFMA group: the barrier's skew and the LDS
//     latency run under MFMAs already in the pipe;
//   * every fragment of chunk c is in registers by then, so the barrier also frees chunk c's
//     OWN stage: the DMA pieces issued behind it are those of chunk c + NS (a ring of NS stages
//     keeps NS chunks in flight, not NS - 1);
//   * those pieces go BETWEEN the MFMAs of the last group, one behind an MFMA each (a dependent
//     MFMA cannot issue for 64 cycles anyway).
// The order of the products within an accumulator is unchanged (bit-identical results).
// chunk(c) -- called once per chunk, c = 0, 1, 2, ... in order -- returns the wave-uniform context
// of chunk c (its offsets); issue(ctx, stage, j) starts DMA piece j (of this wave's T::PPW).
template <class T, class Chunk, class Issue>
__device__ __forceinline__ void kk_mainloop(int nchunk, float *smem, f32x16 (&acc)[T::MI][T::NI],
                                            int wn, int ks, int i, int kk, Chunk chunk,
                                            Issue issue) {
#ifndef SCAE_PIPE_ABL
#define SCAE_PIPE_ABL 0
#endif
  // X6 (where a wave has an even number of 4-k groups per chunk): a group is a PAIR of quads --
  // the lane's eight k of one bf16 MFMA -- and its products the six exact partial products of
  // bf16x6.h, the five small ones into an accumulator of their own
  constexpr bool X6 = SCAE_PIPE_X6 != 0 && (4 / T::KS) % 2 == 0;
  constexpr int NS = T::NS, QG = X6 ? 2 : 1, G = 4 / T::KS / QG;
  constexpr int NM = (X6 ? 6 : 4) * T::MI * T::NI;
  const int sw = (i >> 2) & 3;
  const int aoff = kk * T::TA * BKH + i * BKH;
  const int boff = T::TA * BK + (kk * T::TB + wn * 32 * T::NI + i) * BKH;
  float4 a[2][QG][T::MI], b[2][QG][T::NI];
  auto load = [&](const float *st, int gg, int buf) {
#pragma unroll
    for (int h = 0; h < QG; ++h) {
      const int qo = ((ks * G * QG + gg * QG + h) ^ sw) << 2;
#pragma unroll
      for (int mi = 0; mi < T::MI; ++mi) a[buf][h][mi] = lds4(st + aoff + mi * 32 * BKH + qo);
#pragma unroll
      for (int ni = 0; ni < T::NI; ++ni) b[buf][h][ni] = lds4(st + boff + ni * 32 * BKH + qo);
    }
  };
  f32x16 accl[X6 ? T::MI : 1][X6 ? T::NI : 1];
  if (X6) {
#pragma unroll
    for (int mi = 0; mi < T::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < T::NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) accl[mi][ni][e] = 0.f;
  }
#pragma unroll
  for (int c = 0; c < NS; ++c)
    if (c < nchunk) {
      const auto ctx = chunk(c);
#pragma unroll
      for (int j = 0; j < T::PPW; ++j) issue(ctx, smem + c * T::STAGE, j);
    }
  if (nchunk <= 0) return;
  wait_chunk<T::PPW, NS>(min(NS - 1, nchunk - 1));   // chunk 0 has landed
  wg_barrier();
  load(smem, 0, 0);
  int s = 0;   // stage of chunk c
  // (`last` is a compile-time constant: a run-time test around the barrier + reads makes the
  // compiler wait for ALL outstanding LDS reads -- the next chunk's too -- before the last group)
  auto body = [&](int c, auto last_t) {
    constexpr bool last = decltype(last_t)::value;
    const bool more = SCAE_PIPE_ABL != 1 && !last && c + NS < nchunk;
    const auto ctx = chunk(c + NS);   // (a few scalar instructions; unused past the end)
    float *st = smem + s * T::STAGE;
    const int sn = s + 1 == NS ? 0 : s + 1;
    const float *stn = smem + sn * T::STAGE;
#pragma unroll
    for (int gg = 0; gg < G; ++gg) {
      const int cur = gg & 1;
      if (gg + 1 < G) {
        load(st, gg + 1, cur ^ 1);
      } else if (!last) {
        // chunk c + 1 has landed (chunks c + 2 .. may still be in flight); everyone's pieces
        // of it, and everyone done reading chunk c
        wait_chunk<T::PPW, NS>(min(NS - 2, nchunk - 2 - c));
        wg_barrier();
        load(stn, 0, cur ^ 1);
      }
      const bool dma_here = gg + 1 == G && more;
      scae_x6::Split3 as[X6 ? T::MI : 1], bs[X6 ? T::NI : 1];
      if (X6) {
#pragma unroll
        for (int mi = 0; mi < T::MI; ++mi) as[mi] = scae_x6::split3(a[cur][0][mi], a[cur][QG - 1][mi]);
#pragma unroll
        for (int ni = 0; ni < T::NI; ++ni) bs[ni] = scae_x6::split3(b[cur][0][ni], b[cur][QG - 1][ni]);
      }
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        if (X6) {
          // product by product over the wave's tiles (m / tiles: hi lo, lo hi, mid mid, hi mid,
          // mid hi -> the small accumulator; hi hi -> the tile's)
          constexpr int NTL = T::MI * T::NI;
          const int pr = m / NTL, mi = (m % NTL) / T::NI, ni = (m % NTL) % T::NI;
          const bf16x8 av = pr == 1 ? as[mi].lo : (pr == 2 || pr == 4) ? as[mi].mid : as[mi].hi;
          const bf16x8 bv = pr == 0 ? bs[ni].lo : (pr == 2 || pr == 3) ? bs[ni].mid : bs[ni].hi;
          if (pr < 5)
            accl[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, accl[mi][ni], 0, 0, 0);
          else
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[mi][ni], 0, 0, 0);
        } else {
          const int mi = (m >> 2) / T::NI, ni = (m >> 2) % T::NI, e = m & 3;
#if SCAE_PIPE_ABL == 2
          acc[mi][ni][0] += a[cur][0][mi][e] * b[cur][0][ni][e];
#else
          const float4 &aq = a[cur][0][mi], &bq = b[cur][0][ni];
          const float av = e == 0 ? aq.x : e == 1 ? aq.y : e == 2 ? aq.z : aq.w;
          const float bv = e == 0 ? bq.x : e == 1 ? bq.y : e == 2 ? bq.z : bq.w;
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mi][ni], 0, 0, 0);
#endif
        }
        if (dma_here && m + 1 < NM) {   // pieces spread over the NM - 1 gaps between the MFMAs
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = m * T::PPW / (NM - 1); j < (m + 1) * T::PPW / (NM - 1); ++j) issue(ctx, st, j);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (G & 1) {   // the next chunk's first fragments were read into the other buffer
#pragma unroll
      for (int h = 0; h < QG; ++h) {
#pragma unroll
        for (int mi = 0; mi < T::MI; ++mi) a[0][h][mi] = a[1][h][mi];
#pragma unroll
        for (int ni = 0; ni < T::NI; ++ni) b[0][h][ni] = b[1][h][ni];
      }
    }
    s = sn;
  };
  for (int c = 0; c + 1 < nchunk; ++c) body(c, std::false_type{});
  body(nchunk - 1, std::true_type{});
  if (X6) {
#pragma unroll
    for (int mi = 0; mi < T::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < T::NI; ++ni) acc[mi][ni] += accl[mi][ni];
  }
}

// Accumulators out: epi(row, col, value) per element (row-major C, 32 consecutive
// columns per store instruction).  With KS > 1 the wave groups meet in LDS first.
// epi(mi, ni, e, row, col, value): the accumulator's indices as compile-time constants too
template <class T, class Epi>
__device__ __forceinline__ void kk_epilogue_idx(float *smem, f32x16 (&acc)[T::MI][T::NI], int wid,
                                                int wn, int ks, int i, int kk, Epi epi) {
  if (T::KS > 1) {
    // [ks - 1][wn][mi][ni][reg][lane]: conflict-free, 64 consecutive floats per store
    wg_barrier();   // the ring is dead
    float *slab = smem + ((ks - 1) * T::WN + wn) * (T::MI * T::NI * 16 * 64);
    if (ks > 0) {
#pragma unroll
      for (int mi = 0; mi < T::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < T::NI; ++ni)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            slab[((mi * T::NI + ni) * 16 + e) * 64 + kk * 32 + i] = acc[mi][ni][e];
    }
    wg_barrier();
    if (ks > 0) return;
#pragma unroll
    for (int k2 = 1; k2 < T::KS; ++k2) {
      const float *src = smem + ((k2 - 1) * T::WN + wn) * (T::MI * T::NI * 16 * 64);
#pragma unroll
      for (int mi = 0; mi < T::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < T::NI; ++ni)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            acc[mi][ni][e] += src[((mi * T::NI + ni) * 16 + e) * 64 + kk * 32 + i];
    }
  }
#pragma unroll
  for (int mi = 0; mi < T::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < T::NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        epi(mi, ni, e, mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk, (wn * T::NI + ni) * 32 + i,
            acc[mi][ni][e]);
}
template <class T, class Epi>
__device__ __forceinline__ void kk_epilogue(float *smem, f32x16 (&acc)[T::MI][T::NI], int wid,
                                            int wn, int ks, int i, int kk, Epi epi) {
  kk_epilogue_idx<T>(smem, acc, wid, wn, ks, i, kk,
                     [&](int, int, int, int row, int col, float v) { epi(row, col, v); });
}

// ---------------------------------------------------------------------------------
// k-strided x k-strided tile (weight gradient): C[TA x TB] += sum_k A[k][TA] B[k][TB]
//   LDS [k][rows]; 2 x 2 waves, each (TA/2) x (TB/2); a DMA piece = 256 consecutive
//   floats of the [k][rows] tile.
template <int TA_, int TB_, int BKW_, int NS_>
struct SS {
  static constexpr int TA = TA_, TB = TB_, BKW = BKW_, NS = NS_;   // BKW: K chunk
  static constexpr int MI = TA / 64, NI = TB / 64;
  static constexpr int STAGE = (TA + TB) * BKW;
  static constexpr int PA = TA * BKW / 256, PB = TB * BKW / 256;   // pieces per chunk
  static constexpr int PPW = (PA + PB) / 4;
  static constexpr int SMEM = NS * STAGE;
  static_assert(MI >= 1 && NI >= 1 && (PA + PB) % 4 == 0 && (NS >= 2 && NS <= 7), "tile shape");
};

// The K loop of an SS tile (same structure as kk_mainloop: the wait + barrier for chunk c + 1
// and the read of its first fragments sit in front of chunk c's LAST fragment batch, under
// MFMAs already in the pipe).  issue(c, stage, j): DMA piece j of chunk c; each(st): called once
// per chunk with its landed stage (the weight gradient's bias column sums).
template <class T, class Issue, class Each>
__device__ __forceinline__ void ss_mainloop(int nchunk, float *smem, f32x16 (&acc)[T::MI][T::NI],
                                            int wm, int wn, int i, int kk, Issue issue, Each each) {
  constexpr bool X6 = SCAE_PIPE_X6 != 0;
  // a batch = the lane's U k of each fragment: 4 fp32 MFMA steps, or (X6) the 8 k of one bf16 MFMA
  constexpr int LA = T::NS - 1, S = T::BKW / 2, U = X6 ? 8 : 4, NB = S / U;
  static_assert(NB >= 1 && S % U == 0 && T::PPW % NB == 0, "fragment batches per chunk");
  const int aoff = kk * T::TA + wm * 32 * T::MI + i;
  const int boff = T::TA * T::BKW + kk * T::TB + wn * 32 * T::NI + i;
  float a[2][U][T::MI], b[2][U][T::NI];
  auto load = [&](const float *st, int blk, int buf) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = blk * U + u;
#pragma unroll
      for (int mi = 0; mi < T::MI; ++mi) a[buf][u][mi] = st[aoff + 2 * s * T::TA + mi * 32];
#pragma unroll
      for (int ni = 0; ni < T::NI; ++ni) b[buf][u][ni] = st[boff + 2 * s * T::TB + ni * 32];
    }
  };
  // X6: the small products' accumulator
  // (one: in the launch these tiles share with the data gradient's, the registers of two more
  // accumulators cost a workgroup per CU -- B = 128, the three pair launches: 145 us with one,
  // 159 with two or three, 152 on the fp32 MFMA chain)
  constexpr int NSM = 1;
  f32x16 accl[X6 ? T::MI : 1][X6 ? T::NI : 1][NSM];
  if (X6) {
#pragma unroll
    for (int mi = 0; mi < T::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < T::NI; ++ni)
#pragma unroll
        for (int n = 0; n < NSM; ++n)
#pragma unroll
          for (int e = 0; e < 16; ++e) accl[mi][ni][n][e] = 0.f;
  }
#pragma unroll
  for (int c = 0; c < LA; ++c)
    if (c < nchunk) {
#pragma unroll
      for (int j = 0; j < T::PPW; ++j) issue(c, smem + c * T::STAGE, j);
    }
  if (nchunk <= 0) return;
  wait_chunk<T::PPW, LA>(min(LA - 1, nchunk - 1));
  wg_barrier();
  load(smem, 0, 0);
  int s = 0;
  auto body = [&](int c, auto last_t) {
    constexpr bool last = decltype(last_t)::value;
    const bool more = !last && c + LA < nchunk;
    const float *st = smem + s * T::STAGE;
    float *s2 = smem + (s >= 1 ? s - 1 : T::NS - 1) * T::STAGE;   // stage of chunk c - 1: free
    const int sn = s + 1 == T::NS ? 0 : s + 1;
    each(st);
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      const int cur = blk & 1;
      if (blk + 1 < NB) {
        load(st, blk + 1, cur ^ 1);
      } else if (!last) {
        // chunk c + 1 has landed: what is younger are the pieces of chunk c + LA issued so far
        // in this iteration (and, with a deeper ring, the chunks between)
        if (more)
          wait_vm<(LA - 2) * T::PPW + (NB - 1) * (T::PPW / NB)>();
        else
          wait_chunk<T::PPW, LA>(min(LA - 2, nchunk - 2 - c));
        wg_barrier();
        load(smem + sn * T::STAGE, 0, cur ^ 1);
      }
      if (X6) {
        scae_x6::Split3 as[T::MI], bs[T::NI];
#pragma unroll
        for (int mi = 0; mi < T::MI; ++mi) {
          const float x[8] = {a[cur][0][mi], a[cur][1][mi], a[cur][2][mi], a[cur][3][mi],
                              a[cur][4 % U][mi], a[cur][5 % U][mi], a[cur][6 % U][mi], a[cur][7 % U][mi]};
          as[mi] = scae_x6::split3(x);
        }
#pragma unroll
        for (int ni = 0; ni < T::NI; ++ni) {
          const float x[8] = {b[cur][0][ni], b[cur][1][ni], b[cur][2][ni], b[cur][3][ni],
                              b[cur][4 % U][ni], b[cur][5 % U][ni], b[cur][6 % U][ni], b[cur][7 % U][ni]};
          bs[ni] = scae_x6::split3(x);
        }
#define SCAE_SS_STEP(AP, BP, N)                                                                   \
  _Pragma("unroll") for (int mi = 0; mi < T::MI; ++mi)                                            \
  _Pragma("unroll") for (int ni = 0; ni < T::NI; ++ni) accl[mi][ni][(N) % NSM] =                  \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[mi].AP, bs[ni].BP, accl[mi][ni][(N) % NSM], 0, 0, 0);
        SCAE_SS_STEP(hi, lo, 0)
        SCAE_SS_STEP(lo, hi, 1)
        SCAE_SS_STEP(mid, mid, 2)
        SCAE_SS_STEP(hi, mid, 0)
        SCAE_SS_STEP(mid, hi, 1)
#undef SCAE_SS_STEP
#pragma unroll
        for (int mi = 0; mi < T::MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < T::NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[mi].hi, bs[ni].hi, acc[mi][ni],
                                                                  0, 0, 0);
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int mi = 0; mi < T::MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < T::NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][u][mi], b[cur][u][ni],
                                                                acc[mi][ni], 0, 0, 0);
      }
      if (more) {
#pragma unroll
        for (int j = blk * (T::PPW / NB); j < (blk + 1) * (T::PPW / NB); ++j) issue(c + LA, s2, j);
      }
    }
    if (NB & 1) {   // the next chunk's first batch was read into the other buffer
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int mi = 0; mi < T::MI; ++mi) a[0][u][mi] = a[1][u][mi];
#pragma unroll
        for (int ni = 0; ni < T::NI; ++ni) b[0][u][ni] = b[1][u][ni];
      }
    }
    s = sn;
  };
  for (int c = 0; c + 1 < nchunk; ++c) body(c, std::false_type{});
  body(nchunk - 1, std::true_type{});
  if (X6) {
#pragma unroll
    for (int mi = 0; mi < T::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < T::NI; ++ni) {
        f32x16 small = accl[mi][ni][0];
#pragma unroll
        for (int n = 1; n < NSM; ++n) small += accl[mi][ni][n];
        acc[mi][ni] += small;
      }
  }
}

}  // namespace scae_pipe
