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
#include "common.h"

namespace scae_pipe {
typedef float f32x16 __attribute__((ext_vector_type(16)));
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

// The MFMAs of one chunk from ring stage `st`.  The fragments of 4-k group g + 1 are
// read while the MFMAs of group g run (one wave cannot hide a ds_read behind anything
// else), and `dma(j)` -- the DMA pieces of a later chunk, with their address
// arithmetic -- is spread between the groups instead of idling the matrix pipe at
// the head of the chunk.
template <class T, class Dma>
__device__ __forceinline__ void kk_compute(const float *st, f32x16 (&acc)[T::MI][T::NI], int wn,
                                           int ks, int i, int kk, bool more, Dma dma) {
  const float *As = st + kk * T::TA * BKH + i * BKH;
  const float *Bs = st + T::TA * BK + (kk * T::TB + wn * 32 * T::NI + i) * BKH;
  const int sw = (i >> 2) & 3;
  constexpr int G = 4 / T::KS;
  float4 a[2][T::MI], b[2][T::NI];
  auto load = [&](int gg, int buf) {
    const int qo = ((ks * G + gg) ^ sw) << 2;
#pragma unroll
    for (int mi = 0; mi < T::MI; ++mi) a[buf][mi] = lds4(As + mi * 32 * BKH + qo);
#pragma unroll
    for (int ni = 0; ni < T::NI; ++ni) b[buf][ni] = lds4(Bs + ni * 32 * BKH + qo);
  };
  load(0, 0);
#pragma unroll
  for (int gg = 0; gg < G; ++gg) {
    const int cur = gg & 1;
    if (gg + 1 < G) load(gg + 1, cur ^ 1);
#pragma unroll
    for (int mi = 0; mi < T::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < T::NI; ++ni) {
#if SCAE_PIPE_ABL == 2
        acc[mi][ni][0] += a[cur][mi].x * b[cur][ni].x + a[cur][mi].y * b[cur][ni].y +
                          a[cur][mi].z * b[cur][ni].z + a[cur][mi].w * b[cur][ni].w;
#else
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][mi].x, b[cur][ni].x, acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][mi].y, b[cur][ni].y, acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][mi].z, b[cur][ni].z, acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][mi].w, b[cur][ni].w, acc[mi][ni], 0, 0, 0);
#endif
      }
    if (more) {   // workgroup-uniform
#pragma unroll
      for (int j = gg * T::PPW / G; j < (gg + 1) * T::PPW / G; ++j) dma(j);
    }
  }
}

// The K loop.  chunk(c) -- called once per chunk, c = 0, 1, 2, ... in order -- returns
// the wave-uniform context of chunk c (its offsets); issue(ctx, stage, j) starts DMA
// piece j (of this wave's T::PPW) of that chunk.
template <class T, class Chunk, class Issue>
__device__ __forceinline__ void kk_mainloop(int nchunk, float *smem, f32x16 (&acc)[T::MI][T::NI],
                                            int wn, int ks, int i, int kk, Chunk chunk,
                                            Issue issue) {
  constexpr int LA = T::NS - 1;   // chunks in flight beyond the one being multiplied
#pragma unroll
  for (int c = 0; c < LA; ++c)
    if (c < nchunk) {
      const auto ctx = chunk(c);
#pragma unroll
      for (int j = 0; j < T::PPW; ++j) issue(ctx, smem + c * T::STAGE, j);
    }
  int s = 0;   // stage of chunk c
  for (int c = 0; c < nchunk; ++c) {
#ifndef SCAE_PIPE_ABL
#define SCAE_PIPE_ABL 0
#endif
    const bool more = SCAE_PIPE_ABL != 1 && c + LA < nchunk;
    const auto ctx = chunk(c + LA);   // (a few scalar instructions; unused past the end)
    // chunk c has landed; younger chunks may still be in flight
    wait_chunk<T::PPW, LA>(min(LA - 1, nchunk - 1 - c));
    wg_barrier();   // everyone's pieces of chunk c; everyone done with chunk c - 1
    float *s2 = smem + (s >= 1 ? s - 1 : T::NS - 1) * T::STAGE;   // stage of chunk c - 1
    kk_compute<T>(smem + s * T::STAGE, acc, wn, ks, i, kk, more,
                  [&](int j) { issue(ctx, s2, j); });
    s = s + 1 == T::NS ? 0 : s + 1;
  }
}

// Accumulators out: epi(row, col, value) per element (row-major C, 32 consecutive
// columns per store instruction).  With KS > 1 the wave groups meet in LDS first.
template <class T, class Epi>
__device__ __forceinline__ void kk_epilogue(float *smem, f32x16 (&acc)[T::MI][T::NI], int wid,
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
        epi(mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk, (wn * T::NI + ni) * 32 + i, acc[mi][ni][e]);
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

template <class T, class Dma>
__device__ __forceinline__ void ss_compute(const float *st, f32x16 (&acc)[T::MI][T::NI], int wm,
                                           int wn, int i, int kk, bool more, Dma dma) {
  const float *As = st + kk * T::TA + wm * 32 * T::MI + i;
  const float *Bs = st + T::TA * T::BKW + kk * T::TB + wn * 32 * T::NI + i;
  constexpr int S = T::BKW / 2, U = 4;   // MFMA steps per chunk, per fragment batch
  float a[2][U][T::MI], b[2][U][T::NI];
  auto load = [&](int blk, int buf) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = blk * U + u;
#pragma unroll
      for (int mi = 0; mi < T::MI; ++mi) a[buf][u][mi] = As[2 * s * T::TA + mi * 32];
#pragma unroll
      for (int ni = 0; ni < T::NI; ++ni) b[buf][u][ni] = Bs[2 * s * T::TB + ni * 32];
    }
  };
  load(0, 0);
#pragma unroll
  for (int blk = 0; blk < S / U; ++blk) {
    const int cur = blk & 1;
    if (blk + 1 < S / U) load(blk + 1, cur ^ 1);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int mi = 0; mi < T::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < T::NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][u][mi], b[cur][u][ni],
                                                            acc[mi][ni], 0, 0, 0);
    if (more) {
#pragma unroll
      for (int j = blk * T::PPW / (S / U); j < (blk + 1) * T::PPW / (S / U); ++j) dma(j);
    }
  }
}

}  // namespace scae_pipe
