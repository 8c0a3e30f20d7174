// Shared fp32 MFMA tile loop pieces (v_mfma_f32_16x16x4_f32, wave64) for the
// implicit-GEMM convolutions (K8) and the batched GEMM (K7).
//
// K is walked in 32-wide chunks staged through LDS by 256 threads, in two
// workgroup shapes selected by the SK template flag:
//   !SK: 64 x 64 output tile, 2 x 2 waves each owning a 32 x 32 sub-tile;
//    SK: 32 x 32 output tile; the 4 waves split every K chunk four ways and are
//        summed through LDS at the end -- 4x the workgroups for the small
//        problems of this model, which would otherwise leave CUs idle.
// Operand tiles are k-contiguous [rows][BK + 2] (stride = 2 mod 32 banks) or
// k-strided [BK][T + 16] (stride = 16 mod 32), both conflict-free for the
// row-per-lane fragment reads.  Accumulators leave through per-wave LDS slabs
// as float4 rows.
#pragma once
#include "common.h"

namespace scae_tile {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 256;
constexpr int BK = 32;
constexpr int LDK = BK + 2;  // k-contiguous tile [rows][LDK]
constexpr int LDR = 36;      // epilogue slab [32][LDR]: b128 rows on distinct banks

template <bool SK>
struct Tile {
  static constexpr int T = SK ? 32 : 64;  // tile rows = tile cols
  static constexpr int NQ = T / 32;       // float4 per thread per operand per chunk
  static constexpr int LDT = T + 16;      // k-strided tile [BK][LDT]
  static constexpr int OPER = BK * LDT > T * LDK ? BK * LDT : T * LDK;
  static constexpr int SMEM = 2 * OPER > 4 * 32 * LDR ? 2 * OPER : 4 * 32 * LDR;
};

template <int NQ>
struct Quads {
  float4 v[NQ];
};

__device__ __forceinline__ float4 ld4(const float *p) {
  return *reinterpret_cast<const float4 *>(p);
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// registers -> LDS.  KC: tile[row][k] (row = id / 8, k quad = id % 8);
// otherwise tile[k][row] (k = id / (T/4), row quad = id % (T/4)).
template <bool SK, bool KC>
__device__ __forceinline__ void deposit(float *tile, const Quads<Tile<SK>::NQ> &q) {
  constexpr int T = Tile<SK>::T, LDT = Tile<SK>::LDT;
#pragma unroll
  for (int i = 0; i < Tile<SK>::NQ; ++i) {
    const int id = threadIdx.x + NT * i;
    if (KC) {
      float *p = tile + (id >> 3) * LDK + ((id & 7) << 2);
      *reinterpret_cast<float2 *>(p) = make_float2(q.v[i].x, q.v[i].y);
      *reinterpret_cast<float2 *>(p + 2) = make_float2(q.v[i].z, q.v[i].w);
    } else {
      *reinterpret_cast<float4 *>(tile + (id / (T / 4)) * LDT + 4 * (id % (T / 4))) = q.v[i];
    }
  }
}

// the MFMAs of one K chunk.  !SK: wave (wid>>1, wid&1) owns a 32x32 sub-tile and
// runs all 8 k-steps; SK: every wave covers the whole 32x32 tile for k-steps
// 2*wid, 2*wid+1.  AK / BKC: operand tile is k-contiguous.
template <bool SK, bool AK, bool BKC>
__device__ __forceinline__ void mma_chunk(const float *As, const float *Bs, f32x4 (&acc)[2][2],
                                          int wid, int r, int q) {
  constexpr int LDT = Tile<SK>::LDT;
  const int ro = SK ? 0 : 32 * (wid >> 1), co = SK ? 0 : 32 * (wid & 1);
  const int kb = SK ? 8 * wid : 0;
#pragma unroll
  for (int s = 0; s < (SK ? 2 : 8); ++s) {
    const int kk = kb + 4 * s;
    float a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = ro + 16 * i + r, col = co + 16 * i + r;
      a[i] = AK ? As[row * LDK + kk + q] : As[(kk + q) * LDT + row];
      b[i] = BKC ? Bs[col * LDK + kk + q] : Bs[(kk + q) * LDT + col];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

// accumulators -> per-wave LDS slab -> epi(tile row, tile col (multiple of 4), float4)
template <bool SK, class Epi>
__device__ __forceinline__ void tile_epilogue(float *smem, const f32x4 (&acc)[2][2], int wid,
                                              int r, int q, Epi epi) {
  __syncthreads();  // operand tiles are dead: the slabs alias them
  float *slab = smem + wid * 32 * LDR;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        slab[(16 * i + 4 * q + reg) * LDR + 16 * j + r] = acc[i][j][reg];
  __syncthreads();
  const int row = threadIdx.x >> 3, c4 = (threadIdx.x & 7) << 2;
  const float *src = smem + row * LDR + c4;
  if (SK) {
    float4 v = ld4(src);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 u = ld4(src + w * 32 * LDR);
      v.x += u.x, v.y += u.y, v.z += u.z, v.w += u.w;
    }
    epi(row, c4, v);
  } else {
#pragma unroll
    for (int w = 0; w < 4; ++w) epi(32 * (w >> 1) + row, 32 * (w & 1) + c4, ld4(src + w * 32 * LDR));
  }
}

}  // namespace scae_tile
