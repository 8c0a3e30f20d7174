// Shared fp32 MFMA tile loop pieces (v_mfma_f32_16x16x4_f32, wave64) for the
// implicit-GEMM convolutions (K8) and the batched GEMM (K7).
//
// K is walked in 32-wide chunks staged through LDS by 256 threads, in two
// workgroup shapes selected by the SK template flag:
//   !SK: 64 x 64 output tile, 2 x 2 waves each owning a 32 x 32 sub-tile;
//    SK: 32 x 32 output tile; the 4 waves split every K chunk four ways and are
//        summed through LDS at the end -- 4x the workgroups for the small
//        problems of this model, which would otherwise leave CUs idle.
// Operand tiles are k-contiguous [rows][BK + 4 | BK + 2] or k-strided
// [BK][T + 16] (stride = 16 mod 32), conflict-free for the fragment reads.  Accumulators leave through per-wave LDS slabs
// as float4 rows.
#pragma once
#include "common.h"

namespace scae_tile {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 256;
#ifndef SCAE_TILE_BK
#define SCAE_TILE_BK 32
#endif
constexpr int BK = SCAE_TILE_BK;  // K chunk per LDS stage
constexpr int QPR = BK / 4;        // float4 quads per k-contiguous tile row
// k-contiguous tile [rows][LDK].  When both operands are k-contiguous ("VEC")
// every lane owns a run of consecutive k of its rows and reads it with
// ds_read_b128 (row stride BK+4 floats = an odd number of 16-byte units);
// otherwise lanes read single floats at k = 4s + q (stride BK+2 = 2 mod 32).
constexpr int LDKV = BK + 4, LDKS = BK + 2;
constexpr int LDR = 36;  // epilogue slab [32][LDR]: b128 rows on distinct banks
#ifndef SCAE_TILE_STAGES
#define SCAE_TILE_STAGES 4
#endif
constexpr int STAGES = SCAE_TILE_STAGES;  // register prefetch depth of the K loop

template <bool SK>
struct Tile {
  static constexpr int T = SK ? 32 : 64;   // tile rows = tile cols
  static constexpr int NQ = T * QPR / NT;  // float4 per thread per operand per chunk
  static constexpr int LDT = T + 16;       // k-strided tile [BK][LDT]
  static constexpr int OPER = BK * LDT > T * LDKV ? BK * LDT : T * LDKV;
  static constexpr int SMEM = 2 * OPER > 4 * 32 * LDR ? 2 * OPER : 4 * 32 * LDR;
};

// vector fragment reads need both operands k-contiguous and >= 4 k per lane
template <bool SK, bool AK, bool BKC>
constexpr bool use_vec() {
  return AK && BKC && (SK ? BK / 16 : BK / 4) >= 4;
}

template <int NQ>
struct Quads {
  float4 v[NQ];
};

__device__ __forceinline__ float4 ld4(const float *p) {
  return *reinterpret_cast<const float4 *>(p);
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// registers -> LDS.  KC: tile[row][k] (row = id / QPR, k quad = id % QPR);
// otherwise tile[k][row] (k = id / (T/4), row quad = id % (T/4)).
template <bool SK, bool KC, bool VEC>
__device__ __forceinline__ void deposit(float *tile, const Quads<Tile<SK>::NQ> &q) {
  constexpr int T = Tile<SK>::T, LDT = Tile<SK>::LDT;
#pragma unroll
  for (int i = 0; i < Tile<SK>::NQ; ++i) {
    const int id = threadIdx.x + NT * i;
    if (KC && VEC) {
      *reinterpret_cast<float4 *>(tile + (id / QPR) * LDKV + 4 * (id % QPR)) = q.v[i];
    } else if (KC) {
      float *p = tile + (id / QPR) * LDKS + 4 * (id % QPR);  // 8-byte aligned
      *reinterpret_cast<float2 *>(p) = make_float2(q.v[i].x, q.v[i].y);
      *reinterpret_cast<float2 *>(p + 2) = make_float2(q.v[i].z, q.v[i].w);
    } else {
      *reinterpret_cast<float4 *>(tile + (id / (T / 4)) * LDT + 4 * (id % (T / 4))) = q.v[i];
    }
  }
}

// The MFMAs of one K chunk.  !SK: wave (wid>>1, wid&1) owns a 32x32 sub-tile
// and covers all BK k; SK: every wave covers the whole 32x32 tile for its
// quarter of the chunk.  v_mfma_f32_16x16x4_f32 takes A[r][k_q], B[c][k_q] from
// lane (r, q): which four k one instruction contracts is free as long as A
// and B agree, so in VEC mode lane q owns KL consecutive k (vector LDS reads,
// 4 MFMA steps per ds_read_b128), otherwise k = 4s + q.
template <bool SK, bool AK, bool BKC>
__device__ __forceinline__ void mma_chunk(const float *As, const float *Bs, f32x4 (&acc)[2][2],
                                          int wid, int r, int q) {
  constexpr int LDT = Tile<SK>::LDT;
  constexpr bool VEC = use_vec<SK, AK, BKC>();
  constexpr int KW = SK ? BK / 4 : BK;  // k covered by this wave
  constexpr int KL = KW / 4;            // per lane
  const int ro = SK ? 0 : 32 * (wid >> 1), co = SK ? 0 : 32 * (wid & 1);
  const int kw0 = SK ? KW * wid : 0;
  constexpr int US = KL < 4 ? KL : 4;  // MFMA steps per fragment load
#pragma unroll
  for (int s4 = 0; s4 < KL; s4 += US) {
    float a[2][US], b[2][US];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = ro + 16 * i + r, col = co + 16 * i + r;
      if (VEC) {
        const float4 av = ld4(As + row * LDKV + kw0 + KL * q + s4);
        const float4 bv = ld4(Bs + col * LDKV + kw0 + KL * q + s4);
        a[i][0] = av.x, a[i][1] = av.y, a[i][2] = av.z, a[i][3] = av.w;
        b[i][0] = bv.x, b[i][1] = bv.y, b[i][2] = bv.z, b[i][3] = bv.w;
      } else {
#pragma unroll
        for (int u = 0; u < US; ++u) {
          const int k = kw0 + 4 * (s4 + u) + q;
          a[i][u] = AK ? As[row * LDKS + k] : As[k * LDT + row];
          b[i][u] = BKC ? Bs[col * LDKS + k] : Bs[k * LDT + col];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < US; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][u], b[j][u], acc[i][j], 0, 0, 0);
  }
}

// The K loop: chunk c is fetched into registers ST chunks before it is needed
// (fetch(c, ra, rb)), so that ST-1 chunks of MFMAs -- times the waves resident
// on the SIMD -- cover the L2 / HBM latency of the operand loads; `staged(ra)`
// sees every A chunk as it is deposited (bias-gradient hook).
template <int ST, bool SK, bool AK, bool BKC, class Fetch, class Staged>
__device__ __forceinline__ void tile_mainloop(int nchunk, float *As, float *Bs,
                                              f32x4 (&acc)[2][2], int wid, int r, int q,
                                              Fetch fetch, Staged staged) {
  constexpr int NQ = Tile<SK>::NQ;
  Quads<NQ> ra[ST], rb[ST];
#pragma unroll
  for (int s = 0; s < ST; ++s)
    if (s < nchunk) fetch(s, ra[s], rb[s]);
  for (int c0 = 0; c0 < nchunk; c0 += ST) {
#pragma unroll
    for (int s = 0; s < ST; ++s) {
      const int c = c0 + s;
      if (c < nchunk) {   // workgroup-uniform
        __syncthreads();  // the previous chunk's fragment reads are done
        deposit<SK, AK, use_vec<SK, AK, BKC>()>(As, ra[s]);
        deposit<SK, BKC, use_vec<SK, AK, BKC>()>(Bs, rb[s]);
        staged(ra[s]);
        __syncthreads();
        if (c + ST < nchunk) fetch(c + ST, ra[s], rb[s]);
        mma_chunk<SK, AK, BKC>(As, Bs, acc, wid, r, q);
      }
    }
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
