// Shared fp32 MFMA tile loop pieces (v_mfma_f32_16x16x4_f32, wave64) for the
// implicit-GEMM convolutions (K8) and the batched GEMM (K7).
//
// K is walked in 32-wide chunks staged through LDS by 256 threads, in the
// workgroup shapes of struct Tile below: 64 x 64 tiles, or 32 x 32 / 32 x 64
// tiles whose 4 waves split every K chunk four ways and are summed through LDS
// at the end -- more workgroups for the small problems of this model, which
// would otherwise leave CUs idle.
// Operand tiles are k-contiguous [rows][BK + 4 | BK + 2] or k-strided
// [BK][rows + 16] (stride = 16 mod 32), conflict-free for the fragment reads.
// Accumulators leave through per-wave LDS slabs as float4 rows.
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
#ifndef SCAE_TILE_STAGES
#define SCAE_TILE_STAGES 4
#endif
constexpr int STAGES = SCAE_TILE_STAGES;  // register prefetch depth of the K loop

// Workgroup shapes (256 threads = 4 waves):
//   MODE 0: 64 x 64 output tile, 2 x 2 waves each owning a 32 x 32 sub-tile, full K;
//   MODE 1: 32 x 32 tile, every wave covers the whole tile for a quarter of each K
//           chunk, the four partial tiles are summed through LDS at the end;
//   MODE 2: 32 x 64 tile, K split like MODE 1 (one third less operand traffic per
//           flop than MODE 1, twice the MFMAs between barriers).
// A `bool SK` template argument of the users selects MODE 0 / 1.
template <int MODE>
struct Tile {
  static constexpr bool SK = MODE != 0;
  static constexpr int TA = MODE == 0 ? 64 : 32;  // output rows = rows of the A tile
  static constexpr int TB = MODE == 1 ? 32 : 64;  // output cols = rows of the B tile
  static constexpr int T = TA;                    // (for the square shapes)
  static constexpr int NQ = TA * QPR / NT;        // float4 per thread per chunk, A
  static constexpr int NQB = TB * QPR / NT;       //                              B
  static constexpr int LDT = TA + 16, LDTB = TB + 16;  // k-strided tiles [BK][LD]
  static constexpr int OPER = BK * LDT > TA * LDKV ? BK * LDT : TA * LDKV;  // floats, A
  static constexpr int OPB = BK * LDTB > TB * LDKV ? BK * LDTB : TB * LDKV;
  static constexpr int WC = SK ? TB : 32;   // output columns per wave
  static constexpr int NJ = WC / 16;        // 16-column MFMA tiles per wave
  static constexpr int LDR = WC + 4;        // epilogue slab [32][LDR]: odd number of 16 B
  static constexpr int SMEM = OPER + OPB > 4 * 32 * LDR ? OPER + OPB : 4 * 32 * LDR;
  typedef f32x4 Acc;
};

// MODE 3: 128 x 128 output tile, bf16 operands: fp32 values from global memory are rounded
// to bf16 (nearest even) when they are deposited in LDS, the products run on
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate) -- 16x the matrix rate of the fp32 forms for
// BASELINE.json configs[2] ("bs=1024 bf16").  2 x 2 waves, each 64 x 64 = 2 x 2 MFMA tiles;
// LDS tiles are k-contiguous bf16 [128][BK + 8] (80-byte rows: conflict-free 16-byte
// fragment reads) whatever the operand's layout in memory (k-strided operands are
// transposed in registers on their way in), double buffered: one barrier per chunk.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <>
struct Tile<3> {
  static constexpr bool SK = false;
  static constexpr int TA = 128, TB = 128, T = 128;
  static constexpr int NQ = TA * QPR / NT, NQB = TB * QPR / NT;   // 4 float4 per thread
  static constexpr int LDH = BK + 8;                      // bf16 elements per LDS row
  static constexpr int OPER = TA * LDH / 2, OPB = TB * LDH / 2;   // floats per stage
  static constexpr int NJ = 2;
  static constexpr int LDR = 36;                          // epilogue slab [32][36] per wave
  static constexpr int SMEM = 2 * (OPER + OPB) > 4 * 32 * LDR ? 2 * (OPER + OPB) : 4 * 32 * LDR;
  typedef f32x16 Acc;
};
static_assert(BK == 32, "the bf16 tile loop is written for 32-wide K chunks");

// position of the i-th k-strided quad of thread `tid` in a [BK][rows] operand chunk: k within
// the chunk and the quad's first row.  MODE 3 gives a thread four CONSECUTIVE k of one row
// quad (a 4 x 4 block it can transpose in registers).
template <int MODE, int ROWS>
__device__ __forceinline__ void kstr_pos(int tid, int i, int &k, int &row) {
  if (MODE == 3) {
    k = 4 * (tid / (ROWS / 4)) + i;
    row = 4 * (tid % (ROWS / 4));
  } else {
    const int id = tid + NT * i;
    k = id / (ROWS / 4);
    row = 4 * (id % (ROWS / 4));
  }
}

// vector fragment reads need both operands k-contiguous and >= 4 k per lane
template <int MODE, bool AK, bool BKC>
constexpr bool use_vec() {
  return AK && BKC && (MODE != 0 ? BK / 16 : BK / 4) >= 4;
}

template <int NQ>
struct Quads {
  float4 v[NQ];
};

__device__ __forceinline__ float4 ld4(const float *p) {
  return *reinterpret_cast<const float4 *>(p);
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// registers -> bf16 LDS tile [ROWS][LDH].  KC: quad i = 4 consecutive k of row id / QPR;
// otherwise the thread's four quads are a (4 k) x (4 rows) block (kstr_pos<3>).
template <int ROWS, bool KC>
__device__ __forceinline__ void deposit_bf16(unsigned short *tile, const Quads<ROWS * QPR / NT> &q) {
  constexpr int LDH = Tile<3>::LDH;
  if (KC) {
#pragma unroll
    for (int i = 0; i < ROWS * QPR / NT; ++i) {
      const int id = threadIdx.x + NT * i;
      *reinterpret_cast<uint2 *>(tile + (id / QPR) * LDH + 4 * (id % QPR)) =
          make_uint2(pack_bf16(q.v[i].x, q.v[i].y), pack_bf16(q.v[i].z, q.v[i].w));
    }
  } else {
    static_assert(ROWS * QPR / NT == 4, "4 x 4 register blocks");
    const int kg = threadIdx.x / (ROWS / 4), row = 4 * (threadIdx.x % (ROWS / 4));
    unsigned short *p = tile + row * LDH + 4 * kg;
    *reinterpret_cast<uint2 *>(p) =
        make_uint2(pack_bf16(q.v[0].x, q.v[1].x), pack_bf16(q.v[2].x, q.v[3].x));
    *reinterpret_cast<uint2 *>(p + LDH) =
        make_uint2(pack_bf16(q.v[0].y, q.v[1].y), pack_bf16(q.v[2].y, q.v[3].y));
    *reinterpret_cast<uint2 *>(p + 2 * LDH) =
        make_uint2(pack_bf16(q.v[0].z, q.v[1].z), pack_bf16(q.v[2].z, q.v[3].z));
    *reinterpret_cast<uint2 *>(p + 3 * LDH) =
        make_uint2(pack_bf16(q.v[0].w, q.v[1].w), pack_bf16(q.v[2].w, q.v[3].w));
  }
}

// the MFMAs of one chunk from bf16 tiles
__device__ __forceinline__ void mma_chunk_bf16(const unsigned short *As, const unsigned short *Bs,
                                               f32x16 (&acc)[2][2], int wid, int lane) {
  constexpr int LDH = Tile<3>::LDH;
  const int i = lane & 31, kk = lane >> 5, wm = wid >> 1, wn = wid & 1;
#pragma unroll
  for (int s = 0; s < BK / 16; ++s) {
    bf16x8 a[2], b[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      a[t] = *reinterpret_cast<const bf16x8 *>(As + (wm * 64 + t * 32 + i) * LDH + 16 * s + 8 * kk);
      b[t] = *reinterpret_cast<const bf16x8 *>(Bs + (wn * 64 + t * 32 + i) * LDH + 16 * s + 8 * kk);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[u], acc[t][u], 0, 0, 0);
  }
}


// registers -> LDS for a tile of ROWS rows.  KC: tile[row][k] (row = id / QPR,
// k quad = id % QPR); otherwise tile[k][row] (k = id / (ROWS/4), row quad = id % (ROWS/4)).
template <int ROWS, bool KC, bool VEC>
__device__ __forceinline__ void deposit(float *tile, const Quads<ROWS * QPR / NT> &q) {
  constexpr int LDT = ROWS + 16;
#pragma unroll
  for (int i = 0; i < ROWS * QPR / NT; ++i) {
    const int id = threadIdx.x + NT * i;
    if (KC && VEC) {
      *reinterpret_cast<float4 *>(tile + (id / QPR) * LDKV + 4 * (id % QPR)) = q.v[i];
    } else if (KC) {
      float *p = tile + (id / QPR) * LDKS + 4 * (id % QPR);  // 8-byte aligned
      *reinterpret_cast<float2 *>(p) = make_float2(q.v[i].x, q.v[i].y);
      *reinterpret_cast<float2 *>(p + 2) = make_float2(q.v[i].z, q.v[i].w);
    } else {
      *reinterpret_cast<float4 *>(tile + (id / (ROWS / 4)) * LDT + 4 * (id % (ROWS / 4))) =
          q.v[i];
    }
  }
}

// The MFMAs of one K chunk.  v_mfma_f32_16x16x4_f32 takes A[r][k_q], B[c][k_q] from
// lane (r, q): which four k one instruction contracts is free as long as A and B
// agree, so in VEC mode lane q owns KL consecutive k (vector LDS reads, 4 MFMA
// steps per ds_read_b128), otherwise k = 4s + q.
template <int MODE, bool AK, bool BKC>
__device__ __forceinline__ void mma_chunk(const float *As, const float *Bs,
                                          f32x4 (&acc)[2][Tile<MODE>::NJ], int wid, int r,
                                          int q) {
  using TL = Tile<MODE>;
  constexpr bool SK = TL::SK;
  constexpr int LDT = TL::LDT, LDTB = TL::LDTB, NJ = TL::NJ;
  constexpr bool VEC = use_vec<MODE, AK, BKC>();
  constexpr int KW = SK ? BK / 4 : BK;  // k covered by this wave
  constexpr int KL = KW / 4;            // per lane
  const int ro = SK ? 0 : 32 * (wid >> 1), co = SK ? 0 : 32 * (wid & 1);
  const int kw0 = SK ? KW * wid : 0;
  constexpr int US = KL < 4 ? KL : 4;  // MFMA steps per fragment load
#pragma unroll
  for (int s4 = 0; s4 < KL; s4 += US) {
    float a[2][US], b[NJ][US];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = ro + 16 * i + r;
      if (VEC) {
        const float4 av = ld4(As + row * LDKV + kw0 + KL * q + s4);
        a[i][0] = av.x, a[i][1] = av.y, a[i][2] = av.z, a[i][3] = av.w;
      } else {
#pragma unroll
        for (int u = 0; u < US; ++u) {
          const int k = kw0 + 4 * (s4 + u) + q;
          a[i][u] = AK ? As[row * LDKS + k] : As[k * LDT + row];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = co + 16 * j + r;
      if (VEC) {
        const float4 bv = ld4(Bs + col * LDKV + kw0 + KL * q + s4);
        b[j][0] = bv.x, b[j][1] = bv.y, b[j][2] = bv.z, b[j][3] = bv.w;
      } else {
#pragma unroll
        for (int u = 0; u < US; ++u) {
          const int k = kw0 + 4 * (s4 + u) + q;
          b[j][u] = BKC ? Bs[col * LDKS + k] : Bs[k * LDTB + col];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < US; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][u], b[j][u], acc[i][j], 0, 0, 0);
  }
}

// The K loop: chunk c is fetched into registers ST chunks before it is needed
// (fetch(c, ra, rb)), so that ST-1 chunks of MFMAs -- times the waves resident
// on the SIMD -- cover the L2 / HBM latency of the operand loads; `staged(ra)`
// sees every A chunk as it is deposited (bias-gradient hook).
// MODE 3: chunks are fetched two ahead into registers, rounded to bf16 into one of two LDS
// stages, one barrier per chunk (`As` is the start of the workgroup's LDS; Bs is ignored)
template <bool AK, bool BKC, class Fetch, class Staged>
__device__ __forceinline__ void tile_mainloop_bf16(int nchunk, float *smem, f32x16 (&acc)[2][2],
                                                   int wid, Fetch fetch, Staged staged) {
  using TL = Tile<3>;
#ifndef SCAE_BF16_ST
#define SCAE_BF16_ST 1
#endif
  constexpr int ST = SCAE_BF16_ST;   // chunks fetched ahead into registers
  unsigned short *base = reinterpret_cast<unsigned short *>(smem);
  constexpr int STAGE = 2 * (TL::OPER + TL::OPB);   // bf16 elements per stage
  Quads<TL::NQ> ra[ST];
  Quads<TL::NQB> rb[ST];
#pragma unroll
  for (int s = 0; s < ST; ++s)
    if (s < nchunk) fetch(s, ra[s], rb[s]);
  const int lane = threadIdx.x & 63;
  for (int c0 = 0; c0 < nchunk; c0 += ST) {
#pragma unroll
    for (int s = 0; s < ST; ++s) {
      const int c = c0 + s;
      if (c < nchunk) {   // workgroup-uniform
        unsigned short *as = base + (c & 1) * STAGE, *bs = as + 2 * TL::OPER;
        deposit_bf16<TL::TA, AK>(as, ra[s]);
        deposit_bf16<TL::TB, BKC>(bs, rb[s]);
        staged(ra[s]);
        __syncthreads();   // stage s is complete; everyone is done with the other stage's
                           // previous contents (they deposit into it only after this barrier)
        if (c + ST < nchunk) fetch(c + ST, ra[s], rb[s]);
        mma_chunk_bf16(as, bs, acc, wid, lane);
      }
    }
  }
}

template <int ST, int MODE, bool AK, bool BKC, class Fetch, class Staged>
__device__ __forceinline__ void tile_mainloop(int nchunk, float *As, float *Bs,
                                              typename Tile<MODE>::Acc (&acc)[2][Tile<MODE>::NJ],
                                              int wid, int r, int q, Fetch fetch, Staged staged) {
  if constexpr (MODE == 3) {
    tile_mainloop_bf16<AK, BKC>(nchunk, As, acc, wid, fetch, staged);
  } else {
  using TL = Tile<MODE>;
  constexpr bool VEC = use_vec<MODE, AK, BKC>();
  Quads<TL::NQ> ra[ST];
  Quads<TL::NQB> rb[ST];
#pragma unroll
  for (int s = 0; s < ST; ++s)
    if (s < nchunk) fetch(s, ra[s], rb[s]);
  for (int c0 = 0; c0 < nchunk; c0 += ST) {
#pragma unroll
    for (int s = 0; s < ST; ++s) {
      const int c = c0 + s;
      if (c < nchunk) {   // workgroup-uniform
        __syncthreads();  // the previous chunk's fragment reads are done
        deposit<TL::TA, AK, VEC>(As, ra[s]);
        deposit<TL::TB, BKC, VEC>(Bs, rb[s]);
        staged(ra[s]);
        __syncthreads();
        if (c + ST < nchunk) fetch(c + ST, ra[s], rb[s]);
        mma_chunk<MODE, AK, BKC>(As, Bs, acc, wid, r, q);
      }
    }
  }
  }
}

// accumulators -> per-wave LDS slab -> epi(tile row, tile col (multiple of 4), float4)
// MODE 3: each wave's four 32 x 32 accumulator tiles pass one at a time through its
// private [32][36] slab and leave as float4 rows
template <class Epi>
__device__ __forceinline__ void tile_epilogue_bf16(float *smem, const f32x16 (&acc)[2][2], int wid,
                                                   Epi epi) {
  constexpr int LDR = Tile<3>::LDR;
  const int lane = threadIdx.x & 63, i = lane & 31, kk = lane >> 5, wm = wid >> 1, wn = wid & 1;
  __syncthreads();   // operand stages are dead: the slabs alias them
  float *slab = smem + wid * 32 * LDR;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int e = 0; e < 16; ++e)
        slab[((e & 3) + 8 * (e >> 2) + 4 * kk) * LDR + i] = acc[t][u][e];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own slab
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int id = lane + 64 * pass, row = id / 8, c4 = 4 * (id % 8);
        epi(wm * 64 + t * 32 + row, wn * 64 + u * 32 + c4, ld4(slab + row * LDR + c4));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <int MODE, class Epi>
__device__ __forceinline__ void tile_epilogue(float *smem,
                                              const typename Tile<MODE>::Acc (&acc)[2][Tile<MODE>::NJ],
                                              int wid, int r, int q, Epi epi) {
  if constexpr (MODE == 3) {
    tile_epilogue_bf16(smem, acc, wid, epi);
  } else {
  using TL = Tile<MODE>;
  constexpr int LDR = TL::LDR, NJ = TL::NJ, QR = TL::WC / 4;  // float4 per slab row
  __syncthreads();  // operand tiles are dead: the slabs alias them
  float *slab = smem + wid * 32 * LDR;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        slab[(16 * i + 4 * q + reg) * LDR + 16 * j + r] = acc[i][j][reg];
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 32 * QR / NT; ++pass) {
    const int id = threadIdx.x + NT * pass, row = id / QR, c4 = 4 * (id % QR);
    const float *src = smem + row * LDR + c4;
    if (TL::SK) {
      float4 v = ld4(src);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float4 u = ld4(src + w * 32 * LDR);
        v.x += u.x, v.y += u.y, v.z += u.z, v.w += u.w;
      }
      epi(row, c4, v);
    } else {
#pragma unroll
      for (int w = 0; w < 4; ++w)
        epi(32 * (w >> 1) + row, 32 * (w & 1) + c4, ld4(src + w * 32 * LDR));
    }
  }
  }
}

}  // namespace scae_tile
