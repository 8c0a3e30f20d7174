// Single-wave MFMA building blocks shared by the wave-per-tile kernels of the object
// encoder (set_encoder_wave.hip, seed_attention_wave.hip): v_mfma_f32_16x16x4_f32 products
// whose operands a lane reads as float4s of CONSECUTIVE k (which four k an instruction
// contracts is free as long as A and B agree), DPP reductions over the 16 lanes of a row
// group, LDS tile strides.  Lane l: r = l & 15 (row of A / column of B and of the
// result), q = l >> 4 (k group; result rows 4 q .. 4 q + 3).
#pragma once
#include "common.h"

namespace scae_wave {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int D = 16;
constexpr int RS = 24;    // row stride of "R" tiles [rows][16]: conflict-free b128 row reads
constexpr int TS = 40;    // row stride of "T" tiles [..][32] (transposed or N x N)
constexpr int SLOT = 32 * TS;   // floats of an N x N tile

__device__ __forceinline__ float4 ld4(const float *p) {
  return *reinterpret_cast<const float4 *>(p);
}
__device__ __forceinline__ f32x4 splat(float v) { return (f32x4){v, v, v, v}; }
__device__ __forceinline__ void lds_fence() {
  // the wave's own LDS writes are visible to its later reads once they have completed
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// acc += A B over K = 16 (a, b: the lane's four consecutive k)
__device__ __forceinline__ f32x4 mma16(f32x4 acc, float4 a, float4 b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  return acc;
}
// The same K = 16 product with bf16 operands (BASELINE.json configs[2], "bf16 ... MFMA
// attention path"): the lane's four consecutive k are rounded to bf16 (nearest even) and
// contracted by ONE v_mfma_f32_16x16x16_bf16, fp32 accumulate.
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x4 to_bf16x4(float4 v) {
  // (a compiler-visible conversion, not inline asm: the result feeds an MFMA directly and
  // the hazard recogniser must see the VALU write to place the wait states; two 2-vectors:
  // a 4-vector conversion is scalarised into four v_cvt + two v_perm, these are two
  // v_cvt_pk_bf16_f32)
  const f32x2v a = {v.x, v.y}, b = {v.z, v.w};
  const u32x2v u = {__builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2v)),
                    __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2v))};
  return __builtin_bit_cast(bf16x4, u);
}
template <bool BF>
__device__ __forceinline__ f32x4 mma16p(f32x4 acc, float4 a, float4 b) {
  if (BF) return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(to_bf16x4(a), to_bf16x4(b), acc, 0, 0, 0);
  return mma16(acc, a, b);
}
struct F8 {
  float4 lo, hi;
};
__device__ __forceinline__ f32x4 mma32(f32x4 acc, const F8 &a, const F8 &b) {
  return mma16(mma16(acc, a.lo, b.lo), a.hi, b.hi);
}
// Reductions over the 16 lanes of a row group (= one DPP row): quad butterflies, then the
// half-row and row mirrors pair every lane with the other quads -- VALU-speed lane
// exchanges (ds_bpermute shuffles cost an LDS round trip each: a first version of this
// kernel spent half of its time in them).  All 16 lanes get the result.
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
__device__ __forceinline__ float rsum(float v) {
  v += dpp<DPP_XOR1>(v);
  v += dpp<DPP_XOR2>(v);
  v += dpp<DPP_HALF_MIRROR>(v);
  v += dpp<DPP_MIRROR>(v);
  return v;
}
__device__ __forceinline__ float rmax(float v) {
  v = fmaxf(v, dpp<DPP_XOR1>(v));
  v = fmaxf(v, dpp<DPP_XOR2>(v));
  v = fmaxf(v, dpp<DPP_HALF_MIRROR>(v));
  v = fmaxf(v, dpp<DPP_MIRROR>(v));
  return v;
}
// sum over the 16 rows of an O-layout tile (column sum): every lane of column r gets it
__device__ __forceinline__ float csum(const f32x4 &o) {
  float v = (o[0] + o[1]) + (o[2] + o[3]);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// K = 16 NT operands: the lane's 4 NT consecutive k
template <int NT>
struct FK {
  float4 v[NT];
};
template <int NT>
__device__ __forceinline__ f32x4 mmak(f32x4 acc, const FK<NT> &a, const FK<NT> &b) {
#pragma unroll
  for (int j = 0; j < NT; ++j) acc = mma16(acc, a.v[j], b.v[j]);
  return acc;
}
template <int NT, bool BF>
__device__ __forceinline__ f32x4 mmakp(f32x4 acc, const FK<NT> &a, const FK<NT> &b) {
#pragma unroll
  for (int j = 0; j < NT; ++j) acc = mma16p<BF>(acc, a.v[j], b.v[j]);
  return acc;
}
}  // namespace scae_wave
