// fp32 products on the bf16 matrix cores, without giving up fp32 accuracy.
//
// gfx950 multiplies bf16 operands (v_mfma_f32_32x32x16_bf16: 16 k per instruction, 32 cycles)
// at 16 x the rate of fp32 ones (v_mfma_f32_32x32x2_f32: 2 k, 64 cycles).  An fp32 number is
// the EXACT sum of three bf16 numbers -- its 24 significant bits cut into 8 + 8 + 8:
//     hi = x with the low 16 bits cleared,  mid = (x - hi) with the low 16 bits cleared,
//     lo = (x - hi) - mid          (both differences are exact; lo has <= 8 significant bits)
// -- and a product of two bf16 numbers is exact in fp32.  So a * b = sum of nine exact partial
// products, of which the six with weight >= 2^-16,
//     hi hi  |  hi mid, mid hi  |  hi lo, mid mid, lo hi,
// carry everything but 2^-24 of the result (what is dropped -- mid lo, lo mid, lo lo -- is at
// most 2^-23 relative: below the rounding of an fp32 accumulation).  Six bf16 MFMAs per 16 k
// against eight fp32 MFMAs: 192 cycles of the matrix pipe instead of 512.  The five small
// products go into an accumulator of their own (their sum is ~2^-8 of the result: its rounding
// errors are 2^-8 of an fp32 accumulator's), so the result is, if anything, CLOSER to the
// exact dot product than the fp32 MFMA chain's (tools/x6_probe.py; profiles/r06/x6_probe.txt).
// The loops issue the products one KIND at a time over the wave's tiles -- hi lo, lo hi, mid mid,
// hi mid, mid hi into the small accumulator, hi hi into the tile's -- so that consecutive MFMAs
// write different accumulators.
// The split costs ~5.5 VALU instructions per operand element and is done on the fragments in
// registers, so no tensor changes its layout or its type: the data in HBM and LDS stay fp32.
#pragma once
#include "common.h"

namespace scae_x6 {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Split3 {
  bf16x8 hi, mid, lo;
};

// x[0..7]: the lane's eight k of one MFMA operand
__device__ __forceinline__ Split3 split3(const float (&x)[8]) {
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned xb = __float_as_uint(x[e]);
    h[e] = xb & 0xffff0000u;
    const float r1 = x[e] - __uint_as_float(h[e]);   // exact
    m[e] = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(m[e]);      // exact, <= 8 significant bits
    l[e] = __float_as_uint(r2);
  }
  u32x4 ph, pm, pl;
#pragma unroll
  for (int e = 0; e < 4; ++e) {   // element 2 e in the low half, 2 e + 1 in the high half
    ph[e] = __builtin_amdgcn_perm(h[2 * e + 1], h[2 * e], 0x07060302u);
    pm[e] = __builtin_amdgcn_perm(m[2 * e + 1], m[2 * e], 0x07060302u);
    pl[e] = __builtin_amdgcn_perm(l[2 * e + 1], l[2 * e], 0x07060302u);
  }
  Split3 s;
  s.hi = __builtin_bit_cast(bf16x8, ph);
  s.mid = __builtin_bit_cast(bf16x8, pm);
  s.lo = __builtin_bit_cast(bf16x8, pl);
  return s;
}
__device__ __forceinline__ Split3 split3(float4 a, float4 b) {
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  return split3(x);
}

}  // namespace scae_x6
