// The image layer of the part-capsule CNN encoder (K8, C_in <= 4: nine-tap dot products) as
// device functions, so that its forward -- and the parameter-only filter re-layouts that ride
// with it -- can run as block ranges of the training step's prologue launch
// (step_prologue.hip) as well as in conv_mfma.hip's own kernel.
#pragma once
#include "common.h"

namespace scae_first {
struct ConvGeom {
  int B, IH, IW, OH, OW, Cin, Cout, stride;
};

// W[co][ci][3][3] -> Wf[co][tap][ci] (+ its fragment-major copy behind it), Wd[ci][tap][co] for up
// to 8 layers
struct RelayoutBatch {
  const float *w[8];
  float *wf[8], *wd[8];
  int Cout[8], Cin[8];
  // bf16 copies of wf (Cout,9,Cin) / wd (Cin,9,Cout) for the bf16-resident kernels
  // (conv_bf16.hip), nullable
  unsigned short *wfh[8], *wdh[8];
};
// fp32 -> bf16 bits, round to nearest even (what v_cvt_pk_bf16_f32 does; finite values)
__host__ __device__ inline unsigned short bf16_bits(float v) {
  union { float f; unsigned u; } c;
  c.f = v;
  if ((c.u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((c.u >> 16) | 0x40);   // NaN
  return (unsigned short)((c.u + 0x7fffu + ((c.u >> 16) & 1u)) >> 16);
}
// The FRAGMENT-MAJOR copy the image-resident forward (conv_resident.hip) reads -- behind the first
// Cout*9*Cin floats of a `wf` buffer -- holds the filter ALREADY SPLIT into the three bf16 planes
// of bf16x6.h (hi / mid / lo: the exact 8 + 8 + 8 cut of each fp32 value), so that kernel splits
// no B operand.  Unit = bf16; element (co, tap, ci), plane p at
//   [co / 32][chunk = tap * Cin/32 + ci/32][pair = ci%16 / 8][p][lane = (ci%32 / 16) * 32 + co%32][ci%8]:
// the 64 lanes x 16 bytes of one MFMA operand (v_mfma_f32_32x32x16_bf16: lane = (column, k half),
// eight k each) are 1 KiB contiguous -- one coalesced global_load_dwordx4.  1.5 * Cout*9*Cin floats.
// Needs Cin % 32 == 0 and Cout % 32 == 0.
__host__ __device__ inline size_t packed_index(int Cin, int co, int tap, int ci, int plane) {
  const size_t chunk = (size_t)(co >> 5) * (9 * (Cin >> 5)) + tap * (Cin >> 5) + (ci >> 5);
  return ((((chunk * 2 + ((ci & 15) >> 3)) * 3 + plane) * 64 + ((ci & 31) >> 4) * 32 + (co & 31)) << 3) +
         (ci & 7);
}
// the three bf16 planes of an fp32 value (bf16x6.h's split: masks and exact differences)
__host__ __device__ inline void split3_bits(float v, unsigned short (&out)[3]) {
  union { float f; unsigned u; } a, b, c;
  a.f = v;
  const unsigned h = a.u & 0xffff0000u;
  a.u = h;
  b.f = v - a.f;
  const unsigned m = b.u & 0xffff0000u;
  b.u = m;
  c.f = (v - a.f) - b.f;
  out[0] = (unsigned short)(h >> 16), out[1] = (unsigned short)(m >> 16),
  out[2] = (unsigned short)(c.u >> 16);
}
__host__ __device__ inline bool packed_copy(int Cout, int Cin) { return Cin % 32 == 0 && Cout % 32 == 0; }
__device__ __forceinline__ void relayout_one(const float *w, float *wf, float *wd, int Cout,
                                             int Cin, int e, unsigned short *wfh = nullptr,
                                             unsigned short *wdh = nullptr) {
  if (e >= Cout * Cin * 9) return;
  const int co = e / (Cin * 9), rem = e - co * Cin * 9, ci = rem / 9, tap = rem - ci * 9;
  const float v = w[e];
  wf[((size_t)co * 9 + tap) * Cin + ci] = v;
  wd[((size_t)ci * 9 + tap) * Cout + co] = v;
  if (packed_copy(Cout, Cin)) {
    unsigned short pl[3], *dst = reinterpret_cast<unsigned short *>(wf + (size_t)Cout * 9 * Cin);
    split3_bits(v, pl);
#pragma unroll
    for (int p = 0; p < 3; ++p) dst[packed_index(Cin, co, tap, ci, p)] = pl[p];
  }
  if (wfh) wfh[((size_t)co * 9 + tap) * Cin + ci] = bf16_bits(v);
  if (wdh) wdh[((size_t)ci * 9 + tap) * Cout + co] = bf16_bits(v);
}
__device__ __forceinline__ void relayout_batch(const RelayoutBatch &r, int l, int e) {
  relayout_one(r.w[l], r.wf[l], r.wd[l], r.Cout[l], r.Cin[l], e, r.wfh[l], r.wdh[l]);
}
// fills a RelayoutBatch; returns the workgroups (of 256 elements) per layer or < 0
inline int fill_relayout(RelayoutBatch &r, int n_layers, const float *const *w, float *const *wf,
                         float *const *wd, const int *Cout, const int *Cin,
                         unsigned short *const *wfh = nullptr,
                         unsigned short *const *wdh = nullptr) {
  if (!(n_layers > 0 && n_layers <= 8 && w && wf && wd && Cout && Cin)) return -1;
  int nmax = 0;
  for (int l = 0; l < n_layers; ++l) {
    if (!(w[l] && wf[l] && wd[l] && Cout[l] > 0 && Cin[l] > 0)) return -1;
    r.w[l] = w[l], r.wf[l] = wf[l], r.wd[l] = wd[l], r.Cout[l] = Cout[l], r.Cin[l] = Cin[l];
    r.wfh[l] = wfh ? wfh[l] : nullptr, r.wdh[l] = wdh ? wdh[l] : nullptr;
    nmax = nmax > Cout[l] * Cin[l] * 9 ? nmax : Cout[l] * Cin[l] * 9;
  }
  return (nmax + 255) / 256;
}

// ---- first layer (image, C_in <= 4): direct kernels ----------------------------
// One workgroup per (image, pixel slice): the image is staged in LDS, each wave
// owns 64 output channels (lane = channel: NHWC stores / loads are 256-byte
// coalesced rows) and, when C_out < 256, a share of the slice's pixels.
struct FirstSplit {
  int nchunk, parts, slices;
};
__host__ __device__ inline FirstSplit first_split(int B, int Cout) {
  FirstSplit f;
  f.nchunk = Cout / 64;
  f.parts = (f.nchunk <= 4 && 4 % f.nchunk == 0) ? 4 / f.nchunk : 1;
  int s = (512 + B - 1) / B;  // >= 512 workgroups
  f.slices = s < 1 ? 1 : (s > 8 ? 8 : s);
  return f;
}

// the lane = (pixel, channel quad) forms of the two image-layer kernels: C_out / 4 lanes
// per pixel must divide a wave
__host__ __device__ inline bool first_vec(int Cout) {
  return Cout == 64 || Cout == 128 || Cout == 256;
}
__device__ __forceinline__ void stage_image(float *s_img, const float *img, int n, int count) {
  for (int e = threadIdx.x; e < count; e += 256) s_img[e] = img[(size_t)n * count + e];
  __syncthreads();
}

// workgroup `blk` (256 threads) of the forward: image NCHW (B,Cin,IH,IW),
// w [Cout][Cin][3][3] -> out NHWC, ReLU; s_img: Cin * IH * IW floats of LDS
template <int CIN>
__device__ __forceinline__ void fwd_block(const float *__restrict__ img,
                                          const float *__restrict__ w,
                                          const float *__restrict__ bias,
                                          float *__restrict__ out, const ConvGeom &g, int blk,
                                          float *s_img,
                                          unsigned short *__restrict__ out_h = nullptr) {
  // out_h (nullable): the same values as bf16 INSTEAD of the fp32 tensor (the bf16-resident
  // layers behind it read nothing else)
  const FirstSplit f = first_split(g.B, g.Cout);
  const int n = blk / f.slices, slice = blk % f.slices;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  stage_image(s_img, img, n, CIN * g.IH * g.IW);
  const int P = g.OH * g.OW, per = (P + f.slices - 1) / f.slices;
  const int pbeg = slice * per, pend = min(P, pbeg + per);
  if (first_vec(g.Cout)) {
    // lane = (pixel of the pass, channel quad): a wave stores 1 KiB of consecutive NHWC
    // floats per instruction (lane-per-channel dword stores left the kernel at a third of
    // the write bandwidth)
    const int QL = g.Cout / 4, PPW = 64 / QL, cq = lane % QL, ps = lane / QL;
    float wr[4][CIN * 9], b4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int k = 0; k < CIN * 9; ++k) wr[c][k] = w[(size_t)(4 * cq + c) * CIN * 9 + k];
      b4[c] = bias[4 * cq + c];
    }
    const int step = 4 * PPW;
    int p = pbeg + wave * PPW + ps;
    int oh = p / g.OW, ow = p - oh * g.OW;
    float *dst = out + ((size_t)n * P + p) * g.Cout + 4 * cq;
    unsigned short *dst_h = out_h ? out_h + ((size_t)n * P + p) * g.Cout + 4 * cq : nullptr;
    for (; p < pend; p += step) {
      const float *src = s_img + oh * g.stride * g.IW + ow * g.stride;
      float acc[4] = {b4[0], b4[1], b4[2], b4[3]};
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float x = src[(ci * g.IH + t / 3) * g.IW + t % 3];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = fmaf(x, wr[c][ci * 9 + t], acc[c]);
        }
      if (dst_h) {
        const unsigned lo = bf16_bits(fmaxf(acc[0], 0.f)) | ((unsigned)bf16_bits(fmaxf(acc[1], 0.f)) << 16);
        const unsigned hi = bf16_bits(fmaxf(acc[2], 0.f)) | ((unsigned)bf16_bits(fmaxf(acc[3], 0.f)) << 16);
        *reinterpret_cast<uint2 *>(dst_h) = make_uint2(lo, hi);
        dst_h += (size_t)step * g.Cout;
      } else {
        *reinterpret_cast<float4 *>(dst) = make_float4(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f),
                                                       fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
      }
      dst += (size_t)step * g.Cout;
      ow += step;
      while (ow >= g.OW) ow -= g.OW, ++oh;
    }
    return;
  }
  for (int wi = wave; wi < f.nchunk * f.parts; wi += 4) {
    const int co = (wi % f.nchunk) * 64 + lane, part = wi / f.nchunk;
    float wr[CIN * 9];
#pragma unroll
    for (int k = 0; k < CIN * 9; ++k) wr[k] = w[(size_t)co * CIN * 9 + k];
    const float b = bias[co];
    // (oh, ow) of pixel p carried along instead of divided out per pixel
    int oh = (pbeg + part) / g.OW, ow = (pbeg + part) - oh * g.OW;
    float *dst = out + ((size_t)n * P + pbeg + part) * g.Cout + co;
    for (int p = pbeg + part; p < pend; p += f.parts) {
      const float *src = s_img + oh * g.stride * g.IW + ow * g.stride;
      float acc = b;
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int t = 0; t < 9; ++t)
          acc = fmaf(src[(ci * g.IH + t / 3) * g.IW + t % 3], wr[ci * 9 + t], acc);
      if (out_h)
        out_h[dst - out] = bf16_bits(fmaxf(acc, 0.f));
      else
        *dst = fmaxf(acc, 0.f);
      dst += (size_t)f.parts * g.Cout;
      ow += f.parts;
      while (ow >= g.OW) ow -= g.OW, ++oh;
    }
  }
}
}  // namespace scae_first
