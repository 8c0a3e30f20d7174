// K8r -- the forward of a small 3x3 "valid" convolution layer (part_encoder.py:26-44,
// nn_ext.py:34-59: Conv2d(k=3, stride s, padding 0) + ReLU) with the INPUT IMAGES RESIDENT in
// LDS, for layers whose whole problem is a few microseconds of matrix time (the encoder's
// 9x9 -> 7x7 and 7x7 -> 5x5 layers at B = 128: 1.85 / 0.94 GFLOP on 256 CUs).
//
// What bounds the ring-pipelined tiles of conv_mfma.hip there (phase stamps, tools/fwd_prof.py)
// is not the matrix pipe: every K chunk costs a workgroup barrier and three LDS-DMA
// instructions per wave, each of which holds the wave's issue for 60-180 cycles that no MFMA
// covers, and 200 / 392 tiles on 256 CUs leave a quarter of the chip idle.  Here
//   * a workgroup owns a GROUP OF IMAGES x 32 output channels: the images' input pixels
//     (G x IH x IW x Cin floats, 25-50 KB) are staged ONCE; the nine taps of the implicit GEMM
//     are nine offsets into that LDS image -- no operand of A is ever re-fetched;
//   * the weights come straight from global memory into MFMA fragments: the filter is kept
//     in a FRAGMENT-MAJOR copy (scae_conv3x3_relayout*: the second half of the `wf` buffer),
//     where the 64 lanes x 16 bytes of one fragment quad are 1 KiB contiguous -- one fully
//     coalesced `global_load_dwordx4`, no LDS, a chunk ahead of its use;
//   * the four waves split K by input-channel block and never meet inside the main loop: no
//     barrier, no DMA, no counted waits -- ds_read_b128 + v_mfma_f32_32x32x2_f32 only; they
//     sum their accumulators through LDS once at the end (fixed order: deterministic);
//   * the grid is B / G x Cout / 32 workgroups: 256 or 512 at B = 128, equal work each.
// Pixels are stored with the 16-byte slots of a pixel XOR-swizzled by (pixel & 15), applied on
// the source side of the staging DMA, so the row-per-lane fragment reads (32 pixels x 16 B, a
// pixel = Cin x 4 B apart) are bank-conflict free.
#include <cstdlib>

#include "bf16x6.h"
#include "mfma_pipe.h"
#include "conv_first_dev.h"

namespace {
namespace pipe = scae_pipe;
using scae_first::ConvGeom;
constexpr int NT = 256;

struct ResArgs {
  const float *in, *wp, *bias, *post_bias;
  float *out, *out_post;
  ConvGeom g;
  int G;   // images per workgroup
};

// The products run on the bf16 matrix cores as six exact partial products of a three-way split
// of the fp32 operands (bf16x6.h): fp32 results at 6 / 16 of the fp32 MFMA time.  The filter
// arrives already split (the fragment-major copy holds its three bf16 planes), the pixels are
// split on the fragments in registers.
template <int MI>
__global__ __launch_bounds__(NT, 2) void conv_res_fwd_kernel(ResArgs a) {
  constexpr bool X6 = true;
  extern __shared__ __attribute__((aligned(1024))) float smem[];
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63,
            li = lane & 31, lk = lane >> 5;
  const ConvGeom &g = a.g;
  const int ntiles = g.Cout / 32;
  const int nt = blockIdx.x % ntiles, img0 = (blockIdx.x / ntiles) * a.G;
  const int nimg = min(a.G, g.B - img0);
  const int ipix = g.IH * g.IW, opix = g.OH * g.OW;
  const int SP = g.Cin / 4;   // 16-byte slots per pixel (a multiple of 32)
  // ---- 1. the group's input pixels -> LDS, slot-swizzled ------------------------------------
  {
    const pipe::rsrc_t rin =
        pipe::make_rsrc(a.in, (unsigned)((size_t)g.B * ipix * g.Cin * 4));
    const int units = a.G * ipix * SP, pieces = (units + 63) / 64;
    const int valid = nimg * ipix;
    for (int j = wid; j < pieces; j += 4) {
      const int u = j * 64 + lane, p = u / SP, s = u - p * SP;
      const int vo = p < valid ? (((img0 * ipix + p) * g.Cin) + ((s ^ (p & 15)) << 2)) * 4
                               : pipe::DMA_ZERO;
      pipe::dma16(rin, smem + j * 256, vo, 0);
    }
  }
  // ---- this lane's rows: output pixel r of the group -> its window's first input pixel -----
  const int rows = nimg * opix;
  int pbase[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int r = min(mi * 32 + li, rows - 1);
    const int n = r / opix, rem = r - n * opix, oh = rem / g.OW, ow = rem - oh * g.OW;
    pbase[mi] = n * ipix + oh * g.stride * g.IW + ow * g.stride;
  }
  // ---- 2. main loop: wave `wid` contracts the channel blocks wid, wid + 4, .. of every tap ---
  const int CB = g.Cin / 32;         // 32-channel blocks per tap
  const int CPT = CB / 4;            // ... per wave
  const int NCW = 9 * CPT;           // this wave's chunks
  // (per 32-channel chunk: 2 pairs of quads x 3 planes x 64 lanes x 16 bytes)
  const uint4 *wq = reinterpret_cast<const uint4 *>(a.wp) + (size_t)nt * (9 * CB) * 384 + lane;
  // X6: NS small-product accumulators per tile -- with one tile per wave three, so that two
  // MFMAs on the same accumulator are three instructions apart (a dependent MFMA waits ~64
  // cycles); with more tiles the tiles themselves interleave
  constexpr int NS = !X6 ? 1 : (MI == 1 ? 3 : 1);
  pipe::f32x16 acc[MI], accl[X6 ? MI : 1][NS];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      acc[mi][e] = 0.f;
      if (X6) {
#pragma unroll
        for (int n = 0; n < NS; ++n) accl[mi][n][e] = 0.f;
      }
    }
  uint4 bq[2][2][3];   // [buffer][pair][plane]
  auto load_b = [&](int c, int buf) {
    const int tap = c / CPT, cb = wid + 4 * (c - tap * CPT);
    const uint4 *src = wq + (size_t)(tap * CB + cb) * 384;
#pragma unroll
    for (int qp = 0; qp < 2; ++qp)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bq[buf][qp][pl] = src[(qp * 3 + pl) * 64];
  };
  load_b(0, 0);
  pipe::wait_vm<0>();   // the staging pieces have landed (and the first B quads: one round trip, shared)
  __syncthreads();
  auto chunk = [&](int c, int buf) {
    if (c + 1 < NCW) load_b(c + 1, buf ^ 1);
    const int tap = c / CPT, cb = wid + 4 * (c - tap * CPT);
    const int kh = tap / 3, kw = tap - 3 * kh;
    const int slot0 = cb * 8 + lk * 4;
    const float *ap[MI];
    int sw[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int p = pbase[mi] + kh * g.IW + kw;
      ap[mi] = smem + p * g.Cin;
      sw[mi] = p & 15;
    }
    if (X6) {
      // two quads = the lane's eight k of one bf16 MFMA (the same eight for A and B)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        scae_x6::Split3 bs;
        bs.hi = __builtin_bit_cast(scae_x6::bf16x8, bq[buf][qp][0]);
        bs.mid = __builtin_bit_cast(scae_x6::bf16x8, bq[buf][qp][1]);
        bs.lo = __builtin_bit_cast(scae_x6::bf16x8, bq[buf][qp][2]);
        scae_x6::Split3 as[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const float4 a0 = pipe::lds4(ap[mi] + (((slot0 + 2 * qp) ^ sw[mi]) << 2));
          const float4 a1 = pipe::lds4(ap[mi] + (((slot0 + 2 * qp + 1) ^ sw[mi]) << 2));
          as[mi] = scae_x6::split3(a0, a1);
        }
        // the six products, product by product over the tiles: consecutive MFMAs write
        // different accumulators (smallest products first within an accumulator)
#define SCAE_X6_STEP(AP, BP, N)                                                                   \
  _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) accl[mi][(N) % NS] =                          \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[mi].AP, bs.BP, accl[mi][(N) % NS], 0, 0, 0);
        SCAE_X6_STEP(hi, lo, 0)
        SCAE_X6_STEP(lo, hi, 1)
        SCAE_X6_STEP(mid, mid, 2)
        SCAE_X6_STEP(hi, mid, 0)
        SCAE_X6_STEP(mid, hi, 1)
#undef SCAE_X6_STEP
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[mi].hi, bs.hi, acc[mi], 0, 0, 0);
      }
    }
  };
  for (int c = 0; c < NCW; c += 2) {
    chunk(c, 0);
    if (c + 1 < NCW) chunk(c + 1, 1);
  }
  if (X6) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      pipe::f32x16 small = accl[mi][0];
#pragma unroll
      for (int n = 1; n < NS; ++n) small += accl[mi][n];
      acc[mi] += small;
    }
  }
  // ---- 3. the four K parts meet in LDS (the staged pixels are dead), fixed order ------------
  __syncthreads();
  // slab [wave][mi][reg][lane]
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) smem[((wid * MI + mi) * 16 + e) * 64 + lane] = acc[mi][e];
  const int col = nt * 32 + li;
  const float bn = a.bias[col];
  float pb[MI][4];
  const int row0 = 8 * wid + 4 * lk;   // wave `wid` finishes registers 4 wid .. 4 wid + 3: these rows of a tile
  if (a.out_post) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = min(mi * 32 + row0 + e, rows - 1);
        pb[mi][e] = a.post_bias[(size_t)col * opix + r % opix];
      }
  }
  __syncthreads();
  const unsigned obytes = (unsigned)((size_t)g.B * opix * g.Cout * 4);
  const pipe::rsrc_t ro = pipe::make_rsrc(a.out, obytes);
  const pipe::rsrc_t rp = pipe::make_rsrc(a.out_post ? a.out_post : a.out, obytes);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += smem[((w * MI + mi) * 16 + 4 * wid + e) * 64 + lane];
      const int r = mi * 32 + row0 + e;
      const float o = fmaxf(v + bn, 0.f);
      // rows past the group's last output pixel: an offset no descriptor covers (dropped)
      const int off = r < rows ? ((img0 * opix + r) * g.Cout + col) * 4 : pipe::DMA_ZERO;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, o), ro, off, 0, 0);
      if (a.out_post)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, o + pb[mi][e]), rp, off, 0, 0);
    }
}

struct ResPlan {
  int G, MI;
  size_t lds;
};
// images per workgroup: the best row use of the 32-row MFMA tiles among the groups that fit
// (<= 4 tiles, LDS for two workgroups per CU); ties go to the smaller group (more workgroups)
bool res_plan(const ConvGeom &g, ResPlan &p, int force_g) {
  if (g.stride < 1 || g.stride > 2 || g.Cin % 128 || g.Cout % 32) return false;
  const int opix = g.OH * g.OW, ipix = g.IH * g.IW;
  double best = 0.;
  p.G = 0;
  for (int G = 1; G <= 8 && G <= g.B; ++G) {
    if (force_g > 0 && G != force_g) continue;
    const int MI = (G * opix + 31) / 32;
    // (the staging DMA writes whole pieces of 64 lanes x 16 B: the image block is rounded up to
    // a piece, or an odd G * ipix would put the last piece's zero half past the allocation)
    const size_t stage = ((size_t)G * ipix * g.Cin * 4 + 1023) & ~(size_t)1023;
    const size_t lds = std::max(stage, (size_t)4 * MI * 16 * 64 * 4);
    if (MI > 4 || lds > (force_g > 0 ? 150 : 64) * 1024) break;
    const double use = (double)G * opix / (32. * MI);
    if (use > best + 1e-9) best = use, p.G = G, p.MI = MI, p.lds = lds;
  }
  return p.G > 0;
}
}  // namespace

extern "C" int scae_conv3x3_fwd_res_supported(int B, int IH, int IW, int Cin, int Cout, int stride) {
  if (B <= 0 || IH < 3 || IW < 3 || Cin <= 0 || Cout <= 0 || stride <= 0) return 0;
  ConvGeom g{B, IH, IW, (IH - 3) / stride + 1, (IW - 3) / stride + 1, Cin, Cout, stride};
  ResPlan p;
  return res_plan(g, p, 0) ? 1 : 0;
}

// `group`: images per workgroup (0: chosen by shape -- the production setting; > 0: forced,
// tests and measurements)
extern "C" int scae_conv3x3_fwd_res_f32(const float *in, const float *wp, const float *bias,
                                        float *out, const float *post_bias, float *out_post, int B,
                                        int IH, int IW, int Cin, int Cout, int stride, int group,
                                        void *stream) {
  SCAE_REQUIRE(in && wp && bias && out && (!out_post || post_bias) && B > 0 && IH >= 3 && IW >= 3 &&
               Cin > 0 && Cout > 0 && stride > 0 && group >= 0);
  ConvGeom g{B, IH, IW, (IH - 3) / stride + 1, (IW - 3) / stride + 1, Cin, Cout, stride};
  ResPlan p;
  if (!res_plan(g, p, group)) return SCAE_ERR_UNSUPPORTED;
  ResArgs a{in, wp, bias, post_bias, out, out_post, g, p.G};
  const dim3 grid((Cout / 32) * ((B + p.G - 1) / p.G));
  hipStream_t st = (hipStream_t)stream;
#define SCAE_RES(M)                                                                              \
  case M: {                                                                                      \
    if (p.lds > 48 * 1024) {                                                                     \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_res_fwd_kernel<M>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds); \
      if (e != hipSuccess) return (int)e;                                                        \
    }                                                                                            \
    scae::launch(conv_res_fwd_kernel<M>, grid, dim3(NT), p.lds, st, a);                          \
    break;                                                                                       \
  }
  switch (p.MI) {
    SCAE_RES(1) SCAE_RES(2) SCAE_RES(3) SCAE_RES(4)
    default: return SCAE_ERR_UNSUPPORTED;
  }
#undef SCAE_RES
  return scae_launch_status();
}
