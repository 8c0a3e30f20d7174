// K8 -- the 3x3 "valid" convolutions of the part-capsule CNN encoder
// (part_encoder.py:26-44, nn_ext.py:34-59: Conv2d(k=3, stride s, no padding) +
// ReLU stacks) as implicit GEMMs on the CDNA4 fp32 matrix cores.
//
// Activations are kept NHWC so that one (output pixel, filter tap) row of the
// implicit A matrix is C_in contiguous floats; weights are re-laid once per
// step as Wf[co][tap][ci] (forward) and Wd[ci][tap][co] (data gradient).
//   forward : out[m][co]      = relu(sum_{tap,ci} in[pix(m,tap)][ci] Wf[co][tap][ci] + b)
//   dgrad   : din[m'][ci]     = gate(sum_{tap,co} dpre[opix(m',tap)][co] Wd[ci][tap][co])
//             blockIdx.z = stride-parity class of the input pixel, so that no
//             tile multiplies structurally-zero taps
//   wgrad   : dW[tap][co][ci] = sum_m dpre[m][co] in[pix(m,tap)][ci]   (split over m),
//             db[co] = sum_m dpre[m][co] falls out of the staged A operand
// All passes share one MFMA tile loop (v_mfma_f32_16x16x4_f32, exact fp32
// products, fp32 accumulate) over 32-wide K chunks with register-prefetched
// staging, in two shapes:
//   * 64 x 64 tiles, 2 x 2 waves each owning a 32 x 32 sub-tile (large layers);
//   * 32 x 32 tiles whose 4 waves split every K chunk four ways and are summed
//     through LDS at the end (the encoder's layers have only 3k-10k output
//     pixels x 128 channels: 64 x 64 tiles would leave most of the 256 CUs
//     without a workgroup, 32 x 32 split-K tiles give 4x as many).
// Either way the accumulators leave through LDS as float4 rows (coalesced NHWC
// stores).  The first layer (C_in <= 4: nine-tap dot products) is a direct kernel.
#include "mfma_tile.h"

namespace {
using namespace scae_tile;

struct ConvGeom {
  int B, IH, IW, OH, OW, Cin, Cout, stride;
};

#define SCAE_TILE_PROLOGUE                                                      \
  using TL = Tile<SK>;                                                          \
  constexpr int T = TL::T, NQ = TL::NQ;                                         \
  __shared__ __attribute__((aligned(16))) float smem[TL::SMEM];                 \
  float *As = smem, *Bs = smem + TL::OPER;                                      \
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63, r = lane & 15,  \
            q = lane >> 4;                                                      \
  f32x4 acc[2][2];                                                              \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
      acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

// ---- forward: grid (Cout/T, ceil(M/T)) ------------------------------------------
template <bool SK>
__global__ __launch_bounds__(NT) void conv_fwd_kernel(const float *__restrict__ in,
                                                      const float *__restrict__ wf,
                                                      const float *__restrict__ bias,
                                                      float *__restrict__ out, ConvGeom g) {
  SCAE_TILE_PROLOGUE
  const int M = g.B * g.OH * g.OW, K = 9 * g.Cin;
  const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;
  // each thread stages the same rows every chunk: resolve their pixels once
  long abase[NQ];
  const float *bptr[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int id = tid + NT * i, m = m0 + id / QPR, kq = 4 * (id % QPR);
    abase[i] = -1;
    if (m < M) {
      const int n = m / (g.OH * g.OW), rem = m - n * g.OH * g.OW, oh = rem / g.OW,
                ow = rem - oh * g.OW;
      abase[i] = (((long)n * g.IH + oh * g.stride) * g.IW + ow * g.stride) * g.Cin + kq;
    }
    bptr[i] = wf + (size_t)(n0 + id / QPR) * K + kq;
  }
  auto fetch = [&](int c, Quads<NQ> &ra, Quads<NQ> &rb) {
    const int k0 = c * BK;
    const int tap = k0 / g.Cin, ci0 = k0 - tap * g.Cin, kh = tap / 3, kw = tap - kh * 3;
    const int off = (kh * g.IW + kw) * g.Cin + ci0;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      ra.v[i] = abase[i] >= 0 ? ld4(in + abase[i] + off) : zero4();
      rb.v[i] = ld4(bptr[i] + k0);
    }
  };
  tile_mainloop<STAGES, SK, true, true>(K / BK, As, Bs, acc, wid, r, q, fetch,
                                        [](const Quads<NQ> &) {});
  tile_epilogue<SK>(smem, acc, wid, r, q, [&](int row, int col, float4 v) {
    const int m = m0 + row, n = n0 + col;
    if (m >= M) return;
    const float4 b = ld4(bias + n);
    *reinterpret_cast<float4 *>(out + (size_t)m * g.Cout + n) =
        make_float4(fmaxf(v.x + b.x, 0.f), fmaxf(v.y + b.y, 0.f), fmaxf(v.z + b.z, 0.f),
                    fmaxf(v.w + b.w, 0.f));
  });
}

// ---- data gradient: grid (Cin/T, ceil(M0/T), stride^2) ---------------------------
// din rows of parity class (ph, pw): (n, a, b) -> input pixel (s*a + ph, s*b + pw).
// gate: the ReLU output of the producing layer at the same pixels, or nullptr.
template <bool SK>
__global__ __launch_bounds__(NT) void conv_dgrad_kernel(const float *__restrict__ dpre,
                                                        const float *__restrict__ wd,
                                                        const float *__restrict__ gate,
                                                        float *__restrict__ din, ConvGeom g) {
  const int ph = blockIdx.z / g.stride, pw = blockIdx.z % g.stride;
  const int AH = (g.IH - ph + g.stride - 1) / g.stride, AW = (g.IW - pw + g.stride - 1) / g.stride;
  const int M = g.B * AH * AW, KT = 9 * g.Cout;
  if ((int)blockIdx.y * Tile<SK>::T >= M) return;
  SCAE_TILE_PROLOGUE
  const int m0 = blockIdx.y * T, n0 = blockIdx.x * T, sh = g.stride - 1;  // stride 1 or 2
  // valid taps of this class: kh = ph + s*u, kw = pw + s*v (< 3); each Cout/32 chunks
  const int nw = (2 - pw) / g.stride + 1, ntap = ((2 - ph) / g.stride + 1) * nw;
  const int cpt = g.Cout / BK, nchunk = ntap * cpt;
  int pn[NQ], pih[NQ], piw[NQ];
  const float *bptr[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int id = tid + NT * i, m = m0 + id / QPR, kq = 4 * (id % QPR);
    pn[i] = -1, pih[i] = 0, piw[i] = 0;
    if (m < M) {
      const int n = m / (AH * AW), rem = m - n * AH * AW, a = rem / AW, b = rem - a * AW;
      pn[i] = n * g.OH * g.OW, pih[i] = g.stride * a + ph, piw[i] = g.stride * b + pw;
    }
    bptr[i] = wd + (size_t)(n0 + id / QPR) * KT + kq;
  }
  auto fetch = [&](int c, Quads<NQ> &ra, Quads<NQ> &rb) {
    const int t = c / cpt, co0 = (c - t * cpt) * BK;
    const int kh = ph + g.stride * (t / nw), kw = pw + g.stride * (t % nw);
    const int koff = (kh * 3 + kw) * g.Cout + co0;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int dh = pih[i] - kh, dw = piw[i] - kw, oh = dh >> sh, ow = dw >> sh;
      const bool ok = pn[i] >= 0 && dh >= 0 && dw >= 0 && oh < g.OH && ow < g.OW;
      ra.v[i] = ok ? ld4(dpre + (size_t)(pn[i] + oh * g.OW + ow) * g.Cout + co0 +
                         4 * ((tid + NT * i) % QPR))
                   : zero4();
      rb.v[i] = ld4(bptr[i] + koff);
    }
  };
  tile_mainloop<STAGES, SK, true, true>(nchunk, As, Bs, acc, wid, r, q, fetch,
                                        [](const Quads<NQ> &) {});
  tile_epilogue<SK>(smem, acc, wid, r, q, [&](int row, int col, float4 v) {
    const int m = m0 + row;
    if (m >= M) return;
    const int nb = m / (AH * AW), rem = m - nb * AH * AW, a = rem / AW, b = rem - a * AW;
    const size_t o =
        (((size_t)nb * g.IH + g.stride * a + ph) * g.IW + g.stride * b + pw) * g.Cin + n0 + col;
    if (gate) {
      const float4 gt = ld4(gate + o);
      v.x = gt.x > 0.f ? v.x : 0.f, v.y = gt.y > 0.f ? v.y : 0.f;
      v.z = gt.z > 0.f ? v.z : 0.f, v.w = gt.w > 0.f ? v.w : 0.f;
    }
    *reinterpret_cast<float4 *>(din + o) = v;
  });
}

// ---- weight gradient: grid (Cin/T, Cout/T, 9 taps * S splits) --------------------
// partial[(split*9 + tap)][co][ci], then bias partials [split][co] after 9*S slabs
template <bool SK>
__global__ __launch_bounds__(NT) void conv_wgrad_kernel(const float *__restrict__ dpre,
                                                        const float *__restrict__ in,
                                                        float *__restrict__ partial, ConvGeom g,
                                                        int splits) {
  SCAE_TILE_PROLOGUE
  const int M = g.B * g.OH * g.OW;
  const int tap = blockIdx.z % 9, split = blockIdx.z / 9, kh = tap / 3, kw = tap - kh * 3;
  const int per = ((M + splits - 1) / splits + BK - 1) / BK * BK;
  const int kbeg = split * per, kend = min(M, kbeg + per);
  const int co0 = blockIdx.y * T, ci0 = blockIdx.x * T;
  const bool want_bias = tap == 0 && blockIdx.x == 0;  // workgroup-uniform
  float4 bsum = zero4();
  auto fetch = [&](int c, Quads<NQ> &ra, Quads<NQ> &rb) {
    const int k0 = kbeg + c * BK;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int id = tid + NT * i, m = k0 + id / (T / 4), rq = 4 * (id % (T / 4));
      ra.v[i] = rb.v[i] = zero4();
      if (m < kend) {
        const int n = m / (g.OH * g.OW), rem = m - n * g.OH * g.OW, oh = rem / g.OW,
                  ow = rem - oh * g.OW;
        const size_t pix = ((size_t)n * g.IH + oh * g.stride + kh) * g.IW + ow * g.stride + kw;
        ra.v[i] = ld4(dpre + (size_t)m * g.Cout + co0 + rq);
        rb.v[i] = ld4(in + pix * g.Cin + ci0 + rq);
      }
    }
  };
  const int nchunk = kbeg < kend ? (kend - kbeg + BK - 1) / BK : 0;
  tile_mainloop<STAGES, SK, false, false>(nchunk, As, Bs, acc, wid, r, q, fetch,
                                          [&](const Quads<NQ> &ra) {
    if (want_bias) {
#pragma unroll
      for (int i = 0; i < NQ; ++i)
        bsum.x += ra.v[i].x, bsum.y += ra.v[i].y, bsum.z += ra.v[i].z, bsum.w += ra.v[i].w;
    }
  });
  float *dst = partial + (size_t)(split * 9 + tap) * g.Cout * g.Cin;
  tile_epilogue<SK>(smem, acc, wid, r, q, [&](int row, int col, float4 v) {
    *reinterpret_cast<float4 *>(dst + (size_t)(co0 + row) * g.Cin + ci0 + col) = v;
  });
  if (want_bias) {  // column sums of the staged dpre rows: threads tid % (T/4) share a quad
    __syncthreads();
    reinterpret_cast<float4 *>(smem)[tid] = bsum;
    __syncthreads();
    if (tid < T) {
      float sum = 0.f;
      for (int j = 0; j < NT / (T / 4); ++j) sum += smem[4 * (tid / 4 + (T / 4) * j) + (tid & 3)];
      partial[(size_t)splits * 9 * g.Cout * g.Cin + (size_t)split * g.Cout + co0 + tid] = sum;
    }
  }
}

// ---- small helpers ---------------------------------------------------------------
// W[co][ci][3][3] -> Wf[co][tap][ci], Wd[ci][tap][co]
__global__ void relayout_weights_kernel(const float *__restrict__ w, float *__restrict__ wf,
                                        float *__restrict__ wd, int Cout, int Cin) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Cout * Cin * 9) return;
  const int co = e / (Cin * 9), rem = e - co * Cin * 9, ci = rem / 9, tap = rem - ci * 9;
  const float v = w[e];
  wf[((size_t)co * 9 + tap) * Cin + ci] = v;
  wd[((size_t)ci * 9 + tap) * Cout + co] = v;
}

// dW[co][ci][tap] = sum_split partial[(split*9+tap)][co][ci]; db[co] = sum_split bias partials
__global__ void reduce_wgrad_kernel(const float *__restrict__ partial, float *__restrict__ dw,
                                    float *__restrict__ db, int Cout, int Cin, int splits) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;  // over (tap, co, ci): coalesced reads
  const int n = 9 * Cout * Cin;
  if (e < n) {
    const int tap = e / (Cout * Cin), rem = e - tap * Cout * Cin, co = rem / Cin,
              ci = rem - co * Cin;
    float acc = 0.f;
    for (int s = 0; s < splits; ++s)
      acc += partial[((size_t)(s * 9 + tap) * Cout + co) * Cin + ci];
    dw[((size_t)co * Cin + ci) * 9 + tap] = acc;
  } else if (e < n + Cout && db) {
    float acc = 0.f;
    for (int s = 0; s < splits; ++s) acc += partial[(size_t)splits * n + (size_t)s * Cout + e - n];
    db[e - n] = acc;
  }
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

__device__ __forceinline__ void stage_image(float *s_img, const float *img, int n, int count) {
  for (int e = threadIdx.x; e < count; e += 256) s_img[e] = img[(size_t)n * count + e];
  __syncthreads();
}

// image NCHW (B,Cin,IH,IW), w [Cout][Cin][3][3] -> out NHWC, ReLU
template <int CIN>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(const float *__restrict__ img,
                                                             const float *__restrict__ w,
                                                             const float *__restrict__ bias,
                                                             float *__restrict__ out,
                                                             ConvGeom g) {
  extern __shared__ float s_img[];
  const FirstSplit f = first_split(g.B, g.Cout);
  const int n = blockIdx.x / f.slices, slice = blockIdx.x % f.slices;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  stage_image(s_img, img, n, CIN * g.IH * g.IW);
  const int P = g.OH * g.OW, per = (P + f.slices - 1) / f.slices;
  const int pbeg = slice * per, pend = min(P, pbeg + per);
  for (int wi = wave; wi < f.nchunk * f.parts; wi += 4) {
    const int co = (wi % f.nchunk) * 64 + lane, part = wi / f.nchunk;
    float wr[CIN * 9];
#pragma unroll
    for (int k = 0; k < CIN * 9; ++k) wr[k] = w[(size_t)co * CIN * 9 + k];
    const float b = bias[co];
    for (int p = pbeg + part; p < pend; p += f.parts) {
      const int oh = p / g.OW, ow = p - oh * g.OW;
      const float *src = s_img + oh * g.stride * g.IW + ow * g.stride;
      float acc = b;
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int t = 0; t < 9; ++t)
          acc = fmaf(src[(ci * g.IH + t / 3) * g.IW + t % 3], wr[ci * 9 + t], acc);
      out[((size_t)n * P + p) * g.Cout + co] = fmaxf(acc, 0.f);
    }
  }
}

// weight/bias gradient partials: partial[(n*slices + slice)*parts + part] = one row
// [dW (Cout, CIN*9) | db (Cout)]; the caller sums the rows.
template <int CIN>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(const float *__restrict__ dpre,
                                                               const float *__restrict__ img,
                                                               float *__restrict__ partial,
                                                               ConvGeom g) {
  extern __shared__ float s_img[];
  constexpr int K1 = CIN * 9 + 1;
  const FirstSplit f = first_split(g.B, g.Cout);
  const int n = blockIdx.x / f.slices, slice = blockIdx.x % f.slices;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  stage_image(s_img, img, n, CIN * g.IH * g.IW);
  const int P = g.OH * g.OW, per = (P + f.slices - 1) / f.slices;
  const int pbeg = slice * per, pend = min(P, pbeg + per);
  for (int wi = wave; wi < f.nchunk * f.parts; wi += 4) {
    const int co = (wi % f.nchunk) * 64 + lane, part = wi / f.nchunk;
    float acc[K1];
#pragma unroll
    for (int k = 0; k < K1; ++k) acc[k] = 0.f;
    for (int p = pbeg + part; p < pend; p += f.parts) {
      const int oh = p / g.OW, ow = p - oh * g.OW;
      const float *src = s_img + oh * g.stride * g.IW + ow * g.stride;
      const float d = dpre[((size_t)n * P + p) * g.Cout + co];
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int t = 0; t < 9; ++t)
          acc[ci * 9 + t] = fmaf(d, src[(ci * g.IH + t / 3) * g.IW + t % 3], acc[ci * 9 + t]);
      acc[K1 - 1] += d;
    }
    // row layout [dW (Cout x CIN*9) | db (Cout)]
    float *row = partial + ((size_t)blockIdx.x * f.parts + part) * g.Cout * K1;
#pragma unroll
    for (int k = 0; k < K1 - 1; ++k) row[(size_t)co * (K1 - 1) + k] = acc[k];
    row[(size_t)g.Cout * (K1 - 1) + co] = acc[K1 - 1];
  }
}

int check_geom(const ConvGeom &g, bool gemm) {
  if (g.B <= 0 || g.IH <= 0 || g.IW <= 0 || g.Cin <= 0 || g.Cout <= 0 || g.stride <= 0)
    return SCAE_ERR_BAD_ARG;
  if (g.OH != (g.IH - 3) / g.stride + 1 || g.OW != (g.IW - 3) / g.stride + 1 || g.OH <= 0 ||
      g.OW <= 0)
    return SCAE_ERR_BAD_ARG;
  if (gemm && (g.Cin % 64 || g.Cout % 64 || g.stride > 2)) return SCAE_ERR_UNSUPPORTED;
  return SCAE_OK;
}

// 64x64 tiles once they fill the chip a few times over, 32x32 split-K tiles below
inline bool small_tiles(long tiles64) { return tiles64 < 1024; }

struct WgradPlan {
  bool small;
  int splits;
};
WgradPlan wgrad_plan(int M, int Cin, int Cout) {
  // 64x64 tiles (half the L2 traffic per flop of the 32x32 shape); the grid is
  // filled by splitting the pixel (K) dimension instead: >= 512 workgroups of
  // >= 8 K chunks each, measured best on the encoder's 128-channel layers
  WgradPlan p;
  const long tiles64 = (long)(Cin / 64) * (Cout / 64) * 9;
  p.small = false;
  long s = (512 + tiles64 - 1) / tiles64;
  const long cap = (M / BK) / 8;
  s = s > cap ? cap : s;
  p.splits = (int)(s < 1 ? 1 : (s > 32 ? 32 : s));
  return p;
}
}  // namespace

extern "C" int scae_conv3x3_relayout_f32(const float *w, float *wf, float *wd, int Cout,
                                         int Cin, void *stream) {
  SCAE_REQUIRE(w && wf && wd && Cout > 0 && Cin > 0);
  const int n = Cout * Cin * 9;
  hipLaunchKernelGGL(relayout_weights_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, w, wf, wd, Cout, Cin);
  return scae_launch_status();
}

extern "C" int scae_conv3x3_first_fwd_f32(const float *img, const float *w, const float *bias,
                                          float *out, int B, int Cin, int IH, int IW, int Cout,
                                          int stride, void *stream) {
  ConvGeom g{B, IH, IW, (IH - 3) / stride + 1, (IW - 3) / stride + 1, Cin, Cout, stride};
  int rc = check_geom(g, false);
  if (rc) return rc;
  SCAE_REQUIRE(img && w && bias && out);
  if (Cout % 64) return SCAE_ERR_UNSUPPORTED;
  const FirstSplit f = first_split(B, Cout);
  const size_t lds = (size_t)Cin * IH * IW * sizeof(float);
  if (lds > 64 * 1024) return SCAE_ERR_UNSUPPORTED;
#define SCAE_FIRST_FWD(CI)                                                                   \
  case CI:                                                                                   \
    hipLaunchKernelGGL(conv_first_fwd_kernel<CI>, dim3(B * f.slices), dim3(256), lds,        \
                       (hipStream_t)stream, img, w, bias, out, g);                           \
    break;
  switch (Cin) {
    SCAE_FIRST_FWD(1) SCAE_FIRST_FWD(2) SCAE_FIRST_FWD(3) SCAE_FIRST_FWD(4)
    default: return SCAE_ERR_UNSUPPORTED;
  }
#undef SCAE_FIRST_FWD
  return scae_launch_status();
}

extern "C" int scae_conv3x3_first_wgrad_rows(int B, int Cout) {
  if (B <= 0 || Cout <= 0 || Cout % 64) return 0;
  const FirstSplit f = first_split(B, Cout);
  return B * f.slices * f.parts;
}

extern "C" int scae_conv3x3_first_wgrad_f32(const float *dpre, const float *img, float *partial,
                                            int B, int Cin, int IH, int IW, int Cout, int stride,
                                            void *stream) {
  ConvGeom g{B, IH, IW, (IH - 3) / stride + 1, (IW - 3) / stride + 1, Cin, Cout, stride};
  int rc = check_geom(g, false);
  if (rc) return rc;
  SCAE_REQUIRE(dpre && img && partial);
  if (Cout % 64) return SCAE_ERR_UNSUPPORTED;
  const FirstSplit f = first_split(B, Cout);
  const size_t lds = (size_t)Cin * IH * IW * sizeof(float);
  if (lds > 64 * 1024) return SCAE_ERR_UNSUPPORTED;
#define SCAE_FIRST_WGRAD(CI)                                                                 \
  case CI:                                                                                   \
    hipLaunchKernelGGL(conv_first_wgrad_kernel<CI>, dim3(B * f.slices), dim3(256), lds,      \
                       (hipStream_t)stream, dpre, img, partial, g);                          \
    break;
  switch (Cin) {
    SCAE_FIRST_WGRAD(1) SCAE_FIRST_WGRAD(2) SCAE_FIRST_WGRAD(3) SCAE_FIRST_WGRAD(4)
    default: return SCAE_ERR_UNSUPPORTED;
  }
#undef SCAE_FIRST_WGRAD
  return scae_launch_status();
}

extern "C" int scae_conv3x3_fwd_f32(const float *in, const float *wf, const float *bias,
                                    float *out, int B, int IH, int IW, int Cin, int Cout,
                                    int stride, void *stream) {
  ConvGeom g{B, IH, IW, (IH - 3) / stride + 1, (IW - 3) / stride + 1, Cin, Cout, stride};
  int rc = check_geom(g, true);
  if (rc) return rc;
  SCAE_REQUIRE(in && wf && bias && out);
  const int M = B * g.OH * g.OW;
  if (small_tiles((long)(Cout / 64) * ((M + 63) / 64)))
    hipLaunchKernelGGL(conv_fwd_kernel<true>, dim3(Cout / 32, (M + 31) / 32), dim3(NT), 0,
                       (hipStream_t)stream, in, wf, bias, out, g);
  else
    hipLaunchKernelGGL(conv_fwd_kernel<false>, dim3(Cout / 64, (M + 63) / 64), dim3(NT), 0,
                       (hipStream_t)stream, in, wf, bias, out, g);
  return scae_launch_status();
}

extern "C" int scae_conv3x3_dgrad_f32(const float *dpre, const float *wd, const float *gate,
                                      float *din, int B, int IH, int IW, int Cin, int Cout,
                                      int stride, void *stream) {
  ConvGeom g{B, IH, IW, (IH - 3) / stride + 1, (IW - 3) / stride + 1, Cin, Cout, stride};
  int rc = check_geom(g, true);
  if (rc) return rc;
  SCAE_REQUIRE(dpre && wd && din);
  // class (0,0) is the largest; smaller classes leave their surplus tiles early
  const int M0 = B * ((IH + stride - 1) / stride) * ((IW + stride - 1) / stride);
  const int Z = stride * stride;
  if (small_tiles((long)(Cin / 64) * ((M0 + 63) / 64) * Z))
    hipLaunchKernelGGL(conv_dgrad_kernel<true>, dim3(Cin / 32, (M0 + 31) / 32, Z), dim3(NT), 0,
                       (hipStream_t)stream, dpre, wd, gate, din, g);
  else
    hipLaunchKernelGGL(conv_dgrad_kernel<false>, dim3(Cin / 64, (M0 + 63) / 64, Z), dim3(NT), 0,
                       (hipStream_t)stream, dpre, wd, gate, din, g);
  return scae_launch_status();
}

extern "C" int scae_conv3x3_wgrad_splits(int B, int OH, int OW, int Cin, int Cout) {
  if (B <= 0 || OH <= 0 || OW <= 0 || Cin <= 0 || Cout <= 0) return 0;
  return wgrad_plan(B * OH * OW, Cin, Cout).splits;
}

extern "C" int scae_conv3x3_wgrad_f32(const float *dpre, const float *in, float *partial,
                                      float *dw, float *db, int B, int IH, int IW, int Cin,
                                      int Cout, int stride, void *stream) {
  ConvGeom g{B, IH, IW, (IH - 3) / stride + 1, (IW - 3) / stride + 1, Cin, Cout, stride};
  int rc = check_geom(g, true);
  if (rc) return rc;
  SCAE_REQUIRE(dpre && in && partial && dw);
  const WgradPlan p = wgrad_plan(B * g.OH * g.OW, Cin, Cout);
  if (p.small)
    hipLaunchKernelGGL(conv_wgrad_kernel<true>, dim3(Cin / 32, Cout / 32, 9 * p.splits), dim3(NT),
                       0, (hipStream_t)stream, dpre, in, partial, g, p.splits);
  else
    hipLaunchKernelGGL(conv_wgrad_kernel<false>, dim3(Cin / 64, Cout / 64, 9 * p.splits),
                       dim3(NT), 0, (hipStream_t)stream, dpre, in, partial, g, p.splits);
  const int n = 9 * Cout * Cin + Cout;
  hipLaunchKernelGGL(reduce_wgrad_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, partial, dw, db, Cout, Cin, p.splits);
  return scae_launch_status();
}
