// K9 -- multiple_attention_pooling_2d of the part-capsule encoder head
// (nn_ext.py:76-101, used by part_encoder.py:74): the 1x1 attention conv output
// y (B, HW, A*P) (NHWC, from the K7 GEMM) holds, per part capsule a, P-1
// feature channels and one attention-logit channel (the last).  Per image:
//   mask[a][pix] = softmax_pix(y[pix][a*P + P-1])
//   out[a][p]    = sum_pix y[pix][a*P + p] * mask[a][pix],   p < P-1
// One workgroup per (image, group of capsules): capsules are independent, and an
// image alone would leave half of the 256 CUs idle at B=128; the group's y slab lives in LDS (odd row stride so
// that pixel-per-lane and channel-per-lane accesses are both conflict-free).
// The reference runs this as view / softmax / mul / reshape / sum kernels plus
// their autograd graph; here it is one launch forward and one backward.
#include "geometric_transform.h"
#include "wave_mfma.h"

// (device code of the template colour MLP's backward, for pool_tc_bwd_kernel)
#define SCAE_DEVICE_ONLY
namespace scae_tc {
#include "template_color.hip"
}
#undef SCAE_DEVICE_ONLY

namespace {
constexpr int NT = 512;

struct PoolArgs {
  const float *y, *g;
  float *out, *dy;
  int B, HW, A, P;
  int splits;  // capsule groups per image (grid = B * splits): every capsule is independent
  // fused capsule head (part_encoder.py:75-92), all nullable: the pooled row of
  // capsule a is [pose (6) | presence logit | special features (P-8)]
  const float *noise_u;  // (B,A) U[0,1) draws, logit += (u - .5) * noise_scale
  float noise_scale;
  int similarity;
  float *pose, *presence, *feature;                  // forward outputs
  float *absence;                                    // 1 - presence (nullable)
  const float *pooled, *g_pose, *g_presence, *g_feature, *g_feature2;  // backward inputs
  // fused 1x1 attention conv (part_encoder.py:70-73), forward only: when cx is set the
  // workgroup computes its own y slab = cx (B,HW,C) cw^T (A*P,C) + cb on the matrix
  // cores instead of reading it, and writes it to cy for the backward
  const float *cx, *cw, *cb;
  float *cy;
  int C;
};

__host__ __device__ inline int padded(int AP) { return AP | 1; }
// capsule groups per image: enough workgroups to cover the 256 CUs twice
inline int pool_splits(int B, int A) {
  int s = 1;
  for (int d = 1; d <= A && d <= 8; ++d)
    if (A % d == 0) {
      s = d;
      if ((long)B * d >= 512) break;
    }
  return s;
}
inline size_t lds_floats(int HW, int A, int P, bool bwd) {
  size_t n = (size_t)HW * padded(A * P) + (size_t)A * HW;  // ys, mask
  n += bwd ? (size_t)A * HW + (size_t)A * (P - 1) + A      // t, gs, s
           : (size_t)A * (P - 1);                           // pooled (head mode)
  return n;
}

// stages channels [0, AP) of every pixel row (global row stride ldy) into ys[pix][APp]
__device__ __forceinline__ void stage_y(float *ys, const float *y, int HW, int AP, int APp,
                                        int ldy) {
  if ((AP & 3) == 0 && (ldy & 3) == 0 && ((size_t)y & 15) == 0) {
    const int q = AP / 4;
    for (int e = threadIdx.x; e < HW * q; e += NT) {
      const int pix = e / q, ch = 4 * (e - pix * q);
      const float4 v = *reinterpret_cast<const float4 *>(y + (size_t)pix * ldy + ch);
      float *d = ys + pix * APp + ch;
      d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
    }
  } else {
    for (int e = threadIdx.x; e < HW * AP; e += NT) {
      const int pix = e / AP, ch = e - pix * AP;
      ys[pix * APp + ch] = y[(size_t)pix * ldy + ch];
    }
  }
}

// The workgroup's y slab from the 1x1 conv itself (HW <= 32, C = 64 NB): x of the image
// is staged once into xs[HW][C+4]; a wave owns 16 channels x all pixels, a chain of C/4
// v_mfma_f32_16x16x4_f32 per 16-pixel tile whose weight operand the lane reads straight
// from global as float4s (every weight is used by one workgroup per image only -- nothing
// to share through LDS) -- ALL of them issued before x is staged, so that the one L2
// round trip overlaps the staging.  Result + bias -> ys (for the pooling below) and -> the
// global y (for the backward).
// Measured (B=128, 25 pixels, 128 -> 24 x 23 channels): 17.4 us against 13.4 (K7 GEMM) + 6.7
// (this kernel reading y).  The conv part is bound by the weight reads -- every workgroup
// pulls its 70 KB slab through L2, 72 MB per launch -- not by the MFMAs (replacing them
// with adds changes nothing) nor by latency (prefetching changes nothing); halving C per
// unit to level the waves needs LDS float atomics, which cost 10 us on their own.
template <int NB>
__device__ __forceinline__ void conv_y(float *ys, float *xs, const PoolArgs &k, int b, int a0,
                                       int AP, int APp, int ldy) {
  using namespace scae_wave;
  constexpr int C = 64 * NB, XS = C + 4, q4 = C / 4;
  const int HW = k.HW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const int ncol = (AP + 15) / 16, nrow = (HW + 15) / 16;
  // k of MFMA j in round t = 16 t + 4 q + j (A and B agree): the four k groups of a
  // weight row read 64 contiguous bytes per round
  float4 wf[4 * NB];
  auto load_w = [&](int col) {
    const int ch = col * 16 + r;
    const float *wr = k.cw + (size_t)(a0 * k.P + (ch < AP ? ch : AP - 1)) * C + 4 * q;
#pragma unroll
    for (int t = 0; t < 4 * NB; ++t) wf[t] = ld4(wr + 16 * t);
  };
  if (wave < ncol) load_w(wave);
  const float *x = k.cx + (size_t)b * HW * C;
  for (int e = threadIdx.x; e < HW * q4; e += NT) {
    const int pix = e / q4, c = 4 * (e - pix * q4);
    *reinterpret_cast<float4 *>(xs + pix * XS + c) = ld4(x + (size_t)pix * C + c);
  }
  __syncthreads();
  float *y = k.cy + (size_t)b * HW * ldy + a0 * k.P;
  for (int col = wave; col < ncol; col += NT / 64) {
    if (col != wave) load_w(col);
    const int ch = col * 16 + r;
    const float bias = (k.cb && ch < AP) ? k.cb[a0 * k.P + ch] : 0.f;
    for (int rt = 0; rt < nrow; ++rt) {
      const int pix = rt * 16 + r;
      const float *xr = xs + (pix < HW ? pix : HW - 1) * XS + 4 * q;
      f32x4 acc = splat(0.f);
#pragma unroll
      for (int t = 0; t < 4 * NB; ++t) acc = mma16(acc, ld4(xr + 16 * t), wf[t]);
      if (ch < AP) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int p = rt * 16 + 4 * q + j;
          if (p < HW) {
            const float v = acc[j] + bias;
            ys[p * APp + ch] = v;
            y[(size_t)p * ldy + ch] = v;
          }
        }
      }
    }
  }
}
inline bool conv_y_supported(int HW, int C) { return HW <= 32 && C >= 64 && C % 64 == 0 && C <= 256; }
// xs follows the forward's (ys, mask, pooled), on a 16-byte boundary
__host__ __device__ inline int conv_x_offset(int HW, int A, int P) {
  return (HW * padded(A * P) + A * HW + A * (P - 1) + 3) & ~3;
}

// mask[a][pix] = softmax over pixels of the capsule's logit channel: 32 lanes per
// capsule (a group of capsules is only a handful -- one thread each would leave the
// workgroup waiting on 3 x HW serial LDS round trips)
__device__ __forceinline__ void softmax_masks(float *mask, const float *ys, int HW, int A, int P,
                                              int APp) {
  for (int t = threadIdx.x; t < ((A * 32 + NT - 1) / NT) * NT; t += NT) {
    const int a = t >> 5, l = t & 31;
    const bool ok = a < A;
    const float *col = ys + (ok ? a : 0) * P + P - 1;
    float mx = -INFINITY;
    if (ok)
      for (int pix = l; pix < HW; pix += 32) mx = fmaxf(mx, col[pix * APp]);
    mx = scae::row_max16(mx);   // 32 lanes: two DPP rows + one exchange
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    float s = 0.f;
    if (ok)
      for (int pix = l; pix < HW; pix += 32) {
        const float e = expf(col[pix * APp] - mx);
        mask[a * HW + pix] = e;
        s += e;
      }
    s = scae::row_sum16(s);
    s += __shfl_xor(s, 16, 64);
    const float inv = 1.f / s;
    if (ok)
      for (int pix = l; pix < HW; pix += 32) mask[a * HW + pix] *= inv;  // own elements
  }
}

__device__ __forceinline__ void pool_fwd_body(const PoolArgs &k, float *lds, int blk) {
  // this workgroup: capsules [a0, a0 + A) of image b (A = the group's size from here on)
  const int Af = k.A, A = Af / k.splits, b = blk / k.splits, a0 = (blk % k.splits) * A;
  const int HW = k.HW, P = k.P, AP = A * P, APp = padded(AP), ldy = Af * P;
  const size_t cap0 = (size_t)b * Af + a0;  // global index of the group's first capsule
  float *ys = lds, *mask = ys + HW * APp;
  float *pooled = mask + A * HW;  // [A*(P-1)], head mode only
  if (k.cx) {
    float *xs = lds + conv_x_offset(HW, A, P);
    switch (k.C / 64) {
      case 1: conv_y<1>(ys, xs, k, b, a0, AP, APp, ldy); break;
      case 2: conv_y<2>(ys, xs, k, b, a0, AP, APp, ldy); break;
      case 3: conv_y<3>(ys, xs, k, b, a0, AP, APp, ldy); break;
      default: conv_y<4>(ys, xs, k, b, a0, AP, APp, ldy); break;
    }
  } else
    stage_y(ys, k.y + (size_t)b * HW * ldy + a0 * P, HW, AP, APp, ldy);
  __syncthreads();
  softmax_masks(mask, ys, HW, A, P, APp);
  __syncthreads();
  for (int e = threadIdx.x; e < A * (P - 1); e += NT) {
    const int a = e / (P - 1), p = e - a * (P - 1);
    float s = 0.f;
    for (int pix = 0; pix < HW; ++pix) s = fmaf(ys[pix * APp + a * P + p], mask[a * HW + pix], s);
    k.out[cap0 * (P - 1) + e] = s;
    if (k.pose) pooled[e] = s;
  }
  if (!k.pose) return;
  __syncthreads();
  const int F = P - 8;  // special features per capsule
  for (int a = threadIdx.x; a < A; a += NT) {
    const float *row = pooled + a * (P - 1);
    scae_gt::GtState st;
    scae_gt::gt_eval(row, 1, st);
    float o[6];
    scae_gt::gt_rows(st, k.similarity, o);
#pragma unroll
    for (int j = 0; j < 6; ++j) k.pose[(cap0 + a) * 6 + j] = o[j];
    float logit = row[6];
    if (k.noise_u) logit += (k.noise_u[cap0 + a] - .5f) * k.noise_scale;
    const float pr = scae::sigmoidf_(logit);
    k.presence[cap0 + a] = pr;
    if (k.absence) k.absence[cap0 + a] = 1.f - pr;  // set-transformer input, :113
  }
  if (k.feature)
    for (int e = threadIdx.x; e < A * F; e += NT) {
      const int a = e / F, f = e - a * F;
      k.feature[cap0 * F + e] = pooled[a * (P - 1) + 7 + f];
    }
}
__global__ __launch_bounds__(NT) void pool_fwd_kernel(PoolArgs k) {
  extern __shared__ __align__(16) float lds[];
  pool_fwd_body(k, lds, blockIdx.x);
}

// The head's forward with the template colour MLP's forward (template_color.hip, K10)
// behind it, workgroup by workgroup -- the mirror image of pool_tc_bwd_kernel below: the
// colour MLP of an (image, capsule group) only needs the special features of ITS OWN
// capsules, which this workgroup has just written; tc_fwd (6 us, a dependent launch of its
// own in the step) becomes the tail of the head's workgroups.
__global__ __launch_bounds__(NT) void pool_tc_fwd_kernel(PoolArgs k, scae_tc::TcArgs tk) {
  extern __shared__ __align__(16) float lds[];
  pool_fwd_body(k, lds, blockIdx.x);
  __threadfence_block();
  __syncthreads();   // (the group's feature rows are written; the head's LDS is dead)
  scae_tc::tc_fwd_body<NT>(tk, lds, blockIdx.x);
}

// g (B, A, P-1) -> dy (B, HW, A*P)
__device__ __forceinline__ void pool_bwd_body(const PoolArgs &k, float *lds, int blk) {
  const int Af = k.A, A = Af / k.splits, b = blk / k.splits, a0 = (blk % k.splits) * A;
  const int HW = k.HW, P = k.P, AP = A * P, APp = padded(AP), ldy = Af * P;
  const size_t cap0 = (size_t)b * Af + a0;
  float *ys = lds, *mask = ys + HW * APp, *t = mask + A * HW, *gs = t + A * HW,
        *sa = gs + A * (P - 1);
  stage_y(ys, k.y + (size_t)b * HW * ldy + a0 * P, HW, AP, APp, ldy);
  if (k.pooled) {  // head mode: pull (g_pose, g_presence, g_feature) back to the pooled row
    const int F = P - 8;
    for (int a = threadIdx.x; a < A; a += NT) {
      const float *row = k.pooled + (cap0 + a) * (P - 1);
      float raw[6], go[6], gp[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        raw[j] = row[j];
        go[j] = k.g_pose ? k.g_pose[(cap0 + a) * 6 + j] : 0.f;
      }
      scae_gt::GtState st;
      scae_gt::gt_eval(raw, 1, st);
      scae_gt::gt_backward(st, k.similarity, go, gp);
#pragma unroll
      for (int j = 0; j < 6; ++j) gs[a * (P - 1) + j] = gp[j];
      float logit = row[6];
      if (k.noise_u) logit += (k.noise_u[cap0 + a] - .5f) * k.noise_scale;
      const float sg = scae::sigmoidf_(logit);
      gs[a * (P - 1) + 6] = k.g_presence ? k.g_presence[cap0 + a] * sg * (1.f - sg) : 0.f;
    }
    for (int e = threadIdx.x; e < A * F; e += NT) {
      const int a = e / F, f = e - a * F;
      gs[a * (P - 1) + 7 + f] = (k.g_feature ? k.g_feature[cap0 * F + e] : 0.f) +
                                (k.g_feature2 ? k.g_feature2[cap0 * F + e] : 0.f);
    }
  } else {
    for (int e = threadIdx.x; e < A * (P - 1); e += NT) gs[e] = k.g[cap0 * (P - 1) + e];
  }
  __syncthreads();
  softmax_masks(mask, ys, HW, A, P, APp);
  // t[a][pix] = d out / d mask = sum_p g[a][p] y[pix][a*P + p]   (lanes over pixels)
  for (int e = threadIdx.x; e < A * HW; e += NT) {
    const int a = e / HW, pix = e - a * HW;
    float s = 0.f;
    for (int p = 0; p < P - 1; ++p) s = fmaf(gs[a * (P - 1) + p], ys[pix * APp + a * P + p], s);
    t[e] = s;
  }
  __syncthreads();
  for (int u = threadIdx.x; u < ((A * 32 + NT - 1) / NT) * NT; u += NT) {  // 32 lanes per capsule
    const int a = u >> 5, l = u & 31;
    float s = 0.f;
    if (a < A)
      for (int pix = l; pix < HW; pix += 32) s = fmaf(mask[a * HW + pix], t[a * HW + pix], s);
    s = scae::row_sum16(s);
    s += __shfl_xor(s, 16, 64);
    if (a < A && l == 0) sa[a] = s;
  }
  __syncthreads();
  float *dy = k.dy + (size_t)b * HW * ldy + a0 * P;
  for (int e = threadIdx.x; e < HW * AP; e += NT) {
    const int pix = e / AP, ch = e - pix * AP, a = ch / P, p = ch - a * P;
    const float m = mask[a * HW + pix];
    dy[(size_t)pix * ldy + ch] =
        p < P - 1 ? gs[a * (P - 1) + p] * m : m * (t[a * HW + pix] - sa[a]);
  }
}
__global__ __launch_bounds__(NT) void pool_bwd_kernel(PoolArgs k) {
  extern __shared__ float lds[];
  pool_bwd_body(k, lds, blockIdx.x);
}

// The head's backward with the template colour MLP's backward (template_color.hip, K10) in
// front of it, workgroup by workgroup: both are decomposed into (image, capsule group)
// workgroups over the same groups, and the only thing the head needs from the colour MLP is
// the feature gradient of ITS OWN capsules -- so the colour workgroup of a group runs first
// inside the head's workgroup of that group (512 threads; its sums each have one owner: the
// bits do not depend on the block size), writes g_feature, and the head's part reads it back
// as g_feature2 after a barrier.  The colour kernel's elementwise template-logit blocks are
// the tail of the grid.  11.4 + 12.4 us as two dependent launches.
__global__ __launch_bounds__(NT) void pool_tc_bwd_kernel(PoolArgs k, scae_tc::TcArgs tk, int nA) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x >= nA) {   // workgroup-uniform
    scae_tc::tc_bwdB_body<NT>(tk, (int)blockIdx.x - nA);
    return;
  }
  scae_tc::tc_bwdA_body<NT>(tk, lds, blockIdx.x);
  __syncthreads();   // (its g_feature rows are written; its LDS is dead)
  pool_bwd_body(k, lds, blockIdx.x);
}

int check(const PoolArgs &k) {
  if (k.B <= 0 || k.HW <= 0 || k.A <= 0 || k.P < 2) return SCAE_ERR_BAD_ARG;
  if (!scae_attention_pool_supported(k.HW, k.A, k.P)) return SCAE_ERR_UNSUPPORTED;
  return SCAE_OK;
}
}  // namespace

// scae_capsule_head_bwd_f32 with scae_template_color_bwd_f32 (arguments logits .. color_nonlin;
// its g_feature output is this launch's g_feature2) in front of it
extern "C" int scae_capsule_head_bwd_tc_f32(
    const float *y, const float *pooled, const float *noise_u, float noise_scale, int similarity,
    const float *g_pose, const float *g_presence, const float *g_feature, float *dy, int B,
    int HW, int A, int P, const float *logits, const float *feature, const float *w1,
    const float *b1, const float *w2, const float *b2, const float *color,
    const float *g_templates, const float *g_raw, float *g_logits, float *tc_g_feature,
    float *partial, int C, int hw, int F, int H1, int template_nonlin, int color_nonlin,
    void *stream) {
  PoolArgs k{};
  k.y = y, k.dy = dy, k.B = B, k.HW = HW, k.A = A, k.P = P;
  k.splits = pool_splits(B, A);
  k.noise_u = noise_u, k.noise_scale = noise_scale, k.similarity = similarity;
  k.pooled = pooled, k.g_pose = g_pose, k.g_presence = g_presence, k.g_feature = g_feature;
  k.g_feature2 = tc_g_feature;
  int rc = check(k);
  if (rc) return rc;
  if (P < 8) return SCAE_ERR_UNSUPPORTED;
  SCAE_REQUIRE(y && pooled && dy);
  scae_tc::TcArgs tk;
  rc = scae_tc::bwd_args(tk, logits, feature, w1, b1, w2, b2, color, g_templates, g_raw, g_logits,
                         tc_g_feature, partial, B, A, C, hw, F, H1, template_nonlin,
                         color_nonlin);
  if (rc) return rc;
  // the same groups on both sides, and the colour MLP's features are the head's special ones
  if (tk.splits != k.splits || F != P - 8) return SCAE_ERR_UNSUPPORTED;
  // Large batches (one workgroup per image, >= 512 images) are not latency-bound: there the
  // two kernels are faster apart (B = 1024, 48 capsules: 240 us merged, 140 + 50 apart -- the
  // head's workgroups are LDS-heavy and the colour MLP's serial part sits on top of each)
  if (k.splits == 1) return SCAE_ERR_UNSUPPORTED;
  const size_t lp = lds_floats(HW, A / k.splits, P, true),
               lt = scae_tc::lds_floats(A / k.splits, C, F, H1, true);
  const int nA = B * k.splits;
  scae::launch(pool_tc_bwd_kernel, dim3(nA + scae_tc::bwd_elementwise_blocks(tk)), dim3(NT),
                     (lp > lt ? lp : lt) * sizeof(float), (hipStream_t)stream, k, tk, nA);
  return scae_launch_status();
}

extern "C" int scae_attention_pool_supported(int HW, int A, int P) {
  if (HW <= 0 || A <= 0 || P < 2) return 0;
  return lds_floats(HW, A, P, true) * sizeof(float) <= 128 * 1024;
}

extern "C" int scae_attention_pool_fwd_f32(const float *y, float *out, int B, int HW, int A,
                                           int P, void *stream) {
  PoolArgs k{};
  k.y = y, k.out = out, k.B = B, k.HW = HW, k.A = A, k.P = P;
  k.splits = pool_splits(B, A);
  int rc = check(k);
  if (rc) return rc;
  SCAE_REQUIRE(y && out);
  scae::launch(pool_fwd_kernel, dim3(B * k.splits), dim3(NT),
                     lds_floats(HW, A / k.splits, P, false) * sizeof(float),
                     (hipStream_t)stream, k);
  return scae_launch_status();
}

extern "C" int scae_attention_pool_bwd_f32(const float *y, const float *g, float *dy, int B,
                                           int HW, int A, int P, void *stream) {
  PoolArgs k{};
  k.y = y, k.g = g, k.dy = dy, k.B = B, k.HW = HW, k.A = A, k.P = P;
  k.splits = pool_splits(B, A);
  int rc = check(k);
  if (rc) return rc;
  SCAE_REQUIRE(y && g && dy);
  scae::launch(pool_bwd_kernel, dim3(B * k.splits), dim3(NT),
                     lds_floats(HW, A / k.splits, P, true) * sizeof(float),
                     (hipStream_t)stream, k);
  return scae_launch_status();
}

extern "C" int scae_capsule_head_fwd_f32(const float *y, const float *noise_u, float noise_scale,
                                         int similarity, float *pooled, float *pose,
                                         float *presence, float *feature, float *absence, int B,
                                         int HW, int A, int P, void *stream) {
  PoolArgs k{};
  k.y = y, k.out = pooled, k.B = B, k.HW = HW, k.A = A, k.P = P;
  k.splits = pool_splits(B, A);
  k.noise_u = noise_u, k.noise_scale = noise_scale, k.similarity = similarity;
  k.pose = pose, k.presence = presence, k.feature = feature, k.absence = absence;
  int rc = check(k);
  if (rc) return rc;
  if (P < 8) return SCAE_ERR_UNSUPPORTED;  // 6 pose + presence + attention logit
  SCAE_REQUIRE(y && pooled && pose && presence && (feature || P == 8));
  scae::launch(pool_fwd_kernel, dim3(B * k.splits), dim3(NT),
                     lds_floats(HW, A / k.splits, P, false) * sizeof(float),
                     (hipStream_t)stream, k);
  return scae_launch_status();
}

extern "C" int scae_capsule_head_conv_supported(int HW, int A, int P, int C) {
  return scae_attention_pool_supported(HW, A, P) && P >= 8 && conv_y_supported(HW, C);
}

// Every workgroup reads its capsule group's weight slab once PER IMAGE (through L2): that
// is what bounds the fused form, and it grows with the batch while the K7 GEMM's traffic
// does not.  Preferred while the launch's weight reads stay below 48 MB (36 MB at B = 128,
// 24 x 23 channels, C = 128: 17.4 us against 13.4 + 6.7; at B = 1024, 48 capsules the
// same kernel takes 176 us against 32 + 50) and a wave owns at most two 16-channel tiles.
extern "C" int scae_capsule_head_conv_preferred(int B, int HW, int A, int P, int C) {
  if (B <= 0 || !scae_capsule_head_conv_supported(HW, A, P, C)) return 0;
  const int group = A / pool_splits(B, A);
  return (size_t)B * A * P * C * sizeof(float) <= ((size_t)48 << 20) &&
         (group * P + 15) / 16 <= 2 * (NT / 64);
}

// 1x1 attention conv + capsule head in one launch: y (B,HW,A*P) = x (B,HW,C) w^T + bias is
// produced slab by slab inside the pooling workgroups (and written out for the backward)
extern "C" int scae_capsule_head_conv_fwd_f32(const float *x, const float *w, const float *bias,
                                              int C, float *y, const float *noise_u,
                                              float noise_scale, int similarity, float *pooled,
                                              float *pose, float *presence, float *feature,
                                              float *absence, int B, int HW, int A, int P,
                                              void *stream) {
  PoolArgs k{};
  k.cx = x, k.cw = w, k.cb = bias, k.cy = y, k.C = C;
  k.out = pooled, k.B = B, k.HW = HW, k.A = A, k.P = P;
  k.splits = pool_splits(B, A);
  k.noise_u = noise_u, k.noise_scale = noise_scale, k.similarity = similarity;
  k.pose = pose, k.presence = presence, k.feature = feature, k.absence = absence;
  int rc = check(k);
  if (rc) return rc;
  if (!scae_capsule_head_conv_supported(HW, A, P, C)) return SCAE_ERR_UNSUPPORTED;
  SCAE_REQUIRE(x && w && y && pooled && pose && presence && (feature || P == 8));
  SCAE_REQUIRE((((size_t)x | (size_t)w) & 15) == 0);
  const int Ag = A / k.splits;
  scae::launch(pool_fwd_kernel, dim3(B * k.splits), dim3(NT),
                     (conv_x_offset(HW, Ag, P) + (size_t)HW * (C + 4)) * sizeof(float),
                     (hipStream_t)stream, k);
  return scae_launch_status();
}

// scae_capsule_head_conv_fwd_f32 with scae_template_color_fwd_f32 (arguments logits ..
// color_nonlin; its `feature` input is this launch's `feature` output) behind it
extern "C" int scae_capsule_head_conv_fwd_tc_f32(
    const float *x, const float *w, const float *bias, int C, float *y, const float *noise_u,
    float noise_scale, int similarity, float *pooled, float *pose, float *presence,
    float *feature, float *absence, int B, int HW, int A, int P, const float *logits,
    const float *w1, const float *b1, const float *w2, const float *b2, float *raw,
    float *templates, float *color, int Ct, int hw, int F, int H1, int template_nonlin,
    int color_nonlin, void *stream) {
  PoolArgs k{};
  k.cx = x, k.cw = w, k.cb = bias, k.cy = y, k.C = C;
  k.out = pooled, k.B = B, k.HW = HW, k.A = A, k.P = P;
  k.splits = pool_splits(B, A);
  k.noise_u = noise_u, k.noise_scale = noise_scale, k.similarity = similarity;
  k.pose = pose, k.presence = presence, k.feature = feature, k.absence = absence;
  int rc = check(k);
  if (rc) return rc;
  if (!scae_capsule_head_conv_supported(HW, A, P, C)) return SCAE_ERR_UNSUPPORTED;
  SCAE_REQUIRE(x && w && y && pooled && pose && presence && feature);
  SCAE_REQUIRE((((size_t)x | (size_t)w) & 15) == 0);
  scae_tc::TcArgs tk{};
  tk.logits = logits, tk.feature = feature, tk.w1 = w1, tk.b1 = b1, tk.w2 = w2, tk.b2 = b2;
  tk.raw = raw, tk.templates = templates, tk.color = color;
  tk.B = B, tk.M = A, tk.C = Ct, tk.hw = hw, tk.F = F, tk.H1 = H1;
  tk.tnl = template_nonlin, tk.cnl = color_nonlin;
  tk.splits = scae_tc::tc_splits(B, A);
  rc = scae_tc::check(tk);
  if (rc) return rc;
  SCAE_REQUIRE(logits && w1 && b1 && w2 && b2 && raw && templates && color);
  // the same groups on both sides, the colour MLP's features are the head's special ones;
  // large batches (one workgroup per image) are not latency-bound: the kernels stay apart
  if (tk.splits != k.splits || F != P - 8 || k.splits == 1) return SCAE_ERR_UNSUPPORTED;
  const int Ag = A / k.splits;
  const size_t lp = conv_x_offset(HW, Ag, P) + (size_t)HW * (C + 4),
               lt = scae_tc::lds_floats(Ag, Ct, F, H1, false);
  scae::launch(pool_tc_fwd_kernel, dim3(B * k.splits), dim3(NT),
                     (lp > lt ? lp : lt) * sizeof(float), (hipStream_t)stream, k, tk);
  return scae_launch_status();
}

extern "C" int scae_capsule_head_bwd_f32(const float *y, const float *pooled,
                                         const float *noise_u, float noise_scale, int similarity,
                                         const float *g_pose, const float *g_presence,
                                         const float *g_feature, const float *g_feature2,
                                         float *dy, int B, int HW, int A, int P, void *stream) {
  PoolArgs k{};
  k.y = y, k.dy = dy, k.B = B, k.HW = HW, k.A = A, k.P = P;
  k.splits = pool_splits(B, A);
  k.noise_u = noise_u, k.noise_scale = noise_scale, k.similarity = similarity;
  k.pooled = pooled, k.g_pose = g_pose, k.g_presence = g_presence, k.g_feature = g_feature;
  k.g_feature2 = g_feature2;
  int rc = check(k);
  if (rc) return rc;
  if (P < 8) return SCAE_ERR_UNSUPPORTED;
  SCAE_REQUIRE(y && pooled && dy);
  scae::launch(pool_bwd_kernel, dim3(B * k.splits), dim3(NT),
                     lds_floats(HW, A / k.splits, P, true) * sizeof(float),
                     (hipStream_t)stream, k);
  return scae_launch_status();
}
