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

__global__ __launch_bounds__(NT) void pool_fwd_kernel(PoolArgs k) {
  extern __shared__ float lds[];
  // this workgroup: capsules [a0, a0 + A) of image b (A = the group's size from here on)
  const int Af = k.A, A = Af / k.splits, b = blockIdx.x / k.splits,
            a0 = (blockIdx.x % k.splits) * A;
  const int HW = k.HW, P = k.P, AP = A * P, APp = padded(AP), ldy = Af * P;
  const size_t cap0 = (size_t)b * Af + a0;  // global index of the group's first capsule
  float *ys = lds, *mask = ys + HW * APp;
  stage_y(ys, k.y + (size_t)b * HW * ldy + a0 * P, HW, AP, APp, ldy);
  __syncthreads();
  softmax_masks(mask, ys, HW, A, P, APp);
  __syncthreads();
  float *pooled = mask + A * HW;  // [A*(P-1)], head mode only
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

// g (B, A, P-1) -> dy (B, HW, A*P)
__global__ __launch_bounds__(NT) void pool_bwd_kernel(PoolArgs k) {
  extern __shared__ float lds[];
  const int Af = k.A, A = Af / k.splits, b = blockIdx.x / k.splits,
            a0 = (blockIdx.x % k.splits) * A;
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

int check(const PoolArgs &k) {
  if (k.B <= 0 || k.HW <= 0 || k.A <= 0 || k.P < 2) return SCAE_ERR_BAD_ARG;
  if (!scae_attention_pool_supported(k.HW, k.A, k.P)) return SCAE_ERR_UNSUPPORTED;
  return SCAE_OK;
}
}  // namespace

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
  hipLaunchKernelGGL(pool_fwd_kernel, dim3(B * k.splits), dim3(NT),
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
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(B * k.splits), dim3(NT),
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
  hipLaunchKernelGGL(pool_fwd_kernel, dim3(B * k.splits), dim3(NT),
                     lds_floats(HW, A / k.splits, P, false) * sizeof(float),
                     (hipStream_t)stream, k);
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
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(B * k.splits), dim3(NT),
                     lds_floats(HW, A / k.splits, P, true) * sizeof(float),
                     (hipStream_t)stream, k);
  return scae_launch_status();
}
