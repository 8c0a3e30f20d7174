// K9 -- multiple_attention_pooling_2d of the part-capsule encoder head
// (nn_ext.py:76-101, used by part_encoder.py:74): the 1x1 attention conv output
// y (B, HW, A*P) (NHWC, from the K7 GEMM) holds, per part capsule a, P-1
// feature channels and one attention-logit channel (the last).  Per image:
//   mask[a][pix] = softmax_pix(y[pix][a*P + P-1])
//   out[a][p]    = sum_pix y[pix][a*P + p] * mask[a][pix],   p < P-1
// One workgroup per image; the image's y slab lives in LDS (odd row stride so
// that pixel-per-lane and channel-per-lane accesses are both conflict-free).
// The reference runs this as view / softmax / mul / reshape / sum kernels plus
// their autograd graph; here it is one launch forward and one backward.
#include "geometric_transform.h"

namespace {
constexpr int NT = 256;

struct PoolArgs {
  const float *y, *g;
  float *out, *dy;
  int B, HW, A, P;
  // fused capsule head (part_encoder.py:75-92), all nullable: the pooled row of
  // capsule a is [pose (6) | presence logit | special features (P-8)]
  const float *noise_u;  // (B,A) U[0,1) draws, logit += (u - .5) * noise_scale
  float noise_scale;
  int similarity;
  float *pose, *presence, *feature;                  // forward outputs
  const float *pooled, *g_pose, *g_presence, *g_feature;  // backward inputs
};

__host__ __device__ inline int padded(int AP) { return AP | 1; }
inline size_t lds_floats(int HW, int A, int P, bool bwd) {
  size_t n = (size_t)HW * padded(A * P) + (size_t)A * HW;  // ys, mask
  n += bwd ? (size_t)A * HW + (size_t)A * (P - 1) + A      // t, gs, s
           : (size_t)A * (P - 1);                           // pooled (head mode)
  return n;
}

__device__ __forceinline__ void stage_y(float *ys, const float *y, int HW, int AP, int APp) {
  if ((AP & 3) == 0) {
    for (int e = threadIdx.x; e < HW * AP / 4; e += NT) {
      const float4 v = reinterpret_cast<const float4 *>(y)[e];
      const int pix = (4 * e) / AP, ch = 4 * e - pix * AP;
      float *d = ys + pix * APp + ch;
      d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
    }
  } else {
    for (int e = threadIdx.x; e < HW * AP; e += NT) {
      const int pix = e / AP, ch = e - pix * AP;
      ys[pix * APp + ch] = y[e];
    }
  }
}

// mask[a][pix] = softmax over pixels of the capsule's logit channel
__device__ __forceinline__ void softmax_masks(float *mask, const float *ys, int HW, int A, int P,
                                              int APp) {
  for (int a = threadIdx.x; a < A; a += NT) {
    const float *col = ys + a * P + P - 1;
    float mx = -INFINITY;
    for (int pix = 0; pix < HW; ++pix) mx = fmaxf(mx, col[pix * APp]);
    float s = 0.f;
    for (int pix = 0; pix < HW; ++pix) {
      const float e = expf(col[pix * APp] - mx);
      mask[a * HW + pix] = e;
      s += e;
    }
    const float inv = 1.f / s;
    for (int pix = 0; pix < HW; ++pix) mask[a * HW + pix] *= inv;
  }
}

__global__ __launch_bounds__(NT) void pool_fwd_kernel(PoolArgs k) {
  extern __shared__ float lds[];
  const int HW = k.HW, A = k.A, P = k.P, AP = A * P, APp = padded(AP), b = blockIdx.x;
  float *ys = lds, *mask = ys + HW * APp;
  stage_y(ys, k.y + (size_t)b * HW * AP, HW, AP, APp);
  __syncthreads();
  softmax_masks(mask, ys, HW, A, P, APp);
  __syncthreads();
  float *pooled = mask + A * HW;  // [A*(P-1)], head mode only
  for (int e = threadIdx.x; e < A * (P - 1); e += NT) {
    const int a = e / (P - 1), p = e - a * (P - 1);
    float s = 0.f;
    for (int pix = 0; pix < HW; ++pix) s = fmaf(ys[pix * APp + a * P + p], mask[a * HW + pix], s);
    k.out[(size_t)b * A * (P - 1) + e] = s;
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
    for (int j = 0; j < 6; ++j) k.pose[((size_t)b * A + a) * 6 + j] = o[j];
    float logit = row[6];
    if (k.noise_u) logit += (k.noise_u[(size_t)b * A + a] - .5f) * k.noise_scale;
    k.presence[(size_t)b * A + a] = scae::sigmoidf_(logit);
  }
  if (k.feature)
    for (int e = threadIdx.x; e < A * F; e += NT) {
      const int a = e / F, f = e - a * F;
      k.feature[(size_t)b * A * F + e] = pooled[a * (P - 1) + 7 + f];
    }
}

// g (B, A, P-1) -> dy (B, HW, A*P)
__global__ __launch_bounds__(NT) void pool_bwd_kernel(PoolArgs k) {
  extern __shared__ float lds[];
  const int HW = k.HW, A = k.A, P = k.P, AP = A * P, APp = padded(AP), b = blockIdx.x;
  float *ys = lds, *mask = ys + HW * APp, *t = mask + A * HW, *gs = t + A * HW,
        *sa = gs + A * (P - 1);
  stage_y(ys, k.y + (size_t)b * HW * AP, HW, AP, APp);
  if (k.pooled) {  // head mode: pull (g_pose, g_presence, g_feature) back to the pooled row
    const int F = P - 8;
    for (int a = threadIdx.x; a < A; a += NT) {
      const float *row = k.pooled + ((size_t)b * A + a) * (P - 1);
      float raw[6], go[6], gp[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        raw[j] = row[j];
        go[j] = k.g_pose ? k.g_pose[((size_t)b * A + a) * 6 + j] : 0.f;
      }
      scae_gt::GtState st;
      scae_gt::gt_eval(raw, 1, st);
      scae_gt::gt_backward(st, k.similarity, go, gp);
#pragma unroll
      for (int j = 0; j < 6; ++j) gs[a * (P - 1) + j] = gp[j];
      float logit = row[6];
      if (k.noise_u) logit += (k.noise_u[(size_t)b * A + a] - .5f) * k.noise_scale;
      const float sg = scae::sigmoidf_(logit);
      gs[a * (P - 1) + 6] = k.g_presence ? k.g_presence[(size_t)b * A + a] * sg * (1.f - sg) : 0.f;
    }
    for (int e = threadIdx.x; e < A * F; e += NT) {
      const int a = e / F, f = e - a * F;
      gs[a * (P - 1) + 7 + f] = k.g_feature ? k.g_feature[(size_t)b * A * F + e] : 0.f;
    }
  } else {
    for (int e = threadIdx.x; e < A * (P - 1); e += NT) gs[e] = k.g[(size_t)b * A * (P - 1) + e];
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
  for (int a = threadIdx.x; a < A; a += NT) {
    float s = 0.f;
    for (int pix = 0; pix < HW; ++pix) s = fmaf(mask[a * HW + pix], t[a * HW + pix], s);
    sa[a] = s;
  }
  __syncthreads();
  float *dy = k.dy + (size_t)b * HW * AP;
  for (int e = threadIdx.x; e < HW * AP; e += NT) {
    const int pix = e / AP, ch = e - pix * AP, a = ch / P, p = ch - a * P;
    const float m = mask[a * HW + pix];
    dy[e] = p < P - 1 ? gs[a * (P - 1) + p] * m : m * (t[a * HW + pix] - sa[a]);
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
  int rc = check(k);
  if (rc) return rc;
  SCAE_REQUIRE(y && out);
  hipLaunchKernelGGL(pool_fwd_kernel, dim3(B), dim3(NT), lds_floats(HW, A, P, false) * sizeof(float),
                     (hipStream_t)stream, k);
  return scae_launch_status();
}

extern "C" int scae_attention_pool_bwd_f32(const float *y, const float *g, float *dy, int B,
                                           int HW, int A, int P, void *stream) {
  PoolArgs k{};
  k.y = y, k.g = g, k.dy = dy, k.B = B, k.HW = HW, k.A = A, k.P = P;
  int rc = check(k);
  if (rc) return rc;
  SCAE_REQUIRE(y && g && dy);
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(B), dim3(NT), lds_floats(HW, A, P, true) * sizeof(float),
                     (hipStream_t)stream, k);
  return scae_launch_status();
}

extern "C" int scae_capsule_head_fwd_f32(const float *y, const float *noise_u, float noise_scale,
                                         int similarity, float *pooled, float *pose,
                                         float *presence, float *feature, int B, int HW, int A,
                                         int P, void *stream) {
  PoolArgs k{};
  k.y = y, k.out = pooled, k.B = B, k.HW = HW, k.A = A, k.P = P;
  k.noise_u = noise_u, k.noise_scale = noise_scale, k.similarity = similarity;
  k.pose = pose, k.presence = presence, k.feature = feature;
  int rc = check(k);
  if (rc) return rc;
  if (P < 8) return SCAE_ERR_UNSUPPORTED;  // 6 pose + presence + attention logit
  SCAE_REQUIRE(y && pooled && pose && presence && (feature || P == 8));
  hipLaunchKernelGGL(pool_fwd_kernel, dim3(B), dim3(NT), lds_floats(HW, A, P, false) * sizeof(float),
                     (hipStream_t)stream, k);
  return scae_launch_status();
}

extern "C" int scae_capsule_head_bwd_f32(const float *y, const float *pooled,
                                         const float *noise_u, float noise_scale, int similarity,
                                         const float *g_pose, const float *g_presence,
                                         const float *g_feature, float *dy, int B, int HW, int A,
                                         int P, void *stream) {
  PoolArgs k{};
  k.y = y, k.dy = dy, k.B = B, k.HW = HW, k.A = A, k.P = P;
  k.noise_u = noise_u, k.noise_scale = noise_scale, k.similarity = similarity;
  k.pooled = pooled, k.g_pose = g_pose, k.g_presence = g_presence, k.g_feature = g_feature;
  int rc = check(k);
  if (rc) return rc;
  if (P < 8) return SCAE_ERR_UNSUPPORTED;
  SCAE_REQUIRE(y && pooled && dy);
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(B), dim3(NT), lds_floats(HW, A, P, true) * sizeof(float),
                     (hipStream_t)stream, k);
  return scae_launch_status();
}
