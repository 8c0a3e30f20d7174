// K10 -- TemplateGenerator.forward (part_decoder.py:78-110) for coloured
// templates: raw = nonlin(template_logits) (1,M,C,h,w); per part capsule a
// colour from its special features through MLP([F, H1, C]) (ReLU after both
// layers, nn_ext.py:19-31), colour non-linearity, templates = raw * colour
// (B,M,C,h,w).  The reference runs ~10 small ATen launches forward and ~15 in
// the autograd backward; here: one launch forward, one backward (two kinds of
// workgroups).
//   fwd : one workgroup per image (MLP weights in LDS)
//   bwdA: one workgroup per image: g_colour = <g_templates, raw>, MLP backward
//         -> g_feature and per-image partial weight gradients
//   bwdB: one thread per template texel: g_logits = nonlin'(.) sum_b g_templates * colour
#include "common.h"

namespace {
constexpr int NT = 1024;

struct TcArgs {
  const float *logits;   // (M,C,hw)
  const float *feature;  // (B,M,F)
  const float *w1, *b1, *w2, *b2;  // (H1,F) (H1) (C,H1) (C)
  float *raw;            // (M,C,hw)
  float *templates;      // (B,M,C,hw)
  float *color;          // (B,M,C)
  const float *g_templates, *g_raw;  // (B,M,C,hw); (M,C,hw) nullable
  float *g_feature;      // (B,M,F)
  float *partial;        // (B, H1*F + H1 + C*H1 + C)
  float *g_logits;       // (M,C,hw)
  int B, M, C, hw, F, H1;
  int tnl, cnl;          // 0 sigmoid, 1 relu1
  int splits;            // capsule groups per image (grid = B * splits)
};

// nn_ext.py:139-140: relu6(6x)/6, evaluated as written
__device__ __forceinline__ float relu1(float x) { return fminf(fmaxf(x * 6.f, 0.f), 6.f) / 6.f; }
__device__ __forceinline__ float relu1_grad(float x) {
  const float z = x * 6.f;
  return (z > 0.f && z < 6.f) ? 1.f : 0.f;
}
__device__ __forceinline__ float nonlin(float x, int kind) {
  return kind == 0 ? scae::sigmoidf_(x) : relu1(x);
}
__device__ __forceinline__ float nonlin_grad(float x, int kind) {
  if (kind == 0) {
    const float s = scae::sigmoidf_(x);
    return s * (1.f - s);
  }
  return relu1_grad(x);
}

// w1 rows and h1 rows carry one float of padding: the MLP loops read w1[j][f] with
// lanes over j and h1[m][j] with lanes over m -- unpadded strides of 16 / 32 floats
// put all lanes of a wave on one or two LDS banks
struct Lds {
  float *w1, *b1, *w2, *b2, *feat, *h1, *pre2;
  int ldw, ldh;  // row strides of w1 (F + 1) and h1 (H1 + 1)
};
__device__ __forceinline__ Lds carve(float *base, const TcArgs &k, int M) {
  Lds l;
  l.ldw = k.F + 1, l.ldh = k.H1 + 1;
  l.w1 = base;
  l.b1 = l.w1 + k.H1 * l.ldw;
  l.w2 = l.b1 + k.H1;
  l.b2 = l.w2 + k.C * k.H1;
  l.feat = l.b2 + k.C;
  l.h1 = l.feat + M * k.F;
  l.pre2 = l.h1 + M * l.ldh;
  return l;
}
inline size_t lds_floats(int M, int C, int F, int H1, bool bwd) {
  size_t n = (size_t)H1 * (F + 1) + H1 + (size_t)C * H1 + C + (size_t)M * F +
             (size_t)M * (H1 + 1) + (size_t)M * C;
  if (bwd) n += (size_t)M * C + (size_t)M * H1;  // g_pre2, g_h1
  return n;
}

// stages the MLP and evaluates it for M capsules whose features start at `feature`:
// h1 (post-ReLU), pre2 (second layer pre-activation)
template <int NT>   // block size (any: every sum has one owner)
__device__ __forceinline__ void mlp_forward(const Lds &l, const TcArgs &k, const float *feature,
                                            int M) {
  const int t = threadIdx.x;
  for (int e = t; e < k.H1 * k.F; e += NT) l.w1[(e / k.F) * l.ldw + e % k.F] = k.w1[e];
  for (int e = t; e < k.H1; e += NT) l.b1[e] = k.b1[e];
  for (int e = t; e < k.C * k.H1; e += NT) l.w2[e] = k.w2[e];
  for (int e = t; e < k.C; e += NT) l.b2[e] = k.b2[e];
  for (int e = t; e < M * k.F; e += NT) l.feat[e] = feature[e];
  __syncthreads();
  for (int e = t; e < M * k.H1; e += NT) {
    const int m = e / k.H1, j = e - m * k.H1;
    float s = l.b1[j];
    for (int f = 0; f < k.F; ++f) s = fmaf(l.feat[m * k.F + f], l.w1[j * l.ldw + f], s);
    l.h1[m * l.ldh + j] = fmaxf(s, 0.f);
  }
  __syncthreads();
  for (int e = t; e < M * k.C; e += NT) {
    const int m = e / k.C, c = e - m * k.C;
    float s = l.b2[c];
    for (int j = 0; j < k.H1; ++j) s = fmaf(l.h1[m * l.ldh + j], l.w2[c * k.H1 + j], s);
    l.pre2[e] = s;
  }
  __syncthreads();
}

// colour from the second-layer pre-activation: ReLU, (+.99 for relu1,
// part_decoder.py:97-98), colour non-linearity
__device__ __forceinline__ float color_of(float pre2, int cnl) {
  const float r = fmaxf(pre2, 0.f);
  return cnl == 0 ? scae::sigmoidf_(r) : relu1(r + .99f);
}
__device__ __forceinline__ float color_grad(float pre2, int cnl) {
  if (!(pre2 > 0.f)) return 0.f;
  return cnl == 0 ? nonlin_grad(pre2, 0) : relu1_grad(pre2 + .99f);
}

// One workgroup per (image b, group of M capsules starting at m0): capsules are
// independent, and an image alone would leave half of the CUs idle at B=128.  As a device
// function (block `blk` of B * splits, NT threads -- any: every sum has one owner), so that
// the part-capsule head's forward workgroup of the same group can run it behind itself
// (attention_pool.hip, pool_tc_fwd_kernel).
template <int NT>
__device__ __forceinline__ void tc_fwd_body(const TcArgs &k, float *lds, int blk) {
  const int M = k.M / k.splits, b = blk / k.splits, m0 = (blk % k.splits) * M;
  const Lds l = carve(lds, k, M);
  const int t = threadIdx.x, MC = M * k.C;
  const size_t cap0 = (size_t)b * k.M + m0;  // global index of the group's first capsule
  mlp_forward<NT>(l, k, k.feature + cap0 * k.F, M);
  for (int e = t; e < MC; e += NT) {
    const float col = color_of(l.pre2[e], k.cnl);
    l.pre2[e] = col;
    k.color[cap0 * k.C + e] = col;
  }
  __syncthreads();
  float *dst = k.templates + cap0 * k.C * k.hw;
  const float *logits = k.logits + (size_t)m0 * k.C * k.hw;
  for (int e = t; e < MC * k.hw; e += NT) {
    const float r = nonlin(logits[e], k.tnl);
    dst[e] = r * l.pre2[e / k.hw];
    if (b == 0) k.raw[(size_t)m0 * k.C * k.hw + e] = r;
  }
}
#ifndef SCAE_DEVICE_ONLY   // (attention_pool.hip includes this file for its device code)
__global__ __launch_bounds__(NT) void tc_fwd_kernel(TcArgs k) {
  extern __shared__ float lds[];
  tc_fwd_body<NT>(k, lds, blockIdx.x);
}
#endif

template <int NT>   // block size (any: every sum below has one owner)
__device__ __forceinline__ void tc_bwdA_body(const TcArgs &k, float *lds, int block) {
  const int M = k.M / k.splits, b = block / k.splits, m0 = (block % k.splits) * M;
  const Lds l = carve(lds, k, M);
  float *g2 = l.pre2 + M * k.C, *g1 = g2 + M * k.C;
  const int t = threadIdx.x, MC = M * k.C, wave = t >> 6, lane = t & 63;
  const size_t cap0 = (size_t)b * k.M + m0;
  mlp_forward<NT>(l, k, k.feature + cap0 * k.F, M);
  // g_colour[m,c] = sum_t g_templates[b,m,c,t] * raw[m,c,t]; then through the colour
  // non-linearity and the second ReLU -> g2 (gradient w.r.t. pre2)
  const float *logits = k.logits + (size_t)m0 * k.C * k.hw;
  for (int pair = wave; pair < MC; pair += NT / 64) {
    const float *g = k.g_templates + (cap0 * k.C + pair) * k.hw;
    float s = 0.f;
    for (int i = lane; i < k.hw; i += 64) s = fmaf(g[i], nonlin(logits[pair * k.hw + i], k.tnl), s);
    s = scae::wave_sum(s);
    if (lane == 0) g2[pair] = s * color_grad(l.pre2[pair], k.cnl);
  }
  __syncthreads();
  for (int e = t; e < M * k.H1; e += NT) {  // g1: gradient w.r.t. the first pre-activation
    const int m = e / k.H1, j = e - m * k.H1;
    float s = 0.f;
    for (int c = 0; c < k.C; ++c) s = fmaf(g2[m * k.C + c], l.w2[c * k.H1 + j], s);
    g1[e] = l.h1[m * l.ldh + j] > 0.f ? s : 0.f;
  }
  __syncthreads();
  for (int e = t; e < M * k.F; e += NT) {
    const int m = e / k.F, f = e - m * k.F;
    float s = 0.f;
    for (int j = 0; j < k.H1; ++j) s = fmaf(g1[m * k.H1 + j], l.w1[j * l.ldw + f], s);
    k.g_feature[cap0 * k.F + e] = s;
  }
  // per-workgroup weight-gradient partials: [dW1 | db1 | dW2 | db2]
  const int n1 = k.H1 * k.F, n2 = n1 + k.H1, n3 = n2 + k.C * k.H1, n4 = n3 + k.C;
  float *part = k.partial + (size_t)block * n4;
  for (int e = t; e < n4; e += NT) {
    float s = 0.f;
    if (e < n1) {
      const int j = e / k.F, f = e - j * k.F;
      for (int m = 0; m < M; ++m) s = fmaf(g1[m * k.H1 + j], l.feat[m * k.F + f], s);
    } else if (e < n2) {
      for (int m = 0; m < M; ++m) s += g1[m * k.H1 + e - n1];
    } else if (e < n3) {
      const int c = (e - n2) / k.H1, j = (e - n2) - c * k.H1;
      for (int m = 0; m < M; ++m) s = fmaf(g2[m * k.C + c], l.h1[m * l.ldh + j], s);
    } else {
      for (int m = 0; m < M; ++m) s += g2[m * k.C + e - n3];
    }
    part[e] = s;
  }
}

// workgroup (256 texels, 4 batch parts)
// 256 texels x 4 batch parts per workgroup; a block of fewer than 1024 threads walks
// several parts per thread (same partial sums, same order)
template <int NT>   // block size: 256, 512 or 1024
__device__ __forceinline__ void tc_bwdB_body(const TcArgs &k, int block) {
  __shared__ float red[4][256];
  constexpr int VP = 4 / (NT / 256);
  const int tx = threadIdx.x & 255;
  const int e = block * 256 + tx, MC = k.M * k.C;
  const int n = MC * k.hw;
#pragma unroll
  for (int v = 0; v < VP; ++v) {
    const int part = (threadIdx.x >> 8) * VP + v;
    float s = 0.f;
    if (e < n) {
      const int mc = e / k.hw, per = (k.B + 3) / 4, b0 = part * per, b1 = min(k.B, b0 + per);
#pragma unroll 8
      for (int b = b0; b < b1; ++b)
        s = fmaf(k.g_templates[(size_t)b * n + e], k.color[(size_t)b * MC + mc], s);
    }
    red[part][tx] = s;
  }
  __syncthreads();
  if (threadIdx.x < 256 && e < n) {
    float tot = (red[0][tx] + red[1][tx]) +
                (red[2][tx] + red[3][tx]);
    if (k.g_raw) tot += k.g_raw[e];
    k.g_logits[e] = tot * nonlin_grad(k.logits[e], k.tnl);
  }
}

#ifndef SCAE_DEVICE_ONLY
__global__ __launch_bounds__(NT) void tc_bwd_kernel(TcArgs k, int nA) {
  extern __shared__ float lds[];
  if ((int)blockIdx.x < nA)  // workgroup-uniform
    tc_bwdA_body<NT>(k, lds, blockIdx.x);
  else
    tc_bwdB_body<NT>(k, (int)blockIdx.x - nA);
}
#endif

// capsule groups per image: enough workgroups to cover the 256 CUs twice
inline int tc_splits(int B, int M) {
  int s = 1;
  for (int d = 1; d <= M && d <= 8; ++d)
    if (M % d == 0) {
      s = d;
      if ((long)B * d >= 512) break;
    }
  return s;
}

int check(const TcArgs &k) {
  if (k.B <= 0 || k.M <= 0 || k.C <= 0 || k.hw <= 0 || k.F <= 0 || k.H1 <= 0)
    return SCAE_ERR_BAD_ARG;
  if (k.tnl < 0 || k.tnl > 1 || k.cnl < 0 || k.cnl > 1) return SCAE_ERR_BAD_ARG;
  if (!scae_template_color_supported(k.M, k.C, k.F, k.H1)) return SCAE_ERR_UNSUPPORTED;
  return SCAE_OK;
}
// the arguments of a backward launch, checked
int bwd_args(TcArgs &k, const float *logits, const float *feature, const float *w1,
             const float *b1, const float *w2, const float *b2, const float *color,
             const float *g_templates, const float *g_raw, float *g_logits, float *g_feature,
             float *partial, int B, int M, int C, int hw, int F, int H1, int template_nonlin,
             int color_nonlin) {
  k = TcArgs{};
  k.logits = logits, k.feature = feature, k.w1 = w1, k.b1 = b1, k.w2 = w2, k.b2 = b2;
  k.color = const_cast<float *>(color);
  k.g_templates = g_templates, k.g_raw = g_raw, k.g_logits = g_logits;
  k.g_feature = g_feature, k.partial = partial;
  k.B = B, k.M = M, k.C = C, k.hw = hw, k.F = F, k.H1 = H1;
  k.tnl = template_nonlin, k.cnl = color_nonlin;
  k.splits = tc_splits(B, M);
  int rc = check(k);
  if (rc) return rc;
  SCAE_REQUIRE(logits && feature && w1 && b1 && w2 && b2 && color && g_templates && g_logits &&
               g_feature && partial);
  return SCAE_OK;
}
inline int bwd_elementwise_blocks(const TcArgs &k) { return (k.M * k.C * k.hw + 255) / 256; }
}  // namespace

#ifndef SCAE_DEVICE_ONLY
extern "C" int scae_template_color_supported(int M, int C, int F, int H1) {
  if (M <= 0 || C <= 0 || F <= 0 || H1 <= 0) return 0;
  return lds_floats(M, C, F, H1, true) * sizeof(float) <= 64 * 1024;
}

extern "C" int scae_template_color_fwd_f32(const float *logits, const float *feature,
                                           const float *w1, const float *b1, const float *w2,
                                           const float *b2, float *raw, float *templates,
                                           float *color, int B, int M, int C, int hw, int F,
                                           int H1, int template_nonlin, int color_nonlin,
                                           void *stream) {
  TcArgs k{};
  k.logits = logits, k.feature = feature, k.w1 = w1, k.b1 = b1, k.w2 = w2, k.b2 = b2;
  k.raw = raw, k.templates = templates, k.color = color;
  k.B = B, k.M = M, k.C = C, k.hw = hw, k.F = F, k.H1 = H1;
  k.tnl = template_nonlin, k.cnl = color_nonlin;
  k.splits = tc_splits(B, M);
  int rc = check(k);
  if (rc) return rc;
  SCAE_REQUIRE(logits && feature && w1 && b1 && w2 && b2 && raw && templates && color);
  scae::launch(tc_fwd_kernel, dim3(B * k.splits), dim3(NT),
                     lds_floats(M / k.splits, C, F, H1, false) * sizeof(float),
                     (hipStream_t)stream, k);
  return scae_launch_status();
}

extern "C" int scae_template_color_bwd_f32(const float *logits, const float *feature,
                                           const float *w1, const float *b1, const float *w2,
                                           const float *b2, const float *color,
                                           const float *g_templates, const float *g_raw,
                                           float *g_logits, float *g_feature, float *partial,
                                           int B, int M, int C, int hw, int F, int H1,
                                           int template_nonlin, int color_nonlin, void *stream) {
  TcArgs k;
  int rc = bwd_args(k, logits, feature, w1, b1, w2, b2, color, g_templates, g_raw, g_logits,
                    g_feature, partial, B, M, C, hw, F, H1, template_nonlin, color_nonlin);
  if (rc) return rc;
  const int nA = B * k.splits;
  scae::launch(tc_bwd_kernel, dim3(nA + bwd_elementwise_blocks(k)), dim3(NT),
                     lds_floats(M / k.splits, C, F, H1, true) * sizeof(float),
                     (hipStream_t)stream, k, nA);
  return scae_launch_status();
}

extern "C" int scae_template_color_partial_rows(int B, int M) {
  return B > 0 && M > 0 ? B * tc_splits(B, M) : 0;
}
#endif  // SCAE_DEVICE_ONLY
