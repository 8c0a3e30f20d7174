// Class probabilities of SCAE.forward (stacked_capsule_auto_encoder.py:205-212):
//   prior_cls_prob     = softmax(W caps_presence + b)
//   posterior_cls_prob = softmax(W sum_m posterior[:, :O, m] + b)
// (both through prior_classifier, as the reference does).  One workgroup (one
// wave) per image, lanes over object capsules; replaces two Linear + Softmax
// pairs and a reduction (5 launches) with one.
#include "common.h"

namespace {
constexpr int MAXCLS = 32;

__global__ __launch_bounds__(64) void class_probs_kernel(
    const float *__restrict__ cp, const float *__restrict__ posterior,
    const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ prior_prob,
    float *__restrict__ post_prob, int O, int M, int ncls) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float x0 = 0.f, x1 = 0.f;
  if (lane < O) {
    x0 = cp[(size_t)b * O + lane];
    const float *p = posterior + ((size_t)b * (O + 1) + lane) * M;
    for (int m = 0; m < M; ++m) x1 += p[m];
  }
  float l0[MAXCLS], l1[MAXCLS];
  float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
  for (int c = 0; c < MAXCLS; ++c) {
    l0[c] = l1[c] = -INFINITY;
    if (c < ncls) {
      const float wv = lane < O ? w[c * O + lane] : 0.f;
      l0[c] = scae::wave_sum(x0 * wv) + bias[c];
      l1[c] = scae::wave_sum(x1 * wv) + bias[c];
      m0 = fmaxf(m0, l0[c]);
      m1 = fmaxf(m1, l1[c]);
    }
  }
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int c = 0; c < MAXCLS; ++c)
    if (c < ncls) {
      l0[c] = expf(l0[c] - m0);
      l1[c] = expf(l1[c] - m1);
      s0 += l0[c];
      s1 += l1[c];
    }
#pragma unroll
  for (int c = 0; c < MAXCLS; ++c)
    if (c < ncls && lane == c) {
      prior_prob[(size_t)b * ncls + c] = l0[c] / s0;
      post_prob[(size_t)b * ncls + c] = l1[c] / s1;
    }
}
}  // namespace

extern "C" int scae_class_probs_supported(int O, int ncls) {
  return O > 0 && O <= 64 && ncls > 0 && ncls <= MAXCLS;
}

extern "C" int scae_class_probs_f32(const float *caps_presence, const float *posterior,
                                    const float *w, const float *bias, float *prior_prob,
                                    float *post_prob, int B, int O, int M, int ncls,
                                    void *stream) {
  SCAE_REQUIRE(caps_presence && posterior && w && bias && prior_prob && post_prob && B > 0 &&
               M > 0);
  if (!scae_class_probs_supported(O, ncls)) return SCAE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(class_probs_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream,
                     caps_presence, posterior, w, bias, prior_prob, post_prob, O, M, ncls);
  return scae_launch_status();
}
