// Device body and argument block of the class-probability kernel (class_probs.hip), shared
// with the loss tail (loss_tail.hip), whose per-image launch can carry these workgroups.
#pragma once
#include "common.h"

namespace scae_cp {
constexpr int MAXCLS = 32;
struct ExtraSums {
  scae_scaled_sum j[8];
  int n;
};
struct Args {
  const float *cp, *posterior, *w, *bias;
  float *prior_prob, *post_prob;
  int B, O, M, ncls;
  ExtraSums extra;
};
struct Lds {
  float x[2][64], l[2][MAXCLS];
};

// workgroup b (one wave): b < B an image, else rider b - B
__device__ __forceinline__ void body(const Args &a, Lds &s, int b, int lane) {
  const int B = a.B, O = a.O, M = a.M, ncls = a.ncls;
  if (b >= B) {  // riders: scaled full sums (the scalar outputs of the forward pass)
    const scae_scaled_sum &job = a.extra.j[b - B];
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f, t4 = 0.f, t5 = 0.f, t6 = 0.f, t7 = 0.f;
    int64_t i = lane;  // eight loads in flight per lane: the sum is L2-latency bound
    for (; i + 448 < job.n; i += 512) {
      const float *p = job.src + i;
      t0 += p[0], t1 += p[64], t2 += p[128], t3 += p[192];
      t4 += p[256], t5 += p[320], t6 += p[384], t7 += p[448];
    }
    for (; i < job.n; i += 64) t0 += job.src[i];
    float t = ((t0 + t1) + (t2 + t3)) + ((t4 + t5) + (t6 + t7));
    t = scae::wave_sum(t);
    if (lane == 0) job.dst[0] = t * job.scale;
    return;
  }
  if (lane < O) {  // lane = capsule: the two classifier inputs
    s.x[0][lane] = a.cp[(size_t)b * O + lane];
    const float *p = a.posterior + ((size_t)b * (O + 1) + lane) * M;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;  // four loads in flight
    int m = 0;
    for (; m + 4 <= M; m += 4) m0 += p[m], m1 += p[m + 1], m2 += p[m + 2], m3 += p[m + 3];
    for (; m < M; ++m) m0 += p[m];
    s.x[1][lane] = (m0 + m1) + (m2 + m3);
  }
  __syncthreads();
  if (lane < 2 * ncls) {  // lane = (input, class): one logit each
    const int which = lane / ncls, c = lane - which * ncls;
    const float *wr = a.w + (size_t)c * O;  // four weight loads in flight
    float t0 = a.bias[c], t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int o = 0;
    for (; o + 4 <= O; o += 4) {
      const float w0 = wr[o], w1 = wr[o + 1], w2 = wr[o + 2], w3 = wr[o + 3];
      t0 = fmaf(s.x[which][o], w0, t0), t1 = fmaf(s.x[which][o + 1], w1, t1);
      t2 = fmaf(s.x[which][o + 2], w2, t2), t3 = fmaf(s.x[which][o + 3], w3, t3);
    }
    for (; o < O; ++o) t0 = fmaf(s.x[which][o], wr[o], t0);
    s.l[which][c] = (t0 + t1) + (t2 + t3);
  }
  __syncthreads();
  if (lane < 2 * ncls) {
    const int which = lane / ncls, c = lane - which * ncls;
    float mx = -INFINITY, sum = 0.f;
    for (int k = 0; k < ncls; ++k) mx = fmaxf(mx, s.l[which][k]);
    for (int k = 0; k < ncls; ++k) sum += expf(s.l[which][k] - mx);
    (which ? a.post_prob : a.prior_prob)[(size_t)b * ncls + c] = expf(s.l[which][c] - mx) / sum;
  }
}

// the checked argument block of a launch
inline int fill(Args &a, const float *caps_presence, const float *posterior, const float *w,
                const float *bias, float *prior_prob, float *post_prob, int B, int O, int M,
                int ncls, const scae_scaled_sum *extra_sums, int n_extra) {
  SCAE_REQUIRE(caps_presence && posterior && w && bias && prior_prob && post_prob && B > 0 &&
               M > 0);
  if (!scae_class_probs_supported(O, ncls)) return SCAE_ERR_UNSUPPORTED;
  SCAE_REQUIRE(n_extra >= 0 && n_extra <= 8 && (n_extra == 0 || extra_sums));
  a = Args{caps_presence, posterior, w, bias, prior_prob, post_prob, B, O, M, ncls, {}};
  a.extra.n = n_extra;
  for (int i = 0; i < n_extra; ++i) {
    a.extra.j[i] = extra_sums[i];
    SCAE_REQUIRE(a.extra.j[i].src && a.extra.j[i].dst && a.extra.j[i].n > 0);
  }
  return SCAE_OK;
}
}  // namespace scae_cp
