// Class probabilities of SCAE.forward (stacked_capsule_auto_encoder.py:205-212):
//   prior_cls_prob     = softmax(W caps_presence + b)
//   posterior_cls_prob = softmax(W sum_m posterior[:, :O, m] + b)
// (both through prior_classifier, as the reference does).  One workgroup (one
// wave) per image, lanes over object capsules; replaces two Linear + Softmax
// pairs and a reduction (5 launches) with one.
#include "common.h"

namespace {
constexpr int MAXCLS = 32;
struct ExtraSums {
  scae_scaled_sum j[8];
  int n;
};

__global__ __launch_bounds__(64) void class_probs_kernel(
    const float *__restrict__ cp, const float *__restrict__ posterior,
    const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ prior_prob,
    float *__restrict__ post_prob, int B, int O, int M, int ncls, ExtraSums extra) {
  __shared__ float s_x[2][64], s_l[2][MAXCLS];
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= B) {  // riders: scaled full sums (the scalar outputs of the forward pass)
    const scae_scaled_sum &job = extra.j[b - B];
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
    s_x[0][lane] = cp[(size_t)b * O + lane];
    const float *p = posterior + ((size_t)b * (O + 1) + lane) * M;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;  // four loads in flight
    int m = 0;
    for (; m + 4 <= M; m += 4) m0 += p[m], m1 += p[m + 1], m2 += p[m + 2], m3 += p[m + 3];
    for (; m < M; ++m) m0 += p[m];
    s_x[1][lane] = (m0 + m1) + (m2 + m3);
  }
  __syncthreads();
  if (lane < 2 * ncls) {  // lane = (input, class): one logit each
    const int which = lane / ncls, c = lane - which * ncls;
    const float *wr = w + (size_t)c * O;  // four weight loads in flight
    float t0 = bias[c], t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int o = 0;
    for (; o + 4 <= O; o += 4) {
      const float w0 = wr[o], w1 = wr[o + 1], w2 = wr[o + 2], w3 = wr[o + 3];
      t0 = fmaf(s_x[which][o], w0, t0), t1 = fmaf(s_x[which][o + 1], w1, t1);
      t2 = fmaf(s_x[which][o + 2], w2, t2), t3 = fmaf(s_x[which][o + 3], w3, t3);
    }
    for (; o < O; ++o) t0 = fmaf(s_x[which][o], wr[o], t0);
    s_l[which][c] = (t0 + t1) + (t2 + t3);
  }
  __syncthreads();
  if (lane < 2 * ncls) {
    const int which = lane / ncls, c = lane - which * ncls;
    float mx = -INFINITY, sum = 0.f;
    for (int k = 0; k < ncls; ++k) mx = fmaxf(mx, s_l[which][k]);
    for (int k = 0; k < ncls; ++k) sum += expf(s_l[which][k] - mx);
    (which ? post_prob : prior_prob)[(size_t)b * ncls + c] = expf(s_l[which][c] - mx) / sum;
  }
}
}  // namespace

extern "C" int scae_class_probs_supported(int O, int ncls) {
  return O > 0 && O <= 64 && ncls > 0 && ncls <= MAXCLS;
}

extern "C" int scae_class_probs_f32(const float *caps_presence, const float *posterior,
                                    const float *w, const float *bias, float *prior_prob,
                                    float *post_prob, int B, int O, int M, int ncls,
                                    const scae_scaled_sum *extra_sums, int n_extra,
                                    void *stream) {
  SCAE_REQUIRE(caps_presence && posterior && w && bias && prior_prob && post_prob && B > 0 &&
               M > 0);
  if (!scae_class_probs_supported(O, ncls)) return SCAE_ERR_UNSUPPORTED;
  SCAE_REQUIRE(n_extra >= 0 && n_extra <= 8 && (n_extra == 0 || extra_sums));
  ExtraSums ex;
  ex.n = n_extra;
  for (int i = 0; i < n_extra; ++i) {
    ex.j[i] = extra_sums[i];
    SCAE_REQUIRE(ex.j[i].src && ex.j[i].dst && ex.j[i].n > 0);
  }
  hipLaunchKernelGGL(class_probs_kernel, dim3(B + n_extra), dim3(64), 0, (hipStream_t)stream,
                     caps_presence, posterior, w, bias, prior_prob, post_prob, B, O, M, ncls,
                     ex);
  return scae_launch_status();
}
