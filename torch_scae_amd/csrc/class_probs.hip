// Class probabilities of SCAE.forward (stacked_capsule_auto_encoder.py:205-212):
//   prior_cls_prob     = softmax(W caps_presence + b)
//   posterior_cls_prob = softmax(W sum_m posterior[:, :O, m] + b)
// (both through prior_classifier, as the reference does).  One workgroup (one
// wave) per image, lanes over object capsules; replaces two Linear + Softmax
// pairs and a reduction (5 launches) with one.  (Body: class_probs_dev.h -- in a training
// step these workgroups ride in the loss tail's per-image launch instead.)
#include "class_probs_dev.h"

namespace {
__global__ __launch_bounds__(64) void class_probs_kernel(scae_cp::Args a) {
  __shared__ scae_cp::Lds s;
  scae_cp::body(a, s, blockIdx.x, threadIdx.x);
}
}  // namespace

extern "C" int scae_class_probs_supported(int O, int ncls) {
  return O > 0 && O <= 64 && ncls > 0 && ncls <= scae_cp::MAXCLS;
}

extern "C" int scae_class_probs_f32(const float *caps_presence, const float *posterior,
                                    const float *w, const float *bias, float *prior_prob,
                                    float *post_prob, int B, int O, int M, int ncls,
                                    const scae_scaled_sum *extra_sums, int n_extra,
                                    void *stream) {
  scae_cp::Args a;
  int rc = scae_cp::fill(a, caps_presence, posterior, w, bias, prior_prob, post_prob, B, O, M,
                         ncls, extra_sums, n_extra);
  if (rc) return rc;
  scae::launch(class_probs_kernel, dim3(B + n_extra), dim3(64), 0, (hipStream_t)stream, a);
  return scae_launch_status();
}
