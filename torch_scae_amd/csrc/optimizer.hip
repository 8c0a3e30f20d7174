// RMSprop-with-momentum update of the flat parameter buffer -- the reference's
// default optimiser (base_experiment.py:44-77: torch.optim.RMSprop(lr,
// momentum=0.9, eps)) as one pass over (param, grad, square_avg, buf) instead
// of seven whole-buffer elementwise launches:
//   v   <- alpha v + (1 - alpha) g^2
//   buf <- momentum buf + g / (sqrt(v) + eps)
//   p   <- p - lr buf                    (momentum == 0: p <- p - lr g / (sqrt(v)+eps))
// with g <- g + weight_decay p first, as torch does; lr optionally read from device memory.
// HBM-bound: 4 streams read, 3 written, 28 bytes per parameter.
#include "common.h"

namespace {
struct OptArgs {
  float *p, *v, *buf;
  const float *g;
  const float *lr_dev;  // learning rate in device memory (schedules under graph replay) or NULL
  long n;
  float lr, alpha, eps, momentum, weight_decay, grad_scale;
};

__device__ __forceinline__ void update(float &p, float &v, float &b, float g, const OptArgs &a,
                                       float lr) {
  g *= a.grad_scale;  // e.g. 1/world_size after a SUM all-reduce
  if (a.weight_decay != 0.f) g = fmaf(a.weight_decay, p, g);  // torch: grad.add(param, alpha=wd)
  v = a.alpha * v + (1.f - a.alpha) * g * g;
  const float step = g / (sqrtf(v) + a.eps);
  if (a.momentum > 0.f) {
    b = a.momentum * b + step;
    p -= lr * b;
  } else {
    p -= lr * step;
  }
}

// `head` leading elements bring the (equally misaligned) buffers to a 16-byte
// boundary; then float4 lanes; then the tail
__global__ __launch_bounds__(256) void rmsprop_kernel(OptArgs a, int head) {
  const long stride = (long)gridDim.x * blockDim.x;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const float lr = a.lr_dev ? a.lr_dev[0] : a.lr;
  const long n4 = (a.n - head) >> 2;
  float *p4 = a.p + head, *v4 = a.v + head, *b4 = a.buf ? a.buf + head : nullptr;
  const float *g4 = a.g + head;
  for (long i = tid; i < n4; i += stride) {
    float4 p = reinterpret_cast<float4 *>(p4)[i], v = reinterpret_cast<float4 *>(v4)[i];
    float4 b = b4 ? reinterpret_cast<float4 *>(b4)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 g = reinterpret_cast<const float4 *>(g4)[i];
    update(p.x, v.x, b.x, g.x, a, lr);
    update(p.y, v.y, b.y, g.y, a, lr);
    update(p.z, v.z, b.z, g.z, a, lr);
    update(p.w, v.w, b.w, g.w, a, lr);
    reinterpret_cast<float4 *>(p4)[i] = p;
    reinterpret_cast<float4 *>(v4)[i] = v;
    if (b4) reinterpret_cast<float4 *>(b4)[i] = b;
  }
  // scalar edges: [0, head) and [head + 4*n4, n)
  const long tail0 = head + (n4 << 2), edge = head + (a.n - tail0);
  for (long e = tid; e < edge; e += stride) {
    const long i = e < head ? e : tail0 + (e - head);
    float b = a.buf ? a.buf[i] : 0.f;
    update(a.p[i], a.v[i], b, a.g[i], a, lr);
    if (a.buf) a.buf[i] = b;
  }
}
// The batch hand-over of a training step: image floats and int64 labels into the
// step's resident input buffers, one launch instead of two device copies.
__global__ __launch_bounds__(256) void stage_batch_kernel(float *__restrict__ dst_image,
                                                          const float *__restrict__ src_image,
                                                          long n_image,
                                                          int64_t *__restrict__ dst_label,
                                                          const int64_t *__restrict__ src_label,
                                                          long n_label) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x,
             stride = (long)gridDim.x * blockDim.x;
  const bool vec = (((size_t)dst_image | (size_t)src_image) & 15) == 0;
  const long n4 = vec ? n_image >> 2 : 0;
  for (long i = tid; i < n4; i += stride)
    reinterpret_cast<float4 *>(dst_image)[i] = reinterpret_cast<const float4 *>(src_image)[i];
  for (long i = 4 * n4 + tid; i < n_image; i += stride) dst_image[i] = src_image[i];
  for (long i = tid; i < n_label; i += stride) dst_label[i] = src_label[i];
}
}  // namespace

extern "C" int scae_stage_batch(float *dst_image, const float *src_image, int64_t n_image,
                                int64_t *dst_label, const int64_t *src_label, int64_t n_label,
                                void *stream) {
  SCAE_REQUIRE(dst_image && src_image && n_image > 0 && n_label >= 0 &&
               (n_label == 0 || (dst_label && src_label)));
  long blocks = (n_image / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(stage_batch_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, dst_image, src_image, (long)n_image, dst_label,
                     src_label, (long)n_label);
  return scae_launch_status();
}

extern "C" int scae_rmsprop_step_f32(float *param, const float *grad, float *square_avg,
                                     float *buf, int64_t n, float lr, const float *lr_dev,
                                     float alpha, float eps, float momentum, float weight_decay,
                                     float grad_scale, void *stream) {
  SCAE_REQUIRE(param && grad && square_avg && n > 0);
  if (momentum > 0.f && !buf) return SCAE_ERR_BAD_ARG;
  // the four buffers are slices of equally laid out flat buffers: same phase
  // within a 16-byte line (any 4-byte aligned start is fine)
  const size_t phase = (size_t)param & 15;
  if ((phase & 3) || ((size_t)grad & 15) != phase || ((size_t)square_avg & 15) != phase ||
      (momentum > 0.f && ((size_t)buf & 15) != phase))
    return SCAE_ERR_BAD_ARG;
  int head = (int)((16 - phase) & 15) / 4;
  if (head > n) head = (int)n;
  OptArgs a{param, square_avg, momentum > 0.f ? buf : nullptr, grad, lr_dev, (long)n, lr, alpha,
            eps, momentum, weight_decay, grad_scale};
  long blocks = (n / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(rmsprop_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a,
                     head);
  return scae_launch_status();
}
