// RMSprop-with-momentum update of the flat parameter buffer -- the reference's
// default optimiser (base_experiment.py:44-77: torch.optim.RMSprop(lr,
// momentum=0.9, eps)) as one pass over (param, grad, square_avg, buf) instead
// of seven whole-buffer elementwise launches:
//   v   <- alpha v + (1 - alpha) g^2
//   buf <- momentum buf + g / (sqrt(v) + eps)
//   p   <- p - lr buf                    (momentum == 0: p <- p - lr g / (sqrt(v)+eps))
// HBM-bound: 4 streams read, 3 written, 28 bytes per parameter.
#include "common.h"

namespace {
struct OptArgs {
  float *p, *v, *buf;
  const float *g;
  long n;
  float lr, alpha, eps, momentum;
};

__device__ __forceinline__ void update(float &p, float &v, float &b, float g, const OptArgs &a) {
  v = a.alpha * v + (1.f - a.alpha) * g * g;
  const float step = g / (sqrtf(v) + a.eps);
  if (a.momentum > 0.f) {
    b = a.momentum * b + step;
    p -= a.lr * b;
  } else {
    p -= a.lr * step;
  }
}

__global__ __launch_bounds__(256) void rmsprop_kernel(OptArgs a) {
  const long stride = (long)gridDim.x * blockDim.x;
  const long n4 = a.n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 p = reinterpret_cast<float4 *>(a.p)[i], v = reinterpret_cast<float4 *>(a.v)[i];
    float4 b = a.buf ? reinterpret_cast<float4 *>(a.buf)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 g = reinterpret_cast<const float4 *>(a.g)[i];
    update(p.x, v.x, b.x, g.x, a);
    update(p.y, v.y, b.y, g.y, a);
    update(p.z, v.z, b.z, g.z, a);
    update(p.w, v.w, b.w, g.w, a);
    reinterpret_cast<float4 *>(a.p)[i] = p;
    reinterpret_cast<float4 *>(a.v)[i] = v;
    if (a.buf) reinterpret_cast<float4 *>(a.buf)[i] = b;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
    float b = a.buf ? a.buf[i] : 0.f;
    update(a.p[i], a.v[i], b, a.g[i], a);
    if (a.buf) a.buf[i] = b;
  }
}
}  // namespace

extern "C" int scae_rmsprop_step_f32(float *param, const float *grad, float *square_avg,
                                     float *buf, int64_t n, float lr, float alpha, float eps,
                                     float momentum, void *stream) {
  SCAE_REQUIRE(param && grad && square_avg && n > 0);
  if (momentum > 0.f && !buf) return SCAE_ERR_BAD_ARG;
  if (((size_t)param | (size_t)grad | (size_t)square_avg | (size_t)buf) & 15)
    return SCAE_ERR_BAD_ARG;  // flat buffers are 16-byte aligned allocations
  OptArgs a{param, square_avg, momentum > 0.f ? buf : nullptr, grad, (long)n, lr, alpha, eps,
            momentum};
  long blocks = (n / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(rmsprop_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return scae_launch_status();
}
