// RMSprop-with-momentum update of the flat parameter buffer -- the reference's
// default optimiser (base_experiment.py:44-77: torch.optim.RMSprop(lr,
// momentum=0.9, eps)) as one pass over (param, grad, square_avg, buf) instead
// of seven whole-buffer elementwise launches:
//   v   <- alpha v + (1 - alpha) g^2
//   buf <- momentum buf + g / (sqrt(v) + eps)
//   p   <- p - lr buf                    (momentum == 0: p <- p - lr g / (sqrt(v)+eps))
// with g <- g + weight_decay p first, as torch does; lr optionally read from device memory.
// HBM-bound: 4 streams read, 3 written, 28 bytes per parameter.
#include "sum_rows_dev.h"

namespace {
struct OptArgs {
  float *p, *v, *buf;
  const float *g;
  const float *lr_dev;  // learning rate in device memory (schedules under graph replay) or NULL
  long n;
  float lr, alpha, eps, momentum, weight_decay, grad_scale;
};

// (no FMA contraction: the update is compiled into three kernels -- vector, scalar edge, the
// sum workgroups of rmsprop_sums_kernel -- that must round alike, bit for bit)
#pragma clang fp contract(off)
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
// The step's LAST column sums and the optimiser in one launch.  A training step's backward
// ends in one scae_sum_rows_multi launch whose outputs are slots of the flat gradient buffer
// (parameter gradients nothing else reads), followed by this file's pass over the four flat
// buffers: two latency-bound launches on the step's dependent chain.  Here the sum
// workgroups are the head of the grid and apply the update of the elements they produce
// themselves (the element's offset in the flat buffers is its destination's offset from the
// gradient base); the streaming workgroups behind them skip exactly those elements -- the
// segments' destination ranges, rebuilt from the job table into LDS by every workgroup.  The
// arithmetic per element is unchanged: the results equal the two launches' bit for bit.
constexpr int MAXR = scae_sums::MAXJOBS * 8;
__global__ __launch_bounds__(256) void rmsprop_sums_kernel(OptArgs a, int head,
                                                           scae_sums::Jobs jobs, int sum_blocks) {
  __shared__ float red[scae_sums::NT];
  __shared__ int r_lo[MAXR], r_hi[MAXR];
  __shared__ int r_n;
  const float lr = a.lr_dev ? a.lr_dev[0] : a.lr;
  if ((int)blockIdx.x < sum_blocks) {   // workgroup-uniform
    scae_sums::sum_block(jobs, blockIdx.x, red, [&](float *dst, float v) {
      *dst = v;
      const long off = dst - a.g;
      if (off >= 0 && off < a.n) {
        float b = a.buf ? a.buf[off] : 0.f;
        update(a.p[off], a.v[off], b, v, a, lr);
        if (a.buf) a.buf[off] = b;
      }
    });
    return;
  }
  // the ranges of the flat buffers the sum workgroups own (a thread per segment)
  if (threadIdx.x == 0) r_n = 0;
  __syncthreads();
  if (threadIdx.x < MAXR) {
    const int j = threadIdx.x >> 3, i = threadIdx.x & 7;
    if (j < jobs.n && i < jobs.j[j].n) {
      const scae_sums::Seg &g = jobs.j[j].s[i];
      const long width = g.end - g.begin;
      const long len = g.period > 0 ? (long)(jobs.j[j].cols / g.period) * width : width;
      const long lo = g.dst - a.g;
      if (lo + len > 0 && lo < a.n) {
        const int k = atomicAdd(&r_n, 1);   // (order is irrelevant: membership only)
        r_lo[k] = (int)max(lo, 0l), r_hi[k] = (int)min(lo + len, a.n);
      }
    }
  }
  __syncthreads();
  const int nr = r_n;
  const long stride = (long)(gridDim.x - sum_blocks) * blockDim.x;
  const long tid = (long)(blockIdx.x - sum_blocks) * blockDim.x + threadIdx.x;
  const long n4 = (a.n - head) >> 2;
  float *p4 = a.p + head, *v4 = a.v + head, *b4 = a.buf ? a.buf + head : nullptr;
  const float *g4 = a.g + head;
  for (long i = tid; i < n4; i += stride) {
    // (the loads first: the range scan runs under their latency)
    float4 p = reinterpret_cast<float4 *>(p4)[i], v = reinterpret_cast<float4 *>(v4)[i];
    float4 b = b4 ? reinterpret_cast<float4 *>(b4)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 g = reinterpret_cast<const float4 *>(g4)[i];
    const int e0 = head + 4 * (int)i;
    // bit u: element e0 + u belongs to a sum workgroup
    int own = 0;
    for (int k = 0; k < nr; ++k) {
      const int lo = r_lo[k] - e0, hi = r_hi[k] - e0;   // the range relative to the quad
      if (hi > 0 && lo < 4) own |= ((hi >= 4 ? 15 : (1 << hi) - 1) & ~((lo <= 0 ? 0 : (1 << lo) - 1)));
    }
    if (own == 15) continue;
    update(p.x, v.x, b.x, g.x, a, lr);
    update(p.y, v.y, b.y, g.y, a, lr);
    update(p.z, v.z, b.z, g.z, a, lr);
    update(p.w, v.w, b.w, g.w, a, lr);
    if (own == 0) {
      reinterpret_cast<float4 *>(p4)[i] = p;
      reinterpret_cast<float4 *>(v4)[i] = v;
      if (b4) reinterpret_cast<float4 *>(b4)[i] = b;
    } else {   // (rare: a quad that straddles the edge of an owned range)
      const float pe[4] = {p.x, p.y, p.z, p.w}, ve[4] = {v.x, v.y, v.z, v.w},
                  be[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (!((own >> u) & 1)) {
          p4[4 * i + u] = pe[u], v4[4 * i + u] = ve[u];
          if (b4) b4[4 * i + u] = be[u];
        }
    }
  }
  const long tail0 = head + (n4 << 2), edge = head + (a.n - tail0);
  for (long e = tid; e < edge; e += stride) {
    const long i = e < head ? e : tail0 + (e - head);
    bool owned = false;
    for (int k = 0; k < nr; ++k) owned |= i >= r_lo[k] && i < r_hi[k];
    if (owned) continue;
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
  scae::launch(stage_batch_kernel, dim3((unsigned)blocks), dim3(256), 0,
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
  scae::launch(rmsprop_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a,
                     head);
  return scae_launch_status();
}

// scae_sum_rows_multi_f32(jobs) followed by scae_rmsprop_step_f32(...) as ONE launch
// (rmsprop_sums_kernel above); every job's destinations should be slots of `grad` (others are
// only summed).  weight_decay must be 0 (a parameter without a gradient is then untouched by
// either form).
extern "C" int scae_rmsprop_sums_step_f32(float *param, float *grad, float *square_avg,
                                          float *buf, int64_t n, float lr, const float *lr_dev,
                                          float alpha, float eps, float momentum,
                                          float grad_scale, const scae_sum_job *jobs, int n_jobs,
                                          void *stream) {
  SCAE_REQUIRE(param && grad && square_avg && n > 0);
  if (momentum > 0.f && !buf) return SCAE_ERR_BAD_ARG;
  const size_t phase = (size_t)param & 15;
  if ((phase & 3) || ((size_t)grad & 15) != phase || ((size_t)square_avg & 15) != phase ||
      (momentum > 0.f && ((size_t)buf & 15) != phase))
    return SCAE_ERR_BAD_ARG;
  scae_sums::Jobs js;
  const int sum_blocks = scae_sums::fill_jobs(js, jobs, n_jobs);
  SCAE_REQUIRE(sum_blocks > 0);
  int head = (int)((16 - phase) & 15) / 4;
  if (head > n) head = (int)n;
  OptArgs a{param, square_avg, momentum > 0.f ? buf : nullptr, grad, lr_dev, (long)n, lr, alpha,
            eps, momentum, 0.f, grad_scale};
  // the whole grid resident at once (256 CUs x 8 workgroups of 256 threads): the streaming
  // workgroups take what the sum workgroups leave, in a grid-stride loop
  long blocks = (n / 4 + 255) / 256;
  const long room = 2048 - sum_blocks;
  const long cap = room > 512 ? room : 512;
  blocks = blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
  scae::launch(rmsprop_sums_kernel, dim3((unsigned)(sum_blocks + blocks)), dim3(256), 0,
                     (hipStream_t)stream, a, head, js, sum_blocks);
  return scae_launch_status();
}
