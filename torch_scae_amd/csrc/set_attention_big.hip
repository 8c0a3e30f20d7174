// K2 for sets beyond SCAE_ATTN_MAX_SET (set_transformer.py:24-47 takes any N, M): the same
// masked softmax attention, written for generality, not for speed -- one workgroup per
// (head, batch) problem, a wave per query row (forward, first backward phase) or per key
// (second backward phase), plain fp32 FMAs.  The sizes the benchmarks use (<= 64) never
// reach this file; it exists so that a reference-valid model with, say, 80 part capsules runs.
//   forward : s = q k^T; s -= (1 - presence) 1e32; s /= sqrt_dk (the reference's fp32
//             sequence); p = softmax(s); out = p v; probs kept for the backward
//   backward: dP = dO v^T; dS = p (dP - sum_m p dP) / sqrt_dk (parked in LDS, N x M floats);
//             gq = dS k, gk = dS^T q, gv = p^T dO, gpresence = 1e32 sum_n dS
#include "common.h"

namespace scae_attn_big {
namespace {
constexpr int NT = 256, NW = NT / 64;

__global__ __launch_bounds__(NT) void fwd_kernel(const float *__restrict__ q,
                                                 const float *__restrict__ k,
                                                 const float *__restrict__ v,
                                                 const float *__restrict__ presence,
                                                 float *__restrict__ out, float *__restrict__ probs,
                                                 int N, int M, int dk, int dv, float sqrt_dk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // NW rows of M logits
  const int hb = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float *qb = q + (size_t)hb * N * dk, *kb = k + (size_t)hb * M * dk;
  const float *vb = v + (size_t)hb * M * dv;
  float *row = smem + (size_t)wid * M;
  for (int n = wid; n < N; n += NW) {
    float mx = -INFINITY;
    for (int m = lane; m < M; m += 64) {
      float s = 0.f;
      for (int j = 0; j < dk; ++j) s = fmaf(qb[(size_t)n * dk + j], kb[(size_t)m * dk + j], s);
      if (presence) s = s - (1.f - presence[(size_t)hb * M + m]) * 1e32f;
      s = s / sqrt_dk;
      row[m] = s;
      mx = fmaxf(mx, s);
    }
    mx = scae::wave_max(mx);
    float sum = 0.f;
    for (int m = lane; m < M; m += 64) {
      const float e = expf(row[m] - mx);
      row[m] = e;
      sum += e;
    }
    sum = scae::wave_sum(sum);
    for (int m = lane; m < M; m += 64) {
      const float p = row[m] / sum;
      row[m] = p;
      probs[((size_t)hb * N + n) * M + m] = p;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes
    for (int j = lane; j < dv; j += 64) {
      float acc = 0.f;
      for (int m = 0; m < M; ++m) acc = fmaf(row[m], vb[(size_t)m * dv + j], acc);
      out[((size_t)hb * N + n) * dv + j] = acc;
    }
  }
}

__global__ __launch_bounds__(NT) void bwd_kernel(
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const float *__restrict__ probs, const float *__restrict__ gout, float *__restrict__ gq,
    float *__restrict__ gk, float *__restrict__ gv, float *__restrict__ gpresence, int N, int M,
    int dk, int dv, float sqrt_dk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // dS: N x M
  const int hb = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float *qb = q + (size_t)hb * N * dk, *kb = k + (size_t)hb * M * dk;
  const float *vb = v + (size_t)hb * M * dv, *pb = probs + (size_t)hb * N * M;
  const float *gb = gout + (size_t)hb * N * dv;
  for (int n = wid; n < N; n += NW) {   // a wave per query row
    float *ds = smem + (size_t)n * M;
    float t = 0.f;
    for (int m = lane; m < M; m += 64) {
      float dp = 0.f;
      for (int j = 0; j < dv; ++j) dp = fmaf(gb[(size_t)n * dv + j], vb[(size_t)m * dv + j], dp);
      ds[m] = dp;
      t = fmaf(pb[(size_t)n * M + m], dp, t);
    }
    t = scae::wave_sum(t);
    for (int m = lane; m < M; m += 64) ds[m] = pb[(size_t)n * M + m] * (ds[m] - t) / sqrt_dk;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int j = lane; j < dk; j += 64) {
      float acc = 0.f;
      for (int m = 0; m < M; ++m) acc = fmaf(ds[m], kb[(size_t)m * dk + j], acc);
      gq[((size_t)hb * N + n) * dk + j] = acc;
    }
  }
  __syncthreads();
  for (int m = wid; m < M; m += NW) {   // a wave per key
    for (int j = lane; j < dk; j += 64) {
      float acc = 0.f;
      for (int n = 0; n < N; ++n) acc = fmaf(smem[(size_t)n * M + m], qb[(size_t)n * dk + j], acc);
      gk[((size_t)hb * M + m) * dk + j] = acc;
    }
    for (int j = lane; j < dv; j += 64) {
      float acc = 0.f;
      for (int n = 0; n < N; ++n) acc = fmaf(pb[(size_t)n * M + m], gb[(size_t)n * dv + j], acc);
      gv[((size_t)hb * M + m) * dv + j] = acc;
    }
    if (gpresence) {   // routing -= (1 - p) 1e32  =>  d / dp = +1e32
      float s = 0.f;
      for (int n = lane; n < N; n += 64) s += smem[(size_t)n * M + m];
      s = scae::wave_sum(s);
      if (lane == 0) gpresence[(size_t)hb * M + m] = s * 1e32f;
    }
  }
}

int raise(const void *fn, size_t bytes) {
  if (bytes > 160 * 1024) return SCAE_ERR_UNSUPPORTED;
  if (bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
  }
  return SCAE_OK;
}
}  // namespace

int fwd(const float *q, const float *k, const float *v, const float *presence, float *out,
        float *probs, int HB, int N, int M, int dk, int dv, float sqrt_dk, hipStream_t st) {
  const size_t lds = sizeof(float) * (size_t)NW * M;
  int rc = raise(reinterpret_cast<const void *>(fwd_kernel), lds);
  if (rc) return rc;
  scae::launch(fwd_kernel, dim3(HB), dim3(NT), lds, st, q, k, v, presence, out, probs, N, M,
                     dk, dv, sqrt_dk);
  return scae_launch_status();
}

int bwd(const float *q, const float *k, const float *v, const float *probs, const float *gout,
        float *gq, float *gk, float *gv, float *gpresence, int HB, int N, int M, int dk, int dv,
        float sqrt_dk, hipStream_t st) {
  const size_t lds = sizeof(float) * (size_t)N * M;   // dS of one problem
  int rc = raise(reinterpret_cast<const void *>(bwd_kernel), lds);
  if (rc) return rc;
  scae::launch(bwd_kernel, dim3(HB), dim3(NT), lds, st, q, k, v, probs, gout, gq, gk, gv,
                     gpresence, N, M, dk, dv, sqrt_dk);
  return scae_launch_status();
}
}  // namespace scae_attn_big
