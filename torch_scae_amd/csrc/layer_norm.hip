// LayerNorm over the last dimension for the set-transformer blocks that run module by
// module (nn.LayerNorm(d) in MAB, set_transformer.py:114-131; the fused trunk K2b has its
// own in-register form).  One wave per row: lanes stride over the d columns, mean and
// variance by two passes over the row held in registers (DPP + readlane reductions), so a
// row is read once.  Backward: the same mapping; every workgroup also accumulates its rows'
// contributions to the weight / bias gradients per column and writes ONE partial row
// [gw (d) | gb (d)] -- the caller sums the partial rows (scae_sum_rows_f32; fixed order).
#include "common.h"

namespace {
constexpr int NT = 256, NWV = NT / 64, CPL = 16;   // columns per lane: d <= 64 * CPL

__global__ __launch_bounds__(NT) void ln_fwd_kernel(const float *__restrict__ x,
                                                    const float *__restrict__ w,
                                                    const float *__restrict__ b,
                                                    float *__restrict__ y, float *__restrict__ mean,
                                                    float *__restrict__ rstd, long rows, int d,
                                                    float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long r = (long)blockIdx.x * NWV + wave; r < rows; r += (long)gridDim.x * NWV) {
    const float *xr = x + r * d;
    float v[CPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = lane + 64 * i;
      v[i] = c < d ? xr[c] : 0.f;
      s += v[i];
    }
    const float mu = scae::wave_sum(s) / d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const float t = lane + 64 * i < d ? v[i] - mu : 0.f;
      q = fmaf(t, t, q);
    }
    const float rs = rsqrtf(scae::wave_sum(q) / d + eps);
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = lane + 64 * i;
      if (c < d) y[r * d + c] = (v[i] - mu) * rs * (w ? w[c] : 1.f) + (b ? b[c] : 0.f);
    }
    if (lane == 0) mean[r] = mu, rstd[r] = rs;
  }
}

__global__ __launch_bounds__(NT) void ln_bwd_kernel(const float *__restrict__ x,
                                                    const float *__restrict__ w,
                                                    const float *__restrict__ mean,
                                                    const float *__restrict__ rstd,
                                                    const float *__restrict__ gy,
                                                    float *__restrict__ gx,
                                                    float *__restrict__ partial, long rows, int d) {
  __shared__ float red[NWV][2 * 64 * CPL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float gw[CPL], gb[CPL];
#pragma unroll
  for (int i = 0; i < CPL; ++i) gw[i] = gb[i] = 0.f;
  for (long r = (long)blockIdx.x * NWV + wave; r < rows; r += (long)gridDim.x * NWV) {
    const float mu = mean[r], rs = rstd[r];
    float xh[CPL], gs[CPL];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = lane + 64 * i;
      const bool ok = c < d;
      const float g = ok ? gy[r * d + c] : 0.f;
      xh[i] = ok ? (x[r * d + c] - mu) * rs : 0.f;
      gs[i] = g * (w && ok ? w[c] : (ok ? 1.f : 0.f));
      gw[i] = fmaf(g, xh[i], gw[i]);
      gb[i] += g;
      c1 = fmaf(gs[i], xh[i], c1);
      c2 += gs[i];
    }
    c1 = scae::wave_sum(c1) / d;
    c2 = scae::wave_sum(c2) / d;
    if (gx) {
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int c = lane + 64 * i;
        if (c < d) gx[r * d + c] = rs * (gs[i] - c2 - xh[i] * c1);
      }
    }
  }
  if (!partial) return;   // (uniform)
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    red[wave][lane + 64 * i] = gw[i];
    red[wave][64 * CPL + lane + 64 * i] = gb[i];
  }
  __syncthreads();
  float *row = partial + (size_t)blockIdx.x * 2 * d;
  for (int c = threadIdx.x; c < 2 * d; c += NT) {
    const int at = c < d ? c : 64 * CPL + (c - d);
    float t = 0.f;
#pragma unroll
    for (int wv = 0; wv < NWV; ++wv) t += red[wv][at];
    row[c] = t;
  }
}
}  // namespace

extern "C" int scae_layer_norm_rows(int64_t rows) {
  if (rows <= 0) return 0;
  const int64_t b = (rows + NWV - 1) / NWV;
  return (int)(b < 512 ? b : 512);
}

extern "C" int scae_layer_norm_fwd_f32(const float *x, const float *weight, const float *bias,
                                       float *y, float *mean, float *rstd, int64_t rows, int d,
                                       float eps, void *stream) {
  SCAE_REQUIRE(x && y && mean && rstd && rows > 0 && d > 0);
  if (d > 64 * CPL) return SCAE_ERR_UNSUPPORTED;
  scae::launch(ln_fwd_kernel, dim3(scae_layer_norm_rows(rows)), dim3(NT), 0,
                     (hipStream_t)stream, x, weight, bias, y, mean, rstd, (long)rows, d, eps);
  return scae_launch_status();
}

extern "C" int scae_layer_norm_bwd_f32(const float *x, const float *weight, const float *mean,
                                       const float *rstd, const float *gy, float *gx,
                                       float *partial, int64_t rows, int d, void *stream) {
  SCAE_REQUIRE(x && mean && rstd && gy && (gx || partial) && rows > 0 && d > 0);
  if (d > 64 * CPL) return SCAE_ERR_UNSUPPORTED;
  scae::launch(ln_bwd_kernel, dim3(scae_layer_norm_rows(rows)), dim3(NT), 0,
                     (hipStream_t)stream, x, weight, mean, rstd, gy, gx, partial, (long)rows, d);
  return scae_launch_status();
}
