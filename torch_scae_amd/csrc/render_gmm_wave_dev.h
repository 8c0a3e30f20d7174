// Device code of the wave-form likelihood forward (render_gmm_wave.hip), in a header so that
// the kernel can also run as a block range of a launch it shares with an independent kernel
// (trunk_logprob.hip).  See render_gmm_wave.hip for the design.
#pragma once
#include "common.h"
#include "render_gmm_dev.h"

namespace scae_k1 {
namespace {
constexpr float kLog2e = 1.44269504088896340736f, kLn2 = 0.69314718055994530942f;
constexpr int KC = 8;          // components per register chunk
constexpr float kMasked = -3.0e38f;

template <int C> struct TexelOf { static constexpr int TX = C == 1 ? 2 : (C <= 3 ? 4 : 8); };

__device__ __forceinline__ float ex2(float v) { return __builtin_amdgcn_exp2f(v); }
__device__ __forceinline__ float lg2(float v) { return __builtin_amdgcn_logf(v); }

// Components [k0, k0 + nk) of image b as padded planes of interleaved TX-float texels
// {channel 0 .. C - 1, alpha * alpha_scale, 0 ..}: one padded texel ROW per thread (one
// division per row); the caller synchronises.
template <int C>
__device__ __forceinline__ void stage_planes(float *s_pl, const scae_decoder_desc &d, int b, int k0,
                                             int nk, float alpha_scale, int tid, int nthr) {
  constexpr int TX = TexelOf<C>::TX;
  const int M = d.M, th = d.th, tw = d.tw, tsz = th * tw;
  const int psz = pad_elems(th, tw), pw = pad_w(tw), prow = th + 4;
  const float *g_tmpl = d.templates + (size_t)tb(d, b) * M * C * tsz;
  const float inv_prow = 1.f / (float)prow;
  for (int r = tid; r < nk * prow; r += nthr) {
    const int kl = (int)(((float)r + 0.5f) * inv_prow), yp = r - kl * prow, y = yp - 2, k = k0 + kl;
    float *dst = s_pl + ((size_t)kl * psz + yp * pw) * TX;
    const bool in = y >= 0 && y < th;
    const float *ts = g_tmpl + (size_t)k * C * tsz + y * tw;
    const float *as = d.templates_alpha + (size_t)k * tsz + y * tw;
    for (int xp = 0; xp < pw; ++xp) {
      const int xx = xp - 2;
      float v[TX];
#pragma unroll
      for (int c = 0; c < TX; ++c) v[c] = 0.f;
      if (in && xx >= 0 && xx < tw) {
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = ts[c * tsz + xx];
        v[C] = as[xx] * alpha_scale;
      }
      if (TX == 2) {
        *reinterpret_cast<float2 *>(dst + xp * 2) = make_float2(v[0], v[1]);
      } else {
#pragma unroll
        for (int c = 0; c < TX; c += 4)
          *reinterpret_cast<float4 *>(dst + xp * TX + c) =
              make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
      }
    }
  }
}

// Workgroup (tile, b) of n_tiles per image as a device function (threads 0 .. nthr - 1;
// `smem`: its dynamic LDS), so that the launch can be shared (trunk_logprob.hip).
template <int C>
__device__ __forceinline__ void logprob_wave_body(
    const scae_decoder_desc &d, const float *__restrict__ x, float *__restrict__ log_prob,
    float *__restrict__ lse_post, float *__restrict__ lse_prior, int ppb,
    float *__restrict__ block_sums, float *smem, int tile, int b, int n_tiles, int nthr) {
  constexpr int TX = TexelOf<C>::TX;
  const int tid = threadIdx.x;
  const int M = d.M, W = d.W, HW = d.H * d.W, th = d.th, tw = d.tw;
  const int psz = pad_elems(th, tw), pw = pad_w(tw);
  const Scalars sc = load_scalars(d);
  float *s_pl = smem;                              // M padded planes of TX-float texels
  float *s_coef = s_pl + (size_t)M * psz * TX;     // (M + KC) x 8: texel-space map, presence
  float *s_red = s_coef + (M + KC) * 8;            // 16

  // ---- stage: one padded texel row per thread ---------------------------------------
  {
    stage_planes<C>(s_pl, d, b, 0, M, kLog2e, tid, nthr);
    // texel position of normalised (xn, yn):  ix = ((a0 xn + a1 yn + a2 + 1) tw - 1) / 2
    for (int k = tid; k < M + KC; k += nthr) {
      float co[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, kMasked, 0.f};   // k >= M: masked out
      if (k < M) {
        const float *a = d.pose + ((size_t)b * M + k) * 6;
        const float hx = 0.5f * (float)tw, hy = 0.5f * (float)th;
        co[0] = hx * a[0], co[1] = hx * a[1], co[2] = hx * (a[2] + 1.f) - 0.5f;
        co[3] = hy * a[3], co[4] = hy * a[4], co[5] = hy * (a[5] + 1.f) - 0.5f;
        co[6] = d.presence ? kLog2e * log_safe(d.presence[b * M + k]) : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s_coef[k * 8 + i] = co[i];
    }
  }
  __syncthreads();

  const int p = tile * ppb + tid;
  const bool live = p < HW && tid < ppb;
  const int pc = live ? p : HW - 1;
  const float inv_w = 1.f / (float)W;
  const int pi = (int)(((float)pc + 0.5f) * inv_w), pj = pc - pi * W;   // exact for p < 2^22
  const float xn = (float)(2 * pj + 1) * inv_w - 1.f;
  const float yn = (float)(2 * pi + 1) * (1.f / (float)d.H) - 1.f;
  const float txf = (float)tw, tyf = (float)th, pwf = (float)pw;
  const float c2 = kLog2e * 0.5f * sc.inv_var;

  float xv[C], mpost[C], spost[C];
  // the background component (k = M) opens both running log-sum-exps
  const float u_bg = kLog2e * sc.bg_ml;
  float mprior = u_bg, sprior = 1.f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    xv[c] = x[((size_t)b * C + c) * HW + pc];
    const float bgv = d.bg_image ? d.bg_image[((size_t)b * C + c) * HW + pc] : sc.bg_val;
    const float df = xv[c] - bgv;
    mpost[c] = fmaf(df * df, -c2, u_bg);
    spost[c] = 1.f;
  }

  const float *s_tap = s_pl + (size_t)(2 * pw + 2) * TX;   // tap (0, 0) of plane 0
  for (int k0 = 0; k0 < M; k0 += KC) {
    float uv[KC], pv[KC][C];
#pragma unroll
    for (int q = 0; q < KC; ++q) {
      const int k = k0 + q, kp = k < M ? k : M - 1;   // (wave-uniform)
      const float4 ca = *reinterpret_cast<const float4 *>(s_coef + k * 8);
      const float4 cb = *reinterpret_cast<const float4 *>(s_coef + k * 8 + 4);
      float ix = fmaf(ca.x, xn, fmaf(ca.y, yn, ca.z));
      float iy = fmaf(ca.w, xn, fmaf(cb.x, yn, cb.y));
      ix = fminf(fmaxf(ix, -2.f), txf);   // fmaxf(NaN, -2) = -2: everything outside
      iy = fminf(fmaxf(iy, -2.f), tyf);
      const float x0f = floorf(ix), y0f = floorf(iy);
      const float fx = ix - x0f, fy = iy - y0f;
      const int idx = (int)fmaf(y0f, pwf, x0f);
      const float *q0 = s_tap + ((size_t)kp * psz + idx) * TX;
      const float *q1 = q0 + pw * TX;
      float v[TX];
      if (TX == 2) {
        const float2 v00 = *reinterpret_cast<const float2 *>(q0);
        const float2 v01 = *reinterpret_cast<const float2 *>(q0 + 2);
        const float2 v10 = *reinterpret_cast<const float2 *>(q1);
        const float2 v11 = *reinterpret_cast<const float2 *>(q1 + 2);
        const float t0 = fmaf(fx, v01.x - v00.x, v00.x), a0 = fmaf(fx, v01.y - v00.y, v00.y);
        const float t1 = fmaf(fx, v11.x - v10.x, v10.x), a1 = fmaf(fx, v11.y - v10.y, v10.y);
        v[0] = fmaf(fy, t1 - t0, t0);
        v[1] = fmaf(fy, a1 - a0, a0);
      } else {
#pragma unroll
        for (int c4 = 0; c4 < TX; c4 += 4) {
          const float4 v00 = *reinterpret_cast<const float4 *>(q0 + c4);
          const float4 v01 = *reinterpret_cast<const float4 *>(q0 + TX + c4);
          const float4 v10 = *reinterpret_cast<const float4 *>(q1 + c4);
          const float4 v11 = *reinterpret_cast<const float4 *>(q1 + TX + c4);
          const float e00[4] = {v00.x, v00.y, v00.z, v00.w}, e01[4] = {v01.x, v01.y, v01.z, v01.w};
          const float e10[4] = {v10.x, v10.y, v10.z, v10.w}, e11[4] = {v11.x, v11.y, v11.z, v11.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (c4 + e > C) continue;
            const float t0 = fmaf(fx, e01[e] - e00[e], e00[e]);
            const float t1 = fmaf(fx, e11[e] - e10[e], e10[e]);
            v[c4 + e] = fmaf(fy, t1 - t0, t0);
          }
        }
      }
      const float u = v[C] + cb.z;   // log2-domain mixing logit (part_decoder.py:225-231)
      uv[q] = u;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float df = xv[c] - v[c];
        pv[q][c] = fmaf(df * df, -c2, u);
      }
    }
    // fold the chunk into the running (max, sum) pairs: one exponential per value
    {
      float cm = uv[0];
#pragma unroll
      for (int q = 1; q < KC; ++q) cm = fmaxf(cm, uv[q]);
      const float mn = fmaxf(mprior, cm);
      float s = sprior * ex2(mprior - mn);
#pragma unroll
      for (int q = 0; q < KC; ++q) s += ex2(uv[q] - mn);
      mprior = mn, sprior = s;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float cm = pv[0][c];
#pragma unroll
      for (int q = 1; q < KC; ++q) cm = fmaxf(cm, pv[q][c]);
      const float mn = fmaxf(mpost[c], cm);
      float s = spost[c] * ex2(mpost[c] - mn);
#pragma unroll
      for (int q = 0; q < KC; ++q) s += ex2(pv[q][c] - mn);
      mpost[c] = mn, spost[c] = s;
    }
  }

  // back to natural logarithms; the Normal's constant joins here
  const float knorm = -sc.log_sigma - scae::kHalfLog2Pi;
  const float lprior = kLn2 * (mprior + lg2(sprior));
  float lp_sum = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float lpost = fmaf(kLn2, mpost[c] + lg2(spost[c]), knorm);
    if (live) {
      const size_t o = ((size_t)b * C + c) * HW + p;
      if (log_prob) log_prob[o] = lpost - lprior;
      lse_post[o] = lpost;
      lp_sum += lpost - lprior;
    }
  }
  if (live) lse_prior[(size_t)b * HW + p] = lprior;

  if (block_sums) {   // this tile's sum over pixels and channels, fixed order
    const float ws = scae::wave_sum(lp_sum);
    const int wid = tid >> 6, nw = nthr >> 6;
    if ((tid & 63) == 0) s_red[wid] = ws;
    __syncthreads();
    if (tid == 0) {
      float s = 0.f;
      for (int w = 0; w < nw; ++w) s += s_red[w];
      block_sums[(size_t)b * n_tiles + tile] = s;
    }
  }
}
}  // namespace
}  // namespace scae_k1
