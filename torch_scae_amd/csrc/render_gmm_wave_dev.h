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

constexpr int kLogprobPadLow = 1;   // the likelihood forward's planes (stage_planes)
template <int C> struct TexelOf { static constexpr int TX = C == 1 ? 2 : (C <= 3 ? 4 : 8); };

__device__ __forceinline__ float ex2(float v) { return __builtin_amdgcn_exp2f(v); }
__device__ __forceinline__ float lg2(float v) { return __builtin_amdgcn_logf(v); }

// Components [k0, k0 + nk) of image b as padded planes of interleaved TX-float texels
// {channel 0 .. C - 1, alpha * alpha_scale, 0 ..}: one padded texel ROW per thread (one
// division per row); the caller synchronises.
// PL: zero texels to the left of / above the template (2 to the right / below).  2 lets a
// coordinate be clamped at -2, where the bilinear value AND its derivative are zero (the
// kernels that differentiate); 1 is enough for the value alone (clamp at -1: both taps of
// anything further out are the zero column) and takes the 24 planes of cfg-2 from 43 to
// 37.6 KB -- under the 40 KB that let four workgroups share a CU's LDS.
__host__ __device__ inline int plane_w(int tw, int PL) { return tw + PL + 2; }
__host__ __device__ inline int plane_elems(int th, int tw, int PL) {
  return (th + PL + 2) * (tw + PL + 2);
}
template <int C, int PL = 2>
__device__ __forceinline__ void stage_planes(float *s_pl, const scae_decoder_desc &d, int b, int k0,
                                             int nk, float alpha_scale, int tid, int nthr) {
  constexpr int TX = TexelOf<C>::TX;
  const int M = d.M, th = d.th, tw = d.tw, tsz = th * tw;
  const int psz = plane_elems(th, tw, PL), pw = plane_w(tw, PL), prow = th + PL + 2;
  const float *g_tmpl = d.templates + (size_t)tb(d, b) * M * C * tsz;
  const float inv_prow = 1.f / (float)prow;
  for (int r = tid; r < nk * prow; r += nthr) {
    const int kl = (int)(((float)r + 0.5f) * inv_prow), yp = r - kl * prow, y = yp - PL, k = k0 + kl;
    float *dst = s_pl + ((size_t)kl * psz + yp * pw) * TX;
    const bool in = y >= 0 && y < th;
    const float *ts = g_tmpl + (size_t)k * C * tsz + y * tw;
    const float *as = d.templates_alpha + (size_t)k * tsz + y * tw;
    for (int xp = 0; xp < pw; ++xp) {
      const int xx = xp - PL;
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
  constexpr int PL = kLogprobPadLow;
  const int psz = plane_elems(th, tw, PL), pw = plane_w(tw, PL);
  const Scalars sc = load_scalars(d);
  float *s_pl = smem;                              // M padded planes of TX-float texels
  float *s_coef = s_pl + (size_t)M * psz * TX;     // (M + KC) x 8: texel-space map, presence
  float *s_red = s_coef + (M + KC) * 8;            // 16

  // ---- stage: one padded texel row per thread ---------------------------------------
  {
    stage_planes<C, PL>(s_pl, d, b, 0, M, kLog2e, tid, nthr);
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

  const float *s_tap = s_pl + (size_t)(PL * pw + PL) * TX;   // tap (0, 0) of plane 0
  for (int k0 = 0; k0 < M; k0 += KC) {
    float uv[KC], pv[KC][C];
#pragma unroll
    for (int q = 0; q < KC; ++q) {
      const int k = k0 + q, kp = k < M ? k : M - 1;   // (wave-uniform)
      const float4 ca = *reinterpret_cast<const float4 *>(s_coef + k * 8);
      const float4 cb = *reinterpret_cast<const float4 *>(s_coef + k * 8 + 4);
      float ix = fmaf(ca.x, xn, fmaf(ca.y, yn, ca.z));
      float iy = fmaf(ca.w, xn, fmaf(cb.x, yn, cb.y));
      ix = fminf(fmaxf(ix, -(float)PL), txf);   // fmaxf(NaN, -PL) = -PL: everything outside
      iy = fminf(fmaxf(iy, -(float)PL), tyf);
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
#ifndef SCAE_CELL_ABL
#define SCAE_CELL_ABL 0   // 1: phase 1 only; 2: phase 2 without its pixel loop (timing ablations)
#endif
#ifndef SCAE_CELL_NT320
#define SCAE_CELL_NT320 0
#endif
#ifndef SCAE_CELL_REC_KB
#define SCAE_CELL_REC_KB 16   // LDS budget of a chunk's parked pixel records
#endif
#ifndef SCAE_CELL_ITEMS
#define SCAE_CELL_ITEMS 256
#endif
// ---------------------------------------------------------------------------------------
// Backward of the fused likelihood, cell-gather form.  One workgroup per (component k,
// image b) as in render_gmm.hip's render_bwd_kernel; what differs is how the texel
// gradients are collected.  A pixel's bilinear taps are the four corners of the texel CELL
// its sample position falls in, so the texel gradients of a component are, per cell, four
// moments of the pixel gradients inside it:
//     S0 = sum g,  S1 = sum g fx,  S2 = sum g fy,  S3 = sum g fx fy
//     corner (0,0) += S0 - S1 - S2 + S3,  (0,1) += S1 - S3,  (1,0) += S2 - S3,  (1,1) += S3
// and the pixels of a cell are, row by row, one interval of the inverse affine map.
//   phase 1  lane = pixel: responsibility, d/d(sample), pose / presence sums in registers;
//            the pixel's {g per plane, fx, fy} and its cell id parked in LDS (conflict free);
//   phase 2  lane = (cell, row slice): per row the pixel interval in closed form (+-1 pixel
//            of slack; membership is decided by the parked cell id, so round-off in the
//            inverse map cannot drop or duplicate a pixel), the moments summed in pixel order;
//   phase 3  lane = texel: its four cells' corner terms, fixed order.
// Every address has one writer and every sum a fixed order: bit-reproducible, no atomics,
// no cross-lane traffic in the gradient path.  The segmented DPP scatter this replaces
// spent ~250 of its 431 VALU instructions per (pixel, component) on the scan and the
// tap-by-tap read-add-write of run leaders; here each pixel is visited once more, by its cell.
template <int C> struct RecOf { static constexpr int RS = C == 1 ? 4 : ((C + 4) & ~1); };

__device__ __forceinline__ int fdiv(int n, float inv) { return (int)(((float)n + 0.5f) * inv); }

// Workgroup (component k, image b) as a device function (NTB threads, `smem`: its dynamic
// LDS): its own launch below, or a block range of the launch it shares with the capsule
// likelihood's backward (render_bwd_likelihood.hip).
template <int C, int NTB>
__device__ __forceinline__ void bwd_cell_body(
    const scae_decoder_desc &d, const float *__restrict__ x, const float *__restrict__ lse_post,
    const float *__restrict__ lse_prior, const float *__restrict__ g_logprob,
    const float *__restrict__ g_tile, int lp_tiles, int lp_ppb,
    float *__restrict__ g_templates, float *__restrict__ g_alpha_partial,
    float *__restrict__ g_pose, float *__restrict__ g_presence,
    float *__restrict__ g_bg_image, float *__restrict__ g_scalar_partial, int chunk_rows,
    int max_items, int item_budget, float *smem, int k, int b) {
  constexpr int TX = TexelOf<C>::TX, NV = C + 1, RS = RecOf<C>::RS, NM = 4 * NV;
  const int tid = threadIdx.x;
  const int M = d.M, K = M + 1, W = d.W, H = d.H, HW = H * W, tw = d.tw, th = d.th;
  const int tsz = th * tw, psz = pad_elems(th, tw), pw = pad_w(tw);
  // (only what this workgroup's branch needs of load_scalars: the softplus / sigmoid
  // chains cost ~150 instructions per thread, a fifth of a component's pixel loop)
  const bool has_scale = d.out_scale != nullptr;
  const float sigma = has_scale ? softplusf_(d.out_scale[0]) + 1e-4f : 1.f;
  const float inv_sigma = has_scale ? 1.f / sigma : 1.f, inv_var = inv_sigma * inv_sigma;
  const float knorm = (has_scale ? -logf(sigma) : 0.f) - scae::kHalfLog2Pi, hvar = 0.5f * inv_var;
  const float inv_wf = 1.f / (float)W, inv_hf = 1.f / (float)H;
  const float inv_ppb = __builtin_amdgcn_rcpf((float)lp_ppb);   // (quotients of small integers)
  const int chunk_px = chunk_rows * W;

  float *s_pl = smem;                                   // one padded plane of TX-float texels
  float *s_rec = s_pl + ((psz * TX + 3) & ~3);          // chunk_px x RS
  int *s_id = reinterpret_cast<int *>(s_rec + (((size_t)chunk_px * RS + 3) & ~(size_t)3));
  float *s_part = reinterpret_cast<float *>(s_id + ((chunk_px + 3) & ~3));   // max_items x NM
  float *s_tex = s_part + (size_t)max_items * NM;       // NV x tsz
  float *s_red = s_tex + ((NV * tsz + 3) & ~3);         // 8 x (NTB / 64) <= 64

  if (k == M) {   // background component: no texels, three scalar sums
    struct { float bg_ml, bg_val, inv_var; } sc = {softplusf_(d.bg_mixing_logit[0]),
                                                  d.bg_image ? 0.f : sigmoidf_(d.bg_value[0]), inv_var};
    float acc[3] = {0.f, 0.f, 0.f};   // bg_value, bg_mixing_logit, sigma
    for (int p = tid; p < HW; p += NTB) {
      const float sp = __expf(sc.bg_ml - lse_prior[(size_t)b * HW + p]);
      float gml = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const size_t o = ((size_t)b * C + c) * HW + p;
        const float gc = g_tile ? g_tile[b * lp_tiles + fdiv(p, inv_ppb)] : g_logprob[o];
        const float tv = d.bg_image ? d.bg_image[o] : sc.bg_val;
        const float diff = x[o] - tv;
        const float w = __expf(fmaf(diff * diff, -hvar, knorm) + sc.bg_ml - lse_post[o]);
        const float gtt = gc * w * diff * sc.inv_var;
        gml += gc * (w - sp);
        acc[2] += gc * w * (diff * diff * sc.inv_var - 1.f) * inv_sigma;
        if (d.bg_image) {
          if (g_bg_image) g_bg_image[o] = gtt;
        } else {
          acc[0] += gtt;
        }
      }
      acc[1] += gml;
    }
    scae::block_sum<3, NTB>(acc, s_red);
    if (tid == 0) {
      float *sp = g_scalar_partial + ((size_t)b * K + k) * 4;
      sp[0] = d.bg_image ? 0.f : acc[0] * sc.bg_val * (1.f - sc.bg_val);
      sp[1] = acc[1] * scae::softplus_grad(d.bg_mixing_logit[0]);
      sp[2] = 0.f;
      sp[3] = has_scale ? acc[2] * scae::softplus_grad(d.out_scale[0]) : 0.f;
    }
    return;
  }

  // A component whose presence is below log_safe's threshold has the mixing logit
  // -1e8 (math_ops.py:18-22): its responsibilities underflow to exactly 0 in fp32 and with
  // them every gradient of this workgroup -- zeros, without the pixel loop.  (Part capsules
  // that training has switched off cost nothing; on U[0,1) noise images that is all of them
  // after a few hundred steps)
  if (d.presence && d.presence[b * M + k] < scae::kLogSafeEps) {   // (workgroup-uniform)
    float *o_t = g_templates + ((size_t)b * M + k) * C * tsz;
    for (int e = tid; e < C * tsz; e += NTB) o_t[e] = 0.f;
    float *o_a = g_alpha_partial + ((size_t)b * M + k) * tsz;
    for (int e = tid; e < tsz; e += NTB) o_a[e] = 0.f;
    if (tid < 6) g_pose[((size_t)b * M + k) * 6 + tid] = 0.f;
    if (tid == 0) {
      if (g_presence) g_presence[b * M + k] = 0.f;
      float *sp = g_scalar_partial + ((size_t)b * K + k) * 4;
      sp[0] = sp[1] = sp[2] = sp[3] = 0.f;
    }
    return;
  }
  // ---- stage this component's padded, interleaved plane; clear the texel sums ----------
  {
    const float *ts = d.templates + ((size_t)tb(d, b) * M + k) * C * tsz;
    const float *as = d.templates_alpha + (size_t)k * tsz;
    const float inv_pw = __builtin_amdgcn_rcpf((float)pw);
    for (int e = tid; e < psz; e += NTB) {
      const int yp = fdiv(e, inv_pw), y = yp - 2, xx = e - yp * pw - 2;
      const bool in = y >= 0 && y < th && xx >= 0 && xx < tw;
#pragma unroll
      for (int c = 0; c < TX; ++c)
        s_pl[e * TX + c] = !in || c > C ? 0.f : (c < C ? ts[c * tsz + y * tw + xx] : as[y * tw + xx]);
    }
    for (int e = tid; e < NV * tsz; e += NTB) s_tex[e] = 0.f;
  }
  const float *pa = d.pose + ((size_t)b * M + k) * 6;
  const float pa6[6] = {pa[0], pa[1], pa[2], pa[3], pa[4], pa[5]};
  const float hx = 0.5f * (float)tw, hy = 0.5f * (float)th;
  const float A0 = hx * pa[0], A1 = hx * pa[1], A2 = hx * (pa[2] + 1.f) - 0.5f;
  const float A3 = hy * pa[3], A4 = hy * pa[4], A5 = hy * (pa[5] + 1.f) - 0.5f;
  const float lsp = d.presence ? log_safe(d.presence[b * M + k]) : 0.f;
  const float txf = (float)tw, tyf = (float)th, pwf = (float)pw;
  const float *s_tap = s_pl + (size_t)(2 * pw + 2) * TX;
  // the same map over pixel indices (j, i):  ix = ax j + bx i + c0x,  iy = ay j + by i + c0y
  const float ax = A0 * 2.f * inv_wf, bx = A1 * 2.f * inv_hf;
  const float c0x = fmaf(A0, inv_wf - 1.f, fmaf(A1, inv_hf - 1.f, A2));
  const float ay = A3 * 2.f * inv_wf, by = A4 * 2.f * inv_hf;
  const float c0y = fmaf(A3, inv_wf - 1.f, fmaf(A4, inv_hf - 1.f, A5));
  const float det = ax * by - bx * ay;
  // The inverse map only has to give a SUPERSET of a cell's pixels (membership is the
  // parked cell id).  Phase 1's positions and this affine model agree to ~1e-5 texels, so an
  // interval bound is off by 1e-5 / |slope| pixels: with slopes above 1e-3 a slack of 0.02
  // pixels covers it; flatter maps (and NaNs) take the whole row / all rows.
  const float span = fabsf(ax) + fabsf(ay) + fabsf(bx) + fabsf(by);
  const bool det_ok = fabsf(det) > 1e-3f * span && fabsf(det) < 1e30f;   // (false for NaN too)
  const float inv_det = det_ok ? __builtin_amdgcn_rcpf(det) : 0.f;
  const bool ax_ok = fabsf(ax) > 1e-3f && fabsf(ax) < 1e30f, ay_ok = fabsf(ay) > 1e-3f && fabsf(ay) < 1e30f;
  constexpr float kSlack = 0.02f;
  const float inv_ax = ax_ok ? __builtin_amdgcn_rcpf(ax) : 0.f, inv_ay = ay_ok ? __builtin_amdgcn_rcpf(ay) : 0.f;
  const float dix = -ay * inv_det, diy = ax * inv_det;   // d(row) per unit ix / iy
  // log2-domain constants of the two exponentials of a term
  const float hvar2 = kLog2e * hvar, knorm2 = kLog2e * knorm;
  __syncthreads();

  float acc[8];   // 6 pose sums, sum of d/d(mixing logit), sigma
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;

  for (int r0 = 0; r0 < H; r0 += chunk_rows) {
    const int r1 = min(H, r0 + chunk_rows), p0 = r0 * W, np = (r1 - r0) * W;
    // ---- phase 1: lane = pixel ------------------------------------------------------
    for (int pl = tid; pl < np; pl += NTB) {
      const int p = p0 + pl;
      const int pi = fdiv(p, inv_wf), pj = p - pi * W;
      const float xn = (float)(2 * pj + 1) * inv_wf - 1.f, yn = (float)(2 * pi + 1) * inv_hf - 1.f;
      // the sample position in the reference's own operation order (affine_grid, then
      // grid_sample's un-normalisation): d/d(position) jumps at cell boundaries, so a
      // pixel within round-off of one must land on the side the reference puts it
      float ix, iy;
      tex_pos(pa6, xn, yn, tw, th, ix, iy);
      ix = fminf(fmaxf(ix, -2.f), txf);
      iy = fminf(fmaxf(iy, -2.f), tyf);
      const float x0f = floorf(ix), y0f = floorf(iy), fx = ix - x0f, fy = iy - y0f;
      const int idx = (int)fmaf(y0f, pwf, x0f);
      const float *q0 = s_tap + (size_t)idx * TX, *q1 = q0 + pw * TX;
      float v[NV], vdx[NV], vdy[NV];
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        const float v00 = q0[e], v01 = q0[TX + e], v10 = q1[e], v11 = q1[TX + e];
        const float d0 = v01 - v00, d1 = v11 - v10;
        const float t0 = fmaf(fx, d0, v00), t1 = fmaf(fx, d1, v10);
        vdy[e] = t1 - t0;                    // d/diy
        v[e] = fmaf(fy, vdy[e], t0);
        vdx[e] = fmaf(fy, d1 - d0, d0);      // d/dix
      }
      const float mlv2 = kLog2e * (v[C] + lsp);
      const float sp = ex2(fmaf(-kLog2e, lse_prior[(size_t)b * HW + p], mlv2));
      const float gct = g_tile ? g_tile[b * lp_tiles + fdiv(p, inv_ppb)] : 0.f;
      float gtt[C], gml = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const size_t o = ((size_t)b * C + c) * HW + p;
        const float gc = g_tile ? gct : g_logprob[o];
        const float diff = x[o] - v[c];
        const float w = ex2(fmaf(diff * diff, -hvar2, fmaf(-kLog2e, lse_post[o], mlv2 + knorm2)));
        const float gw = gc * w;
        gtt[c] = gw * diff * inv_var;
        gml += gc * (w - sp);
        if (has_scale) acc[7] += gw * (diff * diff * inv_var - 1.f) * inv_sigma;
      }
      float gix = gml * vdx[C], giy = gml * vdy[C];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        gix = fmaf(gtt[c], vdx[c], gix);
        giy = fmaf(gtt[c], vdy[c], giy);
      }
      acc[0] = fmaf(gix, xn, acc[0]);
      acc[1] = fmaf(gix, yn, acc[1]);
      acc[2] += gix;
      acc[3] = fmaf(giy, xn, acc[3]);
      acc[4] = fmaf(giy, yn, acc[4]);
      acc[5] += giy;
      acc[6] += gml;
      float *rec = s_rec + (size_t)pl * RS;
      if (C == 1) {
        *reinterpret_cast<float4 *>(rec) = make_float4(gtt[0], gml, fx, fy);
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) rec[c] = gtt[c];
        rec[C] = gml, rec[C + 1] = fx, rec[C + 2] = fy;
      }
      s_id[pl] = idx;
    }
    __syncthreads();

#if SCAE_CELL_ABL != 1
    // ---- phase 2: lane = (cell, row slice) -------------------------------------------
    // cells the chunk's pixels can lie in: the image of the chunk rectangle's corners;
    // only cells with a corner inside the template matter (cx in [-1, tw - 1])
    int cxlo, cylo, ncx, ncy;
    {
      const float jx = ax * (float)(W - 1), jy = ay * (float)(W - 1);
      const float xa = fmaf(bx, (float)r0, c0x), xb = fmaf(bx, (float)(r1 - 1), c0x);
      const float ya = fmaf(by, (float)r0, c0y), yb = fmaf(by, (float)(r1 - 1), c0y);
      const float xmin = fminf(xa, xb) + fminf(jx, 0.f), xmax = fmaxf(xa, xb) + fmaxf(jx, 0.f);
      const float ymin = fminf(ya, yb) + fminf(jy, 0.f), ymax = fmaxf(ya, yb) + fmaxf(jy, 0.f);
      // (fmaxf / fminf drop NaNs: a NaN pose ends on an in-range box; its pixels were
      // clamped to cell -2 in phase 1 and match no cell id)
      cxlo = (int)fminf(fmaxf(floorf(xmin - 0.01f), -1.f), txf);
      cylo = (int)fminf(fmaxf(floorf(ymin - 0.01f), -1.f), tyf);
      const int cxhi = (int)fmaxf(fminf(floorf(xmax + 0.01f), txf - 1.f), -2.f);
      const int cyhi = (int)fmaxf(fminf(floorf(ymax + 0.01f), tyf - 1.f), -2.f);
      ncx = max(cxhi - cxlo + 1, 0), ncy = max(cyhi - cylo + 1, 0);
    }
    const int ncells = ncx * ncy;   // (workgroup-uniform)
    // A cell's pixels are split over P = S x G lanes: S row slices (rows i = ilo + s, step S)
    // times G segments of each row's interval -- as many as the item budget allows, so that
    // a pose that puts the whole image into a few cells (a collapsed scale: one cell, 1600
    // pixels) still spreads over the workgroup instead of serialising on a handful of lanes.
    int S = 1, G = 1;
    if (ncells > 0) {
      const int L = max(item_budget / ncells, 1);
      const int rows_cell = det_ok ? min(r1 - r0, (int)fminf(fabsf(dix) + fabsf(diy), 1e4f) + 2)
                                   : r1 - r0;
      S = min(min(L, rows_cell), 64);
      const float run = fminf(ax_ok ? fabsf(inv_ax) : 1e4f, ay_ok ? fabsf(inv_ay) : 1e4f);
      const int jspan = min(W, (int)fminf(run, 1e4f) + 2);   // pixels of a row inside one cell
      G = max(1, min(L / S, jspan >> 2));
      const int nparts = S * G, nitems = ncells * nparts;
      const float inv_nc = __builtin_amdgcn_rcpf((float)ncells), inv_ncx = __builtin_amdgcn_rcpf((float)ncx);
      const float inv_G = __builtin_amdgcn_rcpf((float)G);
      for (int item = tid; item < nitems; item += NTB) {
        const int part = fdiv(item, inv_nc), cell = item - part * ncells;
        // (segments fastest: neighbouring lanes read neighbouring records of one row; with
        // rows fastest their records sit W * 16 bytes apart -- the same LDS banks)
        const int sl = fdiv(part, inv_G), seg = part - sl * G;
        const int cyi = fdiv(cell, inv_ncx), cxi = cell - cyi * ncx;
        const float cxf = (float)(cxlo + cxi), cyf = (float)(cylo + cyi);
        const int myid = (int)fmaf(cyf, pwf, cxf);
        const float ux = cxf - c0x, uy = cyf - c0y;
        int ilo = r0, ihi = r1 - 1;
        if (det_ok) {   // rows that cross the cell's parallelogram
          const float i00 = (ax * uy - ay * ux) * inv_det;
          const float imin = i00 + fminf(dix, 0.f) + fminf(diy, 0.f);
          const float imax = i00 + fmaxf(dix, 0.f) + fmaxf(diy, 0.f);
          ilo = max(ilo, (int)fminf(fmaxf(ceilf(imin - kSlack), -1.f), (float)H));
          ihi = min(ihi, (int)fmaxf(fminf(floorf(imax + kSlack), (float)H), -1.f));
        }
        float m[NM];
#pragma unroll
        for (int q = 0; q < NM; ++q) m[q] = 0.f;
        for (int i = ilo + sl; i <= ihi; i += S) {
          // along the row  ix - cx = ax j - rx,  iy - cy = ay j - ry: the cell's pixels are
          // the j with both in [0, 1)
          const float rx = fmaf(-bx, (float)i, ux), ry = fmaf(-by, (float)i, uy);
          float lo = 0.f, hi = (float)(W - 1);
          if (ax_ok) {
            const float t0 = rx * inv_ax, t1 = t0 + inv_ax;
            lo = fmaxf(lo, ceilf(fminf(t0, t1) - kSlack));
            hi = fminf(hi, floorf(fmaxf(t0, t1) + kSlack));
          }
          if (ay_ok) {
            const float t0 = ry * inv_ay, t1 = t0 + inv_ay;
            lo = fmaxf(lo, ceilf(fminf(t0, t1) - kSlack));
            hi = fminf(hi, floorf(fmaxf(t0, t1) + kSlack));
          }
          if (!(lo <= hi)) continue;
          int jl = (int)lo, jh = (int)hi;
          if (G > 1) {   // this lane's segment of the interval
            const int len = fdiv(jh - jl + G, inv_G);
            jl += seg * len;
            jh = min(jh, jl + len - 1);
          }
          const int base = (i - r0) * W;
#if SCAE_CELL_ABL == 2
          if (jl > 10000)
#endif
          for (int pl = base + jl; pl <= base + jh; ++pl) {
            if (s_id[pl] != myid) continue;
            const float *rec = s_rec + (size_t)pl * RS;
            float g[NV], fx, fy;
            if (C == 1) {
              const float4 r4 = *reinterpret_cast<const float4 *>(rec);
              g[0] = r4.x, g[1] = r4.y, fx = r4.z, fy = r4.w;
            } else {
#pragma unroll
              for (int e = 0; e < NV; ++e) g[e] = rec[e];
              fx = rec[NV], fy = rec[NV + 1];
            }
            const float fxy = fx * fy;
#pragma unroll
            for (int e = 0; e < NV; ++e) {
              m[4 * e] += g[e];
              m[4 * e + 1] = fmaf(g[e], fx, m[4 * e + 1]);
              m[4 * e + 2] = fmaf(g[e], fy, m[4 * e + 2]);
              m[4 * e + 3] = fmaf(g[e], fxy, m[4 * e + 3]);
            }
          }
        }
        float *mp = s_part + (size_t)item * NM;   // [part][cell][NM]
#pragma unroll
        for (int q = 0; q < NM; q += 4)
          *reinterpret_cast<float4 *>(mp + q) = make_float4(m[q], m[q + 1], m[q + 2], m[q + 3]);
      }
    }
    __syncthreads();
    // many parts per cell (few, large cells): fold the parts four to one until a texel's
    // corner sums are short again (fixed order)
    int P = S * G;
    while (P > 6) {   // (workgroup-uniform; phase 3 walks the parts serially per texel)
      const int q = (P + 3) >> 2, stride = q * ncells * (NM / 4);   // parts [j q, j q + q), j = 0..3
      float4 *dst = reinterpret_cast<float4 *>(s_part);
      for (int t = tid; t < stride; t += NTB) {
        float4 a4 = dst[t];
#pragma unroll
        for (int j = 1; j < 4; ++j)
          if (t + j * stride < P * ncells * (NM / 4)) {
            const float4 b4 = dst[t + j * stride];
            a4 = make_float4(a4.x + b4.x, a4.y + b4.y, a4.z + b4.z, a4.w + b4.w);
          }
        dst[t] = a4;
      }
      __syncthreads();
      P = q;
    }

    // ---- phase 3: lane = texel: the corner terms of its four cells ----------------------
    if (ncells > 0) {
      const float inv_tw = __builtin_amdgcn_rcpf((float)tw);
      for (int e = tid; e < tsz; e += NTB) {
        const int ty = fdiv(e, inv_tw), tx = e - ty * tw;
        float g[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) g[q] = 0.f;
#pragma unroll
        for (int corner = 0; corner < 4; ++corner) {
          const int dy = corner >> 1, dx = corner & 1;
          const int cxi = tx - dx - cxlo, cyi = ty - dy - cylo;
          if (cxi < 0 || cxi >= ncx || cyi < 0 || cyi >= ncy) continue;
          const int cell = cyi * ncx + cxi;
          for (int sl = 0; sl < P; ++sl) {
            const float *mp = s_part + (size_t)(sl * ncells + cell) * NM;
#pragma unroll
            for (int q = 0; q < NV; ++q) {
              const float4 mm = *reinterpret_cast<const float4 *>(mp + 4 * q);
              g[q] += corner == 0 ? ((mm.x - mm.y) - mm.z) + mm.w
                                  : (corner == 1 ? mm.y - mm.w : (corner == 2 ? mm.z - mm.w : mm.w));
            }
          }
        }
#pragma unroll
        for (int q = 0; q < NV; ++q) s_tex[q * tsz + e] += g[q];
      }
    }
    __syncthreads();   // the next chunk overwrites the records
#endif
  }

  scae::block_sum<8, NTB>(acc, s_red);   // ends with __syncthreads()
  {
    float *o_t = g_templates + ((size_t)b * M + k) * C * tsz;
    for (int e = tid; e < C * tsz; e += NTB) o_t[e] = s_tex[e];
    float *o_a = g_alpha_partial + ((size_t)b * M + k) * tsz;
    for (int e = tid; e < tsz; e += NTB) o_a[e] = s_tex[C * tsz + e];
  }
  if (tid == 0) {
    float *gp = g_pose + ((size_t)b * M + k) * 6;
    // d ix / d a0 = hx xn, d ix / d a2 = hx, ...
    gp[0] = hx * acc[0], gp[1] = hx * acc[1], gp[2] = hx * acc[2];
    gp[3] = hy * acc[3], gp[4] = hy * acc[4], gp[5] = hy * acc[5];
    if (g_presence && d.presence)
      g_presence[b * M + k] = acc[6] * scae::log_safe_grad(d.presence[b * M + k]);
    float *sp = g_scalar_partial + ((size_t)b * K + k) * 4;
    sp[0] = sp[1] = sp[2] = 0.f;
    sp[3] = has_scale ? acc[7] * scae::softplus_grad(d.out_scale[0]) : 0.f;
  }
}

struct CellGeom {
  int chunk_rows, max_items, item_budget;
  size_t lds;
};
CellGeom cell_geom(const scae_decoder_desc *d) {
  CellGeom g = {0, 0, 0, 0};
  if (!d->templates_alpha || d->C < 1 || d->C > 4 || d->template_repeat > 1) return g;
  const int C = d->C, TX = C == 1 ? 2 : (C <= 3 ? 4 : 8), NV = C + 1;
  const int RS = C == 1 ? 4 : ((C + 4) & ~1);
  // the parked records of a chunk of rows stay below ~32 KB
  // (C = 1: 16 KB of records -- two chunks of rows at 40 x 40 -- keep five workgroups on a CU;
  // wider records take 32 KB: splitting a 32 x 32 image costs more than the occupancy returns)
  int rows = ((NV <= 2 ? SCAE_CELL_REC_KB : 2 * SCAE_CELL_REC_KB) * 1024 / 4) / ((RS + 1) * d->W);
  rows = rows < 1 ? 1 : (rows > d->H ? d->H : rows);
  const int cells = (d->tw + 1) * (d->th + 1);
  g.chunk_rows = rows;
  // (cell, row slice) items per workgroup: ~2 rounds of lanes when their moment partials
  // (16 B per plane and item) stay small beside the records
  g.item_budget = SCAE_CELL_ITEMS;
  g.max_items = cells > g.item_budget ? cells : g.item_budget;   // >= ncells * S
  const size_t chunk_px = (size_t)rows * d->W, tsz = (size_t)d->th * d->tw;
  const size_t floats = ((pad_elems(d->th, d->tw) * TX + 3) & ~3) + ((chunk_px * RS + 3) & ~(size_t)3) +
                        ((chunk_px + 3) & ~(size_t)3) + (size_t)g.max_items * 4 * NV +
                        ((NV * tsz + 3) & ~(size_t)3) + 64;
  g.lds = floats * sizeof(float) <= 64 * 1024 ? floats * sizeof(float) : 0;
  return g;
}

}  // namespace
}  // namespace scae_k1
