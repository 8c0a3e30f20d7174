// K1 -- template render + per-pixel Gaussian-mixture image likelihood for
// gfx950 (MI355X).  Replaces the ATen op cluster of the reference's
//   part_decoder.py:174-237   (affine_grid, 2x grid_sample, background,
//                              presence logits)
//   distributions.py:34-47    (mixture log_prob = logsumexp over components)
//   stacked_capsule_auto_encoder.py:220 (reconstruction log-likelihood)
//
// Design (HBM/VALU bound, no MFMA -- there is no contraction here):
//   * the compact decoder inputs of one image -- M templates (+alpha) of
//     th x tw texels, M poses, M presences -- are staged in LDS once per
//     workgroup; the (B,K,C,H,W) tensors of the reference never exist on the
//     fused path;
//   * forward: one lane owns (pixel, component-subset); the mixture
//     log-sum-exp is an online (max,sum) pair per lane, merged across the
//     KSPLIT lanes of a pixel with wavefront xor-shuffles;
//   * backward: one workgroup per (image, component), pixel-parallel: a lane
//     recomputes its pixel's responsibility and d/d(sample); the 6 pose
//     gradients / presence gradient are reduced with wave shuffles; the texel
//     gradients by a SEGMENTED SCATTER -- the pixels of an image row cross each
//     texel cell in one contiguous run, so a segmented suffix sum (DPP, within
//     16-lane rows) leaves each run's weighted sums in its first lane, which
//     adds them to padded accumulator planes private to its 16-lane row.  When
//     those planes do not fit LDS (C > 1) the round-1 form runs instead: pixel
//     gradients parked in LDS, then every (texel, row slice) GATHERS the pixels
//     of its inverse-affine footprint.  No atomics in either (LDS float adds
//     cost ~160 cycles per instruction under load, measured twice: 190 us of a
//     230 us kernel with one add per pixel, 63 of 99 us with one per run),
//     one writer per address, fixed summation order: bit-reproducible.
#include "common.h"
#include "render_gmm_dev.h"

namespace {

using namespace scae_k1;

constexpr int NT = 256;

struct Taps {
  int i00, i01, i10, i11;      // texel offsets inside one th*tw plane (clamped)
  float m00, m01, m10, m11;    // 1 if that texel is inside the template
  float fx, fy;                // fractional position
  float xn, yn;                // normalised output-pixel coordinates
};

// affine_grid (align_corners=False) + grid_sample's un-normalisation, as
// derived in SURVEY.md 8c: x_j = (2j+1)/W - 1; g = theta [x,y,1];
// ix = ((gx+1) w - 1)/2; bilinear taps, zero padding.
// normalised coordinate of output pixel index j on an axis of n pixels
__device__ __forceinline__ float norm_coord(int j, float inv_n) {
  return (float)(2 * j + 1) * inv_n - 1.f;
}
__device__ __forceinline__ void make_taps(const float *a, int p, int W, int H,
                                          int tw, int th, Taps &t) {
  const float inv_w = 1.f / (float)W;
  const int i = (int)(((float)p + 0.5f) * inv_w), j = p - i * W;  // exact for p < 2^22
  t.xn = norm_coord(j, inv_w);
  t.yn = norm_coord(i, 1.f / (float)H);
  float ix, iy;
  tex_pos(a, t.xn, t.yn, tw, th, ix, iy);
  float x0f = floorf(ix), y0f = floorf(iy);
  t.fx = ix - x0f;
  t.fy = iy - y0f;
  // keep the int conversion defined for wild poses; such taps are all outside
  x0f = fminf(fmaxf(x0f, -2.f), (float)(tw + 1));
  y0f = fminf(fmaxf(y0f, -2.f), (float)(th + 1));
  if (!(ix == ix)) x0f = -2.f;  // NaN pose -> everything outside
  if (!(iy == iy)) y0f = -2.f;
  const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
  const bool bx0 = x0 >= 0 && x0 < tw, bx1 = x1 >= 0 && x1 < tw;
  const bool by0 = y0 >= 0 && y0 < th, by1 = y1 >= 0 && y1 < th;
  const int cx0 = min(max(x0, 0), tw - 1), cx1 = min(max(x1, 0), tw - 1);
  const int cy0 = min(max(y0, 0), th - 1), cy1 = min(max(y1, 0), th - 1);
  t.i00 = cy0 * tw + cx0;
  t.i01 = cy0 * tw + cx1;
  t.i10 = cy1 * tw + cx0;
  t.i11 = cy1 * tw + cx1;
  t.m00 = (bx0 && by0) ? 1.f : 0.f;
  t.m01 = (bx1 && by0) ? 1.f : 0.f;
  t.m10 = (bx0 && by1) ? 1.f : 0.f;
  t.m11 = (bx1 && by1) ? 1.f : 0.f;
}

__device__ __forceinline__ float tap_value(const float *plane, const Taps &t) {
  const float wx1 = t.fx, wx0 = 1.f - t.fx, wy1 = t.fy, wy0 = 1.f - t.fy;
  return plane[t.i00] * (wx0 * wy0 * t.m00) + plane[t.i01] * (wx1 * wy0 * t.m01) +
         plane[t.i10] * (wx0 * wy1 * t.m10) + plane[t.i11] * (wx1 * wy1 * t.m11);
}

// value and d/dix, d/diy (grid_sampler_2d_backward's formulas).
__device__ __forceinline__ void tap_value_grad(const float *plane, const Taps &t,
                                               float &v, float &dx, float &dy) {
  const float v00 = plane[t.i00] * t.m00, v01 = plane[t.i01] * t.m01;
  const float v10 = plane[t.i10] * t.m10, v11 = plane[t.i11] * t.m11;
  const float wx1 = t.fx, wx0 = 1.f - t.fx, wy1 = t.fy, wy0 = 1.f - t.fy;
  v = v00 * (wx0 * wy0) + v01 * (wx1 * wy0) + v10 * (wx0 * wy1) + v11 * (wx1 * wy1);
  dx = (v01 - v00) * wy0 + (v11 - v10) * wy1;
  dy = (v10 - v00) * wx0 + (v11 - v01) * wx1;
}

struct PTaps {
  int base;      // offset of tap (y0, x0) inside a padded plane
  float fx, fy;  // fractional position
  float xn, yn;  // normalised output-pixel coordinates
};

__device__ __forceinline__ void make_ptaps(const float *a, int p, int W, int H, int tw, int th,
                                           PTaps &t) {
  const float inv_w = 1.f / (float)W;
  const int i = (int)(((float)p + 0.5f) * inv_w), j = p - i * W;  // exact for p < 2^22
  t.xn = norm_coord(j, inv_w);
  t.yn = norm_coord(i, 1.f / (float)H);
  float ix, iy;
  tex_pos(a, t.xn, t.yn, tw, th, ix, iy);
  ix = fminf(fmaxf(ix, -2.f), (float)tw);  // fmaxf(NaN, -2) = -2
  iy = fminf(fmaxf(iy, -2.f), (float)th);
  const float x0f = floorf(ix), y0f = floorf(iy);
  t.fx = ix - x0f;
  t.fy = iy - y0f;
  t.base = ((int)y0f + 2) * pad_w(tw) + (int)x0f + 2;
}

__device__ __forceinline__ float ptap_value(const float *plane, const PTaps &t, int pw) {
  const float wx1 = t.fx, wx0 = 1.f - t.fx, wy1 = t.fy, wy0 = 1.f - t.fy;
  const float *q = plane + t.base;
  return q[0] * (wx0 * wy0) + q[1] * (wx1 * wy0) + q[pw] * (wx0 * wy1) + q[pw + 1] * (wx1 * wy1);
}

// value and d/dix, d/diy (grid_sampler_2d_backward's formulas).
__device__ __forceinline__ void ptap_value_grad(const float *plane, const PTaps &t, int pw,
                                                float &v, float &dx, float &dy) {
  const float *q = plane + t.base;
  const float v00 = q[0], v01 = q[1], v10 = q[pw], v11 = q[pw + 1];
  const float wx1 = t.fx, wx0 = 1.f - t.fx, wy1 = t.fy, wy0 = 1.f - t.fy;
  v = v00 * (wx0 * wy0) + v01 * (wx1 * wy0) + v10 * (wx0 * wy1) + v11 * (wx1 * wy1);
  dx = (v01 - v00) * wy0 + (v11 - v10) * wy1;
  dy = (v10 - v00) * wx0 + (v11 - v01) * wx1;
}

// stage `n` dense (th x tw) planes from global memory as padded planes (NTHREADS
// threads of the workgroup; the caller synchronises afterwards)
template <int NTHREADS>
__device__ __forceinline__ void stage_padded(float *dst, const float *src, int n, int th, int tw) {
  const int psz = pad_elems(th, tw), pw = pad_w(tw);
  // zero everything (16-byte stores where dst allows), then one thread per texel
  // row: a single division per row
  const int total = n * psz, n4 = (((size_t)dst & 15) == 0) ? total >> 2 : 0;
  for (int i = threadIdx.x; i < n4; i += NTHREADS)
    reinterpret_cast<float4 *>(dst)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = 4 * n4 + threadIdx.x; i < total; i += NTHREADS) dst[i] = 0.f;
  __syncthreads();
  if (!src) return;
  if (n * th >= NTHREADS) {  // many planes: a thread per texel row (one division per row)
    for (int row = threadIdx.x; row < n * th; row += NTHREADS) {
      const int pl = row / th, y = row - pl * th;
      const float *sp = src + (size_t)row * tw;
      float *dp = dst + pl * psz + (y + 2) * pw + 2;
      for (int x = 0; x < tw; ++x) dp[x] = sp[x];
    }
  } else {  // few planes: a thread per texel
    const int tsz = th * tw;
    for (int i = threadIdx.x; i < n * tsz; i += NTHREADS) {
      const int pl = i / tsz, e = i - pl * tsz, y = e / tw, x = e - y * tw;
      dst[pl * psz + (y + 2) * pw + x + 2] = src[i];
    }
  }
}

// ---------------------------------------------------------------------------
// materialising forward: one workgroup per (component k, image b)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void render_fwd_kernel(scae_decoder_desc d,
                                                        float *__restrict__ tt,
                                                        float *__restrict__ ml) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int M = d.M, C = d.C, K = M + 1, HW = d.H * d.W, tsz = d.th * d.tw;
  const bool alpha_mode = d.templates_alpha != nullptr;
  const int Cm = alpha_mode ? 1 : C;
  const Scalars sc = load_scalars(d);
  float *tt_out = tt + (size_t)(b * K + k) * C * HW;
  float *ml_out = ml + (size_t)(b * K + k) * Cm * HW;

  if (k == M) {  // background component, part_decoder.py:189-195, :210-213
    for (int p = tid; p < HW; p += NT) {
      for (int c = 0; c < C; ++c) {
        const float v = d.bg_image ? d.bg_image[(size_t)(b * C + c) * HW + p] : sc.bg_val;
        tt_out[c * HW + p] = v;
        if (!alpha_mode) ml_out[c * HW + p] = v / sc.temperature;
      }
      if (alpha_mode) ml_out[p] = sc.bg_ml;
    }
    return;
  }

  const int psz = pad_elems(d.th, d.tw), pw = pad_w(d.tw);
  float *s_tmpl = smem;             // C padded planes
  float *s_alpha = smem + C * psz;  // one more (alpha mode)
  const float *g_tmpl = d.templates + (size_t)(tb(d, b) * M + k) * C * tsz;
  // (the alpha plane follows the template planes in LDS: one staging pass each)
  stage_padded<NT>(s_tmpl, g_tmpl, C, d.th, d.tw);
  if (alpha_mode)
    stage_padded<NT>(s_alpha, d.templates_alpha + (size_t)k * tsz, 1, d.th, d.tw);
  float a[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) a[i] = d.pose[(size_t)(b * M + k) * 6 + i];
  const float lsp = d.presence ? log_safe(d.presence[b * M + k]) : 0.f;  // :225-231
  __syncthreads();

  for (int p = tid; p < HW; p += NT) {
    PTaps t;
    make_ptaps(a, p, d.W, d.H, d.tw, d.th, t);
    for (int c = 0; c < C; ++c) {
      const float v = ptap_value(s_tmpl + c * psz, t, pw);
      tt_out[c * HW + p] = v;
      if (!alpha_mode) ml_out[c * HW + p] = v / sc.temperature + lsp;
    }
    if (alpha_mode) ml_out[p] = ptap_value(s_alpha, t, pw) + lsp;
  }
}

// ---------------------------------------------------------------------------
// fused forward: log_prob(x) straight from the compact inputs.
// grid (pixel tiles, B).  Lane layout: tid = pixel_local * KSPLIT + kgroup.
// ---------------------------------------------------------------------------
// PAD: the templates are staged as zero-padded planes (see above); chosen by the
// launcher while M * (C + 1) padded planes keep two workgroups per CU.
template <int C, int KSPLIT, bool PAD>
__global__ __launch_bounds__(NT) void logprob_fwd_kernel(
    scae_decoder_desc d, const float *__restrict__ x, float *__restrict__ log_prob,
    float *__restrict__ lse_post, float *__restrict__ lse_prior, int pix_per_block,
    float *__restrict__ block_sums) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float s_sum[NT / 64];
  float lp_sum = 0.f;  // this lane's share of sum_{c,p} log_prob (block_sums mode)
  const int b = blockIdx.y, tid = threadIdx.x;
  const int M = d.M, HW = d.H * d.W, tsz = d.th * d.tw;
  const bool alpha_mode = d.templates_alpha != nullptr;
  constexpr int CM = C;  // register arrays sized for the per-channel mode
  const Scalars sc = load_scalars(d);

  const int psz = PAD ? pad_elems(d.th, d.tw) : tsz, pw = pad_w(d.tw);  // plane stride in LDS
  float *s_tmpl = smem;                                // M*C planes
  float *s_alpha = s_tmpl + M * C * psz;               // M planes (alpha mode)
  float *s_pose = s_alpha + (alpha_mode ? M * psz : 0);  // M*6
  float *s_lsp = s_pose + M * 6;                       // M
  {
    const float *g_tmpl = d.templates + (size_t)tb(d, b) * M * C * tsz;
    if (PAD) {
      stage_padded<NT>(s_tmpl, g_tmpl, M * C, d.th, d.tw);
      if (alpha_mode) stage_padded<NT>(s_alpha, d.templates_alpha, M, d.th, d.tw);
    } else {
      for (int i = tid; i < M * C * tsz; i += NT) s_tmpl[i] = g_tmpl[i];
      if (alpha_mode)
        for (int i = tid; i < M * tsz; i += NT) s_alpha[i] = d.templates_alpha[i];
    }
    for (int i = tid; i < M * 6; i += NT) s_pose[i] = d.pose[(size_t)b * M * 6 + i];
    for (int i = tid; i < M; i += NT)
      s_lsp[i] = d.presence ? log_safe(d.presence[b * M + i]) : 0.f;
  }
  __syncthreads();

  const int kg = tid % KSPLIT;
  const int p_begin = blockIdx.x * pix_per_block;
  const int p_end = min(p_begin + pix_per_block, HW);
  // every lane of a wave runs the same number of iterations (shuffles below)
  const int n_iter = (pix_per_block + (NT / KSPLIT) - 1) / (NT / KSPLIT);
  for (int it = 0; it < n_iter; ++it) {
    const int p = p_begin + it * (NT / KSPLIT) + tid / KSPLIT;
    const bool live = p < p_end;
    const int pc = live ? p : p_begin;
    float xv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) xv[c] = x[(size_t)(b * C + c) * HW + pc];

    scae::Lse post[C], prior[CM];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      post[c].init();
      prior[c].init();
    }
    for (int k = kg; k < M; k += KSPLIT) {
      Taps t;
      PTaps pt;
      if (PAD)
        make_ptaps(s_pose + k * 6, pc, d.W, d.H, d.tw, d.th, pt);
      else
        make_taps(s_pose + k * 6, pc, d.W, d.H, d.tw, d.th, t);
      float mlv = 0.f;
      if (alpha_mode) {
        mlv = (PAD ? ptap_value(s_alpha + k * psz, pt, pw) : tap_value(s_alpha + k * psz, t)) +
              s_lsp[k];
        prior[0].add(mlv);
      }
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float v = PAD ? ptap_value(s_tmpl + (k * C + c) * psz, pt, pw)
                            : tap_value(s_tmpl + (k * C + c) * psz, t);
        if (!alpha_mode) {
          mlv = v / sc.temperature + s_lsp[k];
          prior[c].add(mlv);
        }
        const float diff = xv[c] - v;
        const float lp = -(diff * diff) * (0.5f * sc.inv_var) - sc.log_sigma - scae::kHalfLog2Pi;
        post[c].add(lp + mlv);
      }
    }
    if (kg == 0) {  // background component (k = M)
      float mlv = sc.bg_ml;
      if (alpha_mode) prior[0].add(mlv);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float v = d.bg_image ? d.bg_image[(size_t)(b * C + c) * HW + pc] : sc.bg_val;
        if (!alpha_mode) {
          mlv = v / sc.temperature;
          prior[c].add(mlv);
        }
        const float diff = xv[c] - v;
        const float lp = -(diff * diff) * (0.5f * sc.inv_var) - sc.log_sigma - scae::kHalfLog2Pi;
        post[c].add(lp + mlv);
      }
    }
    // merge the KSPLIT (<= 4) partial (max,sum) pairs of this pixel: the lanes of a pixel
    // are one quad, so the exchanges are DPP quad permutes (no LDS round trip)
    static_assert(KSPLIT == 1 || KSPLIT == 2 || KSPLIT == 4, "KSPLIT lanes within a quad");
#pragma unroll
    for (int off = 1; off < KSPLIT; off <<= 1) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        if (off == 1) {
          post[c].merge(scae::xor1_f(post[c].m), scae::xor1_f(post[c].s));
          if (c == 0 || !alpha_mode) prior[c].merge(scae::xor1_f(prior[c].m), scae::xor1_f(prior[c].s));
        } else {
          post[c].merge(scae::xor2_f(post[c].m), scae::xor2_f(post[c].s));
          if (c == 0 || !alpha_mode) prior[c].merge(scae::xor2_f(prior[c].m), scae::xor2_f(prior[c].s));
        }
      }
    }
    if (live && kg == 0) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float lpost = post[c].value();
        const float lprior = prior[alpha_mode ? 0 : c].value();
        const size_t o = (size_t)(b * C + c) * HW + p;
        if (log_prob) log_prob[o] = lpost - lprior;
        lp_sum += lpost - lprior;
        lse_post[o] = lpost;
        if (!alpha_mode) lse_prior[o] = lprior;
      }
      if (alpha_mode) lse_prior[(size_t)b * HW + p] = prior[0].value();
    }
  }
  if (block_sums) {  // sum over this workgroup's pixels and channels, fixed order
    float a[1] = {lp_sum};
    scae::block_sum<1, NT>(a, s_sum);
    if (tid == 0) block_sums[(size_t)b * gridDim.x + blockIdx.x] = a[0];
  }
}

// ---------------------------------------------------------------------------
// backward: one workgroup per (component k, image b).
//   FUSED: incoming g_logprob, mixture responsibilities recomputed from the
//          saved per-pixel log-sum-exps;
//   else : incoming g_tt / g_ml of the materialised tensors.
// ---------------------------------------------------------------------------
// pixel-index footprint [lo, hi] (clipped to [0, n-1]) of  centre +- half
__device__ __forceinline__ void clip_range(float centre, float half, int n, int &lo, int &hi) {
  // a pixel outside the exact footprint has weight <= 0; the slack only has to
  // cover fp32 error of the inverse map (one more pixel is added by truncation)
  const float lo_f = centre - half - 0.05f, hi_f = centre + half + 0.05f;
  if (!(lo_f == lo_f) || !(hi_f == hi_f)) {  // NaN: degenerate pose, take everything
    lo = 0;
    hi = n - 1;
    return;
  }
  lo = lo_f <= 0.f ? 0 : (lo_f >= (float)n ? n : (int)lo_f);
  hi = hi_f >= (float)(n - 1) ? n - 1 : (hi_f < 0.f ? -1 : (int)hi_f);
}

#ifndef SCAE_K1_SLICES
#define SCAE_K1_SLICES 8
#endif
constexpr int SLICES = SCAE_K1_SLICES;  // row slices per texel in the backward gather

// The texel-gather form of the backward (round 1): phase 1 parks d/d(sample) per pixel in
// LDS, phase 2 gives every (texel, row slice) a lane that collects the bilinear-weighted
// pixel gradients inside the texel's inverse-affine footprint.  Still the faster form when
// the per-lane-group accumulator planes of the scatter form below do not fit LDS beside
// enough workgroups (C > 1: CIFAR 202 vs 309 us).
template <int C, bool FUSED>
__global__ __launch_bounds__(NT) void render_bwd_gather_kernel(
    scae_decoder_desc d, const float *__restrict__ x, const float *__restrict__ lse_post,
    const float *__restrict__ lse_prior, const float *__restrict__ g_logprob,
    const float *__restrict__ G_tt, const float *__restrict__ G_ml,
    float *__restrict__ g_templates, float *__restrict__ g_alpha_partial,
    float *__restrict__ g_pose, float *__restrict__ g_presence,
    float *__restrict__ g_bg_image, float *__restrict__ g_scalar_partial,
    int rows_per_chunk, const float *__restrict__ g_tile, int lp_tiles, int lp_ppb) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int M = d.M, K = M + 1, W = d.W, H = d.H, HW = H * W, tw = d.tw, th = d.th;
  const int tsz = th * tw;
  const bool alpha_mode = d.templates_alpha != nullptr;
  const int planes = C + (alpha_mode ? 1 : 0);  // gradient planes gathered per texel
  const Scalars sc = load_scalars(d);
  const float inv_T = 1.f / sc.temperature;
  const float inv_w = 1.f / (float)W, inv_h = 1.f / (float)H;

  const int psz = pad_elems(th, tw), pw = pad_w(tw);
  float *s_tmpl = smem;                        // C padded planes
  float *s_alpha = s_tmpl + C * psz;           // one padded plane
  float *s_acc = s_alpha + psz;                // SLICES * (C+1) * tsz
  float *s_red = s_acc + SLICES * (C + 1) * tsz;  // 11 * (NT/64)
  float *s_g = s_red + 11 * (NT / 64);         // (C+1) * npc: per-pixel grads, [pixel][C+1]
  const bool g_pair_ok = ((size_t)s_g & 7) == 0;  // 8-byte reads of a pixel's pair (C == 1)

  const bool is_bg = (k == M);
  float a[6] = {0, 0, 0, 0, 0, 0};
  float lsp = 0.f;
  if (!is_bg) {
    const float *g_tmpl = d.templates + (size_t)(tb(d, b) * M + k) * C * tsz;
    stage_padded<NT>(s_tmpl, g_tmpl, C, th, tw);
    stage_padded<NT>(s_alpha, alpha_mode ? d.templates_alpha + (size_t)k * tsz : nullptr, 1, th,
                     tw);
    for (int i = tid; i < SLICES * (C + 1) * tsz; i += NT) s_acc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) a[i] = d.pose[(size_t)(b * M + k) * 6 + i];
    lsp = d.presence ? log_safe(d.presence[b * M + k]) : 0.f;
  }
  __syncthreads();

  // affine map pixel indices (j, i) -> template position:
  //   ix = ax*j + bx*i + cx,  iy = ay*j + by*i + cy   (only for the footprints)
  const float sx = 0.5f * tw, sy = 0.5f * th;
  const float ax = sx * a[0] * 2.f * inv_w, bx = sx * a[1] * 2.f * inv_h;
  const float cx = sx * (a[0] * (inv_w - 1.f) + a[1] * (inv_h - 1.f) + a[2] + 1.f) - 0.5f;
  const float ay = sy * a[3] * 2.f * inv_w, by = sy * a[4] * 2.f * inv_h;
  const float cy = sy * (a[3] * (inv_w - 1.f) + a[4] * (inv_h - 1.f) + a[5] + 1.f) - 0.5f;
  const float det = ax * by - bx * ay;
  const float inv_det = 1.f / det;  // inf / NaN for degenerate poses -> full range
  const float inv_ax = 1.f / ax, inv_ay = 1.f / ay;
  // phase-2 work items are (texel, row slice): slice s takes footprint rows
  // i_lo + s, i_lo + s + SLICES, ... so that magnified templates (few active
  // texels with large footprints) still spread over all four waves
  constexpr int slices = SLICES;

  // accumulators: 6 pose grads, d/d log_safe(presence), bg_value, bg_ml,
  // temperature, sigma
  float acc[11];
#pragma unroll
  for (int i = 0; i < 11; ++i) acc[i] = 0.f;

  for (int r0 = 0; r0 < H; r0 += rows_per_chunk) {
    const int r1 = min(H, r0 + rows_per_chunk);
    const int p0 = r0 * W, np = (r1 - r0) * W;

    // ---- phase 1: pixel-parallel --------------------------------------
    for (int pl = tid; pl < np; pl += NT) {
      const int p = p0 + pl;
      PTaps t;
      float tv[C], tdx[C], tdy[C];
      float av = 0.f, adx = 0.f, ady = 0.f;
      if (!is_bg) {
        make_ptaps(a, p, W, H, tw, th, t);
#pragma unroll
        for (int c = 0; c < C; ++c)
          ptap_value_grad(s_tmpl + c * psz, t, pw, tv[c], tdx[c], tdy[c]);
        if (alpha_mode) ptap_value_grad(s_alpha, t, pw, av, adx, ady);
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c)
          tv[c] = d.bg_image ? d.bg_image[(size_t)(b * C + c) * HW + p] : sc.bg_val;
      }

      float gtt[C];
      float gml_alpha = 0.f;  // alpha mode: grad wrt the (single-channel) logit
      if (FUSED) {
        float mlv = 0.f, sp = 0.f;
        if (alpha_mode) {
          mlv = (is_bg ? sc.bg_ml : av + lsp);
          sp = __expf(mlv - lse_prior[(size_t)b * HW + p]);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const size_t o = (size_t)(b * C + c) * HW + p;
          // per-pixel gradient, or the gradient of the log-prob tile sum the pixel is in
          const float gc = g_tile ? g_tile[b * lp_tiles + p / lp_ppb] : g_logprob[o];
          if (!alpha_mode) {
            mlv = tv[c] / sc.temperature + lsp;
            sp = __expf(mlv - lse_prior[o]);
          }
          const float diff = x[o] - tv[c];
          const float lp =
              -(diff * diff) * (0.5f * sc.inv_var) - sc.log_sigma - scae::kHalfLog2Pi;
          const float w = __expf(lp + mlv - lse_post[o]);
          gtt[c] = gc * w * diff * sc.inv_var;
          const float gml = gc * (w - sp);
          acc[10] += gc * w * (diff * diff * sc.inv_var - 1.f) / sc.sigma;
          if (alpha_mode) {
            gml_alpha += gml;
          } else {
            gtt[c] += gml * inv_T;
            acc[9] += -gml * tv[c] * inv_T * inv_T;
            acc[6] += gml;
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) {
          gtt[c] = G_tt ? G_tt[((size_t)(b * K + k) * C + c) * HW + p] : 0.f;
          if (!alpha_mode && G_ml) {
            const float gml = G_ml[((size_t)(b * K + k) * C + c) * HW + p];
            gtt[c] += gml * inv_T;
            acc[9] += -gml * tv[c] * inv_T * inv_T;
            acc[6] += gml;
          }
        }
        if (alpha_mode && G_ml) gml_alpha = G_ml[(size_t)(b * K + k) * HW + p];
      }
      if (alpha_mode) acc[6] += gml_alpha;

      if (is_bg) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
          if (d.bg_image) {
            if (g_bg_image) g_bg_image[(size_t)(b * C + c) * HW + p] = gtt[c];
          } else {
            acc[7] += gtt[c];
          }
        }
        acc[8] += gml_alpha;
      } else {
        float gix = gml_alpha * adx, giy = gml_alpha * ady;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          gix += gtt[c] * tdx[c];
          giy += gtt[c] * tdy[c];
          s_g[pl * (C + 1) + c] = gtt[c];
        }
        if (alpha_mode) s_g[pl * (C + 1) + C] = gml_alpha;
        gix *= sx;
        giy *= sy;
        acc[0] += gix * t.xn;
        acc[1] += gix * t.yn;
        acc[2] += gix;
        acc[3] += giy * t.xn;
        acc[4] += giy * t.yn;
        acc[5] += giy;
      }
    }
    if (is_bg) continue;  // wave-uniform: the background has no texels
    __syncthreads();

    // ---- phase 2: texel-parallel gather ---------------------------------
    // only texels the chunk's pixels can reach get a lane: their range is the
    // image of the chunk rectangle's corners under the affine map, +-1 texel
    int tx0 = 0, tx1 = tw - 1, ty0 = 0, ty1 = th - 1;
    {
      const float jx = ax * (float)(W - 1), iy0 = (float)r0, iy1 = (float)(r1 - 1);
      const float xa = bx * iy0 + cx, xb = bx * iy1 + cx;
      const float xmin = fminf(xa, xb) + fminf(jx, 0.f), xmax = fmaxf(xa, xb) + fmaxf(jx, 0.f);
      const float jy = ay * (float)(W - 1);
      const float ya = by * iy0 + cy, yb = by * iy1 + cy;
      const float ymin = fminf(ya, yb) + fminf(jy, 0.f), ymax = fmaxf(ya, yb) + fmaxf(jy, 0.f);
      if (xmin == xmin && xmax == xmax && ymin == ymin && ymax == ymax) {  // no NaN
        tx0 = (int)fminf(fmaxf(floorf(xmin - 0.01f), 0.f), (float)tw);
        tx1 = (int)fmaxf(fminf(ceilf(xmax + 0.01f), (float)(tw - 1)), -1.f);
        ty0 = (int)fminf(fmaxf(floorf(ymin - 0.01f), 0.f), (float)th);
        ty1 = (int)fmaxf(fminf(ceilf(ymax + 0.01f), (float)(th - 1)), -1.f);
      }
    }
    const int atw = max(tx1 - tx0 + 1, 0), ath = max(ty1 - ty0 + 1, 0), atsz = atw * ath;
    for (int item = tid; item < slices * atsz; item += NT) {
      const int slice = item / atsz, ea = item - slice * atsz;
      const int ty = ty0 + ea / atw, tx = tx0 + (ea - (ea / atw) * atw);
      const int e = ty * tw + tx;
      // pixels (j, i) whose sample position lies within +-1 texel of (tx, ty)
      const float u = (float)tx - cx, v = (float)ty - cy;
      int j_lo, j_hi, i_lo, i_hi;
      clip_range((by * u - bx * v) * inv_det, (fabsf(by) + fabsf(bx)) * fabsf(inv_det), W,
                 j_lo, j_hi);
      clip_range((ax * v - ay * u) * inv_det, (fabsf(ay) + fabsf(ax)) * fabsf(inv_det), H,
                 i_lo, i_hi);
      i_lo = max(i_lo, r0);
      i_hi = min(i_hi, r1 - 1);
      float gsum[C + 1];
#pragma unroll
      for (int c = 0; c <= C; ++c) gsum[c] = 0.f;
      for (int i = i_lo + slice; i <= i_hi; i += slices) {
        // along this pixel row  ix - tx = ax*j + rx,  iy - ty = ay*j + ry;
        // the texel's support is the j-interval where both are inside (-1, 1)
        const float rx = fmaf(bx, (float)i, cx) - (float)tx;
        const float ry = fmaf(by, (float)i, cy) - (float)ty;
        float lo = (float)j_lo, hi = (float)j_hi;
        if (fabsf(ax) > 1e-12f) {
          const float c0 = (-1.f - rx) * inv_ax, c1 = (1.f - rx) * inv_ax;
          lo = fmaxf(lo, fminf(c0, c1) - 0.05f);
          hi = fminf(hi, fmaxf(c0, c1) + 0.05f);
        } else if (!(fabsf(rx) < 1.f)) {
          continue;
        }
        if (fabsf(ay) > 1e-12f) {
          const float c0 = (-1.f - ry) * inv_ay, c1 = (1.f - ry) * inv_ay;
          lo = fmaxf(lo, fminf(c0, c1) - 0.05f);
          hi = fminf(hi, fmaxf(c0, c1) + 0.05f);
        } else if (!(fabsf(ry) < 1.f)) {
          continue;
        }
        // lo / hi are clamped into [j_lo, j_hi] (fmaxf / fminf drop NaNs)
        const int jl = (int)ceilf(lo), jh = (int)floorf(hi);
        // per-pixel gradients are interleaved ([pixel][C + 1]): one LDS read per pair;
        // the sample offsets advance by (ax, ay) per pixel instead of being re-derived
        const float *gp = s_g + ((i - r0) * W + jl) * (C + 1);
        float px = fmaf(ax, (float)jl, rx), py = fmaf(ay, (float)jl, ry);
#pragma unroll 4
        for (int j = jl; j <= jh; ++j) {
          const float wgt = fmaxf(1.f - fabsf(px), 0.f) * fmaxf(1.f - fabsf(py), 0.f);
          if (C == 1 && g_pair_ok) {
            const float2 g2 = *reinterpret_cast<const float2 *>(gp);
            gsum[0] = fmaf(wgt, g2.x, gsum[0]);
            if (planes > 1) gsum[1] = fmaf(wgt, g2.y, gsum[1]);
          } else {
#pragma unroll
            for (int c = 0; c <= C; ++c)
              if (c < planes) gsum[c] = fmaf(wgt, gp[c], gsum[c]);
          }
          gp += C + 1;
          px += ax;
          py += ay;
        }
      }
#pragma unroll
      for (int c = 0; c <= C; ++c)
        if (c < planes) s_acc[(slice * (C + 1) + c) * tsz + e] += gsum[c];  // sole owner
    }
    __syncthreads();
  }

  scae::block_sum<11, NT>(acc, s_red);  // ends with __syncthreads()

  if (!is_bg) {
    float *o_t = g_templates + (size_t)(b * M + k) * C * tsz;
    for (int i = tid; i < C * tsz; i += NT) {
      const int c = i / tsz, e = i - c * tsz;
      float acc = 0.f;
#pragma unroll
      for (int sl = 0; sl < SLICES; ++sl) acc += s_acc[(sl * (C + 1) + c) * tsz + e];
      o_t[i] = acc;
    }
    if (alpha_mode) {
      float *o_a = g_alpha_partial + (size_t)(b * M + k) * tsz;
      for (int e = tid; e < tsz; e += NT)
      {
        float acc = 0.f;
#pragma unroll
        for (int sl = 0; sl < SLICES; ++sl) acc += s_acc[(sl * (C + 1) + C) * tsz + e];
        o_a[e] = acc;
      }
    }
  }
  if (tid == 0) {
    float *sp = g_scalar_partial + (size_t)(b * K + k) * 4;
    sp[0] = sp[1] = sp[2] = sp[3] = 0.f;
    if (!is_bg) {
#pragma unroll
      for (int i = 0; i < 6; ++i) g_pose[(size_t)(b * M + k) * 6 + i] = acc[i];
      if (g_presence && d.presence)
        g_presence[b * M + k] = acc[6] * scae::log_safe_grad(d.presence[b * M + k]);
    } else {
      if (!d.bg_image) {
        const float s = sc.bg_val;
        sp[0] = acc[7] * s * (1.f - s);
      }
      if (alpha_mode) sp[1] = acc[8] * scae::softplus_grad(d.bg_mixing_logit[0]);
    }
    if (!alpha_mode) sp[2] = acc[9] * scae::softplus_grad(d.temperature_logit[0] + .5f);
    if (FUSED && d.out_scale) sp[3] = acc[10] * scae::softplus_grad(d.out_scale[0]);
  }
}


template <int C, bool FUSED>
__global__ __launch_bounds__(NT) void render_bwd_kernel(
    scae_decoder_desc d, const float *__restrict__ x, const float *__restrict__ lse_post,
    const float *__restrict__ lse_prior, const float *__restrict__ g_logprob,
    const float *__restrict__ G_tt, const float *__restrict__ G_ml,
    float *__restrict__ g_templates, float *__restrict__ g_alpha_partial,
    float *__restrict__ g_pose, float *__restrict__ g_presence,
    float *__restrict__ g_bg_image, float *__restrict__ g_scalar_partial,
    int rows_per_chunk, const float *__restrict__ g_tile, int lp_tiles, int lp_ppb) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int M = d.M, K = M + 1, W = d.W, H = d.H, HW = H * W, tw = d.tw, th = d.th;
  const int tsz = th * tw;
  const bool alpha_mode = d.templates_alpha != nullptr;
  const int planes = C + (alpha_mode ? 1 : 0);  // gradient planes gathered per texel
  const Scalars sc = load_scalars(d);
  const float inv_T = 1.f / sc.temperature;
  const int psz = pad_elems(th, tw), pw = pad_w(tw);
  constexpr int NWV = NT / 16;   // accumulator plane sets: one per 16-lane DPP row
  float *s_tmpl = smem;                        // C padded planes
  float *s_alpha = s_tmpl + C * psz;           // one padded plane
  float *s_acc = s_alpha + psz;                // NWV * (C+1) padded planes: per-wave sums
  float *s_red = s_acc + NWV * (C + 1) * psz;  // 11 * (NT/64)

  const bool is_bg = (k == M);
  float a[6] = {0, 0, 0, 0, 0, 0};
  float lsp = 0.f;
  if (!is_bg) {
    const float *g_tmpl = d.templates + (size_t)(tb(d, b) * M + k) * C * tsz;
    stage_padded<NT>(s_tmpl, g_tmpl, C, th, tw);
    stage_padded<NT>(s_alpha, alpha_mode ? d.templates_alpha + (size_t)k * tsz : nullptr, 1, th,
                     tw);
    for (int i = tid; i < NWV * (C + 1) * psz; i += NT) s_acc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) a[i] = d.pose[(size_t)(b * M + k) * 6 + i];
    lsp = d.presence ? log_safe(d.presence[b * M + k]) : 0.f;
  }
  __syncthreads();
  const float sx = 0.5f * tw, sy = 0.5f * th;
  // integer quotients by run-time divisors through the reciprocal (an integer division
  // is ~25 instructions; the kernel is VALU-issue bound)
  const float inv_wf = 1.f / (float)W, inv_ppb = 1.f / (float)lp_ppb;
  auto fdiv = [](int n, float inv) { return (int)(((float)n + 0.5f) * inv); };
  const int lane = tid & 63;
  float *wacc = s_acc + (tid >> 4) * (C + 1) * psz;   // this lane group's planes

  // accumulators: 6 pose grads, d/d log_safe(presence), bg_value, bg_ml,
  // temperature, sigma
  float acc[11];
#pragma unroll
  for (int i = 0; i < 11; ++i) acc[i] = 0.f;
  (void)rows_per_chunk;

  for (int pl0 = 0; pl0 < HW; pl0 += NT) {   // (workgroup-uniform trip count: DPP inside)
    const int p = pl0 + tid;
    const bool live = p < HW;
    PTaps t;
    t.base = 0, t.fx = t.fy = t.xn = t.yn = 0.f;
    float gtt[C];
    float gml_alpha = 0.f;  // alpha mode: grad wrt the (single-channel) logit
#pragma unroll
    for (int c = 0; c < C; ++c) gtt[c] = 0.f;
    if (live) {
      float tv[C], tdx[C], tdy[C];
      float av = 0.f, adx = 0.f, ady = 0.f;
      if (!is_bg) {
        make_ptaps(a, p, W, H, tw, th, t);
#pragma unroll
        for (int c = 0; c < C; ++c)
          ptap_value_grad(s_tmpl + c * psz, t, pw, tv[c], tdx[c], tdy[c]);
        if (alpha_mode) ptap_value_grad(s_alpha, t, pw, av, adx, ady);
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c)
          tv[c] = d.bg_image ? d.bg_image[(size_t)(b * C + c) * HW + p] : sc.bg_val;
      }
      if (FUSED) {
        float mlv = 0.f, sp = 0.f;
        if (alpha_mode) {
          mlv = (is_bg ? sc.bg_ml : av + lsp);
          sp = __expf(mlv - lse_prior[(size_t)b * HW + p]);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const size_t o = (size_t)(b * C + c) * HW + p;
          // per-pixel gradient, or the gradient of the log-prob tile sum the pixel is in
          const float gc = g_tile ? g_tile[b * lp_tiles + fdiv(p, inv_ppb)] : g_logprob[o];
          if (!alpha_mode) {
            mlv = tv[c] / sc.temperature + lsp;
            sp = __expf(mlv - lse_prior[o]);
          }
          const float diff = x[o] - tv[c];
          const float lp =
              -(diff * diff) * (0.5f * sc.inv_var) - sc.log_sigma - scae::kHalfLog2Pi;
          const float w = __expf(lp + mlv - lse_post[o]);
          gtt[c] = gc * w * diff * sc.inv_var;
          const float gml = gc * (w - sp);
          acc[10] += gc * w * (diff * diff * sc.inv_var - 1.f) / sc.sigma;
          if (alpha_mode) {
            gml_alpha += gml;
          } else {
            gtt[c] += gml * inv_T;
            acc[9] += -gml * tv[c] * inv_T * inv_T;
            acc[6] += gml;
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) {
          gtt[c] = G_tt ? G_tt[((size_t)(b * K + k) * C + c) * HW + p] : 0.f;
          if (!alpha_mode && G_ml) {
            const float gml = G_ml[((size_t)(b * K + k) * C + c) * HW + p];
            gtt[c] += gml * inv_T;
            acc[9] += -gml * tv[c] * inv_T * inv_T;
            acc[6] += gml;
          }
        }
        if (alpha_mode && G_ml) gml_alpha = G_ml[(size_t)(b * K + k) * HW + p];
      }
      if (alpha_mode) acc[6] += gml_alpha;

      if (is_bg) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
          if (d.bg_image) {
            if (g_bg_image) g_bg_image[(size_t)(b * C + c) * HW + p] = gtt[c];
          } else {
            acc[7] += gtt[c];
          }
        }
        acc[8] += gml_alpha;
      } else {
        float gix = gml_alpha * adx, giy = gml_alpha * ady;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          gix += gtt[c] * tdx[c];
          giy += gtt[c] * tdy[c];
        }
        gix *= sx;
        giy *= sy;
        acc[0] += gix * t.xn;
        acc[1] += gix * t.yn;
        acc[2] += gix;
        acc[3] += giy * t.xn;
        acc[4] += giy * t.yn;
        acc[5] += giy;
      }
    }
    if (is_bg) continue;  // workgroup-uniform: the background has no texels

    // ---- texel gradients: segmented scatter ---------------------------------------
    // The pixels of an image row cross every texel cell in one contiguous run, so the
    // lanes (consecutive pixels) that share a cell -- and with it their four bilinear
    // taps -- are neighbours.  Within each 16-lane DPP row a segmented suffix sum over the
    // (image row, cell) runs (row_shl 1, 2, 4, 8: VALU-speed lane exchanges) leaves a
    // run's weighted gradient sums in its first lane, which adds them to padded
    // accumulator planes private to its lane group (out-of-template taps land in the
    // padding, like the reads); the groups' planes are summed in a fixed order at the end:
    // no atomics, one writer per address, bit-reproducible.  (The gather this replaces -- every texel
    // collecting the pixels of its inverse-affine footprint -- spent ~200 instructions of
    // interval set-up per (texel, row-slice) item: 46 of the kernel's 82 us.)
    const int prow = fdiv(p, inv_wf);   // (exact for p < 2^22)
    const int key = live ? (prow << 16) | t.base : -1 - lane;   // (planes: < 2^16 floats)
    const float wx1 = t.fx, wx0 = 1.f - t.fx, wy1 = t.fy, wy0 = 1.f - t.fy;
    const float wt[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
    float val[4 * (C + 1)];
#pragma unroll
    for (int c = 0; c <= C; ++c) {
      const float gc = c < C ? gtt[c < C ? c : 0] : gml_alpha;
#pragma unroll
      for (int q = 0; q < 4; ++q) val[4 * c + q] = gc * wt[q];
    }
#define SCAE_SEG_STEP(N)                                                                   \
  {                                                                                        \
    const int okey = __builtin_amdgcn_update_dpp(0, key, 0x100 + N, 0xf, 0xf, true);       \
    const bool same = okey == key;                                                         \
    _Pragma("unroll") for (int i = 0; i < 4 * (C + 1); ++i) {                              \
      const float ov = __int_as_float(                                                     \
          __builtin_amdgcn_update_dpp(0, __float_as_int(val[i]), 0x100 + N, 0xf, 0xf, true)); \
      val[i] += same ? ov : 0.f;                                                           \
    }                                                                                      \
  }
    SCAE_SEG_STEP(1) SCAE_SEG_STEP(2)
    // (runs are ~W / tw pixels long: the wider steps only when some run of the wave needs them)
    if (__any(__builtin_amdgcn_update_dpp(0, key, 0x104, 0xf, 0xf, true) == key)) {
      SCAE_SEG_STEP(4)
      if (__any(__builtin_amdgcn_update_dpp(0, key, 0x108, 0xf, 0xf, true) == key)) SCAE_SEG_STEP(8)
    }
#undef SCAE_SEG_STEP
    const int pkey = __builtin_amdgcn_update_dpp(0, key, 0x111, 0xf, 0xf, true);   // row_shr:1
    const bool leader = live && ((lane & 15) == 0 || pkey != key);
    {
      // every 16-lane row has accumulator planes of its own: the leaders of one image
      // row sit in distinct cells, so a plain read-add-write per tap has one writer per
      // address; a row of lanes that straddles two image rows (W is not a multiple of 16)
      // takes them in turn.  The compiler must not fuse the taps of neighbouring
      // addresses into one 8-byte access (another leader's tap lies in between).
      const int grow = fdiv(pl0 + (tid & ~15), inv_wf);
      const bool second = leader && prow != grow;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && !__any(second)) break;   // (wave-uniform) no row of lanes straddles
        const bool on = pass == 0 ? leader && !second : second;
        // tap by tap -- neighbouring leaders' taps alias across taps, never within one;
        // the planes are independent, so their read-add-write chains run side by side
        const int offs[4] = {0, 1, pw, pw + 1};
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          if (on) {
            float old[C + 1];
#pragma unroll
            for (int c = 0; c <= C; ++c)
              if (c < planes) old[c] = wacc[c * psz + t.base + offs[q4]];
#pragma unroll
            for (int c = 0; c <= C; ++c)
              if (c < planes) wacc[c * psz + t.base + offs[q4]] = old[c] + val[4 * c + q4];
          }
          asm volatile("" ::: "memory");
        }
      }
    }
  }
  __syncthreads();

  scae::block_sum<11, NT>(acc, s_red);  // ends with __syncthreads()

  if (!is_bg) {   // the waves' padded planes, summed in wave order
    float *o_t = g_templates + (size_t)(b * M + k) * C * tsz;
    for (int i = tid; i < C * tsz; i += NT) {
      const int c = i / tsz, e = i - c * tsz, y = e / tw, xx = e - y * tw;
      const int at = c * psz + (y + 2) * pw + xx + 2;
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < NWV; ++w) acc += s_acc[w * (C + 1) * psz + at];
      o_t[i] = acc;
    }
    if (alpha_mode) {
      float *o_a = g_alpha_partial + (size_t)(b * M + k) * tsz;
      for (int e = tid; e < tsz; e += NT) {
        const int y = e / tw, xx = e - y * tw, at = C * psz + (y + 2) * pw + xx + 2;
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) acc += s_acc[w * (C + 1) * psz + at];
        o_a[e] = acc;
      }
    }
  }
  if (tid == 0) {
    float *sp = g_scalar_partial + (size_t)(b * K + k) * 4;
    sp[0] = sp[1] = sp[2] = sp[3] = 0.f;
    if (!is_bg) {
#pragma unroll
      for (int i = 0; i < 6; ++i) g_pose[(size_t)(b * M + k) * 6 + i] = acc[i];
      if (g_presence && d.presence)
        g_presence[b * M + k] = acc[6] * scae::log_safe_grad(d.presence[b * M + k]);
    } else {
      if (!d.bg_image) {
        const float s = sc.bg_val;
        sp[0] = acc[7] * s * (1.f - s);
      }
      if (alpha_mode) sp[1] = acc[8] * scae::softplus_grad(d.bg_mixing_logit[0]);
    }
    if (!alpha_mode) sp[2] = acc[9] * scae::softplus_grad(d.temperature_logit[0] + .5f);
    if (FUSED && d.out_scale) sp[3] = acc[10] * scae::softplus_grad(d.out_scale[0]);
  }
}

// ---------------------------------------------------------------------------
// generic mixture over materialised (B,K,C,P) tensors: one lane per (b, p)
// ---------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(NT) void gmm_logprob_fwd_kernel(
    const float *__restrict__ loc, const float *__restrict__ ml,
    const float *__restrict__ sigma_p, const float *__restrict__ x,
    float *__restrict__ out, int K, int Cm, int64_t P) {
  const int b = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (p >= P) return;
  const float sigma = sigma_p[0], inv_var = 1.f / (sigma * sigma), ls = logf(sigma);
  float xv[C];
  scae::Lse post[C], prior[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    xv[c] = x[((size_t)b * C + c) * P + p];
    post[c].init();
    prior[c].init();
  }
  for (int k = 0; k < K; ++k) {
    float mlv = 0.f;
    if (Cm == 1) {
      mlv = ml[((size_t)b * K + k) * P + p];
      prior[0].add(mlv);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (Cm != 1) {
        mlv = ml[(((size_t)b * K + k) * C + c) * P + p];
        prior[c].add(mlv);
      }
      const float diff = xv[c] - loc[(((size_t)b * K + k) * C + c) * P + p];
      post[c].add(-(diff * diff) * (0.5f * inv_var) - ls - scae::kHalfLog2Pi + mlv);
    }
  }
#pragma unroll
  for (int c = 0; c < C; ++c)
    out[((size_t)b * C + c) * P + p] = post[c].value() - prior[Cm == 1 ? 0 : c].value();
}

template <int C>
__global__ __launch_bounds__(NT) void gmm_logprob_bwd_kernel(
    const float *__restrict__ loc, const float *__restrict__ ml,
    const float *__restrict__ sigma_p, const float *__restrict__ x,
    const float *__restrict__ g, float *__restrict__ g_loc, float *__restrict__ g_ml,
    float *__restrict__ g_sigma_partial, float *__restrict__ g_x, int K, int Cm,
    int64_t P) {
  __shared__ float s_red[NT / 64];
  const int b = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * NT + threadIdx.x;
  const bool live = p < P;
  const float sigma = sigma_p[0], inv_var = 1.f / (sigma * sigma), ls = logf(sigma);
  float gsig[1] = {0.f};
  if (live) {
    float xv[C], gv[C], gx[C];
    scae::Lse post[C], prior[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      xv[c] = x[((size_t)b * C + c) * P + p];
      gv[c] = g[((size_t)b * C + c) * P + p];
      gx[c] = 0.f;
      post[c].init();
      prior[c].init();
    }
    for (int k = 0; k < K; ++k) {
      float mlv = 0.f;
      if (Cm == 1) {
        mlv = ml[((size_t)b * K + k) * P + p];
        prior[0].add(mlv);
      }
#pragma unroll
      for (int c = 0; c < C; ++c) {
        if (Cm != 1) {
          mlv = ml[(((size_t)b * K + k) * C + c) * P + p];
          prior[c].add(mlv);
        }
        const float diff = xv[c] - loc[(((size_t)b * K + k) * C + c) * P + p];
        post[c].add(-(diff * diff) * (0.5f * inv_var) - ls - scae::kHalfLog2Pi + mlv);
      }
    }
    float lpost[C], lprior[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      lpost[c] = post[c].value();
      lprior[c] = prior[Cm == 1 ? 0 : c].value();
    }
    for (int k = 0; k < K; ++k) {
      float mlv = 0.f, gml1 = 0.f;
      if (Cm == 1) mlv = ml[((size_t)b * K + k) * P + p];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const size_t o = (((size_t)b * K + k) * C + c) * P + p;
        if (Cm != 1) mlv = ml[o];
        const float diff = xv[c] - loc[o];
        const float w = expf(-(diff * diff) * (0.5f * inv_var) - ls - scae::kHalfLog2Pi +
                             mlv - lpost[c]);
        const float sp = expf(mlv - lprior[c]);
        const float gl = gv[c] * w * diff * inv_var;
        g_loc[o] = gl;
        gx[c] -= gl;
        gsig[0] += gv[c] * w * (diff * diff * inv_var - 1.f) / sigma;
        if (Cm != 1)
          g_ml[o] = gv[c] * (w - sp);
        else
          gml1 += gv[c] * (w - sp);
      }
      if (Cm == 1) g_ml[((size_t)b * K + k) * P + p] = gml1;
    }
    if (g_x) {
#pragma unroll
      for (int c = 0; c < C; ++c) g_x[((size_t)b * C + c) * P + p] = gx[c];
    }
  }
  scae::block_sum<1, NT>(gsig, s_red);
  if (threadIdx.x == 0) atomicAdd(&g_sigma_partial[b], gsig[0]);
}

template <int C, bool MODE>
__global__ __launch_bounds__(NT) void gmm_mean_mode_kernel(
    const float *__restrict__ loc, const float *__restrict__ ml,
    const float *__restrict__ sigma_p, float *__restrict__ out, int maximum, int K,
    int Cm, int64_t P) {
  const int b = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (p >= P) return;
  if (MODE) {
    // argmax over K of log_softmax(ml) (+ log N(loc; loc, sigma), a constant
    // per component here) -- first maximum wins, like torch.argmax on CPU.
    (void)sigma_p;
    (void)maximum;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int cm = (Cm == 1) ? 0 : c;
      float best = -INFINITY;
      int bk = 0;
      for (int k = 0; k < K; ++k) {
        const float v = ml[(((size_t)b * K + k) * Cm + cm) * P + p];
        if (v > best) {
          best = v;
          bk = k;
        }
      }
      out[((size_t)b * C + c) * P + p] = loc[(((size_t)b * K + bk) * C + c) * P + p];
    }
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int cm = (Cm == 1) ? 0 : c;
      float m = -INFINITY;
      for (int k = 0; k < K; ++k)
        m = fmaxf(m, ml[(((size_t)b * K + k) * Cm + cm) * P + p]);
      float s = 0.f, acc = 0.f;
      for (int k = 0; k < K; ++k) {
        const float e = expf(ml[(((size_t)b * K + k) * Cm + cm) * P + p] - m);
        s += e;
        acc += e * loc[(((size_t)b * K + k) * C + c) * P + p];
      }
      out[((size_t)b * C + c) * P + p] = acc / s;
    }
  }
}

int check_desc(const scae_decoder_desc *d) {
  if (!d || !d->templates || !d->pose) return SCAE_ERR_BAD_ARG;
  if (d->B <= 0 || d->M <= 0 || d->C <= 0 || d->th <= 0 || d->tw <= 0 || d->H <= 0 ||
      d->W <= 0)
    return SCAE_ERR_BAD_ARG;
  if (!d->bg_image && !d->bg_value) return SCAE_ERR_BAD_ARG;
  if (d->templates_alpha && !d->bg_mixing_logit) return SCAE_ERR_BAD_ARG;
  if (d->template_repeat < 0 || (d->template_repeat > 1 && d->B % d->template_repeat))
    return SCAE_ERR_BAD_ARG;
  if (!d->templates_alpha && !d->temperature_logit) return SCAE_ERR_BAD_ARG;
  if (d->C > SCAE_MAX_CHANNELS) return SCAE_ERR_UNSUPPORTED;
  if ((d->C + 1) * d->th * d->tw > SCAE_RENDER_MAX_TEMPLATE_ELEMS) return SCAE_ERR_UNSUPPORTED;
  return SCAE_OK;
}

template <typename KernelT>
int set_lds(KernelT kernel, size_t bytes) {
  if (bytes > 160 * 1024) return SCAE_ERR_UNSUPPORTED;
  if (bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
  }
  return SCAE_OK;
}

}  // namespace

extern "C" int scae_template_render_fwd_f32(const scae_decoder_desc *d,
                                            float *transformed_templates,
                                            float *mixing_logits, void *stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  SCAE_REQUIRE(transformed_templates && mixing_logits);
#ifndef SCAE_K1_NO_WAVE
  if (render_wave_lds(d) && (((size_t)transformed_templates | (size_t)mixing_logits) & 15) == 0 &&
      (!d->bg_image || ((size_t)d->bg_image & 15) == 0))
    return launch_render_wave(d, transformed_templates, mixing_logits, (hipStream_t)stream);
#endif
  const size_t lds = sizeof(float) * (size_t)(d->C + 1) * pad_elems(d->th, d->tw);
  int rc2 = set_lds(render_fwd_kernel, lds);
  if (rc2) return rc2;
  scae::launch(render_fwd_kernel, dim3(d->M + 1, d->B), dim3(NT), lds,
                     (hipStream_t)stream, *d, transformed_templates, mixing_logits);
  return scae_launch_status();
}

// pixel tiling of the log-prob kernel: lanes per pixel, pixels per workgroup
namespace scae_k1 {
LpTiling lp_tiling(const scae_decoder_desc *d) {
  const int HW = d->H * d->W;
  LpTiling t;
  t.wave = false;
#ifndef SCAE_K1_NO_WAVE
  if (logprob_wave_lds(d)) {
    // wave form: a workgroup = up to 16 waves of 64 consecutive pixels of one image, all M
    // components per lane; enough tiles per image for >= 2 workgroups per CU
    const int waves = (HW + 63) / 64;
    int tiles = (waves + 15) / 16;
    const int want = (512 + d->B - 1) / d->B;
    if (tiles < want) tiles = want;
    if (tiles > waves) tiles = waves;
    // A CU deals a workgroup's waves to its four SIMDs starting at the first one, so it holds
    // floor(waves per SIMD / ceil(workgroup waves / 4)) workgroups (tools/probes/
    // lds_residency.cpp, tools/tl_prof.py): at the 4 waves per SIMD of the launch shared with
    // the object encoder's trunk, 7-wave workgroups are 2 per CU -- 512 places for 128 + 512
    // workgroups, a quarter of the likelihood ran in a second round -- and 4-wave workgroups
    // are 4 per CU.  Small batches take 4-wave workgroups ...
    // ... when four of them also fit the CU's LDS (the planes of all M components per workgroup)
    const int tiles4 = (waves + 3) / 4;
    if ((long)d->B * tiles4 <= 1024 && tiles4 > tiles && logprob_wave_lds(d) <= 40 * 1024)
      tiles = tiles4;
    const int wpt = (waves + tiles - 1) / tiles;
    t.wave = true;
    t.ksplit = 1;
    t.ppb = wpt * 64;
    t.tiles = (HW + t.ppb - 1) / t.ppb;
    return t;
  }
#endif
  // component split across lanes: more lanes per pixel when the batch alone
  // cannot fill 256 CUs
  const long pixels = (long)d->B * HW;
  t.ksplit = pixels >= 256L * 1024 * 4 ? 1 : (pixels >= 256L * 1024 ? 2 : 4);
#ifndef SCAE_LP_ROUNDS
#define SCAE_LP_ROUNDS 6
#endif
  t.ppb = SCAE_LP_ROUNDS * NT / t.ksplit;  // pixel rounds per workgroup amortise the LDS fill
  if (t.ppb > HW) t.ppb = ((HW + (NT / t.ksplit) - 1) / (NT / t.ksplit)) * (NT / t.ksplit);
  t.tiles = (HW + t.ppb - 1) / t.ppb;
  return t;
}
}  // namespace scae_k1

namespace {
template <int C>
int launch_logprob_fwd(const scae_decoder_desc *d, const float *x, float *log_prob,
                       float *lse_post, float *lse_prior, float *block_sums, hipStream_t st) {
  const size_t planes = (size_t)d->M * (d->C + (d->templates_alpha ? 1 : 0));
  // zero-padded planes (no tap masks) while two workgroups still fit a CU's LDS
  const size_t lds_pad = sizeof(float) * (planes * pad_elems(d->th, d->tw) + (size_t)d->M * 7);
#ifndef SCAE_LP_PAD_LIMIT_KB
#define SCAE_LP_PAD_LIMIT_KB 72
#endif
  const bool pad = lds_pad <= (size_t)SCAE_LP_PAD_LIMIT_KB * 1024;
  const size_t lds =
      pad ? lds_pad : sizeof(float) * (planes * d->th * d->tw + (size_t)d->M * 7);
  const LpTiling t = lp_tiling(d);
  if (t.wave) return launch_logprob_wave(d, t, x, log_prob, lse_post, lse_prior, block_sums, st);
  const int ppb = t.ppb;
  const dim3 grid(t.tiles, d->B);
  int rc;
#define SCAE_LAUNCH_LP2(KS, PD)                                                           \
  rc = set_lds(logprob_fwd_kernel<C, KS, PD>, lds);                                       \
  if (rc) return rc;                                                                      \
  scae::launch((logprob_fwd_kernel<C, KS, PD>), grid, dim3(NT), lds, st, *d, x,     \
                     log_prob, lse_post, lse_prior, ppb, block_sums)
#define SCAE_LAUNCH_LP(KS)  \
  if (pad) {                \
    SCAE_LAUNCH_LP2(KS, true);  \
  } else {                  \
    SCAE_LAUNCH_LP2(KS, false); \
  }
  if (t.ksplit == 1) {
    SCAE_LAUNCH_LP(1)
  } else if (t.ksplit == 2) {
    SCAE_LAUNCH_LP(2)
  } else {
    SCAE_LAUNCH_LP(4)
  }
#undef SCAE_LAUNCH_LP
#undef SCAE_LAUNCH_LP2
  return scae_launch_status();
}

template <int C>
int launch_bwd(const scae_decoder_desc *d, const float *x, const float *lse_post,
               const float *lse_prior, const float *g_logprob, const float *g_tt,
               const float *g_ml, float *g_templates, float *g_alpha_partial,
               float *g_pose, float *g_presence, float *g_bg_image,
               float *g_scalar_partial, const float *g_tile, hipStream_t st) {
  const int tsz = d->th * d->tw;
  const LpTiling lt = lp_tiling(d);
  // the scatter form needs a set of accumulator planes per 16-lane row beside >= 4
  // workgroups per CU, and a 16-lane row that spans at most two image rows; else the gather
  const size_t plane = (size_t)(d->C + 1) * pad_elems(d->th, d->tw);
  const bool scatter = d->W >= 16 &&
                       sizeof(float) * ((1 + NT / 16) * plane + 11 * (NT / 64)) <= 40 * 1024;
  int rows = (int)((40 * 1024 / sizeof(float)) / ((size_t)(d->C + 1) * d->W));
  rows = rows < 1 ? 1 : (rows > d->H ? d->H : rows);
  const size_t lds =
      scatter ? sizeof(float) * ((1 + NT / 16) * plane + 11 * (NT / 64))
              : sizeof(float) * ((size_t)(d->C + 1) * pad_elems(d->th, d->tw) +
                                 SLICES * (size_t)(d->C + 1) * tsz + 11 * (NT / 64) +
                                 (size_t)(d->C + 1) * rows * d->W);
  const dim3 grid(d->M + 1, d->B);
  const bool fused = (g_tt == nullptr && g_ml == nullptr);
#ifndef SCAE_K1_NO_CELL
  if (fused && bwd_cell_lds(d))
    return launch_bwd_cell(d, x, lse_post, lse_prior, g_logprob, g_tile, lt.tiles, lt.ppb,
                           g_templates, g_alpha_partial, g_pose, g_presence, g_bg_image,
                           g_scalar_partial, st);
#endif
  int rc;
#define SCAE_LAUNCH_BWD(KERNEL, FU)                                                          \
  rc = set_lds(KERNEL<C, FU>, lds);                                                          \
  if (rc) return rc;                                                                         \
  scae::launch((KERNEL<C, FU>), grid, dim3(NT), lds, st, *d, x, lse_post, lse_prior,  \
                     g_logprob, g_tt, g_ml, g_templates, g_alpha_partial, g_pose, g_presence, \
                     g_bg_image, g_scalar_partial, rows, g_tile, lt.tiles, lt.ppb)
  if (scatter) {
    if (fused) {
      SCAE_LAUNCH_BWD(render_bwd_kernel, true);
    } else {
      SCAE_LAUNCH_BWD(render_bwd_kernel, false);
    }
  } else {
    if (fused) {
      SCAE_LAUNCH_BWD(render_bwd_gather_kernel, true);
    } else {
      SCAE_LAUNCH_BWD(render_bwd_gather_kernel, false);
    }
  }
#undef SCAE_LAUNCH_BWD
  return scae_launch_status();
}
}  // namespace

#define SCAE_DISPATCH_C(Cval, CALL)  \
  switch (Cval) {                    \
    case 1: return CALL(1);          \
    case 2: return CALL(2);          \
    case 3: return CALL(3);          \
    case 4: return CALL(4);          \
    default: return SCAE_ERR_UNSUPPORTED; \
  }

extern "C" int scae_render_gmm_logprob_fwd_f32(const scae_decoder_desc *d, const float *x,
                                               float *log_prob, float *lse_post,
                                               float *lse_prior, void *stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  SCAE_REQUIRE(x && log_prob && lse_post && lse_prior);
#define CALL(CC) \
  launch_logprob_fwd<CC>(d, x, log_prob, lse_post, lse_prior, nullptr, (hipStream_t)stream)
  SCAE_DISPATCH_C(d->C, CALL)
#undef CALL
}

extern "C" int scae_render_gmm_logprob_tiles(const scae_decoder_desc *d) {
  if (check_desc(d)) return 0;
  return lp_tiling(d).tiles;
}

extern "C" int scae_render_gmm_logprob_sums_fwd_f32(const scae_decoder_desc *d, const float *x,
                                                    float *tile_sums, float *lse_post,
                                                    float *lse_prior, void *stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  SCAE_REQUIRE(x && tile_sums && lse_post && lse_prior);
#define CALL(CC) \
  launch_logprob_fwd<CC>(d, x, nullptr, lse_post, lse_prior, tile_sums, (hipStream_t)stream)
  SCAE_DISPATCH_C(d->C, CALL)
#undef CALL
}

extern "C" int scae_render_gmm_sums_bwd_f32(const scae_decoder_desc *d, const float *x,
                                            const float *lse_post, const float *lse_prior,
                                            const float *g_tile_sums, float *g_templates,
                                            float *g_alpha_partial, float *g_pose,
                                            float *g_presence, float *g_bg_image,
                                            float *g_scalar_partial, void *stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  SCAE_REQUIRE(g_templates && g_pose && g_scalar_partial && x && lse_post && lse_prior &&
               g_tile_sums);
  if (d->template_repeat > 1) return SCAE_ERR_UNSUPPORTED;
  if (d->templates_alpha) SCAE_REQUIRE(g_alpha_partial);
#define CALL(CC)                                                                          \
  launch_bwd<CC>(d, x, lse_post, lse_prior, nullptr, nullptr, nullptr, g_templates,       \
                 g_alpha_partial, g_pose, g_presence, g_bg_image, g_scalar_partial,       \
                 g_tile_sums, (hipStream_t)stream)
  SCAE_DISPATCH_C(d->C, CALL)
#undef CALL
}

extern "C" int scae_render_gmm_bwd_f32(const scae_decoder_desc *d, const float *x,
                                       const float *lse_post, const float *lse_prior,
                                       const float *g_logprob, const float *g_tt,
                                       const float *g_ml, float *g_templates,
                                       float *g_alpha_partial, float *g_pose,
                                       float *g_presence, float *g_bg_image,
                                       float *g_scalar_partial, void *stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  SCAE_REQUIRE(g_templates && g_pose && g_scalar_partial);
  if (d->template_repeat > 1) return SCAE_ERR_UNSUPPORTED;  // forward-only (no_grad) feature
  if (d->templates_alpha) SCAE_REQUIRE(g_alpha_partial);
  if (!g_tt && !g_ml) SCAE_REQUIRE(x && lse_post && lse_prior && g_logprob);
#define CALL(CC)                                                                       \
  launch_bwd<CC>(d, x, lse_post, lse_prior, g_logprob, g_tt, g_ml, g_templates,        \
                 g_alpha_partial, g_pose, g_presence, g_bg_image, g_scalar_partial,    \
                 nullptr, (hipStream_t)stream)
  SCAE_DISPATCH_C(d->C, CALL)
#undef CALL
}

namespace {
template <int C>
int launch_gmm_fwd(const float *loc, const float *ml, const float *sigma, const float *x,
                   float *out, int B, int K, int Cm, int64_t P, hipStream_t st) {
  const dim3 grid((unsigned)((P + NT - 1) / NT), B);
  scae::launch((gmm_logprob_fwd_kernel<C>), grid, dim3(NT), 0, st, loc, ml, sigma, x,
                     out, K, Cm, P);
  return scae_launch_status();
}
template <int C>
int launch_gmm_bwd(const float *loc, const float *ml, const float *sigma, const float *x,
                   const float *g, float *g_loc, float *g_ml, float *g_sigma_partial,
                   float *g_x, int B, int K, int Cm, int64_t P, hipStream_t st) {
  hipError_t e = hipMemsetAsync(g_sigma_partial, 0, sizeof(float) * B, st);
  if (e != hipSuccess) return (int)e;
  const dim3 grid((unsigned)((P + NT - 1) / NT), B);
  scae::launch((gmm_logprob_bwd_kernel<C>), grid, dim3(NT), 0, st, loc, ml, sigma, x,
                     g, g_loc, g_ml, g_sigma_partial, g_x, K, Cm, P);
  return scae_launch_status();
}
template <int C>
int launch_gmm_mm(bool mode, const float *loc, const float *ml, const float *sigma,
                  float *out, int maximum, int B, int K, int Cm, int64_t P,
                  hipStream_t st) {
  const dim3 grid((unsigned)((P + NT - 1) / NT), B);
  if (mode)
    scae::launch((gmm_mean_mode_kernel<C, true>), grid, dim3(NT), 0, st, loc, ml,
                       sigma, out, maximum, K, Cm, P);
  else
    scae::launch((gmm_mean_mode_kernel<C, false>), grid, dim3(NT), 0, st, loc, ml,
                       sigma, out, maximum, K, Cm, P);
  return scae_launch_status();
}
int check_gmm(int B, int K, int C, int Cm, int64_t P) {
  if (B <= 0 || K <= 0 || C <= 0 || P <= 0) return SCAE_ERR_BAD_ARG;
  if (Cm != 1 && Cm != C) return SCAE_ERR_BAD_ARG;
  if (C > SCAE_MAX_CHANNELS) return SCAE_ERR_UNSUPPORTED;
  return SCAE_OK;
}
}  // namespace

extern "C" int scae_gmm_log_prob_fwd_f32(const float *loc, const float *mixing_logits,
                                         const float *sigma, const float *x,
                                         float *log_prob, int B, int K, int C, int Cm,
                                         int64_t P, void *stream) {
  int rc = check_gmm(B, K, C, Cm, P);
  if (rc) return rc;
  SCAE_REQUIRE(loc && mixing_logits && sigma && x && log_prob);
#define CALL(CC) \
  launch_gmm_fwd<CC>(loc, mixing_logits, sigma, x, log_prob, B, K, Cm, P, (hipStream_t)stream)
  SCAE_DISPATCH_C(C, CALL)
#undef CALL
}

extern "C" int scae_gmm_log_prob_bwd_f32(const float *loc, const float *mixing_logits,
                                         const float *sigma, const float *x,
                                         const float *g_logprob, float *g_loc,
                                         float *g_ml, float *g_sigma_partial, float *g_x,
                                         int B, int K, int C, int Cm, int64_t P,
                                         void *stream) {
  int rc = check_gmm(B, K, C, Cm, P);
  if (rc) return rc;
  SCAE_REQUIRE(loc && mixing_logits && sigma && x && g_logprob && g_loc && g_ml &&
               g_sigma_partial);
#define CALL(CC)                                                                       \
  launch_gmm_bwd<CC>(loc, mixing_logits, sigma, x, g_logprob, g_loc, g_ml,             \
                     g_sigma_partial, g_x, B, K, Cm, P, (hipStream_t)stream)
  SCAE_DISPATCH_C(C, CALL)
#undef CALL
}

extern "C" int scae_gmm_mean_f32(const float *loc, const float *mixing_logits, float *out,
                                 int B, int K, int C, int Cm, int64_t P, void *stream) {
  int rc = check_gmm(B, K, C, Cm, P);
  if (rc) return rc;
  SCAE_REQUIRE(loc && mixing_logits && out);
#define CALL(CC)                                                                    \
  launch_gmm_mm<CC>(false, loc, mixing_logits, nullptr, out, 0, B, K, Cm, P,        \
                    (hipStream_t)stream)
  SCAE_DISPATCH_C(C, CALL)
#undef CALL
}

extern "C" int scae_gmm_mode_f32(const float *loc, const float *mixing_logits,
                                 const float *sigma, float *out, int maximum, int B,
                                 int K, int C, int Cm, int64_t P, void *stream) {
  int rc = check_gmm(B, K, C, Cm, P);
  if (rc) return rc;
  SCAE_REQUIRE(loc && mixing_logits && out);
  // distributions.py:64-65 adds (B,K,C,..) into (B,K,1,..) in place: the
  // reference raises for C > 1 there; so do we.
  if (maximum && Cm == 1 && C > 1) return SCAE_ERR_UNSUPPORTED;
#define CALL(CC)                                                                    \
  launch_gmm_mm<CC>(true, loc, mixing_logits, sigma, out, maximum, B, K, Cm, P,     \
                    (hipStream_t)stream)
  SCAE_DISPATCH_C(C, CALL)
#undef CALL
}
