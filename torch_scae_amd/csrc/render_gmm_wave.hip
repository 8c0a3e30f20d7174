// K1, likelihood forward in its wave form: log_prob(x) of the per-pixel Gaussian mixture
// straight from the compact decoder inputs (part_decoder.py:174-237, distributions.py:34-47,
// stacked_capsule_auto_encoder.py:220), alpha-channel mode.
//
// One lane = one pixel, ALL M + 1 components; a workgroup = up to 16 waves of 64
// consecutive pixels of one image.  What that buys over the (pixel, component-subset)
// lane layout of render_gmm.hip's logprob_fwd_kernel (137 VALU instructions per
// (pixel, component) term, measured):
//   * the component index is wave-uniform: a component's texel-space affine map and its
//     presence term are one broadcast LDS read per wave, not per-lane index arithmetic;
//   * a texel's channels and its alpha are INTERLEAVED in the padded LDS planes
//     (C = 1: float2 {t, a log2 e}): the four bilinear taps of all planes are two
//     ds_read2_b64 from one address register, and the blend runs on the pairs;
//   * the padded planes of an image are staged a texel ROW per thread (one division per
//     row instead of two per element), by a fifth of the workgroups (4 tiles per image
//     at cfg-2 instead of 5 x 4 lanes per pixel);
//   * the mixture log-sum-exps are two-pass over register-resident chunks of 8
//     components, in the log2 domain (alpha planes and presence terms pre-scaled by
//     log2 e at staging): max, then ONE v_exp_f32 per value -- the online (max, sum)
//     form costs two exponentials per value -- and no cross-lane merge at all.
#include "common.h"
#include "render_gmm_dev.h"
#include "render_gmm_wave_dev.h"

namespace scae_k1 {
namespace {
template <int C>
__global__ __launch_bounds__(1024) void logprob_wave_kernel(
    scae_decoder_desc d, const float *__restrict__ x, float *__restrict__ log_prob,
    float *__restrict__ lse_post, float *__restrict__ lse_prior, int ppb,
    float *__restrict__ block_sums) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  logprob_wave_body<C>(d, x, log_prob, lse_post, lse_prior, ppb, block_sums, smem, blockIdx.x,
                       blockIdx.y, gridDim.x, blockDim.x);
}

template <int C>
int launch_c(const scae_decoder_desc *d, const LpTiling &t, const float *x, float *log_prob,
             float *lse_post, float *lse_prior, float *block_sums, hipStream_t st) {
  const size_t lds = logprob_wave_lds(d);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(logprob_wave_kernel<C>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  scae::launch((logprob_wave_kernel<C>), dim3(t.tiles, d->B), dim3(t.ppb), lds, st, *d, x,
                     log_prob, lse_post, lse_prior, t.ppb, block_sums);
  return scae_launch_status();
}

// ---------------------------------------------------------------------------------------
// Materialising forward (transformed_templates, mixing_logits: part_decoder.py:174-231) in
// the same lane layout.  The tensors are a pure write stream (41 MB at cfg-2, x O more for
// the per-object-capsule reconstructions of reconstruct_alternatives), so the kernel is
// built around the stores: a lane owns FOUR consecutive pixels and every store is a full
// 16-byte quad of one (b, k, c) plane row (1 KiB contiguous per wave instruction); a
// workgroup = all pixel quads of one image x a group of KG components, whose planes it
// stages once -- the per-workgroup set-up of render_gmm.hip's one-component workgroups
// (scalars, staging, 6 pixels per thread) is shared by KG x 4 pixels per thread.
template <int C>
__global__ __launch_bounds__(1024) void render_wave_kernel(scae_decoder_desc d,
                                                           float *__restrict__ tt,
                                                           float *__restrict__ ml, int KG) {
  constexpr int TX = TexelOf<C>::TX;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x;
  const int M = d.M, K = M + 1, W = d.W, HW = d.H * d.W, th = d.th, tw = d.tw;
  const int psz = pad_elems(th, tw), pw = pad_w(tw);
  const int k0 = blockIdx.x * KG, k1 = min(K, k0 + KG), nk = min(M, k1) - k0;   // nk templates
  float *s_pl = smem;                                  // nk padded planes
  float *s_pose = s_pl + (size_t)KG * psz * TX;        // KG x 8: pose (6), log presence
  if (nk > 0) stage_planes<C>(s_pl, d, b, k0, nk, 1.f, tid, nthr);
  for (int kl = tid; kl < nk; kl += nthr) {
    const float *a = d.pose + ((size_t)b * M + k0 + kl) * 6;
#pragma unroll
    for (int i = 0; i < 6; ++i) s_pose[kl * 8 + i] = a[i];
    s_pose[kl * 8 + 6] = d.presence ? log_safe(d.presence[b * M + k0 + kl]) : 0.f;   // :225-231
  }
  __syncthreads();

  const int p0 = 4 * tid;
  if (p0 >= HW) return;
  const float inv_w = 1.f / (float)W, inv_h = 1.f / (float)d.H;
  float xn[4], yn[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int p = p0 + e, pi = (int)(((float)p + 0.5f) * inv_w), pj = p - pi * W;
    xn[e] = (float)(2 * pj + 1) * inv_w - 1.f;
    yn[e] = (float)(2 * pi + 1) * inv_h - 1.f;
  }
  const float txf = (float)tw, tyf = (float)th, pwf = (float)pw;
  const float *s_tap = s_pl + (size_t)(2 * pw + 2) * TX;
  for (int kl = 0; kl < nk; ++kl) {
    const float *a = s_pose + kl * 8;
    const float lsp = a[6];
    float v[TX][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float ix, iy;
      tex_pos(a, xn[e], yn[e], tw, th, ix, iy);
      ix = fminf(fmaxf(ix, -2.f), txf);
      iy = fminf(fmaxf(iy, -2.f), tyf);
      const float x0f = floorf(ix), y0f = floorf(iy), fx = ix - x0f, fy = iy - y0f;
      const float *q0 = s_tap + ((size_t)kl * psz + (int)fmaf(y0f, pwf, x0f)) * TX;
      const float *q1 = q0 + pw * TX;
#pragma unroll
      for (int c = 0; c <= C; ++c) {
        const float v00 = q0[c], v01 = q0[TX + c], v10 = q1[c], v11 = q1[TX + c];
        const float t0 = fmaf(fx, v01 - v00, v00), t1 = fmaf(fx, v11 - v10, v10);
        v[c][e] = fmaf(fy, t1 - t0, t0);
      }
    }
    const size_t kb = (size_t)b * K + k0 + kl;
#pragma unroll
    for (int c = 0; c < C; ++c)
      *reinterpret_cast<float4 *>(tt + (kb * C + c) * HW + p0) =
          make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
    *reinterpret_cast<float4 *>(ml + kb * HW + p0) =
        make_float4(v[C][0] + lsp, v[C][1] + lsp, v[C][2] + lsp, v[C][3] + lsp);
  }
  if (k1 == K) {   // this group ends with the background component, :189-195, :210-213
    const float bg_ml = softplusf_(d.bg_mixing_logit[0]);
    const float bg_val = d.bg_image ? 0.f : sigmoidf_(d.bg_value[0]);
    const size_t kb = (size_t)b * K + M;
#pragma unroll
    for (int c = 0; c < C; ++c)
      *reinterpret_cast<float4 *>(tt + (kb * C + c) * HW + p0) =
          d.bg_image ? *reinterpret_cast<const float4 *>(d.bg_image + ((size_t)b * C + c) * HW + p0)
                     : make_float4(bg_val, bg_val, bg_val, bg_val);
    *reinterpret_cast<float4 *>(ml + kb * HW + p0) = make_float4(bg_ml, bg_ml, bg_ml, bg_ml);
  }
}

#ifndef SCAE_RENDER_WGS
#define SCAE_RENDER_WGS 512
#endif
#ifndef SCAE_RENDER_MINK
#define SCAE_RENDER_MINK 4
#endif
struct RenderGeom {
  int groups, KG, threads;
  size_t lds;
};
RenderGeom render_geom(const scae_decoder_desc *d) {
  RenderGeom g = {0, 0, 0, 0};
  const int HW = d->H * d->W, K = d->M + 1;
  // alpha-channel mode, whole pixel quads, one thread per quad of the image
  if (!d->templates_alpha || d->C < 1 || d->C > 4 || HW % 4 || HW / 4 > 1024) return g;
  const int TX = d->C == 1 ? 2 : (d->C <= 3 ? 4 : 8);
  // component groups: >= 2 workgroups per CU when the batch allows, >= 4 components each
  int groups = (SCAE_RENDER_WGS + d->B - 1) / d->B;
  if (groups > (K + SCAE_RENDER_MINK - 1) / SCAE_RENDER_MINK) groups = (K + SCAE_RENDER_MINK - 1) / SCAE_RENDER_MINK;
  if (groups < 1) groups = 1;
  g.KG = (K + groups - 1) / groups;
  g.groups = (K + g.KG - 1) / g.KG;
  g.threads = ((HW / 4 + 63) / 64) * 64;
  const size_t bytes = sizeof(float) * ((size_t)g.KG * pad_elems(d->th, d->tw) * TX + (size_t)g.KG * 8);
  g.lds = bytes <= 64 * 1024 ? bytes : 0;
  return g;
}

template <int C>
int launch_render_c(const scae_decoder_desc *d, const RenderGeom &g, float *tt, float *ml,
                    hipStream_t st) {
  if (g.lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(render_wave_kernel<C>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
    if (e != hipSuccess) return (int)e;
  }
  scae::launch((render_wave_kernel<C>), dim3(g.groups, d->B), dim3(g.threads), g.lds, st, *d,
                     tt, ml, g.KG);
  return scae_launch_status();
}

template <int C, int NTB>
__global__ __launch_bounds__(NTB) void bwd_cell_kernel(
    scae_decoder_desc d, const float *__restrict__ x, const float *__restrict__ lse_post,
    const float *__restrict__ lse_prior, const float *__restrict__ g_logprob,
    const float *__restrict__ g_tile, int lp_tiles, int lp_ppb,
    float *__restrict__ g_templates, float *__restrict__ g_alpha_partial,
    float *__restrict__ g_pose, float *__restrict__ g_presence,
    float *__restrict__ g_bg_image, float *__restrict__ g_scalar_partial, int chunk_rows,
    int max_items, int item_budget) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  bwd_cell_body<C, NTB>(d, x, lse_post, lse_prior, g_logprob, g_tile, lp_tiles, lp_ppb, g_templates, g_alpha_partial, g_pose, g_presence, g_bg_image, g_scalar_partial, chunk_rows, max_items, item_budget, smem, blockIdx.x, blockIdx.y);
}

// The same work from a FIXED number of resident workgroups, each walking (component, image)
// pairs: a launch that leaves most of every CU's wave slots, registers and LDS to the
// kernels of another stream (train_step's second lane: the whole-grid form floods the chip
// with 3 200 workgroups of 27 KB, and the capsule-MLP chain's 85 KB workgroups on the main
// lane then wait until it has drained).
template <int C, int NTB>
__global__ __launch_bounds__(NTB) void bwd_cell_walk_kernel(
    scae_decoder_desc d, const float *__restrict__ x, const float *__restrict__ lse_post,
    const float *__restrict__ lse_prior, const float *__restrict__ g_logprob,
    const float *__restrict__ g_tile, int lp_tiles, int lp_ppb,
    float *__restrict__ g_templates, float *__restrict__ g_alpha_partial,
    float *__restrict__ g_pose, float *__restrict__ g_presence,
    float *__restrict__ g_bg_image, float *__restrict__ g_scalar_partial, int chunk_rows,
    int max_items, int item_budget) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int K = d.M + 1, total = K * d.B;
  for (int id = blockIdx.x; id < total; id += gridDim.x) {   // (workgroup-uniform)
    const int b = id / K, k = id - b * K;
    bwd_cell_body<C, NTB>(d, x, lse_post, lse_prior, g_logprob, g_tile, lp_tiles, lp_ppb,
                          g_templates, g_alpha_partial, g_pose, g_presence, g_bg_image,
                          g_scalar_partial, chunk_rows, max_items, item_budget, smem, k, b);
    __syncthreads();   // the next pair re-stages the planes
  }
}

template <int C, int NTB>
int launch_bwd_c(const scae_decoder_desc *d, const CellGeom &g, const float *x,
                 const float *lse_post, const float *lse_prior, const float *g_logprob,
                 const float *g_tile, int lp_tiles, int lp_ppb, float *g_templates,
                 float *g_alpha_partial, float *g_pose, float *g_presence, float *g_bg_image,
                 float *g_scalar_partial, hipStream_t st) {
  if (g.lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bwd_cell_kernel<C, NTB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
    if (e != hipSuccess) return (int)e;
  }
  const int resident = d->bwd_resident;
  if (resident > 0 && resident < (d->M + 1) * d->B) {
    if (g.lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bwd_cell_walk_kernel<C, NTB>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
      if (e != hipSuccess) return (int)e;
    }
    scae::launch((bwd_cell_walk_kernel<C, NTB>), dim3(resident), dim3(NTB), g.lds, st, *d, x,
                 lse_post, lse_prior, g_logprob, g_tile, lp_tiles, lp_ppb, g_templates,
                 g_alpha_partial, g_pose, g_presence, g_bg_image, g_scalar_partial,
                 g.chunk_rows, g.max_items, g.item_budget);
    return scae_launch_status();
  }
  scae::launch((bwd_cell_kernel<C, NTB>), dim3(d->M + 1, d->B), dim3(NTB), g.lds, st, *d, x,
                     lse_post, lse_prior, g_logprob, g_tile, lp_tiles, lp_ppb, g_templates,
                     g_alpha_partial, g_pose, g_presence, g_bg_image, g_scalar_partial,
                     g.chunk_rows, g.max_items, g.item_budget);
  return scae_launch_status();
}
}  // namespace

size_t logprob_wave_lds(const scae_decoder_desc *d) {
  if (!d->templates_alpha || d->C < 1 || d->C > 4) return 0;   // alpha-channel mode only
  const int TX = d->C == 1 ? 2 : (d->C <= 3 ? 4 : 8);
  const size_t bytes = sizeof(float) * ((size_t)d->M * plane_elems(d->th, d->tw, kLogprobPadLow) * TX +
                                        (size_t)(d->M + KC) * 8 + 16);
  return bytes <= 160 * 1024 ? bytes : 0;
}

int launch_logprob_wave(const scae_decoder_desc *d, const LpTiling &t, const float *x,
                        float *log_prob, float *lse_post, float *lse_prior, float *block_sums,
                        hipStream_t st) {
  switch (d->C) {
    case 1: return launch_c<1>(d, t, x, log_prob, lse_post, lse_prior, block_sums, st);
    case 2: return launch_c<2>(d, t, x, log_prob, lse_post, lse_prior, block_sums, st);
    case 3: return launch_c<3>(d, t, x, log_prob, lse_post, lse_prior, block_sums, st);
    case 4: return launch_c<4>(d, t, x, log_prob, lse_post, lse_prior, block_sums, st);
    default: return SCAE_ERR_UNSUPPORTED;
  }
}

size_t render_wave_lds(const scae_decoder_desc *d) { return render_geom(d).lds; }

int launch_render_wave(const scae_decoder_desc *d, float *tt, float *ml, hipStream_t st) {
  const RenderGeom g = render_geom(d);
  if (!g.lds) return SCAE_ERR_UNSUPPORTED;
  switch (d->C) {
    case 1: return launch_render_c<1>(d, g, tt, ml, st);
    case 2: return launch_render_c<2>(d, g, tt, ml, st);
    case 3: return launch_render_c<3>(d, g, tt, ml, st);
    case 4: return launch_render_c<4>(d, g, tt, ml, st);
    default: return SCAE_ERR_UNSUPPORTED;
  }
}

size_t bwd_cell_lds(const scae_decoder_desc *d) { return cell_geom(d).lds; }

int launch_bwd_cell(const scae_decoder_desc *d, const float *x, const float *lse_post,
                    const float *lse_prior, const float *g_logprob, const float *g_tile,
                    int lp_tiles, int lp_ppb, float *g_templates, float *g_alpha_partial,
                    float *g_pose, float *g_presence, float *g_bg_image,
                    float *g_scalar_partial, hipStream_t st) {
  const CellGeom g = cell_geom(d);
  if (!g.lds) return SCAE_ERR_UNSUPPORTED;
  // threads per workgroup: the pixel rounds of a chunk of rows with the fewest idle lanes
  // (40 x 40: 320 threads walk the 1600 pixels in exactly 5 rounds; 256 would need 7)
  const int chunk_px = g.chunk_rows * d->W;
  const bool nt320 = SCAE_CELL_NT320 && ((chunk_px + 319) / 320) * 320 < ((chunk_px + 255) / 256) * 256;
#define SCAE_CELL(CC)                                                                         \
  return nt320 ? launch_bwd_c<CC, 320>(d, g, x, lse_post, lse_prior, g_logprob, g_tile,       \
                                       lp_tiles, lp_ppb, g_templates, g_alpha_partial,        \
                                       g_pose, g_presence, g_bg_image, g_scalar_partial, st)  \
               : launch_bwd_c<CC, 256>(d, g, x, lse_post, lse_prior, g_logprob, g_tile,       \
                                       lp_tiles, lp_ppb, g_templates, g_alpha_partial,        \
                                       g_pose, g_presence, g_bg_image, g_scalar_partial, st)
  switch (d->C) {
    case 1: SCAE_CELL(1);
    case 2: SCAE_CELL(2);
    case 3: SCAE_CELL(3);
    case 4: SCAE_CELL(4);
    default: return SCAE_ERR_UNSUPPORTED;
  }
#undef SCAE_CELL
}
}  // namespace scae_k1
