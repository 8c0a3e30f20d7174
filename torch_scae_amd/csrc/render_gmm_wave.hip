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
  hipLaunchKernelGGL((logprob_wave_kernel<C>), dim3(t.tiles, d->B), dim3(t.ppb), lds, st, *d, x,
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
  hipLaunchKernelGGL((render_wave_kernel<C>), dim3(g.groups, d->B), dim3(g.threads), g.lds, st, *d,
                     tt, ml, g.KG);
  return scae_launch_status();
}

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
template <int C> struct RecOf { static constexpr int RS = C == 1 ? 4 : ((C + 4) & ~1); };

__device__ __forceinline__ int fdiv(int n, float inv) { return (int)(((float)n + 0.5f) * inv); }

template <int C, int NTB>
__global__ __launch_bounds__(NTB) void bwd_cell_kernel(
    scae_decoder_desc d, const float *__restrict__ x, const float *__restrict__ lse_post,
    const float *__restrict__ lse_prior, const float *__restrict__ g_logprob,
    const float *__restrict__ g_tile, int lp_tiles, int lp_ppb,
    float *__restrict__ g_templates, float *__restrict__ g_alpha_partial,
    float *__restrict__ g_pose, float *__restrict__ g_presence,
    float *__restrict__ g_bg_image, float *__restrict__ g_scalar_partial, int chunk_rows,
    int max_items, int item_budget) {
  constexpr int TX = TexelOf<C>::TX, NV = C + 1, RS = RecOf<C>::RS, NM = 4 * NV;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
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
      const float mlv = v[C] + lsp;
      const float sp = __expf(mlv - lse_prior[(size_t)b * HW + p]);
      float gtt[C], gml = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const size_t o = ((size_t)b * C + c) * HW + p;
        const float gc = g_tile ? g_tile[b * lp_tiles + fdiv(p, inv_ppb)] : g_logprob[o];
        const float diff = x[o] - v[c];
        const float w = __expf(fmaf(diff * diff, -hvar, knorm) + mlv - lse_post[o]);
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
  hipLaunchKernelGGL((bwd_cell_kernel<C, NTB>), dim3(d->M + 1, d->B), dim3(NTB), g.lds, st, *d, x,
                     lse_post, lse_prior, g_logprob, g_tile, lp_tiles, lp_ppb, g_templates,
                     g_alpha_partial, g_pose, g_presence, g_bg_image, g_scalar_partial,
                     g.chunk_rows, g.max_items, g.item_budget);
  return scae_launch_status();
}
}  // namespace

size_t logprob_wave_lds(const scae_decoder_desc *d) {
  if (!d->templates_alpha || d->C < 1 || d->C > 4) return 0;   // alpha-channel mode only
  const int TX = d->C == 1 ? 2 : (d->C <= 3 ? 4 : 8);
  const size_t bytes = sizeof(float) * ((size_t)d->M * pad_elems(d->th, d->tw) * TX +
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
