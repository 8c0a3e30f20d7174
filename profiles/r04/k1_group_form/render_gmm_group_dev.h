// Backward of the fused likelihood (K1), component-group form: one workgroup per (image b,
// group of G components) -- part_decoder.py:174-237, distributions.py:41-44 differentiated.
//
// The cell-gather algorithm is bwd_cell_body's (render_gmm_wave_dev.h: a pixel's taps are the
// corners of the texel CELL its sample falls in, so the texel gradients are per-cell moments
// of the pixel gradients); what changes is who shares what.  With a workgroup per component
// (3 200 workgroups of 6 pixels per thread at cfg-2) the per-workgroup set-up -- scalars,
// pose constants, staging, the 8-value block reduction, phase 2's box and split, phase 3 --
// was as long as the pixel loop (~800 of ~1 800 instructions per thread), every lane of every
// wave computing the same workgroup-uniform values, and a pixel's x / log-sum-exp values /
// coordinates / incoming gradient were loaded once per component.  Here:
//   * a lane owns ONE pixel of a chunk of NTB consecutive pixels and walks the group's G
//     components with the component index wave-uniform (its pose and presence term are two
//     broadcast LDS reads): the pixel's x, both LSE values, coordinates and gradient are
//     loaded once, in the log2 domain where exponentials follow;
//   * everything per component that is not per pixel is computed by ONE lane (the affine
//     model of phase 2, per chunk its cell box and split) and read back from LDS;
//   * phase 2's (component, cell, part) items of all G components share one item space
//     (lanes stay filled where one component's ~30 cells of a chunk would leave most of a
//     workgroup idle), walk their pixels four at a time without a data-dependent branch, and
//     meet the other parts of their cell through a DPP butterfly inside a 16-lane row -- no
//     partial-moment arrays, no fold passes, phase 3 reads one record per cell;
//   * the texel sums live in registers of their (component, texel) lane across the chunks,
//     the pose / presence / scale sums in registers of the pixel lanes; one block reduction
//     per group.
// One writer per address, fixed summation order: bit-reproducible.
#pragma once
#include "common.h"
#include "render_gmm_dev.h"
#include "render_gmm_wave_dev.h"

namespace scae_k1 {
namespace {
#ifndef SCAE_GROUP_ROUNDS
#define SCAE_GROUP_ROUNDS 1   // lane rounds of phase 2 the per-component item budget aims at
#endif
#ifndef SCAE_GROUP_ABL
#define SCAE_GROUP_ABL 0   // timing ablations: 1 phase 1 only, 2 phase 2 without pixel walks, 3 no phase 3
#endif
constexpr int kCoef = 32;     // floats per component in s_cf
constexpr int kPartMax = 16;  // parts of one cell: one DPP row
constexpr int kGroupRounds = 2;   // lane rounds of one phase-2 pass (register-held moments)

// sum over the aligned group of `P` (1, 2, 4, 8, 16; per lane) consecutive lanes
__device__ __forceinline__ float part_sum(float v, int P) {
  const float a = scae::dpp_f<0xB1>(v);    // lane ^ 1
  v += P >= 2 ? a : 0.f;
  const float b = scae::dpp_f<0x4E>(v);    // lane ^ 2
  v += P >= 4 ? b : 0.f;
  const float c = scae::dpp_f<0x141>(v);   // row_half_mirror: 7 - lane within 8
  v += P >= 8 ? c : 0.f;
  const float e = scae::dpp_f<0x140>(v);   // row_mirror: 15 - lane
  v += P >= 16 ? e : 0.f;
  return v;
}

template <int C, int G, int NTB>
struct GroupLds {
  static constexpr int TX = TexelOf<C>::TX, NV = C + 1, RS = RecOf<C>::RS, NM = 4 * NV;
  static constexpr int SLAB = NTB * (RS + 1);   // records + cell ids of one component's chunk
  // moment records of a component's cells are written over its own (dead) slab
  static constexpr int CELLS_MAX = SLAB / NM;
  static constexpr int NW = NTB / 64;
};

// Workgroup (group `grp` of image b) as a device function (NTB threads, `smem`: its dynamic
// LDS): its own launch (render_gmm_wave.hip), or a block range of the launch it shares with
// the capsule likelihood's backward (render_bwd_likelihood.hip).  grp == n_groups: the
// background component.
template <int C, int G, int NTB>
__device__ __forceinline__ void bwd_group_body(
    const scae_decoder_desc &d, const float *__restrict__ x, const float *__restrict__ lse_post,
    const float *__restrict__ lse_prior, const float *__restrict__ g_logprob,
    const float *__restrict__ g_tile, int lp_tiles, int lp_ppb,
    float *__restrict__ g_templates, float *__restrict__ g_alpha_partial,
    float *__restrict__ g_pose, float *__restrict__ g_presence,
    float *__restrict__ g_bg_image, float *__restrict__ g_scalar_partial, float *smem,
    int grp, int n_groups, int b) {
  using L = GroupLds<C, G, NTB>;
  constexpr int TX = L::TX, NV = L::NV, RS = L::RS, NM = L::NM, SLAB = L::SLAB;
  constexpr int RMAX = kGroupRounds;   // phase-2 lane rounds (group_fits checks the item count)
  const int tid = threadIdx.x;
  const int M = d.M, K = M + 1, W = d.W, H = d.H, HW = H * W, tw = d.tw, th = d.th;
  const int tsz = th * tw, psz = pad_elems(th, tw), pw = pad_w(tw);
  const bool has_scale = d.out_scale != nullptr;
  const float sigma = has_scale ? softplusf_(d.out_scale[0]) + 1e-4f : 1.f;
  const float inv_sigma = has_scale ? 1.f / sigma : 1.f, inv_var = inv_sigma * inv_sigma;
  const float knorm = (has_scale ? -logf(sigma) : 0.f) - scae::kHalfLog2Pi, hvar = 0.5f * inv_var;
  const float inv_wf = 1.f / (float)W, inv_hf = 1.f / (float)H;
  const float inv_ppb = __builtin_amdgcn_rcpf((float)lp_ppb);   // (quotients of small integers)

  float *s_pl = smem;                                             // G padded planes
  float *s_cf = s_pl + (((size_t)G * psz * TX + 3) & ~(size_t)3);   // G x kCoef
  float *s_slab = s_cf + G * kCoef;                               // G x SLAB
  float *s_red = s_slab + (size_t)G * SLAB;                       // 8 G x NW

  if (grp == n_groups) {   // background component: no texels, three scalar sums
    const int k = M;
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

  const int k0 = grp * G, nk = min(G, M - k0);   // components [k0, k0 + nk) (workgroup-uniform)
  const float hx = 0.5f * (float)tw, hy = 0.5f * (float)th;
  const float txf = (float)tw, tyf = (float)th, pwf = (float)pw;
  constexpr float kSlack = 0.02f;

  // ---- set-up: the group's padded planes; per component, by one lane, the affine model ---
  stage_planes<C>(s_pl, d, b, k0, nk, 1.f, tid, NTB);
  if (tid < G) {
    float *cf = s_cf + tid * kCoef;
    if (tid < nk) {
      const int k = k0 + tid;
      const float *pa = d.pose + ((size_t)b * M + k) * 6;
#pragma unroll
      for (int i = 0; i < 6; ++i) cf[i] = pa[i];
      cf[6] = d.presence ? log_safe(d.presence[b * M + k]) : 0.f;
      cf[7] = 0.f;
      const float A0 = hx * pa[0], A1 = hx * pa[1], A2 = hx * (pa[2] + 1.f) - 0.5f;
      const float A3 = hy * pa[3], A4 = hy * pa[4], A5 = hy * (pa[5] + 1.f) - 0.5f;
      // the same map over pixel indices (j, i):  ix = ax j + bx i + c0x,  iy = ay j + by i + c0y
      const float ax = A0 * 2.f * inv_wf, bx = A1 * 2.f * inv_hf;
      const float c0x = fmaf(A0, inv_wf - 1.f, fmaf(A1, inv_hf - 1.f, A2));
      const float ay = A3 * 2.f * inv_wf, by = A4 * 2.f * inv_hf;
      const float c0y = fmaf(A3, inv_wf - 1.f, fmaf(A4, inv_hf - 1.f, A5));
      const float det = ax * by - bx * ay;
      // The inverse map only has to give a SUPERSET of a cell's pixels (membership is the
      // parked cell id).  Phase 1's positions and this affine model agree to ~1e-5 texels, so
      // an interval bound is off by 1e-5 / |slope| pixels: with slopes above 1e-3 a slack of
      // 0.02 pixels covers it; flatter maps (and NaNs) take the whole row / all rows.
      const float span = fabsf(ax) + fabsf(ay) + fabsf(bx) + fabsf(by);
      const bool det_ok = fabsf(det) > 1e-3f * span && fabsf(det) < 1e30f;   // (false for NaN too)
      const float inv_det = det_ok ? 1.f / det : 0.f;
      const bool ax_ok = fabsf(ax) > 1e-3f && fabsf(ax) < 1e30f;
      const bool ay_ok = fabsf(ay) > 1e-3f && fabsf(ay) < 1e30f;
      cf[8] = ax, cf[9] = bx, cf[10] = c0x, cf[11] = ay, cf[12] = by, cf[13] = c0y;
      cf[14] = inv_det;
      cf[15] = -ay * inv_det, cf[16] = ax * inv_det;   // d(row) per unit ix / iy
      cf[17] = ax_ok ? 1.f / ax : 0.f, cf[18] = ay_ok ? 1.f / ay : 0.f;
      cf[19] = __int_as_float((det_ok ? 1 : 0) | (ax_ok ? 2 : 0) | (ay_ok ? 4 : 0));
    } else {
#pragma unroll
      for (int i = 0; i < kCoef; ++i) cf[i] = 0.f;
    }
  }

  float acc[G][8];   // per component: 6 pose sums, sum of d/d(mixing logit), sigma
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[g][i] = 0.f;
  // texel sums of this lane's (component, texel) slots, across the chunks
  constexpr int TSLOTS_MAX = 4;
  float tex[TSLOTS_MAX][NV];
#pragma unroll
  for (int s = 0; s < TSLOTS_MAX; ++s)
#pragma unroll
    for (int q = 0; q < NV; ++q) tex[s][q] = 0.f;
  const int ntex = nk * tsz;   // (host: G * tsz <= TSLOTS_MAX * NTB)
  const float inv_tsz = __builtin_amdgcn_rcpf((float)tsz), inv_tw = __builtin_amdgcn_rcpf((float)tw);

  for (int p0 = 0; p0 < HW; p0 += NTB) {
    const int np = min(NTB, HW - p0);
    const int r0 = fdiv(p0, inv_wf), r1 = fdiv(p0 + np - 1, inv_wf) + 1;   // rows [r0, r1)
    __syncthreads();   // set-up / the previous chunk's phase 3 done
    // ---- per component, by lane g of the first wave: this chunk's cell box and the split
    // of a cell over lanes; the item offsets are a prefix sum through the wave's registers ---
    if (tid < 64) {
      const int g = tid < G ? tid : G - 1;
      float *cf = s_cf + g * kCoef;
      int cxlo = 0, cylo = 0, ncx = 0, ncy = 0, S = 1, Gs = 1;
      if (g < nk) {
        const float4 c8 = *reinterpret_cast<const float4 *>(cf + 8);     // ax bx c0x ay
        const float4 c12 = *reinterpret_cast<const float4 *>(cf + 12);   // by c0y inv_det dix
        const float4 c16 = *reinterpret_cast<const float4 *>(cf + 16);   // diy inv_ax inv_ay flags
        const float ax = c8.x, bx = c8.y, c0x = c8.z, ay = c8.w, by = c12.x, c0y = c12.y;
        const int fl = __float_as_int(c16.w);
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
        const int ncells = ncx * ncy;
        if (ncells > 0) {
          // parts per cell: S row slices x Gs row segments, powers of two, <= 16 lanes --
          // as many as keep the group's items within SCAE_GROUP_ROUNDS lane rounds
          constexpr int budget = (SCAE_GROUP_ROUNDS * NTB) / G;
          const int rows_cell = (fl & 1) ? min(r1 - r0, (int)fminf(fabsf(c12.w) + fabsf(c16.x), 1e4f) + 2)
                                         : r1 - r0;
          const float run = fminf((fl & 2) ? fabsf(c16.y) : 1e4f, (fl & 4) ? fabsf(c16.z) : 1e4f);
          const int jspan = min(W, (int)fminf(run, 1e4f) + 2);   // pixels of a row inside one cell
          int P = 1;
          while (2 * P <= kPartMax && 2 * P * ncells <= budget) P *= 2;
          while (S * 2 <= min(P, rows_cell)) S *= 2;
          const int gmax = max(1, min(P / S, jspan >> 2));
          while (Gs * 2 <= gmax) Gs *= 2;
        }
      }
      const int ncells = ncx * ncy, mine = tid < G ? (ncells * S * Gs + 15) & ~15 : 0;
      // (a cell's parts stay inside a DPP row: every component starts on a multiple of 16)
      int off = 0;
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int nj = __builtin_amdgcn_readlane(mine, j);
        off += j < tid ? nj : 0;
      }
      if (tid < G) {
        *reinterpret_cast<float4 *>(cf + 20) = make_float4(__int_as_float(cxlo), __int_as_float(cylo),
                                                           __int_as_float(ncx), __int_as_float(ncells));
        *reinterpret_cast<float4 *>(cf + 24) = make_float4(__int_as_float(S), __int_as_float(Gs),
                                                           __int_as_float(off), __int_as_float(off + mine));
      }
    }

    // ---- phase 1: lane = pixel, all components of the group -----------------------------
    const bool live = tid < np;
    const int p = p0 + (live ? tid : 0);
    const int pi = fdiv(p, inv_wf), pj = p - pi * W;
    const float xn = (float)(2 * pj + 1) * inv_wf - 1.f, yn = (float)(2 * pi + 1) * inv_hf - 1.f;
    float xv[C], gcv[C], lpk2[C];
    const float gsel = live ? 1.f : 0.f;
    const float lprior2 = kLog2e * lse_prior[(size_t)b * HW + p];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const size_t o = ((size_t)b * C + c) * HW + p;
      xv[c] = x[o];
      gcv[c] = gsel * (g_tile ? g_tile[b * lp_tiles + fdiv(p, inv_ppb)] : g_logprob[o]);
      lpk2[c] = kLog2e * (knorm - lse_post[o]);
    }
    const float hvar2 = kLog2e * hvar;
    __syncthreads();   // planes / coefficients visible; (the records' previous readers are done)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g < nk) {   // (workgroup-uniform)
        const float4 ca = *reinterpret_cast<const float4 *>(s_cf + g * kCoef);
        const float4 cb = *reinterpret_cast<const float4 *>(s_cf + g * kCoef + 4);
        const float pa6[6] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y};
        // the sample position in the reference's own operation order (affine_grid, then
        // grid_sample's un-normalisation): d/d(position) jumps at cell boundaries, so a
        // pixel within round-off of one must land on the side the reference puts it
        float ix, iy;
        tex_pos(pa6, xn, yn, tw, th, ix, iy);
        ix = fminf(fmaxf(ix, -2.f), txf);
        iy = fminf(fmaxf(iy, -2.f), tyf);
        const float x0f = floorf(ix), y0f = floorf(iy), fx = ix - x0f, fy = iy - y0f;
        const int idx = (int)fmaf(y0f, pwf, x0f);
        const float *q0 = s_pl + ((size_t)g * psz + (2 * pw + 2) + idx) * TX, *q1 = q0 + pw * TX;
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
        const float mlv2 = kLog2e * (v[C] + cb.z);
        const float sp = ex2(mlv2 - lprior2);
        float gtt[C], gml = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float diff = xv[c] - v[c];
          const float w = ex2(fmaf(diff * diff, -hvar2, mlv2 + lpk2[c]));
          const float gw = gcv[c] * w;
          gtt[c] = gw * diff * inv_var;
          gml += gcv[c] * (w - sp);
          if (has_scale) acc[g][7] += gw * (diff * diff * inv_var - 1.f) * inv_sigma;
        }
        float gix = gml * vdx[C], giy = gml * vdy[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
          gix = fmaf(gtt[c], vdx[c], gix);
          giy = fmaf(gtt[c], vdy[c], giy);
        }
        acc[g][0] = fmaf(gix, xn, acc[g][0]);
        acc[g][1] = fmaf(gix, yn, acc[g][1]);
        acc[g][2] += gix;
        acc[g][3] = fmaf(giy, xn, acc[g][3]);
        acc[g][4] = fmaf(giy, yn, acc[g][4]);
        acc[g][5] += giy;
        acc[g][6] += gml;
        float *rec = s_slab + (size_t)g * SLAB + (size_t)tid * RS;
        if (C == 1) {
          *reinterpret_cast<float4 *>(rec) = make_float4(gtt[0], gml, fx, fy);
        } else {
#pragma unroll
          for (int c = 0; c < C; ++c) rec[c] = gtt[c];
          rec[C] = gml, rec[C + 1] = fx, rec[C + 2] = fy;
        }
        // (a lane beyond the image's last pixel parks a cell id no item asks for)
        reinterpret_cast<int *>(s_slab + (size_t)g * SLAB + NTB * RS)[tid] = live ? idx : -(1 << 20);
      }
    }
    __syncthreads();

    // ---- phase 2: lane = (component, cell, part), all components in one item space --------
    // The cells' moments are held in registers until every record has been read, then
    // written over the (dead) slabs: RMAX lane rounds per pass.  One pass takes the
    // components [gb, ge) whose items fit; normally that is the whole group.
    for (int gb = SCAE_GROUP_ABL == 1 ? nk : 0; gb < nk;) {   // (workgroup-uniform)
      const int first = __float_as_int(s_cf[gb * kCoef + 26]);
      int ge = gb + 1;
      while (ge < nk && __float_as_int(s_cf[ge * kCoef + 27]) - first <= RMAX * NTB) ++ge;
      const int total = __float_as_int(s_cf[(ge - 1) * kCoef + 27]);   // end of the pass's items
      bool split = false;
      for (int j = gb; j < ge; ++j)
        split = split || __float_as_int(s_cf[j * kCoef + 24]) * __float_as_int(s_cf[j * kCoef + 25]) > 1;
      float mom[RMAX][NM];
      int mslot[RMAX];
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        mslot[r] = -1;
#pragma unroll
        for (int q = 0; q < NM; ++q) mom[r][q] = 0.f;
        if (first + r * NTB < total) {   // (workgroup-uniform)
          const int item = first + r * NTB + tid;
          int g = gb;
#pragma unroll
          for (int j = 1; j < G; ++j)
            g += (j > gb && j < ge && item >= __float_as_int(s_cf[j * kCoef + 26])) ? 1 : 0;
          const float *cf = s_cf + g * kCoef;
          const float4 c20 = *reinterpret_cast<const float4 *>(cf + 20);   // cxlo cylo ncx ncells
          const float4 c24 = *reinterpret_cast<const float4 *>(cf + 24);   // S Gs first end
          const int ncells = __float_as_int(c20.w), S = __float_as_int(c24.x), Gs = __float_as_int(c24.y);
          const int P = S * Gs, local = item - __float_as_int(c24.z);
          const int cell = local >> __builtin_ctz(P), part = local & (P - 1);   // (powers of two)
          const bool act = item < total && cell < ncells;
          if (act) {
            const int ncx = __float_as_int(c20.z);
            const int sl = part >> __builtin_ctz(Gs), seg = part & (Gs - 1);
            const int cyi = fdiv(cell, __builtin_amdgcn_rcpf((float)ncx)), cxi = cell - cyi * ncx;
            const float cxf = (float)(__float_as_int(c20.x) + cxi), cyf = (float)(__float_as_int(c20.y) + cyi);
            const int myid = (int)fmaf(cyf, pwf, cxf);
            const float4 c8 = *reinterpret_cast<const float4 *>(cf + 8);     // ax bx c0x ay
            const float4 c12 = *reinterpret_cast<const float4 *>(cf + 12);   // by c0y inv_det dix
            const float4 c16 = *reinterpret_cast<const float4 *>(cf + 16);   // diy inv_ax inv_ay flags
            const float ax = c8.x, bx = c8.y, ay = c8.w, by = c12.x;
            const float ux = cxf - c8.z, uy = cyf - c12.y;
            const int fl = __float_as_int(c16.w);
            const float inv_ax = c16.y, inv_ay = c16.z;
            int ilo = r0, ihi = r1 - 1;
            if (fl & 1) {   // rows that cross the cell's parallelogram
              const float dix = c12.w, diy = c16.x;
              const float i00 = (ax * uy - ay * ux) * c12.z;
              const float imin = i00 + fminf(dix, 0.f) + fminf(diy, 0.f);
              const float imax = i00 + fmaxf(dix, 0.f) + fmaxf(diy, 0.f);
              ilo = max(ilo, (int)fminf(fmaxf(ceilf(imin - kSlack), -1.f), (float)H));
              ihi = min(ihi, (int)fmaxf(fminf(floorf(imax + kSlack), (float)H), -1.f));
            }
            const float *slab = s_slab + (size_t)g * SLAB;
            const int *ids = reinterpret_cast<const int *>(slab + NTB * RS);
            const float inv_Gs = __builtin_amdgcn_rcpf((float)Gs);
            float m[NM];
#pragma unroll
            for (int q = 0; q < NM; ++q) m[q] = 0.f;
            for (int i = ilo + sl; i <= ihi; i += S) {
              // along the row  ix - cx = ax j - rx,  iy - cy = ay j - ry: the cell's pixels
              // are the j with both in [0, 1)
              const float rx = fmaf(-bx, (float)i, ux), ry = fmaf(-by, (float)i, uy);
              float lo = 0.f, hi = (float)(W - 1);
              if (fl & 2) {
                const float t0 = rx * inv_ax, t1 = t0 + inv_ax;
                lo = fmaxf(lo, ceilf(fminf(t0, t1) - kSlack));
                hi = fminf(hi, floorf(fmaxf(t0, t1) + kSlack));
              }
              if (fl & 4) {
                const float t0 = ry * inv_ay, t1 = t0 + inv_ay;
                lo = fmaxf(lo, ceilf(fminf(t0, t1) - kSlack));
                hi = fminf(hi, floorf(fmaxf(t0, t1) + kSlack));
              }
              if (!(lo <= hi)) continue;
              int jl = (int)lo, jh = (int)hi;
              if (Gs > 1) {   // this lane's segment of the interval
                const int len = fdiv(jh - jl + Gs, inv_Gs);
                jl += seg * len;
                jh = min(jh, jl + len - 1);
              }
              // the chunk's pixels are [p0, p0 + np): its first / last row may be partial
              const int base = i * W - p0;
              const int pl0 = max(base + jl, 0), pl1 = min(base + jh, np - 1);
              // UB pixels at a time, no data-dependent branch: their loads are in flight
              // together (a lane's walk is a chain of dependent LDS round trips)
              constexpr int UB = C == 1 ? 4 : 2;
              for (int pl = SCAE_GROUP_ABL == 2 ? pl1 + 1 : pl0; pl <= pl1; pl += UB) {
                int idq[UB];
                float rq[UB][C == 1 ? 4 : NV + 2];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                  const int q = min(pl + u, pl1);
                  idq[u] = pl + u <= pl1 ? ids[q] : -(1 << 21);
                  if (C == 1) {
                    const float4 r4 = *reinterpret_cast<const float4 *>(slab + (size_t)q * RS);
                    rq[u][0] = r4.x, rq[u][1] = r4.y, rq[u][2] = r4.z, rq[u][3] = r4.w;
                  } else {
#pragma unroll
                    for (int e = 0; e < NV + 2; ++e) rq[u][e] = slab[(size_t)q * RS + e];
                  }
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                  const float sel = idq[u] == myid ? 1.f : 0.f;
                  const float fxq = rq[u][NV], fyq = rq[u][NV + 1], fxy = fxq * fyq;
#pragma unroll
                  for (int e = 0; e < NV; ++e) {
                    const float ge_ = sel * rq[u][e];
                    m[4 * e] += ge_;
                    m[4 * e + 1] = fmaf(ge_, fxq, m[4 * e + 1]);
                    m[4 * e + 2] = fmaf(ge_, fyq, m[4 * e + 2]);
                    m[4 * e + 3] = fmaf(ge_, fxy, m[4 * e + 3]);
                  }
                }
              }
            }
#pragma unroll
            for (int q = 0; q < NM; ++q) mom[r][q] = m[q];
            if (part == 0) mslot[r] = g * SLAB + cell * NM;
          }
          // the parts of a cell meet inside their 16-lane row (inactive lanes add zeros)
          if (split) {   // (workgroup-uniform)
#pragma unroll
            for (int q = 0; q < NM; ++q) mom[r][q] = part_sum(mom[r][q], act ? P : 1);
          }
        }
      }
      __syncthreads();   // every record of the pass has been read: the slabs take the moments
#pragma unroll
      for (int r = 0; r < RMAX; ++r)
        if (mslot[r] >= 0) {
#pragma unroll
          for (int q = 0; q < NM; q += 4)
            *reinterpret_cast<float4 *>(s_slab + mslot[r] + q) =
                make_float4(mom[r][q], mom[r][q + 1], mom[r][q + 2], mom[r][q + 3]);
        }
      __syncthreads();

      // ---- phase 3: lane = (component, texel): the corner terms of its four cells ---------
#pragma unroll
      for (int s = 0; s < TSLOTS_MAX; ++s) {
        const int t = s * NTB + tid;
        if (SCAE_GROUP_ABL != 3 && s * NTB < ntex && t < ntex) {
          const int g = fdiv(t, inv_tsz), e = t - g * tsz;
          if (g >= gb && g < ge) {
            const int ty = fdiv(e, inv_tw), tx = e - ty * tw;
            const float4 c20 = *reinterpret_cast<const float4 *>(s_cf + g * kCoef + 20);
            const int cxlo = __float_as_int(c20.x), cylo = __float_as_int(c20.y);
            const int ncx = __float_as_int(c20.z), ncells = __float_as_int(c20.w);
            const int ncy = ncx > 0 ? fdiv(ncells, __builtin_amdgcn_rcpf((float)ncx)) : 0;
#pragma unroll
            for (int corner = 0; corner < 4; ++corner) {
              const int dy = corner >> 1, dx = corner & 1;
              const int cxi = tx - dx - cxlo, cyi = ty - dy - cylo;
              if (cxi < 0 || cxi >= ncx || cyi < 0 || cyi >= ncy) continue;
              const float *mp = s_slab + (size_t)g * SLAB + (size_t)(cyi * ncx + cxi) * NM;
#pragma unroll
              for (int q = 0; q < NV; ++q) {
                const float4 mm = *reinterpret_cast<const float4 *>(mp + 4 * q);
                tex[s][q] += corner == 0 ? ((mm.x - mm.y) - mm.z) + mm.w
                                         : (corner == 1 ? mm.y - mm.w : (corner == 2 ? mm.z - mm.w : mm.w));
              }
            }
          }
        }
      }
      gb = ge;
      if (gb < nk) __syncthreads();   // (the next pass's moments overwrite nothing this one
                                      // reads -- other slabs -- but keep the phases apart)
    }
  }

  // ---- the group's outputs --------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < TSLOTS_MAX; ++s) {
    const int t = s * NTB + tid;
    if (s * NTB < ntex && t < ntex) {
      const int g = fdiv(t, inv_tsz), e = t - g * tsz, k = k0 + g;
#pragma unroll
      for (int c = 0; c < C; ++c) g_templates[(((size_t)b * M + k) * C + c) * tsz + e] = tex[s][c];
      g_alpha_partial[((size_t)b * M + k) * tsz + e] = tex[s][C];
    }
  }
  __syncthreads();   // (s_red overlaps nothing, but phase 3 of the last chunk must be done
                     // before block_sum's barrier pattern starts)
  float flat[G * 8];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int i = 0; i < 8; ++i) flat[g * 8 + i] = acc[g][i];
  scae::block_sum<G * 8, NTB>(flat, s_red);
  if (tid == 0) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g >= nk) continue;
      const int k = k0 + g;
      const float *a = flat + g * 8;
      float *gp = g_pose + ((size_t)b * M + k) * 6;
      // d ix / d a0 = hx xn, d ix / d a2 = hx, ...
      gp[0] = hx * a[0], gp[1] = hx * a[1], gp[2] = hx * a[2];
      gp[3] = hy * a[3], gp[4] = hy * a[4], gp[5] = hy * a[5];
      if (g_presence && d.presence)
        g_presence[b * M + k] = a[6] * scae::log_safe_grad(d.presence[b * M + k]);
      float *sp = g_scalar_partial + ((size_t)b * K + k) * 4;
      sp[0] = sp[1] = sp[2] = 0.f;
      sp[3] = has_scale ? a[7] * scae::softplus_grad(d.out_scale[0]) : 0.f;
    }
  }
}

// Launch geometry of the group form for a decoder shape (G = 0: not covered -- the
// workgroup-per-component form takes over).
template <int C, int G, int NTB>
size_t group_lds(const scae_decoder_desc *d) {
  using L = GroupLds<C, G, NTB>;
  const size_t floats = (((size_t)G * pad_elems(d->th, d->tw) * L::TX + 3) & ~(size_t)3) + G * kCoef +
                        (size_t)G * L::SLAB + (size_t)8 * G * L::NW;
  return floats * sizeof(float);
}
template <int C, int G, int NTB>
bool group_fits(const scae_decoder_desc *d) {
  using L = GroupLds<C, G, NTB>;
  if (!d->templates_alpha || d->template_repeat > 1 || d->C != C) return false;
  const int cells = (d->tw + 1) * (d->th + 1), tsz = d->th * d->tw;
  const int budget = (SCAE_GROUP_ROUNDS * NTB) / G;
  // a component's items: its cells x a power-of-two split within the budget, padded to 16
  const int items = (cells > budget ? cells : budget) + 15;
  // (a pass of phase 2 takes as many components as fit kGroupRounds lane rounds: one must)
  return cells <= L::CELLS_MAX && G * tsz <= 4 * NTB && items <= kGroupRounds * NTB &&
         group_lds<C, G, NTB>(d) <= 80 * 1024;
}

struct GroupChoice {
  int G, NTB, n_groups;
  size_t lds;
};
// Candidates per channel count: (G, NTB) in {GA, GB} x {320, 256}.  NTB: the chunking of
// the image's pixels with the fewest idle lanes (40 x 40 = 5 x 320, 32 x 32 = 4 x 256); G:
// the fewest padded component slots, then the larger group (SCAE_GROUP_G / SCAE_GROUP_NTB
// override, for measurements).
template <int C> struct GroupCand { static constexpr int GA = C == 1 ? 6 : (C == 3 ? 4 : 2), GB = C == 1 ? 4 : 2; };

template <int C, int G, int NTB>
void group_consider(const scae_decoder_desc *d, int want_g, int want_ntb, GroupChoice &best,
                    long &best_cost) {
  if ((want_g && want_g != G) || (want_ntb && want_ntb != NTB) || !group_fits<C, G, NTB>(d)) return;
  const int HW = d->H * d->W, chunks = (HW + NTB - 1) / NTB, groups = (d->M + G - 1) / G;
  // idle pixel lanes first, padded component slots second, small groups last
  const long cost = ((long)(chunks * NTB - HW) * 64 + (groups * G - d->M)) * 8 + (8 - G);
  if (best.G == 0 || cost < best_cost) {
    best = {G, NTB, groups, group_lds<C, G, NTB>(d)};
    best_cost = cost;
  }
}
template <int C>
GroupChoice group_choice_c(const scae_decoder_desc *d) {
  static const int want_g = getenv("SCAE_GROUP_G") ? atoi(getenv("SCAE_GROUP_G")) : 0;
  static const int want_ntb = getenv("SCAE_GROUP_NTB") ? atoi(getenv("SCAE_GROUP_NTB")) : 0;
  GroupChoice best = {0, 0, 0, 0};
  if (want_g < 0) return best;   // (SCAE_GROUP_G=-1: the workgroup-per-component form)
  long cost = 0;
  group_consider<C, GroupCand<C>::GA, 320>(d, want_g, want_ntb, best, cost);
  group_consider<C, GroupCand<C>::GA, 256>(d, want_g, want_ntb, best, cost);
  if (GroupCand<C>::GB != GroupCand<C>::GA) {
    group_consider<C, GroupCand<C>::GB, 320>(d, want_g, want_ntb, best, cost);
    group_consider<C, GroupCand<C>::GB, 256>(d, want_g, want_ntb, best, cost);
  }
  return best;
}
inline GroupChoice group_choice(const scae_decoder_desc *d) {
  switch (d->C) {
    case 1: return group_choice_c<1>(d);
    case 2: return group_choice_c<2>(d);
    case 3: return group_choice_c<3>(d);
    case 4: return group_choice_c<4>(d);
    default: return GroupChoice{0, 0, 0, 0};
  }
}

}  // namespace
}  // namespace scae_k1
