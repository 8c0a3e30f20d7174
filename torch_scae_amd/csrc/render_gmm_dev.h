// Device helpers shared by the K1 kernels (render_gmm.hip: render / generic likelihood /
// backward; render_gmm_wave.hip: the wave-per-pixel-block likelihood forward).
#pragma once
#include "common.h"

namespace scae_k1 {
using scae::log_safe;
using scae::sigmoidf_;
using scae::softplusf_;

// affine_grid (align_corners=False) + grid_sample's un-normalisation in the reference's
// operation order: g = theta [x, y, 1];  ix = ((gx + 1) w - 1) / 2
__device__ __forceinline__ void tex_pos(const float *a, float xn, float yn, int tw, int th,
                                        float &ix, float &iy) {
  const float gx = a[0] * xn + a[1] * yn + a[2];
  const float gy = a[3] * xn + a[4] * yn + a[5];
  ix = ((gx + 1.f) * tw - 1.f) * 0.5f;
  iy = ((gy + 1.f) * th - 1.f) * 0.5f;
}

// ---- zero-padded template planes in LDS ------------------------------------------
// A (th x tw) plane is staged as (th+4) x (tw+4) with the texels at offset (2, 2) and
// zeros around them.  With the sampling position clamped to [-2, tw] x [-2, th]
// every bilinear tap is then an in-range LDS read and an outside tap reads 0 -- the
// zero padding of grid_sample without per-tap masks, index clamps or compares (the
// unpadded formulation above spends ~2/3 of its instructions on those).  A
// position beyond the clamp has all four taps outside either way; NaN clamps to -2.
__host__ __device__ inline int pad_w(int tw) { return tw + 4; }
__host__ __device__ inline int pad_elems(int th, int tw) { return (th + 4) * (tw + 4); }

// template set of image b: consecutive groups of `template_repeat` images share one
// (stacked_capsule_auto_encoder.py:188-195 decodes every object capsule's votes with the
// image's templates: B*O virtual images, B template sets)
__device__ __forceinline__ int tb(const scae_decoder_desc &d, int b) {
  return d.template_repeat > 1 ? b / d.template_repeat : b;
}

struct Scalars {
  float sigma, inv_var, log_sigma;  // Normal scale of every component
  float temperature;                // temperature mode only
  float bg_ml;                      // alpha mode: softplus(bg_mixing_logit)
  float bg_val;                     // sigmoid(bg_value) when no bg_image
};

__device__ __forceinline__ Scalars load_scalars(const scae_decoder_desc &d) {
  Scalars s;
  s.sigma = d.out_scale ? softplusf_(d.out_scale[0]) + 1e-4f : 1.f;  // :220-223
  s.inv_var = 1.f / (s.sigma * s.sigma);
  s.log_sigma = logf(s.sigma);
  s.temperature = d.templates_alpha ? 1.f
                                    : softplusf_(d.temperature_logit[0] + .5f) + 1e-4f;
  s.bg_ml = d.templates_alpha ? softplusf_(d.bg_mixing_logit[0]) : 0.f;
  s.bg_val = d.bg_image ? 0.f : sigmoidf_(d.bg_value[0]);
  return s;
}


// Pixel tiling of the fused likelihood forward (and of the per-tile log-prob sums it
// hands to the loss tail): `tiles` workgroups per image of `ppb` consecutive pixels each.
struct LpTiling {
  int ksplit, ppb, tiles;
  bool wave;   // the wave-per-pixel-block kernel (render_gmm_wave.hip) covers this shape
};
LpTiling lp_tiling(const scae_decoder_desc *d);

// render_gmm_wave.hip: launch the wave-form forward (the caller checked lp_tiling(d).wave)
int launch_logprob_wave(const scae_decoder_desc *d, const LpTiling &t, const float *x,
                        float *log_prob, float *lse_post, float *lse_prior, float *block_sums,
                        hipStream_t st);
// LDS bytes the wave form needs for this shape (0: shape not covered)
size_t logprob_wave_lds(const scae_decoder_desc *d);

// render_gmm_wave.hip: the materialising forward with 16-byte quad stores (alpha-channel
// mode, H W a multiple of 4); render_wave_lds: LDS bytes, 0 when the shape is not covered.
size_t render_wave_lds(const scae_decoder_desc *d);
int launch_render_wave(const scae_decoder_desc *d, float *tt, float *ml, hipStream_t st);

// render_gmm_wave.hip: the fused backward in its cell-gather form (alpha-channel mode,
// gradient of the per-pixel log-prob or of its tile sums).  bwd_cell_lds: LDS bytes, 0 when
// the shape is not covered (the caller falls back to render_gmm.hip's kernels).
size_t bwd_cell_lds(const scae_decoder_desc *d);
int launch_bwd_cell(const scae_decoder_desc *d, const float *x, const float *lse_post,
                    const float *lse_prior, const float *g_logprob, const float *g_tile,
                    int lp_tiles, int lp_ppb, float *g_templates, float *g_alpha_partial,
                    float *g_pose, float *g_presence, float *g_bg_image,
                    float *g_scalar_partial, hipStream_t st);
}  // namespace scae_k1
