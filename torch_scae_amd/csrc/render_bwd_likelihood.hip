// Two independent backward kernels in ONE launch: the part decoder's likelihood backward
// (K1, bwd_cell_body: 3 200 workgroups of VALU work) and the capsule likelihood's backward
// (K4, likelihood_bwd_body: one workgroup per image, 128 at cfg-2, latency bound).  Both
// only wait for the loss tail's backward; kernels do not overlap on this stack, so K4's
// workgroups ride as the first block range of K1's launch, in a 256-thread form (its loops
// stride by the thread count).  The launch-uniform resources are K1's: 27 KB of LDS against
// K4's 8, 87 VGPRs.  12 us of the cfg-2 step disappear into the 55-64 us of the K1 backward.
#include "common.h"
#include "capsule_likelihood_dev.h"
#include "render_gmm_wave_dev.h"

namespace {
struct LkBwd {   // scae_likelihood_bwd_desc by value
  scae_lk::LkArgs a;
  const int64_t *winner_idx;
  const float *g_lpp, *g_winner, *g_winner_presence, *g_soft_winner, *g_soft_winner_presence,
      *g_posterior, *g_mlp, *g_mlogit;
  float *gvote, *gscale, *gvp, *gx, *gpresence, *gdummy;
};

template <int C>
__global__ __launch_bounds__(256, 5) void bwd_cell_likelihood_kernel(
    scae_decoder_desc d, const float *__restrict__ x, const float *__restrict__ lse_post,
    const float *__restrict__ lse_prior, const float *__restrict__ g_tile, int lp_tiles,
    int lp_ppb, float *__restrict__ g_templates, float *__restrict__ g_alpha_partial,
    float *__restrict__ g_pose, float *__restrict__ g_presence, float *__restrict__ g_bg_image,
    float *__restrict__ g_scalar_partial, int chunk_rows, int max_items, int item_budget,
    LkBwd lk, int n_lk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < n_lk) {   // (workgroup-uniform)
    scae_lk::likelihood_bwd_body<false, 256>(
        lk.a, lk.winner_idx, lk.g_lpp, lk.g_winner, lk.g_winner_presence, lk.g_soft_winner,
        lk.g_soft_winner_presence, lk.g_posterior, lk.g_mlp, lk.g_mlogit, lk.gvote, lk.gscale,
        lk.gvp, lk.gx, lk.gpresence, lk.gdummy, smem, blockIdx.x, n_lk);
    return;
  }
  const int id = (int)blockIdx.x - n_lk, K = d.M + 1, b = id / K, k = id - b * K;
  scae_k1::bwd_cell_body<C, 256>(d, x, lse_post, lse_prior, nullptr, g_tile, lp_tiles, lp_ppb,
                                 g_templates, g_alpha_partial, g_pose, g_presence, g_bg_image,
                                 g_scalar_partial, chunk_rows, max_items, item_budget, smem, k, b);
}

template <int C>
int launch(const scae_decoder_desc *d, const scae_k1::CellGeom &g, const scae_k1::LpTiling &lt,
           const float *x, const float *lse_post, const float *lse_prior, const float *g_tile,
           float *g_templates, float *g_alpha_partial, float *g_pose, float *g_presence,
           float *g_bg_image, float *g_scalar_partial, const LkBwd &lk, hipStream_t st) {
  const size_t lds_k = scae_lk::lk_lds(lk.a.O, lk.a.M, true);
  const size_t lds = g.lds > lds_k ? g.lds : lds_k;
  if (lds > 48 * 1024) {
    hipError_t e =
        hipFuncSetAttribute(reinterpret_cast<const void *>(bwd_cell_likelihood_kernel<C>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int n_lk = lk.a.B < 1024 ? lk.a.B : 1024;
  scae::launch((bwd_cell_likelihood_kernel<C>), dim3(n_lk + (d->M + 1) * d->B), dim3(256),
                     lds, st, *d, x, lse_post, lse_prior, g_tile, lt.tiles, lt.ppb, g_templates,
                     g_alpha_partial, g_pose, g_presence, g_bg_image, g_scalar_partial,
                     g.chunk_rows, g.max_items, g.item_budget, lk, n_lk);
  return scae_launch_status();
}
}  // namespace

extern "C" int scae_render_gmm_sums_bwd_likelihood_f32(
    const scae_decoder_desc *d, const float *x, const float *lse_post, const float *lse_prior,
    const float *g_tile_sums, float *g_templates, float *g_alpha_partial, float *g_pose,
    float *g_presence, float *g_bg_image, float *g_scalar_partial,
    const scae_likelihood_bwd_desc *k, void *stream) {
  SCAE_REQUIRE(d && k);
  const char *e = getenv("SCAE_FUSE_K1_K4_BWD");
  const scae_k1::CellGeom g = scae_k1::cell_geom(d);
  // (the rider must not raise the launch's LDS above what K1's own workgroups need: at 48 / 64
  // capsules its 39.5 KB against K1's 27 took a workgroup per CU from K1 -- B = 1024: 3.81 ms
  // per step merged, 3.78 apart, profiles/r06/cfg3_fuse.txt)
  const bool fits = !(e && *e == '0') && g.lds && (d->C == 1 || d->C == 3) && k->O <= 64 &&
                    scae_lk::lk_lds(k->O, k->M, true) <= g.lds && k->B > 0 && k->O > 0 &&
                    k->M > 0 && d->template_repeat <= 1 && d->bwd_resident <= 0;
  if (!fits) {   // two launches, same results
    int rc = scae_render_gmm_sums_bwd_f32(d, x, lse_post, lse_prior, g_tile_sums, g_templates,
                                          g_alpha_partial, g_pose, g_presence, g_bg_image,
                                          g_scalar_partial, stream);
    if (rc) return rc;
    return scae_capsule_likelihood_bwd_f32(
        k->vote, k->scale, k->vote_presence, k->dummy_vote, k->x, k->presence, k->posterior,
        k->winner_idx, k->g_lpp, k->g_winner, k->g_winner_presence, k->g_soft_winner,
        k->g_soft_winner_presence, k->g_posterior, k->g_mixing_log_prob, k->g_mixing_logit,
        k->gvote, k->gscale, k->gvote_presence, k->gx, k->gpresence, k->gdummy_partial, k->B,
        k->O, k->M, stream);
  }
  SCAE_REQUIRE(x && lse_post && lse_prior && g_tile_sums && g_templates && g_alpha_partial &&
               g_pose && g_scalar_partial);
  SCAE_REQUIRE(k->vote && k->scale && k->vote_presence && k->dummy_vote && k->x &&
               k->posterior && k->winner_idx && k->gvote && k->gscale && k->gvote_presence &&
               k->gx && k->gdummy_partial);
  const scae_k1::LpTiling lt = scae_k1::lp_tiling(d);
  LkBwd lk{{k->vote, k->scale, k->vote_presence, k->dummy_vote, k->x, k->presence, k->B, k->O,
            k->M},
           k->winner_idx,
           k->g_lpp,
           k->g_winner,
           k->g_winner_presence,
           k->g_soft_winner,
           k->g_soft_winner_presence,
           k->g_posterior,
           k->g_mixing_log_prob,
           k->g_mixing_logit,
           k->gvote,
           k->gscale,
           k->gvote_presence,
           k->gx,
           k->gpresence,
           k->gdummy_partial};
  hipStream_t st = (hipStream_t)stream;
  return d->C == 1 ? launch<1>(d, g, lt, x, lse_post, lse_prior, g_tile_sums, g_templates,
                               g_alpha_partial, g_pose, g_presence, g_bg_image, g_scalar_partial,
                               lk, st)
                   : launch<3>(d, g, lt, x, lse_post, lse_prior, g_tile_sums, g_templates,
                               g_alpha_partial, g_pose, g_presence, g_bg_image, g_scalar_partial,
                               lk, st);
}
