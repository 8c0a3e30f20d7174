// K4 -- part-pose mixture likelihood of the object decoder, gfx950.
// Replaces CapsuleLikelihood.__call__, object_decoder.py:257-372: per-vote
// Normal log-prob, dummy component, mixing log-softmax, posterior softmax,
// hard (argmax) and soft winners.
//
// One lane per (image b, part m); the O+1 mixture components are walked in
// registers (two passes: max, then exp-sum), consecutive lanes read
// consecutive parts so every (B,O,M[,6]) access is coalesced.  Small and
// HBM/latency bound -- no matrix cores involved.
#include "common.h"

namespace {
constexpr int NT = 128;
constexpr float kLog001 = -4.605170185988091f;  // np.log(0.01), object_decoder.py:274

// sum over the 6 pose dims of Normal(vote, scale).log_prob(x)   (:263-269)
__device__ __forceinline__ float vote_lp(const float *vt, const float *xv, float sc) {
  const float inv2v = 1.f / (2.f * sc * sc), ls = logf(sc);
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const float d = xv[i] - vt[i];
    acc += -(d * d) * inv2v - ls - scae::kHalfLog2Pi;
  }
  return acc;
}

__global__ __launch_bounds__(NT) void likelihood_fwd_kernel(
    const float *__restrict__ vote, const float *__restrict__ scale,
    const float *__restrict__ vp, const float *__restrict__ dummy_vote,
    const float *__restrict__ x, const float *__restrict__ presence,
    float *__restrict__ lpp, float *__restrict__ binary, float *__restrict__ winner,
    float *__restrict__ winner_presence, int64_t *__restrict__ winner_idx,
    int64_t *__restrict__ is_from_capsule, float *__restrict__ soft_winner,
    float *__restrict__ soft_winner_presence, float *__restrict__ posterior,
    float *__restrict__ mixing_log_prob, float *__restrict__ mixing_logit, int B, int O,
    int M) {
  const int idx = blockIdx.x * NT + threadIdx.x;
  if (idx >= B * M) return;
  const int b = idx / M, m = idx - b * M;
  float xv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) xv[i] = x[(size_t)idx * 6 + i];

  // pass 1: maxima of the mixing logits and of the posterior logits; argmax
  float max_ml = kLog001, max_post = kLog001 + kLog001;
  float best = -INFINITY;
  int best_o = 0;
  for (int o = 0; o < O; ++o) {
    const size_t e = ((size_t)b * O + o) * M + m;
    const float ml = scae::log_safe(vp[e]);
    float vt[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) vt[i] = vote[e * 6 + i];
    const float post = ml + vote_lp(vt, xv, scale[e]);
    max_ml = fmaxf(max_ml, ml);
    max_post = fmaxf(max_post, post);
    if (post > best) {  // first maximum wins (torch.argmax, :310)
      best = post;
      best_o = o;
    }
  }
  // pass 2: exp-sums
  float sum_ml = expf(kLog001 - max_ml), sum_post = expf(kLog001 + kLog001 - max_post);
  for (int o = 0; o < O; ++o) {
    const size_t e = ((size_t)b * O + o) * M + m;
    const float ml = scae::log_safe(vp[e]);
    float vt[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) vt[i] = vote[e * 6 + i];
    sum_ml += expf(ml - max_ml);
    sum_post += expf(ml + vote_lp(vt, xv, scale[e]) - max_post);
  }
  const float lse_ml = max_ml + logf(sum_ml);
  const float lse_post = max_post + logf(sum_post);

  // pass 3: outputs
  float sw[6] = {0, 0, 0, 0, 0, 0};
  float swp = 0.f;
  for (int o = 0; o < O; ++o) {
    const size_t e = ((size_t)b * O + o) * M + m;
    const size_t e1 = ((size_t)b * (O + 1) + o) * M + m;
    const float pv = vp[e];
    const float ml = scae::log_safe(pv);
    float vt[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) vt[i] = vote[e * 6 + i];
    const float post = ml + vote_lp(vt, xv, scale[e]);
    const float pp = expf(post - max_post) / sum_post;  // softmax (:338)
    mixing_logit[e1] = ml;
    mixing_log_prob[e1] = ml - lse_ml;                  // :286
    binary[e] = ml > kLog001 ? 1.f : 0.f;               // :289
    posterior[e1] = pp;
#pragma unroll
    for (int i = 0; i < 6; ++i) sw[i] += pp * vt[i];
    swp += pp * pv;
  }
  {  // dummy component (index O)
    const size_t e1 = ((size_t)b * (O + 1) + O) * M + m;
    const float pp = expf(kLog001 + kLog001 - max_post) / sum_post;
    mixing_logit[e1] = kLog001;
    mixing_log_prob[e1] = kLog001 - lse_ml;
    posterior[e1] = pp;
#pragma unroll
    for (int i = 0; i < 6; ++i) sw[i] += pp * dummy_vote[(size_t)m * 6 + i];
  }
  const float pres = presence ? presence[idx] : 1.f;
  lpp[idx] = presence ? lse_post * pres : lse_post;     // :296-300
  const size_t ew = ((size_t)b * O + best_o) * M + m;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    winner[(size_t)idx * 6 + i] = vote[ew * 6 + i];     // :324
    soft_winner[(size_t)idx * 6 + i] = sw[i];           // :350
  }
  winner_presence[idx] = vp[ew];                        // :328
  soft_winner_presence[idx] = swp;                      // :354
  winner_idx[idx] = best_o;
  is_from_capsule[idx] = best_o / M;                    // :334 (reference quirk)
}

__global__ __launch_bounds__(NT) void likelihood_bwd_kernel(
    const float *__restrict__ vote, const float *__restrict__ scale,
    const float *__restrict__ vp, const float *__restrict__ dummy_vote,
    const float *__restrict__ x, const float *__restrict__ presence,
    const float *__restrict__ posterior, const int64_t *__restrict__ winner_idx,
    const float *__restrict__ g_lpp, const float *__restrict__ g_winner,
    const float *__restrict__ g_winner_presence, const float *__restrict__ g_soft_winner,
    const float *__restrict__ g_soft_winner_presence, const float *__restrict__ g_posterior,
    const float *__restrict__ g_mlp, const float *__restrict__ g_mlogit,
    float *__restrict__ gvote, float *__restrict__ gscale, float *__restrict__ gvp,
    float *__restrict__ gx, float *__restrict__ gpresence, float *__restrict__ gdummy, int B,
    int O, int M) {
  const int idx = blockIdx.x * NT + threadIdx.x;
  if (idx >= B * M) return;
  const int b = idx / M, m = idx - b * M;
  float xv[6], gsw[6], gw[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    xv[i] = x[(size_t)idx * 6 + i];
    gsw[i] = g_soft_winner ? g_soft_winner[(size_t)idx * 6 + i] : 0.f;
    gw[i] = g_winner ? g_winner[(size_t)idx * 6 + i] : 0.f;
  }
  const float gswp = g_soft_winner_presence ? g_soft_winner_presence[idx] : 0.f;
  const float gwp = g_winner_presence ? g_winner_presence[idx] : 0.f;
  const float pres = presence ? presence[idx] : 1.f;
  const float glse = g_lpp ? g_lpp[idx] * pres : 0.f;  // grad wrt logsumexp(post)
  const int win = (int)winner_idx[idx];

  // incoming gradient on every posterior probability, and the softmax-backward
  // inner product  sum_o pp_o * gpp_o  (dummy row included)
  float dot = 0.f, max_ml = kLog001;
  float gmlp_sum = 0.f;  // sum over rows of g_mixing_log_prob (log-softmax bwd)
  for (int o = 0; o <= O; ++o) {
    const size_t e1 = ((size_t)b * (O + 1) + o) * M + m;
    const float pp = posterior[e1];
    float gpp = g_posterior ? g_posterior[e1] : 0.f;
    if (o < O) {
      const size_t e = ((size_t)b * O + o) * M + m;
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) d += gsw[i] * vote[e * 6 + i];
      gpp += d + gswp * vp[e];
      max_ml = fmaxf(max_ml, scae::log_safe(vp[e]));
    } else {
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) d += gsw[i] * dummy_vote[(size_t)m * 6 + i];
      gpp += d;
    }
    dot += pp * gpp;
    if (g_mlp) gmlp_sum += g_mlp[e1];
  }
  float sum_ml = expf(kLog001 - max_ml);
  for (int o = 0; o < O; ++o)
    sum_ml += expf(scae::log_safe(vp[((size_t)b * O + o) * M + m]) - max_ml);

  float gxv[6] = {0, 0, 0, 0, 0, 0};
  // logsumexp(post) = post_o - log(pp_o) for any o: recover it from the most
  // probable component (pp >= 1/(O+1), so the log is safe); needed for the
  // presence gradient
  float pp_best = posterior[((size_t)b * (O + 1) + O) * M + m];
  float post_best = kLog001 + kLog001;
  for (int o = 0; o < O; ++o) {
    const size_t e = ((size_t)b * O + o) * M + m;
    const size_t e1 = ((size_t)b * (O + 1) + o) * M + m;
    const float pv = vp[e], sc = scale[e];
    const float ml = scae::log_safe(pv);
    const float pp = posterior[e1];
    float vt[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) vt[i] = vote[e * 6 + i];
    float gpp = g_posterior ? g_posterior[e1] : 0.f;
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) d += gsw[i] * vt[i];
    gpp += d + gswp * pv;
    // grad wrt the posterior logit  post_o = ml_o + vlp_o
    const float gpost = pp * (gpp - dot) + glse * pp;
    // grad wrt ml_o: via post, via mixing_logit, via mixing_log_prob
    float gml = gpost;
    if (g_mlogit) gml += g_mlogit[e1];
    if (g_mlp) gml += g_mlp[e1] - expf(ml - max_ml) / sum_ml * gmlp_sum;
    // direct terms
    float g_pv = gml * scae::log_safe_grad(pv) + gswp * pp;
    float gsc = 0.f;
    const float inv_var = 1.f / (sc * sc);
    if (pp > pp_best) {
      pp_best = pp;
      post_best = ml + vote_lp(vt, xv, sc);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const float df = xv[i] - vt[i];
      float gv = gpost * df * inv_var + gsw[i] * pp;  // d vlp/d vote = (x - v)/s^2
      if (o == win) gv += gw[i];
      gvote[e * 6 + i] = gv;
      gxv[i] -= gpost * df * inv_var;
      gsc += gpost * (df * df * inv_var - 1.f) / sc;
    }
    if (o == win) g_pv += gwp;
    gvp[e] = g_pv;
    gscale[e] = gsc;
  }
  {  // dummy vote: only the soft winner touches it
    const float ppd = posterior[((size_t)b * (O + 1) + O) * M + m];
#pragma unroll
    for (int i = 0; i < 6; ++i) gdummy[(size_t)idx * 6 + i] = gsw[i] * ppd;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) gx[(size_t)idx * 6 + i] = gxv[i];
  if (gpresence)
    gpresence[idx] = (presence && g_lpp) ? g_lpp[idx] * (post_best - logf(pp_best)) : 0.f;
}
}  // namespace

extern "C" int scae_capsule_likelihood_fwd_f32(
    const float *vote, const float *scale, const float *vote_presence,
    const float *dummy_vote, const float *x, const float *presence,
    float *log_prob_per_point, float *vote_presence_binary, float *winner,
    float *winner_presence, int64_t *winner_idx, int64_t *is_from_capsule,
    float *soft_winner, float *soft_winner_presence, float *posterior,
    float *mixing_log_prob, float *mixing_logit, int B, int O, int M, void *stream) {
  SCAE_REQUIRE(vote && scale && vote_presence && dummy_vote && x);
  SCAE_REQUIRE(log_prob_per_point && vote_presence_binary && winner && winner_presence &&
               winner_idx && is_from_capsule && soft_winner && soft_winner_presence &&
               posterior && mixing_log_prob && mixing_logit);
  SCAE_REQUIRE(B > 0 && O > 0 && M > 0);
  hipLaunchKernelGGL(likelihood_fwd_kernel, dim3((B * M + NT - 1) / NT), dim3(NT), 0,
                     (hipStream_t)stream, vote, scale, vote_presence, dummy_vote, x,
                     presence, log_prob_per_point, vote_presence_binary, winner,
                     winner_presence, winner_idx, is_from_capsule, soft_winner,
                     soft_winner_presence, posterior, mixing_log_prob, mixing_logit, B, O, M);
  return scae_launch_status();
}

extern "C" int scae_capsule_likelihood_bwd_f32(
    const float *vote, const float *scale, const float *vote_presence,
    const float *dummy_vote, const float *x, const float *presence, const float *posterior,
    const int64_t *winner_idx, const float *g_lpp, const float *g_winner,
    const float *g_winner_presence, const float *g_soft_winner,
    const float *g_soft_winner_presence, const float *g_posterior,
    const float *g_mixing_log_prob, const float *g_mixing_logit, float *gvote,
    float *gscale, float *gvote_presence, float *gx, float *gpresence,
    float *gdummy_partial, int B, int O, int M, void *stream) {
  SCAE_REQUIRE(vote && scale && vote_presence && dummy_vote && x && posterior && winner_idx);
  SCAE_REQUIRE(gvote && gscale && gvote_presence && gx && gdummy_partial);
  SCAE_REQUIRE(B > 0 && O > 0 && M > 0);
  hipLaunchKernelGGL(likelihood_bwd_kernel, dim3((B * M + NT - 1) / NT), dim3(NT), 0,
                     (hipStream_t)stream, vote, scale, vote_presence, dummy_vote, x,
                     presence, posterior, winner_idx, g_lpp, g_winner, g_winner_presence,
                     g_soft_winner, g_soft_winner_presence, g_posterior, g_mixing_log_prob,
                     g_mixing_logit, gvote, gscale, gvote_presence, gx, gpresence,
                     gdummy_partial, B, O, M);
  return scae_launch_status();
}
