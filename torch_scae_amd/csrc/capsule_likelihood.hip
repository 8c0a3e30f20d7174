// K4 -- part-pose mixture likelihood of the object decoder, gfx950.
// Replaces CapsuleLikelihood.__call__, object_decoder.py:257-372: per-vote
// Normal log-prob, dummy component, mixing log-softmax, posterior softmax,
// hard (argmax) and soft winners.
//
// One workgroup per image.  Phase A: one lane per (capsule o, part m) pair
// evaluates the mixing logit and the posterior logit (votes read coalesced,
// pair logits parked in LDS).  Phase B: 16 lanes per part reduce the O+1
// components with xor-shuffles (max / exp-sum / first-argmax).  Phase C: lanes
// per pair write the (B,O+1,M) outputs, lanes per (part, pose dim) the winners.
// Small and latency bound -- no matrix cores involved.  (A first version
// walked the O+1 components serially in one lane per part: 35 us + 44 us.)
#include "common.h"
#include "capsule_likelihood_dev.h"

namespace {
using namespace scae_lk;
int lk_check(int B, int O, int M) {
  if (B <= 0 || O <= 0 || M <= 0) return SCAE_ERR_BAD_ARG;
  if (lk_lds(O, M, true) > 160 * 1024) return SCAE_ERR_UNSUPPORTED;
  return SCAE_OK;
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
  int rc = lk_check(B, O, M);
  if (rc) return rc;
  LkArgs a{vote, scale, vote_presence, dummy_vote, x, presence, B, O, M};
  const size_t lds = lk_lds(O, M, false);
  const bool big = O > OMAX;
  const void *fn = big ? reinterpret_cast<const void *>(likelihood_fwd_kernel<true>)
                       : reinterpret_cast<const void *>(likelihood_fwd_kernel<false>);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
#define SCAE_LK_FWD(BG)                                                                        \
  scae::launch(likelihood_fwd_kernel<BG>, dim3(B < 1024 ? B : 1024), dim3(NT), lds,      \
                     (hipStream_t)stream, a, log_prob_per_point, vote_presence_binary, winner, \
                     winner_presence, winner_idx, is_from_capsule, soft_winner,                \
                     soft_winner_presence, posterior, mixing_log_prob, mixing_logit)
  if (big) {
    SCAE_LK_FWD(true);
  } else {
    SCAE_LK_FWD(false);
  }
#undef SCAE_LK_FWD
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
  int rc = lk_check(B, O, M);
  if (rc) return rc;
  LkArgs a{vote, scale, vote_presence, dummy_vote, x, presence, B, O, M};
  const size_t lds = lk_lds(O, M, true);
  const bool big = O > OMAX;
  const void *fn = big ? reinterpret_cast<const void *>(likelihood_bwd_kernel<true>)
                       : reinterpret_cast<const void *>(likelihood_bwd_kernel<false>);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
#define SCAE_LK_BWD(BG)                                                                         \
  scae::launch(likelihood_bwd_kernel<BG>, dim3(B < 1024 ? B : 1024), dim3(NT), lds,       \
                     (hipStream_t)stream, a, winner_idx, g_lpp, g_winner, g_winner_presence,    \
                     g_soft_winner, g_soft_winner_presence, g_posterior, g_mixing_log_prob,     \
                     g_mixing_logit, gvote, gscale, gvote_presence, gx, gpresence,              \
                     gdummy_partial)
  if (big) {
    SCAE_LK_BWD(true);
  } else {
    SCAE_LK_BWD(false);
  }
#undef SCAE_LK_BWD
  return scae_launch_status();
}
