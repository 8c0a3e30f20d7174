// K3 -- object-capsule votes for gfx950.  Replaces the op cluster of
// object_decoder.py:160-225: split of the per-capsule MLP output, OPR/OVR
// geometric_transform (cv_ops.py:20-76), the batched 3x3 OVR x OPR product
// (:189-191; only its top two rows are consumed, :413), presence logits with
// uniform noise (:198-212), vote presences (:217-219) and vote scales (:225).
//
// One workgroup per (b, o) capsule: its A = 8V+7 parameters are read once
// (coalesced), the capsule's OVR 2x3 is computed once and broadcast from LDS
// to the V vote lanes.  Elementwise + one small reduction: HBM bound.
#include "common.h"

#include "capsule_votes_dev.h"

namespace {
using namespace scae_votes;
constexpr int NT = 64;  // one wave per capsule; lanes stride over votes

__global__ __launch_bounds__(NT) void votes_fwd_kernel(
    VoteArgs a, float *__restrict__ vote, float *__restrict__ scale,
    float *__restrict__ vote_presence, float *__restrict__ logit_caps,
    float *__restrict__ logit_vote, float *__restrict__ reg_partial,
    float *__restrict__ caps_presence, int *__restrict__ caps_arg) {
  const int bo = blockIdx.x, o = bo % a.O, V = a.V, lane = threadIdx.x;
  const float *ap = a.all_param + (size_t)bo * a.ldp;

  // OVR (one per capsule): every lane computes it redundantly from 6 floats
  float cv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) cv[i] = ap[6 * V + i] + a.bias_cvr[o * 6 + i];
  Xf C;
  xf_eval(cv, a.similarity, C);

  float lc = ap[6 * V + 6] + a.bias_caps[o];
  if (a.noise_caps) lc += (a.noise_caps[bo] - 0.5f) * a.noise_scale;
  const float pc = scae::sigmoidf_(lc);
  if (lane == 0) logit_caps[bo] = lc;

  float reg = 0.f;
  float best = -INFINITY;  // caps_presence = max_v vote_presence (object_decoder.py:411),
  int best_v = 0x7fffffff;  // first maximiser
  for (int v = lane; v < V; v += NT) {
    float pr[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float dyn = a.allow_deformations ? ap[v * 6 + i] : 0.f;
      reg += dyn * dyn;
      pr[i] = dyn + a.cpr_static[((size_t)o * V + v) * 6 + i];
    }
    Xf P;
    xf_eval(pr, a.similarity, P);
    // [c0 c1 c2; c3 c4 c5; 0 0 1] x [p0 p1 p2; p3 p4 p5; 0 0 1], rows 0..1
    float *vo = vote + ((size_t)bo * V + v) * 6;
    vo[0] = C.o[0] * P.o[0] + C.o[1] * P.o[3];
    vo[1] = C.o[0] * P.o[1] + C.o[1] * P.o[4];
    vo[2] = C.o[0] * P.o[2] + C.o[1] * P.o[5] + C.o[2];
    vo[3] = C.o[3] * P.o[0] + C.o[4] * P.o[3];
    vo[4] = C.o[3] * P.o[1] + C.o[4] * P.o[4];
    vo[5] = C.o[3] * P.o[2] + C.o[4] * P.o[5] + C.o[5];

    float lv = ap[6 * V + 7 + v] + a.bias_vote[o * V + v];
    if (a.noise_vote) lv += (a.noise_vote[(size_t)bo * V + v] - 0.5f) * a.noise_scale;
    logit_vote[(size_t)bo * V + v] = lv;
    const float vpv = pc * scae::sigmoidf_(lv);
    vote_presence[(size_t)bo * V + v] = vpv;
    if (vpv > best) best = vpv, best_v = v;
    float sc = 1.f;
    if (a.learn_vote_scale)
      sc = scae::softplusf_(ap[7 * V + 7 + v] + a.bias_scale[o * V + v] + .5f) + 1e-2f;
    scale[(size_t)bo * V + v] = sc;
  }
  reg = scae::wave_sum(reg);
  if (lane == 0) reg_partial[bo] = reg;
  if (caps_presence) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(best_v, off, 64);
      if (ov > best || (ov == best && oi < best_v)) best = ov, best_v = oi;
    }
    if (lane == 0) {
      caps_presence[bo] = best;
      caps_arg[bo] = best_v;
    }
  }
}

__global__ __launch_bounds__(NT) void votes_bwd_kernel(
    VoteArgs a, const float *__restrict__ gvote, const float *__restrict__ gscale,
    const float *__restrict__ gvp, const float *__restrict__ glc,
    const float *__restrict__ glv, const float *__restrict__ greg,
    float *__restrict__ gall, float *__restrict__ gcpr_in,
    const float *__restrict__ g_caps_presence, const int *__restrict__ caps_arg,
    float *__restrict__ gall_gated) {
  const int bo = blockIdx.x, o = bo % a.O, V = a.V, lane = threadIdx.x;
  const float *ap = a.all_param + (size_t)bo * a.ldp;
  float *ga = gall + (size_t)bo * a.ldp;
  // optional second copy zeroed where all_param <= 0: all_param is a ReLU
  // output, so this is the gradient w.r.t. the producing layer's pre-activation
  float *gg = gall_gated ? gall_gated + (size_t)bo * a.ldp : nullptr;
  auto put = [&](int i, float v) {
    ga[i] = v;
    if (gg) gg[i] = ap[i] > 0.f ? v : 0.f;
  };

  float cv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) cv[i] = ap[6 * V + i] + a.bias_cvr[o * 6 + i];
  Xf C;
  xf_eval(cv, a.similarity, C);
  float lc = ap[6 * V + 6] + a.bias_caps[o];
  if (a.noise_caps) lc += (a.noise_caps[bo] - 0.5f) * a.noise_scale;
  const float pc = scae::sigmoidf_(lc);
  // d cpr_dynamic_reg_loss / d dyn = dyn / B   (l2_loss/B, object_decoder.py:170)
  const float reg_w = greg ? greg[0] / (float)a.B : 0.f;

  float gC[6] = {0, 0, 0, 0, 0, 0};  // grad wrt the capsule's transformed OVR
  float gpc = 0.f;                   // grad wrt sigmoid(caps logit)
  for (int v = lane; v < V; v += NT) {
    float pr[6], dyn[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      dyn[i] = a.allow_deformations ? ap[v * 6 + i] : 0.f;
      pr[i] = dyn[i] + a.cpr_static[((size_t)o * V + v) * 6 + i];
    }
    Xf P;
    xf_eval(pr, a.similarity, P);
    float gv[6] = {0, 0, 0, 0, 0, 0};
    if (gvote) {
#pragma unroll
      for (int i = 0; i < 6; ++i) gv[i] = gvote[((size_t)bo * V + v) * 6 + i];
    }
    gC[0] += gv[0] * P.o[0] + gv[1] * P.o[1] + gv[2] * P.o[2];
    gC[1] += gv[0] * P.o[3] + gv[1] * P.o[4] + gv[2] * P.o[5];
    gC[2] += gv[2];
    gC[3] += gv[3] * P.o[0] + gv[4] * P.o[1] + gv[5] * P.o[2];
    gC[4] += gv[3] * P.o[3] + gv[4] * P.o[4] + gv[5] * P.o[5];
    gC[5] += gv[5];
    float gP[6];
    gP[0] = gv[0] * C.o[0] + gv[3] * C.o[3];
    gP[1] = gv[1] * C.o[0] + gv[4] * C.o[3];
    gP[2] = gv[2] * C.o[0] + gv[5] * C.o[3];
    gP[3] = gv[0] * C.o[1] + gv[3] * C.o[4];
    gP[4] = gv[1] * C.o[1] + gv[4] * C.o[4];
    gP[5] = gv[2] * C.o[1] + gv[5] * C.o[4];
    float gin[6];
    xf_backward(P, a.similarity, gP, gin);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      gcpr_in[((size_t)bo * V + v) * 6 + i] = gin[i];
      put(v * 6 + i, a.allow_deformations ? gin[i] + reg_w * dyn[i] : 0.f);
    }

    float lv = ap[6 * V + 7 + v] + a.bias_vote[o * V + v];
    if (a.noise_vote) lv += (a.noise_vote[(size_t)bo * V + v] - 0.5f) * a.noise_scale;
    const float pv = scae::sigmoidf_(lv);
    float g_vp = gvp ? gvp[(size_t)bo * V + v] : 0.f;
    if (g_caps_presence && caps_arg[bo] == v) g_vp += g_caps_presence[bo];
    gpc += g_vp * pv;
    float g_lv = g_vp * pc * pv * (1.f - pv);
    if (glv) g_lv += glv[(size_t)bo * V + v];
    put(6 * V + 7 + v, g_lv);
    float g_sc = 0.f;
    if (a.learn_vote_scale && gscale)
      g_sc = gscale[(size_t)bo * V + v] *
             scae::softplus_grad(ap[7 * V + 7 + v] + a.bias_scale[o * V + v] + .5f);
    put(7 * V + 7 + v, g_sc);
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) gC[i] = scae::wave_sum(gC[i]);
  gpc = scae::wave_sum(gpc);
  if (lane == 0) {
    float gin[6];
    xf_backward(C, a.similarity, gC, gin);
#pragma unroll
    for (int i = 0; i < 6; ++i) put(6 * V + i, gin[i]);
    float g_lc = gpc * pc * (1.f - pc);
    if (glc) g_lc += glc[bo];
    put(6 * V + 6, g_lc);
  }
}

int check_votes(const VoteArgs &a) {
  if (!a.all_param || !a.cpr_static || !a.bias_cvr || !a.bias_caps || !a.bias_vote ||
      !a.bias_scale)
    return SCAE_ERR_BAD_ARG;
  if (a.B <= 0 || a.O <= 0 || a.V <= 0 || a.ldp < 8 * a.V + 7) return SCAE_ERR_BAD_ARG;
  return SCAE_OK;
}
}  // namespace

extern "C" int scae_capsule_votes_fwd_f32(
    const float *all_param, const float *cpr_static, const float *bias_cvr,
    const float *bias_caps, const float *bias_vote, const float *bias_scale,
    const float *noise_caps, const float *noise_vote, float noise_scale, float *vote,
    float *scale, float *vote_presence, float *logit_caps, float *logit_vote,
    float *reg_partial, float *caps_presence, int *caps_arg, int B, int O, int V,
    int ld_param, int similarity, int learn_vote_scale, int allow_deformations, void *stream) {
  VoteArgs a{all_param, cpr_static, bias_cvr, bias_caps, bias_vote, bias_scale,
             noise_caps, noise_vote, noise_scale, B, O, V, similarity, learn_vote_scale,
             allow_deformations, ld_param > 0 ? ld_param : 8 * V + 7};
  int rc = check_votes(a);
  if (rc) return rc;
  SCAE_REQUIRE(vote && scale && vote_presence && logit_caps && logit_vote && reg_partial);
  SCAE_REQUIRE(!caps_presence == !caps_arg);
  scae::launch(votes_fwd_kernel, dim3(B * O), dim3(NT), 0, (hipStream_t)stream, a,
                     vote, scale, vote_presence, logit_caps, logit_vote, reg_partial,
                     caps_presence, caps_arg);
  return scae_launch_status();
}

extern "C" int scae_capsule_votes_bwd_f32(
    const float *all_param, const float *cpr_static, const float *bias_cvr,
    const float *bias_caps, const float *bias_vote, const float *bias_scale,
    const float *noise_caps, const float *noise_vote, float noise_scale,
    const float *gvote, const float *gscale, const float *gvote_presence,
    const float *glogit_caps, const float *glogit_vote, const float *greg,
    const float *gcaps_presence, const int *caps_arg, float *gall_param, float *gcpr_in,
    float *gall_param_gated, int B, int O, int V, int ld_param, int similarity,
    int learn_vote_scale, int allow_deformations, void *stream) {
  VoteArgs a{all_param, cpr_static, bias_cvr, bias_caps, bias_vote, bias_scale,
             noise_caps, noise_vote, noise_scale, B, O, V, similarity, learn_vote_scale,
             allow_deformations, ld_param > 0 ? ld_param : 8 * V + 7};
  int rc = check_votes(a);
  if (rc) return rc;
  SCAE_REQUIRE(gall_param && gcpr_in && (!gcaps_presence || caps_arg));
  scae::launch(votes_bwd_kernel, dim3(B * O), dim3(NT), 0, (hipStream_t)stream, a,
                     gvote, gscale, gvote_presence, glogit_caps, glogit_vote, greg,
                     gall_param, gcpr_in, gcaps_presence, caps_arg, gall_param_gated);
  return scae_launch_status();
}
