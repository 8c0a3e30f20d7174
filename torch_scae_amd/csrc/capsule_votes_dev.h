// Device code of K3 (capsule_votes.hip) shared with mlp_chain.hip, where the vote kernel's
// work rides at the end of the capsule-MLP chain (forward) and at the head of its
// data-gradient chain (backward): the pose transform of cv_ops.py:20-76 with its local
// derivatives, the argument block, and the two per-block routines that do for 16 capsules
// (16 batch rows of one object capsule, their parameter rows in LDS) what votes_fwd_kernel
// / votes_bwd_kernel do for one.
#pragma once
#include "common.h"

namespace scae_votes {
constexpr float kTwoPi = 6.283185307179586f;

struct Xf {  // a transformed pose and the local derivatives of its 6 outputs
  float sx, sy, sh, c, s, tx, ty;
  float dsx, dsy, dsh, dtx, dty;
  float o[6];
};

__device__ __forceinline__ void xf_eval(const float *p, int similarity, Xf &g) {
  const float ex = scae::sigmoidf_(p[0]), ey = scae::sigmoidf_(p[1]);
  g.sx = ex + 1e-2f;
  g.sy = ey + 1e-2f;
  g.dsx = ex * (1.f - ex);
  g.dsy = ey * (1.f - ey);
  g.sh = tanhf(p[3] * 5.f);
  g.tx = tanhf(p[4] * 5.f);
  g.ty = tanhf(p[5] * 5.f);
  g.dsh = 5.f * (1.f - g.sh * g.sh);
  g.dtx = 5.f * (1.f - g.tx * g.tx);
  g.dty = 5.f * (1.f - g.ty * g.ty);
  const float th = p[2] * kTwoPi;
  g.c = cosf(th);
  g.s = sinf(th);
  if (similarity) {
    g.o[0] = g.sx * g.c;
    g.o[1] = -g.sx * g.s;
    g.o[2] = g.tx;
    g.o[3] = g.sx * g.s;
    g.o[4] = g.sx * g.c;
    g.o[5] = g.ty;
  } else {
    g.o[0] = g.sx * g.c + g.sh * g.sy * g.s;
    g.o[1] = -g.sx * g.s + g.sh * g.sy * g.c;
    g.o[2] = g.tx;
    g.o[3] = g.sy * g.s;
    g.o[4] = g.sy * g.c;
    g.o[5] = g.ty;
  }
}

__device__ __forceinline__ void xf_backward(const Xf &g, int similarity, const float *go,
                                            float *gp) {
  float gsx, gsy, gsh, gth;
  if (similarity) {
    gsx = go[0] * g.c - go[1] * g.s + go[3] * g.s + go[4] * g.c;
    gsy = 0.f;
    gsh = 0.f;
    gth = g.sx * (-go[0] * g.s - go[1] * g.c + go[3] * g.c - go[4] * g.s);
  } else {
    gsx = go[0] * g.c - go[1] * g.s;
    gsy = go[0] * g.sh * g.s + go[1] * g.sh * g.c + go[3] * g.s + go[4] * g.c;
    gsh = go[0] * g.sy * g.s + go[1] * g.sy * g.c;
    gth = go[0] * (-g.sx * g.s + g.sh * g.sy * g.c) +
          go[1] * (-g.sx * g.c - g.sh * g.sy * g.s) + go[3] * g.sy * g.c - go[4] * g.sy * g.s;
  }
  gp[0] = gsx * g.dsx;
  gp[1] = gsy * g.dsy;
  gp[2] = gth * kTwoPi;
  gp[3] = gsh * g.dsh;
  gp[4] = go[2] * g.dtx;
  gp[5] = go[5] * g.dty;
}

struct VoteArgs {
  const float *all_param, *cpr_static, *bias_cvr, *bias_caps, *bias_vote, *bias_scale;
  const float *noise_caps, *noise_vote;
  float noise_scale;
  int B, O, V, similarity, learn_vote_scale, allow_deformations;
  int ldp;  // floats between consecutive capsule rows of all_param (and of its gradients), >= A
};


// Forward-side outputs / backward-side gradient pointers (all nullable where the stand-alone
// launchers allow it).
struct VoteOut {
  float *vote, *scale, *vote_presence, *logit_caps, *logit_vote, *reg_partial, *caps_presence;
  int *caps_arg;
};
struct VoteGrads {
  const float *gvote, *gscale, *gvp, *glc, *glv, *greg, *g_caps_presence;
  const int *caps_arg;
  float *gall, *gcpr_in, *gall_gated;
};

// 16 capsules (b0 .. b0 + 15, o) whose all_param rows sit in LDS at rows[i * ld]; `red`:
// 2 * 16 * V floats of LDS scratch.  NTH threads; ends with the outputs written (the
// caller needs no barrier for them).
template <int NTH>
__device__ __forceinline__ void fwd_block(const VoteArgs &a, const VoteOut &w, const float *rows,
                                          int ld, float *red, int b0, int o) {
  const int V = a.V, tid = threadIdx.x;
  for (int e = tid; e < 16 * V; e += NTH) {
    const int i = e / V, v = e - i * V, b = b0 + i;
    const float *ap = rows + i * ld;
    float regv = 0.f, vpv = -INFINITY;
    if (b < a.B) {
      const int bo = b * a.O + o;
      float cv[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) cv[k] = ap[6 * V + k] + a.bias_cvr[o * 6 + k];
      Xf C;
      xf_eval(cv, a.similarity, C);
      float lc = ap[6 * V + 6] + a.bias_caps[o];
      if (a.noise_caps) lc += (a.noise_caps[bo] - 0.5f) * a.noise_scale;
      const float pc = scae::sigmoidf_(lc);
      if (v == 0) w.logit_caps[bo] = lc;
      float pr[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float dyn = a.allow_deformations ? ap[v * 6 + k] : 0.f;
        regv += dyn * dyn;
        pr[k] = dyn + a.cpr_static[((size_t)o * V + v) * 6 + k];
      }
      Xf P;
      xf_eval(pr, a.similarity, P);
      float *vo = w.vote + ((size_t)bo * V + v) * 6;
      vo[0] = C.o[0] * P.o[0] + C.o[1] * P.o[3];
      vo[1] = C.o[0] * P.o[1] + C.o[1] * P.o[4];
      vo[2] = C.o[0] * P.o[2] + C.o[1] * P.o[5] + C.o[2];
      vo[3] = C.o[3] * P.o[0] + C.o[4] * P.o[3];
      vo[4] = C.o[3] * P.o[1] + C.o[4] * P.o[4];
      vo[5] = C.o[3] * P.o[2] + C.o[4] * P.o[5] + C.o[5];
      float lv = ap[6 * V + 7 + v] + a.bias_vote[o * V + v];
      if (a.noise_vote) lv += (a.noise_vote[(size_t)bo * V + v] - 0.5f) * a.noise_scale;
      w.logit_vote[(size_t)bo * V + v] = lv;
      vpv = pc * scae::sigmoidf_(lv);
      w.vote_presence[(size_t)bo * V + v] = vpv;
      float sc = 1.f;
      if (a.learn_vote_scale)
        sc = scae::softplusf_(ap[7 * V + 7 + v] + a.bias_scale[o * V + v] + .5f) + 1e-2f;
      w.scale[(size_t)bo * V + v] = sc;
    }
    red[e] = regv;
    red[16 * V + e] = vpv;
  }
  __syncthreads();
  if (tid < 256) {   // per capsule: reg sum, first maximiser of vote_presence; 16 lanes each
    const int i = tid >> 4, l = tid & 15;
    float reg = 0.f, best = -INFINITY;
    int best_v = 0x7fffffff;
    for (int v = l; v < V; v += 16) {
      reg += red[i * V + v];
      const float p = red[16 * V + i * V + v];
      if (p > best) best = p, best_v = v;
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      reg += __shfl_xor(reg, off, 64);
      const float ov = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(best_v, off, 64);
      if (ov > best || (ov == best && oi < best_v)) best = ov, best_v = oi;
    }
    if (l == 0 && b0 + i < a.B) {
      const int bo = (b0 + i) * a.O + o;
      w.reg_partial[bo] = reg;
      if (w.caps_presence) {
        w.caps_presence[bo] = best;
        w.caps_arg[bo] = best_v;
      }
    }
  }
}

// Backward for the same 16 capsules: all_param rows from global memory (a.all_param), the
// gradient rows into LDS at grows[i * ld] -- the gated ones when gall_gated is set, which is
// what the data-gradient chain behind it consumes -- and to global memory (gall,
// gall_gated, gcpr_in).  `red`: 16 * V * 7 floats of LDS scratch.  Ends with a barrier.
template <int NTH>
__device__ __forceinline__ void bwd_block(const VoteArgs &a, const VoteGrads &w, float *grows,
                                          int ld, float *red, int b0, int o) {
  const int V = a.V, tid = threadIdx.x, A = 8 * V + 7;
  const float reg_w = w.greg ? w.greg[0] / (float)a.B : 0.f;
  auto put = [&](const float *ap, int i, int bo, int col, float v) {
    w.gall[(size_t)bo * a.ldp + col] = v;
    const float gv = ap[col] > 0.f ? v : 0.f;
    if (w.gall_gated) w.gall_gated[(size_t)bo * a.ldp + col] = gv;
    grows[i * ld + col] = w.gall_gated ? gv : v;
  };
  for (int e = tid; e < 16 * V; e += NTH) {
    const int i = e / V, v = e - i * V, b = b0 + i;
    float part[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // gC[6], gpc
    if (b < a.B) {
      const int bo = b * a.O + o;
      const float *ap = a.all_param + (size_t)bo * a.ldp;
      float cv[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) cv[k] = ap[6 * V + k] + a.bias_cvr[o * 6 + k];
      Xf C;
      xf_eval(cv, a.similarity, C);
      float lc = ap[6 * V + 6] + a.bias_caps[o];
      if (a.noise_caps) lc += (a.noise_caps[bo] - 0.5f) * a.noise_scale;
      const float pc = scae::sigmoidf_(lc);
      float pr[6], dyn[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        dyn[k] = a.allow_deformations ? ap[v * 6 + k] : 0.f;
        pr[k] = dyn[k] + a.cpr_static[((size_t)o * V + v) * 6 + k];
      }
      Xf P;
      xf_eval(pr, a.similarity, P);
      float gv[6] = {0, 0, 0, 0, 0, 0};
      if (w.gvote) {
#pragma unroll
        for (int k = 0; k < 6; ++k) gv[k] = w.gvote[((size_t)bo * V + v) * 6 + k];
      }
      part[0] = gv[0] * P.o[0] + gv[1] * P.o[1] + gv[2] * P.o[2];
      part[1] = gv[0] * P.o[3] + gv[1] * P.o[4] + gv[2] * P.o[5];
      part[2] = gv[2];
      part[3] = gv[3] * P.o[0] + gv[4] * P.o[1] + gv[5] * P.o[2];
      part[4] = gv[3] * P.o[3] + gv[4] * P.o[4] + gv[5] * P.o[5];
      part[5] = gv[5];
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
      for (int k = 0; k < 6; ++k) {
        w.gcpr_in[((size_t)bo * V + v) * 6 + k] = gin[k];
        put(ap, i, bo, v * 6 + k, a.allow_deformations ? gin[k] + reg_w * dyn[k] : 0.f);
      }
      float lv = ap[6 * V + 7 + v] + a.bias_vote[o * V + v];
      if (a.noise_vote) lv += (a.noise_vote[(size_t)bo * V + v] - 0.5f) * a.noise_scale;
      const float pv = scae::sigmoidf_(lv);
      float g_vp = w.gvp ? w.gvp[(size_t)bo * V + v] : 0.f;
      if (w.g_caps_presence && w.caps_arg[bo] == v) g_vp += w.g_caps_presence[bo];
      part[6] = g_vp * pv;
      float g_lv = g_vp * pc * pv * (1.f - pv);
      if (w.glv) g_lv += w.glv[(size_t)bo * V + v];
      put(ap, i, bo, 6 * V + 7 + v, g_lv);
      float g_sc = 0.f;
      if (a.learn_vote_scale && w.gscale)
        g_sc = w.gscale[(size_t)bo * V + v] *
               scae::softplus_grad(ap[7 * V + 7 + v] + a.bias_scale[o * V + v] + .5f);
      put(ap, i, bo, 7 * V + 7 + v, g_sc);
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) red[(size_t)e * 7 + k] = part[k];
  }
  __syncthreads();
  {  // sums over the votes of a capsule: (capsule, value, quarter of the votes) per thread
    float t = 0.f;
    const int i = tid >> 5, k = (tid >> 2) & 7, part = tid & 3;
    if (tid < 512 && k < 7)
      for (int v = part; v < V; v += 4) t += red[(size_t)(i * V + v) * 7 + k];
    t += __shfl_xor(t, 1, 64);
    t += __shfl_xor(t, 2, 64);
    __syncthreads();   // every partial has been read
    if (tid < 512 && k < 7 && part == 0) red[i * 8 + k] = t;
  }
  __syncthreads();
  if (tid < 16) {   // per capsule: the OVR and capsule-logit gradients
    const int i = tid, b = b0 + i;
    if (b < a.B) {
      const int bo = b * a.O + o;
      const float *ap = a.all_param + (size_t)bo * a.ldp;
      float gC[6], gpc = red[i * 8 + 6];
#pragma unroll
      for (int k = 0; k < 6; ++k) gC[k] = red[i * 8 + k];
      float cv[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) cv[k] = ap[6 * V + k] + a.bias_cvr[o * 6 + k];
      Xf C;
      xf_eval(cv, a.similarity, C);
      float lc = ap[6 * V + 6] + a.bias_caps[o];
      if (a.noise_caps) lc += (a.noise_caps[bo] - 0.5f) * a.noise_scale;
      const float pc = scae::sigmoidf_(lc);
      float gin[6];
      xf_backward(C, a.similarity, gC, gin);
#pragma unroll
      for (int k = 0; k < 6; ++k) put(ap, i, bo, 6 * V + k, gin[k]);
      float g_lc = gpc * pc * (1.f - pc);
      if (w.glc) g_lc += w.glc[bo];
      put(ap, i, bo, 6 * V + 6, g_lc);
    } else {
      for (int col = 0; col < A; ++col) grows[i * ld + col] = 0.f;
    }
  }
  // rows past the batch and the columns between A and the next multiple of 16: zeros
  const int a16 = (A + 15) & ~15;
  for (int e = tid; e < 16 * (a16 - A); e += NTH) {
    const int i = e / (a16 - A), col = A + e - i * (a16 - A);
    grows[i * ld + col] = 0.f;
  }
  __syncthreads();
}
}  // namespace scae_votes
