// Device code of K4 (capsule_likelihood.hip) in a header: the backward pass can also run as a
// block range of the K1 backward's launch (render_bwd_likelihood.hip).
#pragma once
#include "common.h"

namespace scae_lk {
namespace {
constexpr int NT = 1024;   // threads of the stand-alone launches
constexpr int OMAX = 64;  // the register form of phase B: 16 lanes x 4 capsules (more: the LDS two-pass form)
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

template <int G>
__device__ __forceinline__ float group_sum(float v) {
  static_assert(G == 16, "one DPP row");
  return scae::row_sum16(v);
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
  static_assert(G == 16, "one DPP row");
  return scae::row_max16(v);
}
template <int G>
__device__ __forceinline__ int group_min_int(int v) {
  static_assert(G == 16, "one DPP row");
  return scae::row_min16(v);
}

struct LkArgs {
  const float *vote, *scale, *vp, *dummy_vote, *x, *presence;
  int B, O, M;
};

// LDS: ml [O][M], post [O][M], stats per part: max_ml, lse_ml, max_post, sum_post,
// win (as float), then x [M][6]
struct LkSmem {
  float *ml, *post, *max_ml, *lse_ml, *max_post, *sum_post, *win, *x, *aux;
};
__device__ __forceinline__ LkSmem lk_carve(float *s, int O, int M) {
  LkSmem r;
  r.ml = s;
  r.post = r.ml + O * M;
  r.max_ml = r.post + O * M;
  r.lse_ml = r.max_ml + M;
  r.max_post = r.lse_ml + M;
  r.sum_post = r.max_post + M;
  r.win = r.sum_post + M;
  r.x = r.win + M;
  r.aux = r.x + M * 6;  // backward: gpp [O+1][M], dot [M], gmlp_sum [M]
  return r;
}

// phases A + B for image b: fills the LDS statistics.  Ends with a barrier.
// BIG: more than OMAX object capsules -- a lane group walks its ceil(O / 16) capsules per lane
// twice through LDS (maxima / arg-max, then the exponential sums) instead of holding them in
// registers; same order of operations, any O the LDS holds.
template <bool BIG, int NT>
__device__ __forceinline__ void lk_stats(const LkArgs &a, const LkSmem &s, int b) {
  const int O = a.O, M = a.M, tid = threadIdx.x;
  for (int i = tid; i < M * 6; i += NT) s.x[i] = a.x[(size_t)b * M * 6 + i];
  __syncthreads();
  for (int e = tid; e < O * M; e += NT) {  // phase A
    const int o = e / M, m = e - o * M;
    const size_t g = (size_t)b * O * M + e;
    float vt[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) vt[i] = a.vote[g * 6 + i];
    const float ml = scae::log_safe(a.vp[g]);
    s.ml[e] = ml;
    s.post[e] = ml + vote_lp(vt, s.x + m * 6, a.scale[g]);
  }
  __syncthreads();
  for (int e = tid; e < ((M * 16 + NT - 1) / NT) * NT; e += NT) {  // phase B
    const int m = e >> 4, l = e & 15;
    constexpr int KR = BIG ? 1 : OMAX / 16;
    const int kpl = BIG ? (O + 15) >> 4 : OMAX / 16;
    float vml[KR], vpo[KR];
    float mx_ml = l == 0 ? kLog001 : -INFINITY;            // lane 0 owns the dummy
    float mx_po = l == 0 ? kLog001 + kLog001 : -INFINITY;
    float best = -INFINITY;
    int best_o = 1 << 30;
    if (BIG) {
      for (int k = 0; k < kpl; ++k) {
        const int o = l + 16 * k;
        if (m < M && o < O) {
          const float pm = s.ml[o * M + m], pp = s.post[o * M + m];
          mx_ml = fmaxf(mx_ml, pm);
          mx_po = fmaxf(mx_po, pp);
          if (pp > best) best = pp, best_o = o;   // ascending o within the lane: first maximum
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        const int o = l + 16 * k;
        const bool in = m < M && o < O;
        vml[k] = in ? s.ml[o * M + m] : -INFINITY;
        vpo[k] = in ? s.post[o * M + m] : -INFINITY;
        mx_ml = fmaxf(mx_ml, vml[k]);
        mx_po = fmaxf(mx_po, vpo[k]);
        if (in && vpo[k] > best) {  // ascending o within the lane: first maximum
          best = vpo[k];
          best_o = o;
        }
      }
    }
    mx_ml = group_max<16>(mx_ml);
    mx_po = group_max<16>(mx_po);
    const float gbest = group_max<16>(best);
    const int win = group_min_int<16>(best == gbest ? best_o : (1 << 30));  // torch.argmax: first
    float sm = l == 0 ? expf(kLog001 - mx_ml) : 0.f;
    float sp = l == 0 ? expf(kLog001 + kLog001 - mx_po) : 0.f;
    if (BIG) {
      for (int k = 0; k < kpl; ++k) {
        const int o = l + 16 * k;
        if (m < M && o < O) {
          const float pm = s.ml[o * M + m], pp = s.post[o * M + m];
          if (pm != -INFINITY) sm += expf(pm - mx_ml);
          if (pp != -INFINITY) sp += expf(pp - mx_po);
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        if (vml[k] != -INFINITY) sm += expf(vml[k] - mx_ml);
        if (vpo[k] != -INFINITY) sp += expf(vpo[k] - mx_po);
      }
    }
    sm = group_sum<16>(sm);
    sp = group_sum<16>(sp);
    if (m < M && l == 0) {
      s.max_ml[m] = mx_ml;
      s.lse_ml[m] = mx_ml + logf(sm);
      s.max_post[m] = mx_po;
      s.sum_post[m] = sp;
      s.win[m] = (float)win;
    }
  }
  __syncthreads();
}

template <bool BIG>
__global__ __launch_bounds__(1024) void likelihood_fwd_kernel(
    LkArgs a, float *__restrict__ lpp, float *__restrict__ binary, float *__restrict__ winner,
    float *__restrict__ winner_presence, int64_t *__restrict__ winner_idx,
    int64_t *__restrict__ is_from_capsule, float *__restrict__ soft_winner,
    float *__restrict__ soft_winner_presence, float *__restrict__ posterior,
    float *__restrict__ mixing_log_prob, float *__restrict__ mixing_logit) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int O = a.O, M = a.M, tid = threadIdx.x;
  const LkSmem s = lk_carve(smem, O, M);
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    __syncthreads();
    lk_stats<BIG, NT>(a, s, b);
    // phase C1: per pair outputs (incl. the dummy row o == O); the posterior
    // probability of every real pair replaces its mixing logit in LDS
    for (int e = tid; e < (O + 1) * M; e += NT) {
      const int o = e / M, m = e - o * M;
      const size_t g1 = (size_t)b * (O + 1) * M + e;
      const float ml = o < O ? s.ml[e] : kLog001;
      const float post = o < O ? s.post[e] : kLog001 + kLog001;
      const float pp = expf(post - s.max_post[m]) / s.sum_post[m];       // :338
      mixing_logit[g1] = ml;
      mixing_log_prob[g1] = ml - s.lse_ml[m];                        // :286
      posterior[g1] = pp;
      if (o < O) {
        binary[(size_t)b * O * M + e] = ml > kLog001 ? 1.f : 0.f;  // :289
        s.ml[e] = pp;
      }
    }
    __syncthreads();
    // phase C2: winners.  Four lanes share the sum over o of one output (loads
    // of a lane's O/4 votes are in flight together), met by two shuffles
    for (int t = tid; t < ((M * 7 * 4 + NT - 1) / NT) * NT; t += NT) {
      const int e = t >> 2, part = t & 3;  // e < M*6: soft_winner[m][i]; else soft presence
      const bool pose = e < M * 6, live = e < M * 7;
      const int m = pose ? e / 6 : e - M * 6, i = pose ? e - m * 6 : 0;
      float acc = 0.f;
      if (live) {
        if (pose) {
#pragma unroll 6
          for (int o = part; o < O; o += 4)
            acc = fmaf(s.ml[o * M + m], a.vote[(((size_t)b * O + o) * M + m) * 6 + i], acc);
        } else {
#pragma unroll 6
          for (int o = part; o < O; o += 4)
            acc = fmaf(s.ml[o * M + m], a.vp[((size_t)b * O + o) * M + m], acc);
        }
      }
      acc += scae::xor1_f(acc);
      acc += scae::xor2_f(acc);
      if (!live || part != 0) continue;
      const size_t idx = (size_t)b * M + m;
      const int win = (int)s.win[m];
      if (pose) {
        const float ppd = expf(kLog001 + kLog001 - s.max_post[m]) / s.sum_post[m];
        soft_winner[idx * 6 + i] = fmaf(ppd, a.dummy_vote[m * 6 + i], acc);   // :350
        winner[idx * 6 + i] = a.vote[(((size_t)b * O + win) * M + m) * 6 + i];  // :324
      } else {
        const float lse_post = s.max_post[m] + logf(s.sum_post[m]);
        lpp[idx] = a.presence ? lse_post * a.presence[idx] : lse_post;  // :296-300
        soft_winner_presence[idx] = acc;                                // :354
        winner_presence[idx] = a.vp[((size_t)b * O + win) * M + m];     // :328
        winner_idx[idx] = win;
        is_from_capsule[idx] = win / M;                                 // :334 (reference quirk)
      }
    }
  }
}

// The backward pass as a device function of workgroup `blk` of `nblk` with NTH threads
// (`smem`: its dynamic LDS): 1024 in its own launch; 256 when it rides in the launch of
// the K1 backward (render_bwd_likelihood.hip).
template <bool BIG, int NT>
__device__ __forceinline__ void likelihood_bwd_body(
    LkArgs a, const int64_t *__restrict__ winner_idx, const float *__restrict__ g_lpp,
    const float *__restrict__ g_winner, const float *__restrict__ g_winner_presence,
    const float *__restrict__ g_soft_winner, const float *__restrict__ g_soft_winner_presence,
    const float *__restrict__ g_posterior, const float *__restrict__ g_mlp,
    const float *__restrict__ g_mlogit, float *__restrict__ gvote, float *__restrict__ gscale,
    float *__restrict__ gvp, float *__restrict__ gx, float *__restrict__ gpresence,
    float *__restrict__ gdummy, float *smem, int blk, int nblk) {
  const int O = a.O, M = a.M, tid = threadIdx.x;
  const LkSmem s = lk_carve(smem, O, M);
  float *s_gpp = s.aux;                 // [O+1][M] incoming grad on posterior probs
  float *s_dot = s_gpp + (O + 1) * M;   // [M] softmax-backward inner product
  float *s_gmlp = s_dot + M;            // [M] column sums of g_mixing_log_prob
  for (int b = blk; b < a.B; b += nblk) {
    __syncthreads();
    lk_stats<BIG, NT>(a, s, b);
    // incoming gradient on every posterior probability (dummy row included)
    for (int e = tid; e < (O + 1) * M; e += NT) {
      const int o = e / M, m = e - o * M;
      const size_t idx = (size_t)b * M + m;
      float gpp = g_posterior ? g_posterior[(size_t)b * (O + 1) * M + e] : 0.f;
      if (g_soft_winner) {
        const float *vt = o < O ? a.vote + (((size_t)b * O + o) * M + m) * 6
                                : a.dummy_vote + (size_t)m * 6;
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) d = fmaf(g_soft_winner[idx * 6 + i], vt[i], d);
        gpp += d;
      }
      if (g_soft_winner_presence && o < O)
        gpp += g_soft_winner_presence[idx] * a.vp[((size_t)b * O + o) * M + m];
      s_gpp[e] = gpp;
    }
    __syncthreads();
    // softmax-backward inner product and the column sums of g_mixing_log_prob:
    // 16 lanes per part capsule m share the sum over the O + 1 components
    for (int t = tid; t < ((M * 16 + NT - 1) / NT) * NT; t += NT) {
      const int m = t >> 4, l = t & 15;
      float dot = 0.f, gs = 0.f;
      if (m < M) {
        for (int o = l; o <= O; o += 16) {
          const float post = o < O ? s.post[o * M + m] : kLog001 + kLog001;
          dot = fmaf(expf(post - s.max_post[m]) / s.sum_post[m], s_gpp[o * M + m], dot);
          if (g_mlp) gs += g_mlp[((size_t)b * (O + 1) + o) * M + m];
        }
      }
      dot = group_sum<16>(dot);
      gs = group_sum<16>(gs);
      if (m < M && l == 0) {
        s_dot[m] = dot;
        s_gmlp[m] = gs;
      }
    }
    __syncthreads();
    for (int e = tid; e < O * M; e += NT) {  // per pair gradients
      const int o = e / M, m = e - o * M;
      const size_t g = (size_t)b * O * M + e, g1 = ((size_t)b * (O + 1) + o) * M + m;
      const size_t idx = (size_t)b * M + m;
      const float pv = a.vp[g], sc = a.scale[g], ml = s.ml[e];
      const float pp = expf(s.post[e] - s.max_post[m]) / s.sum_post[m];
      const float pres = a.presence ? a.presence[idx] : 1.f;
      const float glse = g_lpp ? g_lpp[idx] * pres : 0.f;
      const float gpost = pp * (s_gpp[e] - s_dot[m]) + glse * pp;
      float gml = gpost;
      if (g_mlogit) gml += g_mlogit[g1];
      if (g_mlp) gml += g_mlp[g1] - expf(ml - s.lse_ml[m]) * s_gmlp[m];
      const float gswp = g_soft_winner_presence ? g_soft_winner_presence[idx] : 0.f;
      float g_pv = gml * scae::log_safe_grad(pv) + gswp * pp;
      const bool is_win = o == (int)winner_idx[idx];
      if (is_win && g_winner_presence) g_pv += g_winner_presence[idx];
      const float inv_var = 1.f / (sc * sc);
      float gsc = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const float vt = a.vote[g * 6 + i];
        const float df = s.x[m * 6 + i] - vt;
        float gv = gpost * df * inv_var;
        if (g_soft_winner) gv = fmaf(g_soft_winner[idx * 6 + i], pp, gv);
        if (is_win && g_winner) gv += g_winner[idx * 6 + i];
        gvote[g * 6 + i] = gv;
        gsc += gpost * (df * df * inv_var - 1.f) / sc;
      }
      gvp[g] = g_pv;
      gscale[g] = gsc;
      s.ml[e] = gpost * inv_var;  // reuse: (g wrt posterior logit) / s^2, for gx below
    }
    __syncthreads();
    // gx[m][i] = -sum_o gpost (x - v) / s^2: four lanes per output split the sum
    // over o (their votes are in flight together), met by two shuffles
    for (int t = tid; t < ((M * 6 * 4 + NT - 1) / NT) * NT; t += NT) {
      const int e = t >> 2, part = t & 3;
      const bool live = e < M * 6;
      const int m = live ? e / 6 : 0, i = live ? e - m * 6 : 0;
      float acc = 0.f;
      if (live) {
        const float xv = s.x[e];
#pragma unroll 6
        for (int o = part; o < O; o += 4)
          acc -= s.ml[o * M + m] * (xv - a.vote[(((size_t)b * O + o) * M + m) * 6 + i]);
      }
      acc += scae::xor1_f(acc);
      acc += scae::xor2_f(acc);
      if (!live || part != 0) continue;
      const size_t idx = (size_t)b * M + m;
      gx[idx * 6 + i] = acc;
      const float ppd = expf(kLog001 + kLog001 - s.max_post[m]) / s.sum_post[m];
      gdummy[idx * 6 + i] = g_soft_winner ? g_soft_winner[idx * 6 + i] * ppd : 0.f;
    }
    if (gpresence)
      for (int m = tid; m < M; m += NT) {
        const size_t idx = (size_t)b * M + m;
        gpresence[idx] = (a.presence && g_lpp)
                             ? g_lpp[idx] * (s.max_post[m] + logf(s.sum_post[m]))
                             : 0.f;
      }
  }
}

template <bool BIG>
__global__ __launch_bounds__(1024) void likelihood_bwd_kernel(
    LkArgs a, const int64_t *__restrict__ winner_idx, const float *__restrict__ g_lpp,
    const float *__restrict__ g_winner, const float *__restrict__ g_winner_presence,
    const float *__restrict__ g_soft_winner, const float *__restrict__ g_soft_winner_presence,
    const float *__restrict__ g_posterior, const float *__restrict__ g_mlp,
    const float *__restrict__ g_mlogit, float *__restrict__ gvote, float *__restrict__ gscale,
    float *__restrict__ gvp, float *__restrict__ gx, float *__restrict__ gpresence,
    float *__restrict__ gdummy) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  likelihood_bwd_body<BIG, 1024>(a, winner_idx, g_lpp, g_winner, g_winner_presence, g_soft_winner, g_soft_winner_presence, g_posterior, g_mlp, g_mlogit, gvote, gscale, gvp, gx, gpresence, gdummy, smem, blockIdx.x, gridDim.x);
}
}  // namespace


inline size_t lk_lds(int O, int M, bool bwd) {
  size_t f = 2 * (size_t)O * M + 5 * M + 6 * M;
  if (bwd) f += (size_t)(O + 1) * M + 2 * M;
  return f * sizeof(float);
}
}  // namespace scae_lk
