// K6 -- fused tail of SCAE.loss for gfx950.  Replaces ~125 launch-bound ATen
// ops (forward + autograd backward) of stacked_capsule_auto_encoder.py:238-285
// and object_decoder.py:433-493:
//   capsule log-likelihood term, prior sparsity (l2 | entropy | kl) on
//   caps_presence, posterior sparsity on the capsule mass / n_points, and the
//   two "cross_entropy over probabilities" classification terms (both through
//   prior_classifier, as the reference does, :207-212, :281-282).
// Everything is O(B*O) data: one workgroup per independent group of terms,
// tensors staged in LDS, block reductions by wave shuffles.  The backward kernel recomputes the forward
// statistics and writes every gradient once (no atomics).
#include "common.h"

namespace {
constexpr int NT = 1024;
constexpr int MAXCLS = 32;

struct TailArgs {
  const float *lpp;        // (B,M)   log_prob_per_point
  const float *posterior;  // (B,O+1,M)
  const float *cp;         // (B,O)   caps_presence
  const float *cls_w;      // (ncls,O) nullable
  const float *cls_b;      // (ncls)
  const int64_t *label;    // (B) nullable
  int B, O, M, ncls;
  int prior_type, post_type;  // 0 l2, 1 entropy, 2 kl
  int sparsity_on;            // reference gate: prior weights > 0
  float w_ll, w_pw, w_pb, w_qw, w_qb;  // loss weights
  float l2_within_const, l2_between_const, l2_within_const_post, l2_between_const_post;
};

__device__ __forceinline__ float block_total(float v, float *red) {
  float a[1] = {v};
  scae::block_sum<1, NT>(a, red);
  if (threadIdx.x == 0) red[31] = a[0];
  __syncthreads();
  const float r = red[31];
  __syncthreads();
  return r;
}

// -sum p log_safe(p*k) terms: value and d/dp
__device__ __forceinline__ float ent_term(float p, float k) {
  return -p * scae::log_safe(p * k);
}
__device__ __forceinline__ float ent_term_grad(float p, float k) {
  const float q = p * k;
  return q < scae::kLogSafeEps ? 1e8f : -(logf(q) + 1.f);
}

// shared statistics of one (B,O) activation matrix x
struct Stats {
  float *x;     // [B*O]
  float *row;   // [B]  sum over o
  float *col;   // [O]  sum over b
};

__device__ void row_col_sums(const Stats &s, int B, int O) {
  for (int b = threadIdx.x; b < B; b += NT) {
    float t = 0.f;
    for (int o = 0; o < O; ++o) t += s.x[b * O + o];
    s.row[b] = t;
  }
  // column sums: 16 lanes per column, each takes every 16th row
  for (int e = threadIdx.x; e < ((O * 16 + NT - 1) / NT) * NT; e += NT) {
    const int o = e >> 4, l = e & 15;
    float t = 0.f;
    if (o < O)
      for (int b = l; b < B; b += 16) t += s.x[b * O + o];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    if (o < O && l == 0) s.col[o] = t;
  }
  __syncthreads();
}

// (within, between) of sparsity_loss(type, x); object_decoder.py:433-493
__device__ void sparsity_fwd(const Stats &s, int B, int O, int type, float cw, float cb,
                             float *red, float &within, float &between) {
  float w = 0.f, bt = 0.f;
  if (type == 0) {
    for (int b = threadIdx.x; b < B; b += NT) {
      const float d = s.row[b] - cw;
      w += d * d;
    }
    for (int o = threadIdx.x; o < O; o += NT) {
      const float d = s.col[o] - cb;
      bt += d * d;
    }
    within = block_total(w, red) / B;
    between = block_total(bt, red) / O;
  } else {
    const float k = type == 2 ? (float)O : 1.f;
    for (int i = threadIdx.x; i < B * O; i += NT) {
      const int b = i / O;
      w += ent_term(s.x[i] / (s.row[b] + 1e-8f), k);
    }
    float tot = 0.f;
    for (int o = 0; o < O; ++o) tot += s.col[o];
    for (int o = threadIdx.x; o < O; o += NT) bt += ent_term(s.col[o] / (tot + 1e-8f), k);
    within = block_total(w, red) / B;
    between = -block_total(bt, red);
  }
}

// g[b,o] += gw * d within/dx + gb * d between/dx
__device__ void sparsity_bwd(const Stats &s, int B, int O, int type, float cw, float cb,
                             float gw, float gb, float *g, float *tmp_row /*[B]*/) {
  if (type == 0) {
    for (int i = threadIdx.x; i < B * O; i += NT) {
      const int b = i / O, o = i - b * O;
      g[i] += gw * 2.f * (s.row[b] - cw) / B + gb * 2.f * (s.col[o] - cb) / O;
    }
  } else {
    const float k = type == 2 ? (float)O : 1.f;
    float tot = 0.f;
    for (int o = 0; o < O; ++o) tot += s.col[o];
    const float tinv = 1.f / (tot + 1e-8f);
    // between: d(-H(bp))/dx[b,o] = -(sum_j dH/dbp_j dbp_j/dt_o), t_o = col sums
    float dot_b = 0.f;
    for (int o = 0; o < O; ++o) dot_b += ent_term_grad(s.col[o] * tinv, k) * s.col[o] * tinv;
    for (int b = threadIdx.x; b < B; b += NT) {
      const float rinv = 1.f / (s.row[b] + 1e-8f);
      float dot_w = 0.f;
      for (int j = 0; j < O; ++j)
        dot_w += ent_term_grad(s.x[b * O + j] * rinv, k) * s.x[b * O + j] * rinv;
      tmp_row[b] = dot_w;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < B * O; i += NT) {
      const int b = i / O, o = i - b * O;
      const float rinv = 1.f / (s.row[b] + 1e-8f);
      const float dot_w = tmp_row[b];
      const float dw = (ent_term_grad(s.x[i] * rinv, k) - dot_w) * rinv / B;
      const float db = -(ent_term_grad(s.col[o] * tinv, k) - dot_b) * tinv;
      g[i] += gw * dw + gb * db;
    }
  }
}

// softmax(prior_classifier(x[b])) then cross_entropy(probs, label): value and
// gradient w.r.t. the classifier logits (x is detached in the reference)
__device__ __forceinline__ float cls_xe(const TailArgs &a, const float *xrow, int label,
                                        float (&glogit)[MAXCLS]) {
  float z[MAXCLS], p[MAXCLS];
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < MAXCLS; ++c) {
    if (c < a.ncls) {
      float t = a.cls_b[c];
      for (int o = 0; o < a.O; ++o) t = fmaf(xrow[o], a.cls_w[c * a.O + o], t);
      z[c] = t;
      mx = fmaxf(mx, t);
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < MAXCLS; ++c)
    if (c < a.ncls) {
      p[c] = expf(z[c] - mx);
      sum += p[c];
    }
  float mx2 = -INFINITY;
#pragma unroll
  for (int c = 0; c < MAXCLS; ++c)
    if (c < a.ncls) {
      p[c] /= sum;
      mx2 = fmaxf(mx2, p[c]);
    }
  float sum2 = 0.f, plabel = 0.f;
  float q[MAXCLS];
#pragma unroll
  for (int c = 0; c < MAXCLS; ++c)
    if (c < a.ncls) {
      q[c] = expf(p[c] - mx2);
      sum2 += q[c];
      if (c == label) plabel = p[c];
    }
  const float xe = mx2 + logf(sum2) - plabel;  // -log_softmax(p)[label]
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < MAXCLS; ++c)
    if (c < a.ncls) {
      q[c] = q[c] / sum2 - (c == label ? 1.f : 0.f);  // d xe / d p_c
      dot = fmaf(p[c], q[c], dot);
    }
#pragma unroll
  for (int c = 0; c < MAXCLS; ++c) glogit[c] = c < a.ncls ? p[c] * (q[c] - dot) : 0.f;
  return xe;
}

struct Carve {
  float *cp, *mass, *row_c, *col_c, *row_m, *col_m, *red, *gl;
};
__device__ Carve carve(float *smem, int B, int O, int ncls) {
  Carve c;
  c.cp = smem;
  c.mass = c.cp + B * O;
  c.row_c = c.mass + B * O;
  c.col_c = c.row_c + B;
  c.row_m = c.col_c + O;
  c.col_m = c.row_m + B;
  c.red = c.col_m + O;
  c.gl = c.red + 32;  // [2][B][ncls] classifier logit grads (backward only)
  return c;
}

// stages caps_presence and / or the capsule mass with their row / column sums
__device__ void load_stats(const TailArgs &a, const Carve &c, bool want_cp, bool want_mass,
                           bool sums = true) {
  const int B = a.B, O = a.O, M = a.M;
  for (int i = threadIdx.x; i < B * O; i += NT) {
    const int b = i / O, o = i - b * O;
    if (want_cp) c.cp[i] = a.cp[i];
    if (!want_mass) continue;
    const float *pr = a.posterior + ((size_t)b * (O + 1) + o) * M;
    float t = 0.f;
    if ((M & 3) == 0) {
      const float4 *p4 = reinterpret_cast<const float4 *>(pr);
      for (int m = 0; m < M / 4; ++m) {
        const float4 v = p4[m];
        t += (v.x + v.y) + (v.z + v.w);
      }
    } else {
      for (int m = 0; m < M; ++m) t += pr[m];
    }
    c.mass[i] = t / M;  // mass_explained_by_capsule / n_points (:260-266)
  }
  __syncthreads();
  if (!sums) return;
  if (want_cp) row_col_sums(Stats{c.cp, c.row_c, c.col_c}, B, O);
  if (want_mass) row_col_sums(Stats{c.mass, c.row_m, c.col_m}, B, O);
}

// out: [0] tail loss  [1] log_prob  [2] prior_within [3] prior_between
//      [4] post_within [5] post_between [6] prior_cls_xe [7] posterior_cls_xe
// The terms are independent until the final weighted sum, so they run as three
// workgroups (blockIdx.x = role) on three CUs; tail_combine_kernel then forms
// the scalar.  role 0: capsule log-likelihood, reconstruction term, regulariser;
// role 1: prior sparsity + prior classification; role 2: the posterior pair.
__global__ __launch_bounds__(NT) void tail_fwd_kernel(TailArgs a, scae_loss_extras x, float *out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int B = a.B, O = a.O, role = blockIdx.x;
  const Carve c = carve(smem, B, O, a.ncls);
  if (role == 0) {
    float lp = 0.f;
    for (int i = threadIdx.x; i < B * a.M; i += NT) lp += a.lpp[i];
    const float log_prob = block_total(lp, c.red) / B;
    // reconstruction term (stacked_capsule_auto_encoder.py:222-224) from K1's tile
    // sums, and the dynamic-regularisation scalar
    float rec = 0.f;
    if (x.rec_sums) {
      float t = 0.f;
      for (int i = threadIdx.x; i < x.n_rec; i += NT) t += x.rec_sums[i];
      rec = block_total(t, c.red) / B;
    }
    if (threadIdx.x == 0) {
      out[1] = log_prob;
      out[8] = rec;
      out[9] = -rec;
      out[10] = -log_prob;
      out[11] = x.reg ? x.reg[0] : 0.f;
    }
    return;
  }
  const bool prior = role == 1;
  load_stats(a, c, prior, !prior);
  float within = 0.f, between = 0.f;
  if (a.sparsity_on) {
    if (prior)
      sparsity_fwd(Stats{c.cp, c.row_c, c.col_c}, B, O, a.prior_type, a.l2_within_const,
                   a.l2_between_const, c.red, within, between);
    else
      sparsity_fwd(Stats{c.mass, c.row_m, c.col_m}, B, O, a.post_type, a.l2_within_const_post,
                   a.l2_between_const_post, c.red, within, between);
  }
  float xe = 0.f;
  if (a.label) {
    // posterior classifier input: the un-normalised capsule mass (:210-212)
    if (!prior) {
      for (int i = threadIdx.x; i < B * O; i += NT) c.cp[i] = c.mass[i] * a.M;
      __syncthreads();
    }
    float t = 0.f;
    float gl[MAXCLS];
    for (int b = threadIdx.x; b < B; b += NT) t += cls_xe(a, c.cp + b * O, (int)a.label[b], gl);
    xe = block_total(t, c.red) / B;
  }
  if (threadIdx.x == 0) {
    out[prior ? 2 : 4] = within;
    out[prior ? 3 : 5] = between;
    out[prior ? 6 : 7] = xe;
  }
}

__global__ void tail_combine_kernel(TailArgs a, scae_loss_extras x, float *out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float loss = -a.w_ll * out[1] + a.w_pw * out[2] + a.w_pb * out[3] + a.w_qw * out[4] +
                     a.w_qb * out[5] + out[6] + out[7] - out[8] + x.w_reg * out[11];
  out[0] = loss;
  if (x.loss) x.loss[0] = loss;
}

// Backward: four independent workgroups (blockIdx.x = role), disjoint outputs.
// role 0: g_lpp, g_rec_sums, g_reg; role 1: prior sparsity -> g_caps_presence;
// role 2: posterior sparsity -> g_posterior; role 3: both classification terms
// -> g_cls_w, g_cls_b.
__global__ __launch_bounds__(NT) void tail_bwd_kernel(TailArgs a, scae_loss_extras x,
                                                      const float *gout /*[12]*/, float *g_lpp,
                                                      float *g_post, float *g_cp, float *g_w,
                                                      float *g_b) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int B = a.B, O = a.O, M = a.M, role = blockIdx.x;
  const Carve c = carve(smem, B, O, a.ncls);
  // d(total)/d(component): the loss plus whatever flowed into the individually
  // exposed log entries; d/d(loss) may arrive on the 12-vector, on the separate
  // scalar, or both
  float go[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) go[i] = gout ? gout[i] : 0.f;
  if (x.g_loss) go[0] += x.g_loss[0];
  const float g0 = go[0];
  if (role == 0) {
    const float g_lp = -a.w_ll * g0 + go[1] - go[10];
    if (x.g_rec_sums) {
      const float gr = (-g0 + go[8] - go[9]) / B;
      for (int i = threadIdx.x; i < x.n_rec; i += NT) x.g_rec_sums[i] = gr;
    }
    if (x.g_reg && threadIdx.x == 0) x.g_reg[0] = x.w_reg * g0 + go[11];
    for (int i = threadIdx.x; i < B * M; i += NT) g_lpp[i] = g_lp / B;
    return;
  }
  // reuse LDS: a gradient accumulator over (B,O) and a per-row scratch
  float *gacc = c.gl + 2 * B * MAXCLS;  // [B*O]
  float *tmp = gacc + 2 * B * O;        // [B]
  if (role == 1 || role == 2) {
    const bool prior = role == 1;
    load_stats(a, c, prior, !prior);
    for (int i = threadIdx.x; i < B * O; i += NT) gacc[i] = 0.f;
    __syncthreads();
    if (a.sparsity_on) {
      if (prior)
        sparsity_bwd(Stats{c.cp, c.row_c, c.col_c}, B, O, a.prior_type, a.l2_within_const,
                     a.l2_between_const, a.w_pw * g0 + go[2], a.w_pb * g0 + go[3], gacc, tmp);
      else
        sparsity_bwd(Stats{c.mass, c.row_m, c.col_m}, B, O, a.post_type,
                     a.l2_within_const_post, a.l2_between_const_post, a.w_qw * g0 + go[4],
                     a.w_qb * g0 + go[5], gacc, tmp);
    }
    __syncthreads();
    if (prior) {
      for (int i = threadIdx.x; i < B * O; i += NT) g_cp[i] = gacc[i];
    } else {
      // posterior (B,O+1,M): mass/M = sum_m post / M, i.e. one value per (b, o) row
      // of M entries; the dummy row gets zero.  One row per thread: no per-element
      // index divisions, 16-byte stores when M allows
      for (int bo = threadIdx.x; bo < B * (O + 1); bo += NT) {
        const int b = bo / (O + 1), o = bo - b * (O + 1);
        const float v = o < O ? gacc[b * O + o] / M : 0.f;
        float *row = g_post + (size_t)bo * M;
        if ((M & 3) == 0) {
          const float4 v4 = make_float4(v, v, v, v);
          for (int m = 0; m < M / 4; ++m) reinterpret_cast<float4 *>(row)[m] = v4;
        } else {
          for (int m = 0; m < M; ++m) row[m] = v;
        }
      }
    }
    return;
  }
  // role 3: classifier parameter gradients (inputs are detached)
  if (!(a.label && g_w)) return;
  const float g_x1 = g0 + go[6], g_x2 = g0 + go[7];
  load_stats(a, c, true, true, false);
  float gl[MAXCLS];
  for (int i = threadIdx.x; i < B * O; i += NT) gacc[i] = c.mass[i] * M;  // second input
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * B; e += NT) {
    const int which = e / B, b = e - which * B;
    cls_xe(a, (which ? gacc : c.cp) + b * O, (int)a.label[b], gl);
    const float gx = which ? g_x2 : g_x1;
#pragma unroll
    for (int cc = 0; cc < MAXCLS; ++cc)
      if (cc < a.ncls) c.gl[e * MAXCLS + cc] = gl[cc] * gx / B;
  }
  __syncthreads();
  // g_w[cc][o] = sum_b glogit * input, g_b[cc] = sum_b glogit: few outputs with a
  // 2B-long sum each -> four lanes per output, interleaved over b, meet by shuffle
  const int nout = a.ncls * O + a.ncls;
  for (int e = threadIdx.x; e < ((nout * 4 + NT - 1) / NT) * NT; e += NT) {
    const int out = e >> 2, part = e & 3;
    float t = 0.f;
    if (out < a.ncls * O) {
      const int cc = out / O, o = out - cc * O;
      for (int b = part; b < B; b += 4)
        t += c.gl[b * MAXCLS + cc] * c.cp[b * O + o] +
             c.gl[(B + b) * MAXCLS + cc] * gacc[b * O + o];
    } else if (out < nout) {
      const int cc = out - a.ncls * O;
      for (int b = part; b < B; b += 4) t += c.gl[b * MAXCLS + cc] + c.gl[(B + b) * MAXCLS + cc];
    }
    t += __shfl_xor(t, 1, 64);
    t += __shfl_xor(t, 2, 64);
    if (part == 0 && out < nout) {
      if (out < a.ncls * O)
        g_w[out] = t;
      else
        g_b[out - a.ncls * O] = t;
    }
  }
}

size_t tail_lds(int B, int O, bool bwd) {
  size_t f = 2 * (size_t)B * O + 2 * B + 2 * O + 32;
  if (bwd) f += 2 * (size_t)B * MAXCLS + 2 * (size_t)B * O + B;
  return f * sizeof(float);
}
}  // namespace

extern "C" int scae_loss_tail_supported(int B, int O, int ncls) {
  return (B > 0 && O > 0 && ncls <= MAXCLS && tail_lds(B, O, true) <= 150 * 1024) ? 1 : 0;
}

static int fill_tail(TailArgs &a, const float *lpp, const float *posterior, const float *cp,
                     const float *cls_w, const float *cls_b, const int64_t *label, int B,
                     int O, int M, int ncls, int n_classes_cfg, int prior_type, int post_type,
                     int sparsity_on, const float *weights /*5*/, float within_const) {
  if (!lpp || !posterior || !cp || !weights || B <= 0 || O <= 0 || M <= 0)
    return SCAE_ERR_BAD_ARG;
  if (label && (!cls_w || !cls_b || ncls <= 0)) return SCAE_ERR_BAD_ARG;
  if (prior_type < 0 || prior_type > 2 || post_type < 0 || post_type > 2) return SCAE_ERR_BAD_ARG;
  if (!scae_loss_tail_supported(B, O, ncls)) return SCAE_ERR_UNSUPPORTED;
  a = TailArgs{lpp, posterior, cp, cls_w, cls_b, label, B, O, M, ncls, prior_type, post_type,
               sparsity_on, weights[0], weights[1], weights[2], weights[3], weights[4],
               0.f, 0.f, 0.f, 0.f};
  const float nc = n_classes_cfg > 0 ? (float)n_classes_cfg : 1.f;
  // capsule_l2_loss constants (object_decoder.py:443-449); the posterior call
  // never passes within_example_constant (:262-266)
  a.l2_within_const = within_const == within_const ? within_const : (float)O / nc;
  a.l2_between_const = (float)B / nc;
  a.l2_within_const_post = (float)O / nc;
  a.l2_between_const_post = (float)B / nc;
  return SCAE_OK;
}

extern "C" int scae_loss_tail_fwd_f32(const float *lpp, const float *posterior,
                                      const float *caps_presence, const float *cls_w,
                                      const float *cls_b, const int64_t *label,
                                      const scae_loss_extras *extras, float *out12, int B,
                                      int O, int M, int ncls, int n_classes_cfg, int prior_type,
                                      int post_type, int sparsity_on, const float *weights5,
                                      float within_const, void *stream) {
  TailArgs a;
  int rc = fill_tail(a, lpp, posterior, caps_presence, cls_w, cls_b, label, B, O, M, ncls,
                     n_classes_cfg, prior_type, post_type, sparsity_on, weights5, within_const);
  if (rc) return rc;
  SCAE_REQUIRE(out12);
  scae_loss_extras x{};
  if (extras) x = *extras;
  if (x.rec_sums && x.n_rec <= 0) return SCAE_ERR_BAD_ARG;
  const size_t lds = tail_lds(B, O, false);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tail_fwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(tail_fwd_kernel, dim3(3), dim3(NT), lds, (hipStream_t)stream, a, x, out12);
  hipLaunchKernelGGL(tail_combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, x, out12);
  return scae_launch_status();
}

extern "C" int scae_loss_tail_bwd_f32(const float *lpp, const float *posterior,
                                      const float *caps_presence, const float *cls_w,
                                      const float *cls_b, const int64_t *label,
                                      const scae_loss_extras *extras, const float *gout12,
                                      float *g_lpp, float *g_posterior,
                                      float *g_caps_presence, float *g_cls_w, float *g_cls_b,
                                      int B, int O, int M, int ncls, int n_classes_cfg,
                                      int prior_type, int post_type, int sparsity_on,
                                      const float *weights5, float within_const, void *stream) {
  TailArgs a;
  int rc = fill_tail(a, lpp, posterior, caps_presence, cls_w, cls_b, label, B, O, M, ncls,
                     n_classes_cfg, prior_type, post_type, sparsity_on, weights5, within_const);
  if (rc) return rc;
  SCAE_REQUIRE(g_lpp && g_posterior && g_caps_presence);
  if (!gout12 && !(extras && extras->g_loss)) return SCAE_ERR_BAD_ARG;
  scae_loss_extras x{};
  if (extras) x = *extras;
  if (x.rec_sums && (x.n_rec <= 0 || !x.g_rec_sums)) return SCAE_ERR_BAD_ARG;
  if (x.reg && !x.g_reg) return SCAE_ERR_BAD_ARG;
  if (label) SCAE_REQUIRE(g_cls_w && g_cls_b);
  const size_t lds = tail_lds(B, O, true);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tail_bwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(tail_bwd_kernel, dim3(4), dim3(NT), lds, (hipStream_t)stream, a, x, gout12,
                     g_lpp, g_posterior, g_caps_presence, g_cls_w, g_cls_b);
  return scae_launch_status();
}
