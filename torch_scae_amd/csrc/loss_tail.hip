// K6 -- fused tail of SCAE.loss for gfx950.  Replaces ~125 launch-bound ATen
// ops (forward + autograd backward) of stacked_capsule_auto_encoder.py:238-285
// and object_decoder.py:433-493:
//   capsule log-likelihood term, prior sparsity (l2 | entropy | kl) on
//   caps_presence, posterior sparsity on the capsule mass / n_points, and the
//   two "cross_entropy over probabilities" classification terms (both through
//   prior_classifier, as the reference does, :207-212, :281-282).
// Decomposition: everything except the between-example terms is per image, so
//   tail_image_kernel   one wave per image: row statistics, within-example
//                       terms, both classification terms (lanes = classes),
//                       per-image partials + the capsule mass into a workspace;
//   tail_combine_kernel one workgroup: batch sums of the partials, column
//                       (between-example) statistics, the training scalar;
//   tail_bwd_kernel     one workgroup per image writes that image's gradient
//                       rows from the saved statistics, a few more workgroups
//                       reduce the classifier-parameter gradients over the batch.
// No atomics, every output has one writer, fixed summation orders.
#include "class_probs_dev.h"
#include "common.h"

namespace {
constexpr int NTI = 64;    // tail_image_kernel: one wave per image
// Workgroup sizes of the combine (ONE workgroup: its batch / column sums are chains of L2
// loads, more waves keep more of them in flight) and of the backward: at small batches
// (scae_loss_tail_defer_preferred) both run NT_SMALL threads, so that the combine can be a
// workgroup of the backward launch and an image's workgroup can form the column sums
// exactly as the combine does; at large batches 1024 and 256 (measured best at B = 1024).
// (A 1024-thread form of the small-batch backward took 16 us instead of 10.)
constexpr int NT_SMALL = 512, NTC_LARGE = 1024, NTB_LARGE = 256;
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

// workspace (floats): part (B,8) | mass (B,O) | gl (B,2,ncls) | col (2,O)
//   part[b] = {sum_m lpp, prior within_b, posterior within_b, prior xe_b,
//              posterior xe_b, row sum of caps_presence, row sum of mass / M, -}
//   mass[b][o] = sum_m posterior[b,o,m]  (un-normalised)
//   gl[b][which][c] = d xe_b / d logit_c (which: 0 prior, 1 posterior input)
//   col[0][o] = sum_b caps_presence, col[1][o] = sum_b mass / M
struct Ws {
  float *part, *mass, *gl, *col;
};
__host__ __device__ inline Ws carve_ws(float *w, int B, int O, int ncls) {
  Ws s;
  s.part = w;
  s.mass = s.part + (size_t)B * 8;
  s.gl = s.mass + (size_t)B * O;
  s.col = s.gl + (size_t)B * 2 * (ncls > 0 ? ncls : 1);
  return s;
}
inline size_t ws_floats(int B, int O, int ncls) {
  return (size_t)B * 8 + (size_t)B * O + (size_t)B * 2 * (ncls > 0 ? ncls : 1) + 2 * (size_t)O;
}

// -sum p log_safe(p*k) terms: value and d/dp
__device__ __forceinline__ float ent_term(float p, float k) {
  return -p * scae::log_safe(p * k);
}
__device__ __forceinline__ float ent_term_grad(float p, float k) {
  const float q = p * k;
  return q < scae::kLogSafeEps ? 1e8f : -(logf(q) + 1.f);
}

// within-example sparsity term of one image: x[o] in LDS, r = its row sum
__device__ __forceinline__ float within_term(const float *x, float r, int O, int type,
                                             float cw, int lane) {
  if (type == 0) return (r - cw) * (r - cw);
  const float k = type == 2 ? (float)O : 1.f;
  float w = 0.f;
  for (int o = lane; o < O; o += NTI) w += ent_term(x[o] / (r + 1e-8f), k);
  return scae::wave_sum(w);
}

// softmax(prior_classifier(x)) then cross_entropy(probs, label) (:281-282): lanes
// are classes.  Returns xe (all lanes); gl = d xe / d logit of this lane's class.
__device__ __forceinline__ float cls_xe(const TailArgs &a, const float *x, int label, int lane,
                                        float &gl) {
  const bool on = lane < a.ncls;
  float z = -INFINITY;
  if (on) {  // four weight loads in flight (the row sits in L2, not in LDS)
    const float *wr = a.cls_w + (size_t)lane * a.O;
    float z0 = a.cls_b[lane], z1 = 0.f, z2 = 0.f, z3 = 0.f;
    int o = 0;
    for (; o + 4 <= a.O; o += 4) {
      const float w0 = wr[o], w1 = wr[o + 1], w2 = wr[o + 2], w3 = wr[o + 3];
      z0 = fmaf(x[o], w0, z0), z1 = fmaf(x[o + 1], w1, z1);
      z2 = fmaf(x[o + 2], w2, z2), z3 = fmaf(x[o + 3], w3, z3);
    }
    for (; o < a.O; ++o) z0 = fmaf(x[o], wr[o], z0);
    z = (z0 + z1) + (z2 + z3);
  }
  const float mx = scae::wave_max(z);
  float p = on ? expf(z - mx) : 0.f;
  p /= scae::wave_sum(p);
  const float mx2 = scae::wave_max(on ? p : -INFINITY);
  const float q = on ? expf(p - mx2) : 0.f;
  const float sum2 = scae::wave_sum(q);
  const float plabel = scae::wave_sum(lane == label ? p : 0.f);
  const float dq = on ? q / sum2 - (lane == label ? 1.f : 0.f) : 0.f;  // d xe / d p_c
  const float dot = scae::wave_sum(p * dq);
  gl = p * (dq - dot);
  return mx2 + logf(sum2) - plabel;  // -log_softmax(p)[label]
}

// Workgroups [B, B + n_cp) are the class-probability kernel's (class_probs_dev.h): in a
// training step that launch -- same one-wave-per-image shape, independent of this one --
// rides here.
__global__ __launch_bounds__(NTI) void tail_image_kernel(TailArgs a, Ws ws, scae_cp::Args cpa,
                                                        int n_cp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int B = a.B, O = a.O, M = a.M, b = blockIdx.x, lane = threadIdx.x;
  if (b >= B) {   // workgroup-uniform
    __shared__ scae_cp::Lds s_cpl;
    scae_cp::body(cpa, s_cpl, b - B, lane);
    return;
  }
  float *s_cp = smem, *s_mass = smem + O, *s_raw = smem + 2 * O;
  float rc = 0.f, rm = 0.f;
  for (int o = lane; o < O; o += NTI) {
    const float xc = a.cp[(size_t)b * O + o];
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
    s_cp[o] = xc;
    s_raw[o] = t;       // classifier input (:210-212)
    s_mass[o] = t / M;  // mass_explained_by_capsule / n_points (:260-266)
    ws.mass[(size_t)b * O + o] = t;
    rc += xc;
    rm += t / M;
  }
  rc = scae::wave_sum(rc);
  rm = scae::wave_sum(rm);
  float lp = 0.f;
  for (int m = lane; m < M; m += NTI) lp += a.lpp[(size_t)b * M + m];
  lp = scae::wave_sum(lp);
  __syncthreads();
  float wc = 0.f, wm = 0.f;
  if (a.sparsity_on) {
    wc = within_term(s_cp, rc, O, a.prior_type, a.l2_within_const, lane);
    wm = within_term(s_mass, rm, O, a.post_type, a.l2_within_const_post, lane);
  }
  float xe_c = 0.f, xe_m = 0.f;
  if (a.label) {
    const int label = (int)a.label[b];
    float g0, g1;
    xe_c = cls_xe(a, s_cp, label, lane, g0);
    xe_m = cls_xe(a, s_raw, label, lane, g1);
    if (lane < a.ncls) {
      ws.gl[((size_t)b * 2 + 0) * a.ncls + lane] = g0;
      ws.gl[((size_t)b * 2 + 1) * a.ncls + lane] = g1;
    }
  }
  if (lane == 0) {
    float *p = ws.part + (size_t)b * 8;
    p[0] = lp, p[1] = wc, p[2] = wm, p[3] = xe_c, p[4] = xe_m, p[5] = rc, p[6] = rm, p[7] = 0.f;
  }
}

// between-example term from the column sums in LDS (first wave; result in all lanes)
__device__ __forceinline__ float between_term(const float *col, int O, int type, float cb,
                                              int lane) {
  float t = 0.f;
  if (type == 0) {
    for (int o = lane; o < O; o += 64) t += (col[o] - cb) * (col[o] - cb);
    return scae::wave_sum(t) / O;
  }
  const float k = type == 2 ? (float)O : 1.f;
  float tot = 0.f;
  for (int o = 0; o < O; ++o) tot += col[o];
  for (int o = lane; o < O; o += 64) t += ent_term(col[o] / (tot + 1e-8f), k);
  return -scae::wave_sum(t);
}

// out: [0] loss  [1] log_prob  [2] prior_within [3] prior_between [4] post_within
//      [5] post_between [6] prior_cls_xe [7] posterior_cls_xe [8] rec_ll [9] -rec_ll
//      [10] -log_prob [11] reg
// column sums over the batch into col[2 O] (LDS; also published to ws.col when asked): 16
// lanes per column, each takes every 16th image: the summation order -- and the result, bit
// for bit -- does not depend on who forms it (nor on NTC).
template <int NTC>
__device__ __forceinline__ void column_sums(const TailArgs &a, const Ws &ws, float *col, int tid,
                                            bool publish) {
  const int B = a.B, O = a.O;
  for (int e = tid; e < ((2 * O * 16 + NTC - 1) / NTC) * NTC; e += NTC) {
    const int c = e >> 4, l = e & 15, which = c / O, o = c - which * O;
    float t = 0.f;
    if (c < 2 * O) {  // (loads kept in flight: four independent partial sums)
      auto at = [&](int b) {
        return which == 0 ? a.cp[(size_t)b * O + o] : ws.mass[(size_t)b * O + o] / a.M;
      };
      float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
      int b = l;
      for (; b + 48 < B; b += 64) {
        const float u0 = at(b), u1 = at(b + 16), u2 = at(b + 32), u3 = at(b + 48);
        t0 += u0, t1 += u1, t2 += u2, t3 += u3;
      }
      for (; b < B; b += 16) t0 += at(b);
      t = (t0 + t1) + (t2 + t3);
    }
    t = scae::row_sum16(t);
    if (c < 2 * O && l == 0) {
      col[c] = t;
      if (publish) ws.col[c] = t;
    }
  }
}

// out: [0] loss  [1] log_prob  [2] prior_within [3] prior_between [4] post_within
//      [5] post_between [6] prior_cls_xe [7] posterior_cls_xe [8] rec_ll [9] -rec_ll
//      [10] -log_prob [11] reg
template <int NTC>
__device__ __forceinline__ void combine_body(const TailArgs &a, const scae_loss_extras &x,
                                             const Ws &ws, float *out, float *smem) {
  const int B = a.B, O = a.O, tid = threadIdx.x;
  float *col = smem, *red = smem + 2 * O;  // red: 6 * (NTC/64) floats
  float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int b = tid; b < B; b += NTC) {
    const float4 p = *reinterpret_cast<const float4 *>(ws.part + (size_t)b * 8);
    v[0] += p.x, v[1] += p.y, v[2] += p.z, v[3] += p.w;
    v[4] += ws.part[(size_t)b * 8 + 4];
  }
  if (x.rec_sums)
    for (int i = tid; i < x.n_rec; i += NTC) v[5] += x.rec_sums[i];
  column_sums<NTC>(a, ws, col, tid, true);
  scae::block_sum<6, NTC>(v, red);  // (contains the barriers that publish col[])
  if (tid >= 64) return;
  float pb = 0.f, qb = 0.f;
  if (a.sparsity_on) {
    pb = between_term(col, O, a.prior_type, a.l2_between_const, tid);
    qb = between_term(col + O, O, a.post_type, a.l2_between_const_post, tid);
  }
  if (tid != 0) return;
  const float log_prob = v[0] / B, pw = v[1] / B, qw = v[2] / B, xe1 = v[3] / B, xe2 = v[4] / B;
  const float rec = x.rec_sums ? v[5] / B : 0.f, reg = x.reg ? x.reg[0] : 0.f;
  out[1] = log_prob, out[2] = pw, out[3] = pb, out[4] = qw, out[5] = qb;
  out[6] = xe1, out[7] = xe2, out[8] = rec, out[9] = -rec, out[10] = -log_prob, out[11] = reg;
  const float loss = -a.w_ll * log_prob + a.w_pw * pw + a.w_pb * pb + a.w_qw * qw +
                     a.w_qb * qb + xe1 + xe2 - rec + x.w_reg * reg;
  out[0] = loss;
  if (x.loss) x.loss[0] = loss;
}
template <int NTC>
__global__ __launch_bounds__(NTC) void tail_combine_kernel(TailArgs a, scae_loss_extras x, Ws ws,
                                                          float *out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  combine_body<NTC>(a, x, ws, out, smem);
}

// d(gw * within + gb * between) / d x[b,o] for one image: x[o], col[o] in LDS
__device__ __forceinline__ void sparsity_grad(const float *x, const float *col, float r, int B,
                                              int O, int type, float cw, float cb, float gw,
                                              float gb, float *g, int lane) {
  if (type == 0) {
    for (int o = lane; o < O; o += 64)
      g[o] = gw * 2.f * (r - cw) / B + gb * 2.f * (col[o] - cb) / O;
    return;
  }
  const float k = type == 2 ? (float)O : 1.f;
  float tot = 0.f;
  for (int o = 0; o < O; ++o) tot += col[o];
  const float tinv = 1.f / (tot + 1e-8f), rinv = 1.f / (r + 1e-8f);
  float dot_b = 0.f, dot_w = 0.f;
  for (int o = lane; o < O; o += 64) {
    dot_b += ent_term_grad(col[o] * tinv, k) * col[o] * tinv;
    dot_w += ent_term_grad(x[o] * rinv, k) * x[o] * rinv;
  }
  dot_b = scae::wave_sum(dot_b);
  dot_w = scae::wave_sum(dot_w);
  for (int o = lane; o < O; o += 64) {
    const float dw = (ent_term_grad(x[o] * rinv, k) - dot_w) * rinv / B;
    const float db = -(ent_term_grad(col[o] * tinv, k) - dot_b) * tinv;
    g[o] = gw * dw + gb * db;
  }
}

// blockIdx.x < B: the gradient rows of image b.  blockIdx.x >= B: 64 outputs each
// of the classifier-parameter gradients (4 lanes per output, interleaved over b).
// x.defer_combine (a training step: nothing reads the forward's scalars before the backward
// has run): the forward left the combine workgroup out -- it is the LAST workgroup of this
// launch instead, and the image workgroups form the column sums they need themselves.
template <int NT>
__global__ __launch_bounds__(NT) void tail_bwd_kernel(TailArgs a, scae_loss_extras x, Ws ws,
                                                      const float *gout /*[12]*/, float *g_lpp,
                                                      float *g_post, float *g_cp, float *g_w,
                                                      float *g_b) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int B = a.B, O = a.O, M = a.M, tid = threadIdx.x;
  if (x.defer_combine && blockIdx.x + 1 == gridDim.x) {   // workgroup-uniform
    combine_body<NT>(a, x, ws, x.out12, smem);
    return;
  }
  // d(total)/d(component): the loss plus whatever flowed into the individually
  // exposed log entries; d/d(loss) may arrive on the 12-vector, on the separate
  // scalar, or both
  float go[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) go[i] = gout ? gout[i] : 0.f;
  if (x.g_loss) go[0] += x.g_loss[0];
  const float g0 = go[0];
  if ((int)blockIdx.x >= B) {  // classifier parameters (their inputs are detached)
    if (!(a.label && g_w)) return;
    const float g_x1 = (g0 + go[6]) / B, g_x2 = (g0 + go[7]) / B;
    const int nout = a.ncls * O + a.ncls;
    const int out = ((int)blockIdx.x - B) * (NT / 4) + (tid >> 2), part = tid & 3;
    float t = 0.f;
    // (independent loads kept in flight: the sums are L2-latency bound)
    if (out < a.ncls * O) {
      const int cc = out / O, o = out - cc * O;
      float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
      int b = part;
      for (; b + 12 < B; b += 16) {
#define SCAE_TERM(bb)                                                                  \
  (ws.gl[((size_t)(bb) * 2) * a.ncls + cc] * g_x1 * a.cp[(size_t)(bb) * O + o] +      \
   ws.gl[((size_t)(bb) * 2 + 1) * a.ncls + cc] * g_x2 * ws.mass[(size_t)(bb) * O + o])
        const float u0 = SCAE_TERM(b), u1 = SCAE_TERM(b + 4), u2 = SCAE_TERM(b + 8),
                    u3 = SCAE_TERM(b + 12);
        t0 += u0, t1 += u1, t2 += u2, t3 += u3;
      }
      for (; b < B; b += 4) t0 += SCAE_TERM(b);
#undef SCAE_TERM
      t = (t0 + t1) + (t2 + t3);
    } else if (out < nout) {
      const int cc = out - a.ncls * O;
#pragma unroll 4
      for (int b = part; b < B; b += 4)
        t += ws.gl[((size_t)b * 2) * a.ncls + cc] * g_x1 +
             ws.gl[((size_t)b * 2 + 1) * a.ncls + cc] * g_x2;
    }
    t += scae::xor1_f(t);
    t += scae::xor2_f(t);
    if (part == 0 && out < nout) {
      if (out < a.ncls * O)
        g_w[out] = t;
      else
        g_b[out - a.ncls * O] = t;
    }
    return;
  }
  const int b = blockIdx.x;
  const float g_lp = (-a.w_ll * g0 + go[1] - go[10]) / B;
  for (int m = tid; m < M; m += NT) g_lpp[(size_t)b * M + m] = g_lp;
  if (x.g_rec_sums) {  // image b fills its share
    const float gr = (-g0 + go[8] - go[9]) / B;
    const int per = (x.n_rec + B - 1) / B, i1 = min(x.n_rec, (b + 1) * per);
    for (int i = b * per + tid; i < i1; i += NT) x.g_rec_sums[i] = gr;
  }
  if (x.g_reg && b == 0 && tid == 0) x.g_reg[0] = x.w_reg * g0 + go[11];
  float *s_cp = smem, *s_mass = s_cp + O, *col = s_mass + O, *gc = col + 2 * O, *gm = gc + O;
  for (int o = tid; o < O; o += NT) {
    s_cp[o] = a.cp[(size_t)b * O + o];
    s_mass[o] = ws.mass[(size_t)b * O + o] / M;
    if (!x.defer_combine) {
      col[o] = ws.col[o];
      col[O + o] = ws.col[O + o];
    }
    gc[o] = gm[o] = 0.f;
  }
  if (x.defer_combine) column_sums<NT>(a, ws, col, tid, false);
  __syncthreads();
  if (a.sparsity_on && tid < 64) {
    const float *p = ws.part + (size_t)b * 8;
    sparsity_grad(s_cp, col, p[5], B, O, a.prior_type, a.l2_within_const, a.l2_between_const,
                  a.w_pw * g0 + go[2], a.w_pb * g0 + go[3], gc, tid);
    sparsity_grad(s_mass, col + O, p[6], B, O, a.post_type, a.l2_within_const_post,
                  a.l2_between_const_post, a.w_qw * g0 + go[4], a.w_qb * g0 + go[5], gm, tid);
  }
  __syncthreads();
  for (int o = tid; o < O; o += NT) g_cp[(size_t)b * O + o] = gc[o];
  // posterior (O+1, M) rows of image b: mass / M = sum_m post / M, one value per
  // (b, o) row; the dummy row gets zero
  float *gp = g_post + (size_t)b * (O + 1) * M;
  if ((M & 3) == 0) {
    const int M4 = M / 4;
    for (int e = tid; e < (O + 1) * M4; e += NT) {
      const int o = e / M4;
      const float v = o < O ? gm[o] / M : 0.f;
      reinterpret_cast<float4 *>(gp)[e] = make_float4(v, v, v, v);
    }
  } else {
    for (int e = tid; e < (O + 1) * M; e += NT) {
      const int o = e / M;
      gp[e] = o < O ? gm[o] / M : 0.f;
    }
  }
}
}  // namespace

extern "C" int scae_loss_tail_supported(int B, int O, int ncls) {
  return (B > 0 && O > 0 && O <= 4096 && ncls <= MAXCLS) ? 1 : 0;
}

extern "C" int64_t scae_loss_tail_workspace_floats(int B, int O, int ncls) {
  if (B <= 0 || O <= 0 || ncls < 0) return 0;
  return (int64_t)ws_floats(B, O, ncls);
}

static size_t combine_lds(int O) { return (2 * O + 6 * (NTC_LARGE / 64)) * sizeof(float); }
// the combine workgroup on its own, in the shape this batch size takes
static void launch_combine(const TailArgs &a, const scae_loss_extras &x, const Ws &ws, float *out12,
                           hipStream_t st) {
  if (scae_loss_tail_defer_preferred(a.B, a.O))
    scae::launch(tail_combine_kernel<NT_SMALL>, dim3(1), dim3(NT_SMALL), combine_lds(a.O), st,
                       a, x, ws, out12);
  else
    scae::launch(tail_combine_kernel<NTC_LARGE>, dim3(1), dim3(NTC_LARGE), combine_lds(a.O),
                       st, a, x, ws, out12);
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

static int tail_fwd(const float *lpp, const float *posterior, const float *caps_presence,
                    const float *cls_w, const float *cls_b, const int64_t *label,
                    const scae_loss_extras *extras, float *out12, float *workspace, int B, int O,
                    int M, int ncls, int n_classes_cfg, int prior_type, int post_type,
                    int sparsity_on, const float *weights5, float within_const,
                    const scae_cp::Args *cpa, void *stream) {
  TailArgs a;
  int rc = fill_tail(a, lpp, posterior, caps_presence, cls_w, cls_b, label, B, O, M, ncls,
                     n_classes_cfg, prior_type, post_type, sparsity_on, weights5, within_const);
  if (rc) return rc;
  SCAE_REQUIRE(out12 && workspace);
  scae_loss_extras x{};
  if (extras) x = *extras;
  if (x.rec_sums && x.n_rec <= 0) return SCAE_ERR_BAD_ARG;
  const Ws ws = carve_ws(workspace, B, O, ncls);
  hipStream_t st = (hipStream_t)stream;
  const int n_cp = cpa ? cpa->B + cpa->extra.n : 0;
  scae::launch(tail_image_kernel, dim3(B + n_cp), dim3(NTI), 3 * O * sizeof(float), st, a,
                     ws, cpa ? *cpa : scae_cp::Args{}, n_cp);
  if (!x.defer_combine)   // (else: the backward launch -- or scae_loss_tail_combine_f32)
    launch_combine(a, x, ws, out12, st);
  return scae_launch_status();
}

// With defer_combine every image workgroup of the backward sums the 2 O columns over the whole
// batch itself: B^2 2 O loads per launch, 3 MB at B = 128 (and a dependent launch saved) but
// 540 MB at B = 1024, where the same kernel took 288 us instead of 60 + 30.  Preferred below 8 MB.
extern "C" int scae_loss_tail_defer_preferred(int B, int O) {
  return B > 0 && O > 0 && (long)B * B * 2 * O * (long)sizeof(float) <= (8l << 20);
}

extern "C" int scae_loss_tail_combine_f32(const float *lpp, const float *posterior,
                                          const float *caps_presence, const float *cls_w,
                                          const float *cls_b, const int64_t *label,
                                          const scae_loss_extras *extras, float *out12,
                                          float *workspace, int B, int O, int M, int ncls,
                                          int n_classes_cfg, int prior_type, int post_type,
                                          int sparsity_on, const float *weights5,
                                          float within_const, void *stream) {
  TailArgs a;
  int rc = fill_tail(a, lpp, posterior, caps_presence, cls_w, cls_b, label, B, O, M, ncls,
                     n_classes_cfg, prior_type, post_type, sparsity_on, weights5, within_const);
  if (rc) return rc;
  SCAE_REQUIRE(out12 && workspace);
  scae_loss_extras x{};
  if (extras) x = *extras;
  if (x.rec_sums && x.n_rec <= 0) return SCAE_ERR_BAD_ARG;
  launch_combine(a, x, carve_ws(workspace, B, O, ncls), out12, (hipStream_t)stream);
  return scae_launch_status();
}

extern "C" int scae_loss_tail_fwd_f32(const float *lpp, const float *posterior,
                                      const float *caps_presence, const float *cls_w,
                                      const float *cls_b, const int64_t *label,
                                      const scae_loss_extras *extras, float *out12,
                                      float *workspace, int B, int O, int M, int ncls,
                                      int n_classes_cfg, int prior_type, int post_type,
                                      int sparsity_on, const float *weights5,
                                      float within_const, void *stream) {
  return tail_fwd(lpp, posterior, caps_presence, cls_w, cls_b, label, extras, out12, workspace,
                  B, O, M, ncls, n_classes_cfg, prior_type, post_type, sparsity_on, weights5,
                  within_const, nullptr, stream);
}

extern "C" int scae_loss_tail_fwd_class_probs_f32(
    const float *lpp, const float *posterior, const float *caps_presence, const float *cls_w,
    const float *cls_b, const int64_t *label, const scae_loss_extras *extras, float *out12,
    float *workspace, int B, int O, int M, int ncls, int n_classes_cfg, int prior_type,
    int post_type, int sparsity_on, const float *weights5, float within_const,
    const float *cp_caps_presence, const float *cp_posterior, const float *cp_w,
    const float *cp_bias, float *prior_prob, float *post_prob, int cp_B, int cp_O, int cp_M,
    int cp_ncls, const scae_scaled_sum *extra_sums, int n_extra, void *stream) {
  scae_cp::Args cpa;
  int rc = scae_cp::fill(cpa, cp_caps_presence, cp_posterior, cp_w, cp_bias, prior_prob,
                         post_prob, cp_B, cp_O, cp_M, cp_ncls, extra_sums, n_extra);
  if (rc) return rc;
  return tail_fwd(lpp, posterior, caps_presence, cls_w, cls_b, label, extras, out12, workspace,
                  B, O, M, ncls, n_classes_cfg, prior_type, post_type, sparsity_on, weights5,
                  within_const, &cpa, stream);
}

extern "C" int scae_loss_tail_bwd_f32(const float *lpp, const float *posterior,
                                      const float *caps_presence, const float *cls_w,
                                      const float *cls_b, const int64_t *label,
                                      const scae_loss_extras *extras, const float *gout12,
                                      const float *workspace, float *g_lpp, float *g_posterior,
                                      float *g_caps_presence, float *g_cls_w, float *g_cls_b,
                                      int B, int O, int M, int ncls, int n_classes_cfg,
                                      int prior_type, int post_type, int sparsity_on,
                                      const float *weights5, float within_const, void *stream) {
  TailArgs a;
  int rc = fill_tail(a, lpp, posterior, caps_presence, cls_w, cls_b, label, B, O, M, ncls,
                     n_classes_cfg, prior_type, post_type, sparsity_on, weights5, within_const);
  if (rc) return rc;
  SCAE_REQUIRE(g_lpp && g_posterior && g_caps_presence && workspace);
  if (!gout12 && !(extras && extras->g_loss)) return SCAE_ERR_BAD_ARG;
  scae_loss_extras x{};
  if (extras) x = *extras;
  if (x.rec_sums && (x.n_rec <= 0 || !x.g_rec_sums)) return SCAE_ERR_BAD_ARG;
  if (x.reg && !x.g_reg) return SCAE_ERR_BAD_ARG;
  if (label) SCAE_REQUIRE(g_cls_w && g_cls_b);
  const Ws ws = carve_ws(const_cast<float *>(workspace), B, O, ncls);
  if (x.defer_combine) SCAE_REQUIRE(x.out12);
  const size_t lds = 6 * O * sizeof(float) > combine_lds(O) ? 6 * O * sizeof(float) : combine_lds(O);
#define SCAE_TAIL_BWD(NTH)                                                                       \
  do {                                                                                           \
    const int cls_blocks = label ? (ncls * O + ncls + NTH / 4 - 1) / (NTH / 4) : 0;              \
    scae::launch(tail_bwd_kernel<NTH>, dim3(B + cls_blocks + (x.defer_combine ? 1 : 0)),   \
                       dim3(NTH), lds, (hipStream_t)stream, a, x, ws, gout12, g_lpp, g_posterior, \
                       g_caps_presence, g_cls_w, g_cls_b);                                       \
  } while (0)
  if (x.defer_combine || scae_loss_tail_defer_preferred(B, O))
    SCAE_TAIL_BWD(NT_SMALL);
  else
    SCAE_TAIL_BWD(NTB_LARGE);
#undef SCAE_TAIL_BWD
  return scae_launch_status();
}
