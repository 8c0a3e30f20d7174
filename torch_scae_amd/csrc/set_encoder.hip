// K2b -- fused set-transformer encoder trunk for gfx950:
//     fc1 -> L x SAB (single head, optional LayerNorm) -> fc2
// i.e. SetTransformer.forward up to (not including) the output attention,
// set_transformer.py:212-219 with SAB = MAB(x, x, presence) (:107-142).
//
// Why one kernel: at the reference's sizes (24 capsules x 16 hidden dims) a
// SAB is ~15 launch-bound ATen ops forward and ~40 backward; three of them
// plus fc1/fc2 were ~190 of the ~500 launches of a training step while doing
// < 0.1 GFLOP.  Here one wavefront owns one set: lane n holds row n of every
// (N x D) activation in registers, K/V/P tiles live in LDS, all weights of the
// trunk (a few thousand floats) are staged in LDS once per workgroup.  The
// D = 16 contractions are far below an MFMA tile, so this trunk uses VALU
// FMAs; the matrix cores are used by the wide (d = 256) output attention
// (set_attention.hip).  The backward kernel recomputes each block from its
// saved input, needs no atomics (every weight-gradient entry has one owning
// lane; per-workgroup partial sums are reduced by the caller) and is
// bit-reproducible.
#include "common.h"

namespace {

constexpr int NT = 64;        // one wavefront per workgroup; lane = set element
constexpr int NMAX = 64;      // max set size
constexpr int MAXSEG = 4;     // input given as up to 4 column segments
constexpr float kLnEps = 1e-5f;

struct Seg {
  const float *ptr;   // (B, N, width) view: element (b, n, j) at ptr[b*bs + n*rs + j]
  float *grad;        // nullable, contiguous (B, N, width)
  int width, rs;
  long bs;
};

struct StArgs {
  Seg seg[MAXSEG];
  int nseg;
  const float *presence;  // (B, N) nullable
  const float *params;    // packed, layout below
  float *z;               // (B, N, Dout)
  float *hsave;           // (B, L+1, N, D): input of every block + trunk output
  const float *gz;        // bwd: (B, N, Dout)
  float *pg_partial;      // bwd: (gridDim.x, P) per-workgroup parameter grads
  int B, N, Din, Dout, L, layer_norm;
  float sqrt_d;
};

// packed global parameter layout (floats), D = hidden width:
//   W1 [D][Din], b1 [D]
//   per layer: Wq,bq, Wk,bk, Wv,bv, Wo,bo, (ln0w, ln0b), Wf,bf, (ln1w, ln1b)
//   W2 [Dout][D], b2 [Dout]
template <int D>
struct Layout {
  int Din, Dout, L, ln;
  __host__ __device__ int layer_size() const { return 5 * (D * D + D) + (ln ? 4 * D : 0); }
  __host__ __device__ int off_w1() const { return 0; }
  __host__ __device__ int off_b1() const { return D * Din; }
  __host__ __device__ int off_layer(int l) const { return D * Din + D + l * layer_size(); }
  // within a layer
  __host__ __device__ int o_wq() const { return 0; }
  __host__ __device__ int o_bq() const { return D * D; }
  __host__ __device__ int o_wk() const { return D * D + D; }
  __host__ __device__ int o_bk() const { return 2 * D * D + D; }
  __host__ __device__ int o_wv() const { return 2 * (D * D + D); }
  __host__ __device__ int o_bv() const { return 3 * D * D + 2 * D; }
  __host__ __device__ int o_wo() const { return 3 * (D * D + D); }
  __host__ __device__ int o_bo() const { return 4 * D * D + 3 * D; }
  __host__ __device__ int o_ln0() const { return 4 * (D * D + D); }
  __host__ __device__ int o_wf() const { return 4 * (D * D + D) + (ln ? 2 * D : 0); }
  __host__ __device__ int o_bf() const { return o_wf() + D * D; }
  __host__ __device__ int o_ln1() const { return o_bf() + D; }
  __host__ __device__ int off_w2() const { return off_layer(L); }
  __host__ __device__ int off_b2() const { return off_w2() + Dout * D; }
  __host__ __device__ int total() const { return off_b2() + Dout; }
  // LDS copy of the weights: W1 is stored transposed ([Din][D]) and everything
  // is shifted so that each matrix starts 16-byte aligned
  __host__ __device__ int lds_w1t() const { return 0; }
  __host__ __device__ int lds_b1() const { return D * Din; }
  __host__ __device__ int lds_layer(int l) const { return D * Din + D + l * layer_size(); }
  __host__ __device__ int lds_total() const { return lds_layer(L); }  // fc2 stays in L2
};

// y = W x + b,  W [D][D] row-major in LDS (16-byte aligned rows)
template <int D>
__device__ __forceinline__ void linear(const float *W, const float *bias, const float (&x)[D],
                                       float (&y)[D]) {
#pragma unroll
  for (int i = 0; i < D; ++i) {
    float acc = bias[i];
    const float4 *w4 = reinterpret_cast<const float4 *>(W + i * D);
#pragma unroll
    for (int j = 0; j < D / 4; ++j) {
      const float4 w = w4[j];
      acc = fmaf(x[4 * j], w.x, acc);
      acc = fmaf(x[4 * j + 1], w.y, acc);
      acc = fmaf(x[4 * j + 2], w.z, acc);
      acc = fmaf(x[4 * j + 3], w.w, acc);
    }
    y[i] = acc;
  }
}

// gx += W^T gy
template <int D>
__device__ __forceinline__ void linear_t(const float *W, const float (&gy)[D], float (&gx)[D]) {
#pragma unroll
  for (int i = 0; i < D; ++i) {
    const float4 *w4 = reinterpret_cast<const float4 *>(W + i * D);
#pragma unroll
    for (int j = 0; j < D / 4; ++j) {
      const float4 w = w4[j];
      gx[4 * j] = fmaf(gy[i], w.x, gx[4 * j]);
      gx[4 * j + 1] = fmaf(gy[i], w.y, gx[4 * j + 1]);
      gx[4 * j + 2] = fmaf(gy[i], w.z, gx[4 * j + 2]);
      gx[4 * j + 3] = fmaf(gy[i], w.w, gx[4 * j + 3]);
    }
  }
}

template <int D>
__device__ __forceinline__ void layer_norm(const float (&x)[D], const float *gamma,
                                           const float *beta, float (&xhat)[D], float &rstd,
                                           float (&y)[D]) {
  float mean = 0.f;
#pragma unroll
  for (int i = 0; i < D; ++i) mean += x[i];
  mean *= (1.f / D);
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    const float d = x[i] - mean;
    var = fmaf(d, d, var);
  }
  rstd = 1.f / sqrtf(var * (1.f / D) + kLnEps);
#pragma unroll
  for (int i = 0; i < D; ++i) {
    xhat[i] = (x[i] - mean) * rstd;
    y[i] = fmaf(xhat[i], gamma[i], beta[i]);
  }
}

// gx = LayerNorm backward of gy (row-wise)
template <int D>
__device__ __forceinline__ void layer_norm_bwd(const float (&gy)[D], const float *gamma,
                                               const float (&xhat)[D], float rstd,
                                               float (&gx)[D]) {
  float s1 = 0.f, s2 = 0.f;
  float gh[D];
#pragma unroll
  for (int i = 0; i < D; ++i) {
    gh[i] = gy[i] * gamma[i];
    s1 += gh[i];
    s2 = fmaf(gh[i], xhat[i], s2);
  }
  s1 *= (1.f / D);
  s2 *= (1.f / D);
#pragma unroll
  for (int i = 0; i < D; ++i) gx[i] = (gh[i] - s1 - xhat[i] * s2) * rstd;
}

template <int D>
__device__ __forceinline__ void store_row(float *tile, int n, const float (&x)[D]) {
  float4 *p = reinterpret_cast<float4 *>(tile + n * (D + 4));
#pragma unroll
  for (int j = 0; j < D / 4; ++j) p[j] = make_float4(x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]);
}

// acc += sum_j a[j] * row[j]   (row: 16-byte aligned LDS)
template <int D>
__device__ __forceinline__ float dot_row(const float (&a)[D], const float *row) {
  const float4 *p = reinterpret_cast<const float4 *>(row);
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < D / 4; ++j) {
    const float4 r = p[j];
    acc = fmaf(a[4 * j], r.x, acc);
    acc = fmaf(a[4 * j + 1], r.y, acc);
    acc = fmaf(a[4 * j + 2], r.z, acc);
    acc = fmaf(a[4 * j + 3], r.w, acc);
  }
  return acc;
}
template <int D>
__device__ __forceinline__ void axpy_row(float s, const float *row, float (&y)[D]) {
  const float4 *p = reinterpret_cast<const float4 *>(row);
#pragma unroll
  for (int j = 0; j < D / 4; ++j) {
    const float4 r = p[j];
    y[4 * j] = fmaf(s, r.x, y[4 * j]);
    y[4 * j + 1] = fmaf(s, r.y, y[4 * j + 1]);
    y[4 * j + 2] = fmaf(s, r.z, y[4 * j + 2]);
    y[4 * j + 3] = fmaf(s, r.w, y[4 * j + 3]);
  }
}

// One SAB forward for lane-row n.  K / V rows go through LDS tiles s_k / s_v,
// the attention probabilities through s_p ([NMAX][NMAX+1], own row per lane).
// Everything the backward needs is returned by reference.
template <int D>
struct BlockState {
  float q[D], a[D], xhat0[D], h1n[D], t[D], xhat1[D];
  float rstd0, rstd1, pn;
};

template <int D>
__device__ __forceinline__ void sab_forward(const Layout<D> &lay, const float *W /*layer base in LDS*/,
                                            const float *presence_b, int N, int lane,
                                            float sqrt_d, const float (&h)[D], float *s_k,
                                            float *s_v, float *s_p, BlockState<D> &st,
                                            float (&out)[D]) {
  constexpr int TS = D + 4;
  const int ps = N + 1;           // row stride of the probability tile
  const bool live = lane < N;     // lanes beyond the set only keep barriers company
  float k[D], v[D];
  linear<D>(W + lay.o_wq(), W + lay.o_bq(), h, st.q);
  linear<D>(W + lay.o_wk(), W + lay.o_bk(), h, k);
  linear<D>(W + lay.o_wv(), W + lay.o_bv(), h, v);
  __syncthreads();  // previous readers of the tiles are done
  if (live) {
    store_row<D>(s_k, lane, k);
    store_row<D>(s_v, lane, v);
  }
  __syncthreads();
  // routing = (q k^T - (1 - presence) 1e32) / sqrt(d); softmax over keys
  float *prow = s_p + (live ? lane : 0) * ps;
  const int Nl = live ? N : 0;
  float mx = -INFINITY;
  for (int m = 0; m < Nl; ++m) {
    float s = dot_row<D>(st.q, s_k + m * TS);
    if (presence_b) s = s - (1.f - presence_b[m]) * 1e32f;
    s = s / sqrt_d;
    prow[m] = s;
    mx = fmaxf(mx, s);
  }
  float sum = 0.f;
  for (int m = 0; m < Nl; ++m) {
    const float e = expf(prow[m] - mx);
    prow[m] = e;
    sum += e;
  }
#pragma unroll
  for (int i = 0; i < D; ++i) st.a[i] = 0.f;
  for (int m = 0; m < Nl; ++m) {
    const float p = prow[m] / sum;
    prow[m] = p;
    axpy_row<D>(p, s_v + m * TS, st.a);
  }
  float o[D], r[D];
  linear<D>(W + lay.o_wo(), W + lay.o_bo(), st.a, o);
  st.pn = presence_b ? presence_b[lane < N ? lane : 0] : 1.f;
#pragma unroll
  for (int i = 0; i < D; ++i) r[i] = (o[i] + h[i]) * st.pn;  // residual, presence gate
  if (lay.ln) {
    layer_norm<D>(r, W + lay.o_ln0(), W + lay.o_ln0() + D, st.xhat0, st.rstd0, st.h1n);
  } else {
#pragma unroll
    for (int i = 0; i < D; ++i) st.h1n[i] = r[i];
  }
  linear<D>(W + lay.o_wf(), W + lay.o_bf(), st.h1n, st.t);
  float h2[D];
#pragma unroll
  for (int i = 0; i < D; ++i) h2[i] = st.h1n[i] + fmaxf(st.t[i], 0.f);
  if (lay.ln) {
    layer_norm<D>(h2, W + lay.o_ln1(), W + lay.o_ln1() + D, st.xhat1, st.rstd1, out);
  } else {
#pragma unroll
    for (int i = 0; i < D; ++i) out[i] = h2[i];
  }
}

template <int D>
__device__ __forceinline__ void stage_weights(const Layout<D> &lay, const float *params,
                                              float *s_w) {
  const int Din = lay.Din;
  for (int i = threadIdx.x; i < D * Din; i += NT) {  // W1 -> transposed
    const int r = i / Din, c = i - r * Din;
    s_w[lay.lds_w1t() + c * D + r] = params[lay.off_w1() + i];
  }
  const int rest = lay.off_w2() - lay.off_b1();
  for (int i = threadIdx.x; i < rest; i += NT) s_w[lay.lds_b1() + i] = params[lay.off_b1() + i];
}

template <int D>
__device__ __forceinline__ void stage_input(const StArgs &a, int b, float *s_x, int DinP) {
  int col = 0;
  for (int s = 0; s < a.nseg; ++s) {
    const Seg &sg = a.seg[s];
    for (int i = threadIdx.x; i < a.N * sg.width; i += NT) {
      const int n = i / sg.width, j = i - n * sg.width;
      s_x[n * DinP + col + j] = sg.ptr[(size_t)b * sg.bs + (size_t)n * sg.rs + j];
    }
    col += sg.width;
  }
}

// LDS carve shared by forward and backward
template <int D>
struct Smem {
  int w, x, k, v, p, total_fwd;
  // with_x: the forward keeps the input rows in LDS; the backward reads them
  // from global memory instead (it only needs them column-wise, coalesced)
  __host__ __device__ Smem(const Layout<D> &lay, int N, int DinP, bool with_x) {
    w = 0;
    x = (lay.lds_total() + 3) & ~3;
    k = (x + (with_x ? N * DinP : 0) + 3) & ~3;
    v = k + N * (D + 4);
    p = v + N * (D + 4);
    total_fwd = (p + N * (N + 1) + 3) & ~3;
  }
};

template <int D>
__global__ __launch_bounds__(NT) void st_fwd_kernel(StArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Layout<D> lay{a.Din, a.Dout, a.L, a.layer_norm};
  const int N = a.N, Din = a.Din, DinP = Din | 1, lane = threadIdx.x;
  const Smem<D> sm(lay, N, DinP, true);
  float *s_w = smem + sm.w, *s_x = smem + sm.x, *s_k = smem + sm.k, *s_v = smem + sm.v,
        *s_p = smem + sm.p;
  constexpr int TS = D + 4;
  stage_weights<D>(lay, a.params, s_w);
  const int n = lane < N ? lane : 0;

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    __syncthreads();
    stage_input<D>(a, b, s_x, DinP);
    __syncthreads();
    const float *presence_b = a.presence ? a.presence + (size_t)b * N : nullptr;
    float h[D];
#pragma unroll
    for (int i = 0; i < D; ++i) h[i] = s_w[lay.lds_b1() + i];
    for (int j = 0; j < Din; ++j) axpy_row<D>(s_x[n * DinP + j], s_w + lay.lds_w1t() + j * D, h);

    float *hs = a.hsave + (size_t)b * (a.L + 1) * N * D;
    for (int l = 0; l < a.L; ++l) {
      if (lane < N) {
#pragma unroll
        for (int i = 0; i < D; ++i) hs[((size_t)l * N + lane) * D + i] = h[i];
      }
      BlockState<D> st;
      float out[D];
      sab_forward<D>(lay, s_w + lay.lds_layer(l), presence_b, N, lane, a.sqrt_d, h, s_k, s_v,
                     s_p, st, out);
#pragma unroll
      for (int i = 0; i < D; ++i) h[i] = out[i];
    }
    if (lane < N) {
#pragma unroll
      for (int i = 0; i < D; ++i) hs[((size_t)a.L * N + lane) * D + i] = h[i];
    }
    // fc2: lanes own output columns, rows come from an LDS tile
    __syncthreads();
    if (lane < N) store_row<D>(s_k, lane, h);
    __syncthreads();
    for (int c = lane; c < a.Dout; c += NT) {
      float w[D];
#pragma unroll
      for (int j = 0; j < D; ++j) w[j] = a.params[lay.off_w2() + c * D + j];
      const float bias = a.params[lay.off_b2() + c];
      for (int m = 0; m < N; ++m)
        a.z[((size_t)b * N + m) * a.Dout + c] = bias + dot_row<D>(w, s_k + m * TS);
    }
  }
}

// dW[i][j] += sum_n gy[n][i] x[n][j] for one D x D matrix (+ bias grad).
// gy / x are LDS tiles; lane owns row i = lane % D, column block lane / D.
template <int D>
__device__ __forceinline__ void weight_grad(const float *s_gy, const float *s_xin, int N,
                                            float *pg_w, float *pg_b) {
  constexpr int TS = D + 4;
  constexpr int NB = NT / D;        // column blocks handled in parallel
  constexpr int CW = D / NB;        // columns per lane
  const int i = threadIdx.x % D, cb = threadIdx.x / D;
  float acc[CW];
#pragma unroll
  for (int c = 0; c < CW; ++c) acc[c] = 0.f;
  float bacc = 0.f;
  for (int m = 0; m < N; ++m) {
    const float g = s_gy[m * TS + i];
    bacc += g;
#pragma unroll
    for (int c = 0; c < CW; ++c) acc[c] = fmaf(g, s_xin[m * TS + cb * CW + c], acc[c]);
  }
#pragma unroll
  for (int c = 0; c < CW; ++c) pg_w[i * D + cb * CW + c] += acc[c];
  if (pg_b && cb == 0) pg_b[i] += bacc;
}

// per-feature sums over rows of a tile product:  gamma += sum_n gy*xhat, beta += sum_n gy
template <int D>
__device__ __forceinline__ void ln_param_grad(const float *s_gy, const float *s_xhat, int N,
                                              float *pg_gamma) {
  constexpr int TS = D + 4;
  if (threadIdx.x < D) {
    float ga = 0.f, be = 0.f;
    for (int m = 0; m < N; ++m) {
      const float g = s_gy[m * TS + threadIdx.x];
      ga = fmaf(g, s_xhat[m * TS + threadIdx.x], ga);
      be += g;
    }
    pg_gamma[threadIdx.x] += ga;
    pg_gamma[D + threadIdx.x] += be;
  }
}

template <int D>
__global__ __launch_bounds__(NT) void st_bwd_kernel(StArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Layout<D> lay{a.Din, a.Dout, a.L, a.layer_norm};
  const int N = a.N, Din = a.Din, DinP = Din | 1, lane = threadIdx.x;
  const Smem<D> sm(lay, N, DinP, false);
  constexpr int TS = D + 4;
  const int ps = N + 1;
  float *s_w = smem + sm.w, *s_k = smem + sm.k, *s_v = smem + sm.v, *s_p = smem + sm.p;
  float *s_gs = smem + sm.total_fwd;               // [N][max(N, 64) + 1]
  const int gs_stride = (N > NT ? N : NT) + 1;
  float *s_t0 = s_gs + ((N * gs_stride + 3) & ~3); // four more [N][TS] tiles
  float *s_t1 = s_t0 + N * TS;
  float *s_t2 = s_t1 + N * TS;
  float *s_t3 = s_t2 + N * TS;
  float *s_pg = s_t3 + N * TS;                     // [P] parameter-gradient accumulators
  const int P = lay.total();
  stage_weights<D>(lay, a.params, s_w);
  for (int i = lane; i < P; i += NT) s_pg[i] = 0.f;
  const int n = lane < N ? lane : 0;
  const bool live = lane < N;

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    __syncthreads();
    const float *presence_b = a.presence ? a.presence + (size_t)b * N : nullptr;
    const float *hs = a.hsave + (size_t)b * (a.L + 1) * N * D;
    const float *gzb = a.gz + (size_t)b * N * a.Dout;

    // ---- fc2 backward ---------------------------------------------------
    float hL[D];
#pragma unroll
    for (int i = 0; i < D; ++i) hL[i] = hs[((size_t)a.L * N + n) * D + i];
    if (live) store_row<D>(s_k, lane, hL);
    __syncthreads();
    float g[D];  // gradient w.r.t. the current block output, lane-row n
#pragma unroll
    for (int i = 0; i < D; ++i) g[i] = 0.f;
    for (int c0 = 0; c0 < a.Dout; c0 += NT) {
      const int c = c0 + lane;
      const bool cl = c < a.Dout;
      // (1) lanes = columns: dW2 / db2, and park the gz tile in LDS
      float acc[D];
#pragma unroll
      for (int j = 0; j < D; ++j) acc[j] = 0.f;
      float bacc = 0.f;
      __syncthreads();
      for (int m = 0; m < N; ++m) {
        const float gv = cl ? gzb[(size_t)m * a.Dout + c] : 0.f;
        s_gs[m * gs_stride + lane] = gv;
        bacc += gv;
        axpy_row<D>(gv, s_k + m * TS, acc);
      }
      if (cl) {
#pragma unroll
        for (int j = 0; j < D; ++j) s_pg[lay.off_w2() + c * D + j] += acc[j];
        s_pg[lay.off_b2() + c] += bacc;
      }
      __syncthreads();
      // (2) lanes = rows: g += gz[n][c] * W2[c][:]
      const int cn = min(NT, a.Dout - c0);
      for (int cc = 0; cc < cn; ++cc) {
        const float gv = s_gs[n * gs_stride + cc];
        const float *w2 = a.params + lay.off_w2() + (size_t)(c0 + cc) * D;
#pragma unroll
        for (int j = 0; j < D; ++j) g[j] = fmaf(gv, w2[j], g[j]);
      }
    }

    // ---- SAB blocks, last to first ---------------------------------------
    for (int l = a.L - 1; l >= 0; --l) {
      const float *W = s_w + lay.lds_layer(l);
      float *PG = s_pg + lay.off_layer(l);
      float h[D];
#pragma unroll
      for (int i = 0; i < D; ++i) h[i] = hs[((size_t)l * N + n) * D + i];
      BlockState<D> st;
      float out[D];
      sab_forward<D>(lay, W, presence_b, N, lane, a.sqrt_d, h, s_k, s_v, s_p, st, out);
      if (!live) {
#pragma unroll
        for (int i = 0; i < D; ++i) g[i] = 0.f;
      }
      // LN1 backward
      float g_h2[D];
      if (lay.ln) {
        __syncthreads();
        if (live) {
          store_row<D>(s_t0, lane, g);
          store_row<D>(s_t1, lane, st.xhat1);
        }
        __syncthreads();
        ln_param_grad<D>(s_t0, s_t1, N, PG + lay.o_ln1());
        layer_norm_bwd<D>(g, W + lay.o_ln1(), st.xhat1, st.rstd1, g_h2);
      } else {
#pragma unroll
        for (int i = 0; i < D; ++i) g_h2[i] = g[i];
      }
      // h2 = h1n + relu(Wf h1n + bf)
      float g_t[D], g_h1n[D];
#pragma unroll
      for (int i = 0; i < D; ++i) {
        g_t[i] = st.t[i] > 0.f ? g_h2[i] : 0.f;
        g_h1n[i] = g_h2[i];
      }
      linear_t<D>(W + lay.o_wf(), g_t, g_h1n);
      __syncthreads();
      if (live) {
        store_row<D>(s_t0, lane, g_t);
        store_row<D>(s_t1, lane, st.h1n);
      }
      __syncthreads();
      weight_grad<D>(s_t0, s_t1, N, PG + lay.o_wf(), PG + lay.o_bf());
      // LN0 backward
      float g_r[D];
      if (lay.ln) {
        __syncthreads();
        if (live) {
          store_row<D>(s_t0, lane, g_h1n);
          store_row<D>(s_t1, lane, st.xhat0);
        }
        __syncthreads();
        ln_param_grad<D>(s_t0, s_t1, N, PG + lay.o_ln0());
        layer_norm_bwd<D>(g_h1n, W + lay.o_ln0(), st.xhat0, st.rstd0, g_r);
      } else {
#pragma unroll
        for (int i = 0; i < D; ++i) g_r[i] = g_h1n[i];
      }
      // r = (o + h) * presence_n
      float g_o[D], g_h[D];
#pragma unroll
      for (int i = 0; i < D; ++i) {
        g_o[i] = live ? g_r[i] * st.pn : 0.f;
        g_h[i] = g_o[i];
      }
      // o = Wo a + bo
      float g_a[D];
#pragma unroll
      for (int i = 0; i < D; ++i) g_a[i] = 0.f;
      linear_t<D>(W + lay.o_wo(), g_o, g_a);
      __syncthreads();
      if (live) {
        store_row<D>(s_t0, lane, g_o);
        store_row<D>(s_t1, lane, st.a);
        store_row<D>(s_t2, lane, g_a);
        store_row<D>(s_t3, lane, st.q);
      }
      __syncthreads();
      weight_grad<D>(s_t0, s_t1, N, PG + lay.o_wo(), PG + lay.o_bo());
      // a = P v ; softmax ; s = q k^T / sqrt(d)
      const float *prow = s_p + n * ps;
      float *gsrow = s_gs + n * gs_stride;
      const int Nl = live ? N : 0;
      float dotp = 0.f;
      for (int m = 0; m < Nl; ++m) {
        const float gp = dot_row<D>(g_a, s_v + m * TS);  // dL/dP[n][m]
        gsrow[m] = gp;
        dotp = fmaf(prow[m], gp, dotp);
      }
      float g_q[D];
#pragma unroll
      for (int i = 0; i < D; ++i) g_q[i] = 0.f;
      for (int m = 0; m < Nl; ++m) {
        const float gs = prow[m] * (gsrow[m] - dotp) / a.sqrt_d;
        gsrow[m] = gs;
        axpy_row<D>(gs, s_k + m * TS, g_q);
      }
      __syncthreads();
      // transposed sums: lane m gathers over query rows
      float g_k[D], g_v[D];
#pragma unroll
      for (int i = 0; i < D; ++i) g_k[i] = g_v[i] = 0.f;
      if (live) {
        for (int r = 0; r < N; ++r) {
          axpy_row<D>(s_gs[r * gs_stride + lane], s_t3 + r * TS, g_k);
          axpy_row<D>(s_p[r * ps + lane], s_t2 + r * TS, g_v);
        }
      }
      linear_t<D>(W + lay.o_wq(), g_q, g_h);
      linear_t<D>(W + lay.o_wk(), g_k, g_h);
      linear_t<D>(W + lay.o_wv(), g_v, g_h);
      // weight grads of the three input projections (x = h)
      __syncthreads();
      if (live) {
        store_row<D>(s_t0, lane, g_q);
        store_row<D>(s_t1, lane, h);
        store_row<D>(s_t2, lane, g_k);
        store_row<D>(s_t3, lane, g_v);
      }
      __syncthreads();
      weight_grad<D>(s_t0, s_t1, N, PG + lay.o_wq(), PG + lay.o_bq());
      weight_grad<D>(s_t2, s_t1, N, PG + lay.o_wk(), PG + lay.o_bk());
      weight_grad<D>(s_t3, s_t1, N, PG + lay.o_wv(), PG + lay.o_bv());
#pragma unroll
      for (int i = 0; i < D; ++i) g[i] = g_h[i];
    }

    // ---- fc1 backward -----------------------------------------------------
    if (!live) {
#pragma unroll
      for (int i = 0; i < D; ++i) g[i] = 0.f;
    }
    __syncthreads();
    if (live) store_row<D>(s_t0, lane, g);
    __syncthreads();
    for (int j = lane; j < Din; j += NT) {  // dW1[:, j]; x read column-wise from global
      int sj = 0, cj = j;
      while (cj >= a.seg[sj].width) cj -= a.seg[sj++].width;
      const Seg &xs = a.seg[sj];
      const float *xcol = xs.ptr + (size_t)b * xs.bs + cj;
      float acc[D];
#pragma unroll
      for (int i = 0; i < D; ++i) acc[i] = 0.f;
      for (int m = 0; m < N; ++m) axpy_row<D>(xcol[(size_t)m * xs.rs], s_t0 + m * TS, acc);
#pragma unroll
      for (int i = 0; i < D; ++i) s_pg[lay.off_w1() + i * Din + j] += acc[i];
    }
    if (lane < D) {
      float bacc = 0.f;
      for (int m = 0; m < N; ++m) bacc += s_t0[m * TS + lane];
      s_pg[lay.off_b1() + lane] += bacc;
    }
    // input gradients for the segments that want one
    int col = 0;
    for (int s = 0; s < a.nseg; ++s) {
      const Seg &sg = a.seg[s];
      if (sg.grad && live) {
        for (int j = 0; j < sg.width; ++j)
          sg.grad[((size_t)b * N + lane) * sg.width + j] =
              dot_row<D>(g, s_w + lay.lds_w1t() + (col + j) * D);
      }
      col += sg.width;
    }
  }
  __syncthreads();
  float *dst = a.pg_partial + (size_t)blockIdx.x * P;
  for (int i = lane; i < P; i += NT) dst[i] = s_pg[i];
}

template <int D>
size_t lds_bytes(const StArgs &a, bool bwd) {
  const Layout<D> lay{a.Din, a.Dout, a.L, a.layer_norm};
  const Smem<D> sm(lay, a.N, a.Din | 1, !bwd);
  size_t f = sm.total_fwd;
  const int gs_stride = (a.N > NT ? a.N : NT) + 1;
  if (bwd) f += ((a.N * gs_stride + 3) & ~3) + 4 * a.N * (D + 4) + lay.total();
  return f * sizeof(float);
}

template <int D>
int launch(const StArgs &a, bool bwd, int grid, hipStream_t st) {
  const size_t lds = lds_bytes<D>(a, bwd);
  if (lds > 160 * 1024) return SCAE_ERR_UNSUPPORTED;
  const void *fn = bwd ? reinterpret_cast<const void *>(st_bwd_kernel<D>)
                       : reinterpret_cast<const void *>(st_fwd_kernel<D>);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  if (bwd)
    hipLaunchKernelGGL(st_bwd_kernel<D>, dim3(grid), dim3(NT), lds, st, a);
  else
    hipLaunchKernelGGL(st_fwd_kernel<D>, dim3(grid), dim3(NT), lds, st, a);
  return scae_launch_status();
}

int fill_args(StArgs &a, int nseg, const float *const *seg_ptr, const int *seg_width,
              const int *seg_row_stride, const int64_t *seg_batch_stride,
              float *const *seg_grad, const float *presence, const float *params, int B, int N,
              int D, int Din, int Dout, int L, int layer_norm) {
  if (nseg < 1 || nseg > MAXSEG || !seg_ptr || !seg_width || !seg_row_stride ||
      !seg_batch_stride || !params)
    return SCAE_ERR_BAD_ARG;
  if (B <= 0 || N <= 0 || Din <= 0 || Dout <= 0 || L < 0) return SCAE_ERR_BAD_ARG;
  if (N > NMAX || (D != 8 && D != 16 && D != 32)) return SCAE_ERR_UNSUPPORTED;
  int tot = 0;
  for (int s = 0; s < nseg; ++s) {
    if (!seg_ptr[s] || seg_width[s] <= 0) return SCAE_ERR_BAD_ARG;
    a.seg[s] = Seg{seg_ptr[s], seg_grad ? seg_grad[s] : nullptr, seg_width[s],
                   seg_row_stride[s], (long)seg_batch_stride[s]};
    tot += seg_width[s];
  }
  if (tot != Din) return SCAE_ERR_BAD_ARG;
  a.nseg = nseg;
  a.presence = presence;
  a.params = params;
  a.B = B;
  a.N = N;
  a.Din = Din;
  a.Dout = Dout;
  a.L = L;
  a.layer_norm = layer_norm;
  a.sqrt_d = sqrtf((float)D);
  return SCAE_OK;
}

}  // namespace

extern "C" int scae_set_encoder_param_count(int D, int Din, int Dout, int L, int layer_norm) {
  return D * Din + D + L * (5 * (D * D + D) + (layer_norm ? 4 * D : 0)) + Dout * D + Dout;
}

extern "C" int scae_set_encoder_grid(int B) { return B < 256 ? B : 256; }

extern "C" int scae_set_encoder_supported(int N, int D, int Din, int Dout, int L,
                                          int layer_norm) {
  if (N <= 0 || N > NMAX || Din <= 0 || Dout <= 0 || L < 0) return 0;
  StArgs a{};
  a.N = N;
  a.Din = Din;
  a.Dout = Dout;
  a.L = L;
  a.layer_norm = layer_norm;
  size_t need;
  switch (D) {
    case 8: need = lds_bytes<8>(a, true); break;
    case 16: need = lds_bytes<16>(a, true); break;
    case 32: need = lds_bytes<32>(a, true); break;
    default: return 0;
  }
  return need <= 160 * 1024 ? 1 : 0;
}

extern "C" int scae_set_encoder_fwd_f32(int nseg, const float *const *seg_ptr,
                                        const int *seg_width, const int *seg_row_stride,
                                        const int64_t *seg_batch_stride, const float *presence,
                                        const float *params, float *z, float *hsave, int B,
                                        int N, int D, int Din, int Dout, int L, int layer_norm,
                                        void *stream) {
  StArgs a{};
  int rc = fill_args(a, nseg, seg_ptr, seg_width, seg_row_stride, seg_batch_stride, nullptr,
                     presence, params, B, N, D, Din, Dout, L, layer_norm);
  if (rc) return rc;
  SCAE_REQUIRE(z && hsave);
  a.z = z;
  a.hsave = hsave;
  const int grid = scae_set_encoder_grid(B);
  switch (D) {
    case 8: return launch<8>(a, false, grid, (hipStream_t)stream);
    case 16: return launch<16>(a, false, grid, (hipStream_t)stream);
    default: return launch<32>(a, false, grid, (hipStream_t)stream);
  }
}

extern "C" int scae_set_encoder_bwd_f32(int nseg, const float *const *seg_ptr,
                                        const int *seg_width, const int *seg_row_stride,
                                        const int64_t *seg_batch_stride,
                                        float *const *seg_grad, const float *presence,
                                        const float *params, const float *hsave,
                                        const float *gz, float *pg_partial, int B, int N, int D,
                                        int Din, int Dout, int L, int layer_norm, void *stream) {
  StArgs a{};
  int rc = fill_args(a, nseg, seg_ptr, seg_width, seg_row_stride, seg_batch_stride, seg_grad,
                     presence, params, B, N, D, Din, Dout, L, layer_norm);
  if (rc) return rc;
  SCAE_REQUIRE(hsave && gz && pg_partial);
  a.hsave = const_cast<float *>(hsave);
  a.gz = gz;
  a.pg_partial = pg_partial;
  const int grid = scae_set_encoder_grid(B);
  switch (D) {
    case 8: return launch<8>(a, true, grid, (hipStream_t)stream);
    case 16: return launch<16>(a, true, grid, (hipStream_t)stream);
    default: return launch<32>(a, true, grid, (hipStream_t)stream);
  }
}
