// K2c -- output attention of the set transformer, gfx950:
//   MultiHeadQKVAttention(seeds, z, z, presence) with z = fc2(h)
//   (set_transformer.py:218-223 + :68-104 with n_heads = 1).
//
// Algebra used (exact up to fp32 re-association): z enters the attention only
// through the linear maps k = Wk z + bk and v = Wv z + bv, and the output
// projection is linear too, so with h the (N x D) trunk output
//     K' = h (Wk W2)^T + (Wk b2 + bk)                         (N x C)
//     V' = h (Wo Wv W2)^T + (Wo (Wv b2 + bv) + bo)            (N x C)
//     out = softmax((q K'^T - (1 - presence) 1e32) / sqrt(C)) V'
// (rows of the softmax sum to one, which is what lets bo ride inside V').
// The three C x C = 256 x 256 projections of 3072 rows each -- ~1.2 GFLOP
// forward, the largest GEMMs of the object encoder -- collapse into two
// (C x D) = 256 x 16 maps that are evaluated INSIDE this kernel from LDS; the
// folding products themselves are a few tiny batch-invariant GEMMs done by the
// caller (and differentiated by autograd).  q = Wq seeds + bq is batch
// invariant as well and is passed in.
//
// One workgroup (16 waves) per set, element-parallel stages on LDS tiles like
// set_encoder.hip; the presence mask arithmetic is the reference's fp32
// sequence.  Backward recomputes K', V', writes h-gradients once and leaves
// per-workgroup partial parameter gradients for the caller to sum.
#include <algorithm>

#include "common.h"

namespace {
constexpr int NT = 1024;
constexpr int NMAX = 64;

struct SaArgs {
  const float *h;         // (B,N,D)
  const float *q;         // (O,C)
  const float *wk, *bk;   // (C,D), (C)
  const float *wv, *bv;   // (C,D), (C)
  const float *presence;  // (B,N) nullable
  float *out;             // (B,O,C)
  float *probs;           // (B,O,N)
  const float *gout;      // bwd (B,O,C)
  float *gh;              // bwd (B,N,D)
  float *partial;         // bwd (grid, O*C + 2*C*D + 2*C): [gq | gwk | gbk | gwv | gbv]
  int B, N, O, C;
  float sqrt_c;
  int splits;  // query groups per set (grid = B * splits workgroups)
};

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

__device__ __forceinline__ float dot4(const float *a, const float *b, int n4) {
  const float4 *pa = reinterpret_cast<const float4 *>(a);
  const float4 *pb = reinterpret_cast<const float4 *>(b);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
  for (int j = 0; j < n4; ++j) {
    const float4 x = pa[j], y = pb[j];
    a0 = fmaf(x.x, y.x, a0);
    a1 = fmaf(x.y, y.y, a1);
    a2 = fmaf(x.z, y.z, a2);
    a3 = fmaf(x.w, y.w, a3);
  }
  return (a0 + a1) + (a2 + a3);
}

__device__ __forceinline__ float4 ld4s(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4s(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ void fma4(float4 &acc, float s, const float4 &v) {
  acc.x = fmaf(s, v.x, acc.x), acc.y = fmaf(s, v.y, acc.y);
  acc.z = fmaf(s, v.z, acc.z), acc.w = fmaf(s, v.w, acc.w);
}
// a C-long dot product shared by 4 consecutive lanes, met by shuffles: lane `part`
// takes the float4s part, part + 4, ... (interleaved, so that the four lanes read
// consecutive 16-byte units: no bank conflict among them).  Needs C % 16 == 0.
__device__ __forceinline__ float dot_quarter(const float *a, const float *b, int C, int part) {
  const float4 *pa = reinterpret_cast<const float4 *>(a) + part;
  const float4 *pb = reinterpret_cast<const float4 *>(b) + part;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
  for (int j = 0; j < C / 16; ++j) {
    const float4 x = pa[4 * j], y = pb[4 * j];
    a0 = fmaf(x.x, y.x, a0);
    a1 = fmaf(x.y, y.y, a1);
    a2 = fmaf(x.z, y.z, a2);
    a3 = fmaf(x.w, y.w, a3);
  }
  float s = (a0 + a1) + (a2 + a3);
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  return s;
}

template <int D>
struct Carve {
  float *h, *q, *wk, *wv, *bk, *bv, *K, *V, *S, *GO, *GS;
  int CS;  // padded row stride of the (.. x C) tiles
};

template <int D>
__host__ __device__ size_t carve(int N, int O, int C, bool bwd, float *base, Carve<D> *out) {
  constexpr int TS = D + 4;
  size_t o = 0;
  auto take = [&](size_t n) {
    float *p = base ? base + o : nullptr;
    o += (n + 3) & ~(size_t)3;
    return p;
  };
  Carve<D> r{};
  r.CS = C + 4;  // C multiple of 8 -> (C+4)/4 odd: conflict-free row-per-lane b128 reads
  r.h = take((size_t)N * TS);
  r.q = take((size_t)O * r.CS);
  r.wk = take((size_t)C * TS);
  r.wv = take((size_t)C * TS);
  r.bk = take(C);
  r.bv = take(C);
  r.K = take((size_t)N * r.CS);
  r.V = take((size_t)N * r.CS);
  r.S = take((size_t)O * (N + 1));
  if (bwd) {
    r.GO = take((size_t)O * r.CS);
    r.GS = take((size_t)O * (N + 1));
  }
  if (out) *out = r;
  return o;
}

// weights, and the Og query rows starting at o0 (local row index in the tiles)
template <int D>
__device__ __forceinline__ void stage_common(const SaArgs &a, const Carve<D> &c, int o0, int Og) {
  constexpr int TS = D + 4;
  const int C = a.C, O = Og, tid = threadIdx.x;
  const bool vec = (((size_t)a.wk | (size_t)a.wv | (size_t)a.q) & 15) == 0;  // C % 8 == 0
  if (vec) {  // 16-byte copies: (C x D) dense rows -> TS-strided rows, q rows -> CS-strided
    constexpr int Q = D / 4;
    for (int e = tid; e < C * Q; e += NT) {
      const int r = e / Q, j4 = 4 * (e - r * Q);
      st4s(c.wk + r * TS + j4, ld4s(a.wk + (size_t)r * D + j4));
      st4s(c.wv + r * TS + j4, ld4s(a.wv + (size_t)r * D + j4));
    }
    const int C4 = C / 4;
    for (int e = tid; e < O * C4; e += NT) {
      const int o = e / C4, c4 = 4 * (e - o * C4);
      st4s(c.q + o * c.CS + c4, ld4s(a.q + ((size_t)o0 + o) * C + c4));
    }
  } else {
#pragma unroll 4
    for (int e = tid; e < C * D; e += NT) {
      const int r = e / D, j = e - r * D;
      c.wk[r * TS + j] = a.wk[e];
      c.wv[r * TS + j] = a.wv[e];
    }
#pragma unroll 4
    for (int e = tid; e < O * C; e += NT) {
      const int o = e / C, cc = e - o * C;
      c.q[o * c.CS + cc] = a.q[(size_t)o0 * C + e];
    }
  }
  for (int e = tid; e < C; e += NT) {
    c.bk[e] = a.bk[e];
    c.bv[e] = a.bv[e];
  }
}

// K', V', routing, softmax for sample b; leaves P in c.S.  Ends with a barrier.
template <int D>
__device__ __forceinline__ void forward_core(const SaArgs &a, const Carve<D> &c, int b, int Og) {
  constexpr int TS = D + 4;
  const int N = a.N, O = Og, C = a.C, CS = c.CS, NS = N + 1, tid = threadIdx.x;
  for (int e = tid; e < N * D; e += NT)
    c.h[(e / D) * TS + (e % D)] = a.h[(size_t)b * N * D + e];
  __syncthreads();
  if (NT % C == 0) {  // K' and V': a thread keeps its column's two weight rows in registers
    const int cc = tid % C;
    float4 wkr[D / 4], wvr[D / 4];
#pragma unroll
    for (int j = 0; j < D / 4; ++j) {
      wkr[j] = ld4s(c.wk + cc * TS + 4 * j);
      wvr[j] = ld4s(c.wv + cc * TS + 4 * j);
    }
    const float bkc = c.bk[cc], bvc = c.bv[cc];
    for (int m = tid / C; m < N; m += NT / C) {
      float k0 = 0.f, k1 = 0.f, v0 = 0.f, v1 = 0.f;
#pragma unroll
      for (int j = 0; j < D / 4; ++j) {
        const float4 hv = ld4s(c.h + m * TS + 4 * j);
        k0 = fmaf(hv.x, wkr[j].x, k0), k1 = fmaf(hv.y, wkr[j].y, k1);
        k0 = fmaf(hv.z, wkr[j].z, k0), k1 = fmaf(hv.w, wkr[j].w, k1);
        v0 = fmaf(hv.x, wvr[j].x, v0), v1 = fmaf(hv.y, wvr[j].y, v1);
        v0 = fmaf(hv.z, wvr[j].z, v0), v1 = fmaf(hv.w, wvr[j].w, v1);
      }
      c.K[m * CS + cc] = bkc + (k0 + k1);
      c.V[m * CS + cc] = bvc + (v0 + v1);
    }
  } else {
    for (int e = tid; e < N * C; e += NT) {
      const int m = e / C, cc = e - m * C;
      c.K[m * CS + cc] = c.bk[cc] + dot4(c.h + m * TS, c.wk + cc * TS, D / 4);
      c.V[m * CS + cc] = c.bv[cc] + dot4(c.h + m * TS, c.wv + cc * TS, D / 4);
    }
  }
  __syncthreads();
  const float *pres = a.presence ? a.presence + (size_t)b * N : nullptr;
  if ((C & 15) == 0) {  // routing (set_transformer.py:40-43): 4 lanes per logit
    for (int t = tid; t < ((O * N * 4 + NT - 1) / NT) * NT; t += NT) {
      const int e = t >> 2, part = t & 3;
      const bool ok = e < O * N;
      const int o = ok ? e / N : 0, m = ok ? e - o * N : 0;
      float s = dot_quarter(c.q + o * CS, c.K + m * CS, C, part);
      if (ok && part == 0) {
        if (pres) s = s - (1.f - pres[m]) * 1e32f;
        c.S[o * NS + m] = s / a.sqrt_c;
      }
    }
  } else {
    for (int e = tid; e < O * N; e += NT) {
      const int o = e / N, m = e - o * N;
      float s = dot4(c.q + o * CS, c.K + m * CS, C / 4);
      if (pres) s = s - (1.f - pres[m]) * 1e32f;
      c.S[o * NS + m] = s / a.sqrt_c;
    }
  }
  __syncthreads();
  for (int e = tid; e < ((O * 16 + NT - 1) / NT) * NT; e += NT) {  // softmax, 16 lanes / row
    const int o = e >> 4, l = e & 15;
    float v[NMAX / 16];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NMAX / 16; ++k) {
      const int m = l + 16 * k;
      v[k] = (o < O && m < N) ? c.S[o * NS + m] : -INFINITY;
      mx = fmaxf(mx, v[k]);
    }
    mx = group_max<16>(mx);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NMAX / 16; ++k) {
      v[k] = v[k] == -INFINITY ? 0.f : expf(v[k] - mx);
      sum += v[k];
    }
    sum = group_sum<16>(sum);
#pragma unroll
    for (int k = 0; k < NMAX / 16; ++k) {
      const int m = l + 16 * k;
      if (o < O && m < N) c.S[o * NS + m] = v[k] / sum;
    }
  }
  __syncthreads();
}

// Workgroup (set b, query group s): the O queries are independent given K', V',
// so a set is shared out over `splits` workgroups (each recomputes the cheap K', V')
// -- one workgroup per set would leave half of the CUs idle at B = 128.
template <int D>
__global__ __launch_bounds__(NT) void sa_fwd_kernel(SaArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Og = a.O / a.splits, o0 = (blockIdx.x % a.splits) * Og;
  Carve<D> c;
  carve<D>(a.N, Og, a.C, false, smem, &c);
  const int N = a.N, O = Og, C = a.C, CS = c.CS, NS = N + 1, tid = threadIdx.x;
  stage_common<D>(a, c, o0, Og);
  for (int b = blockIdx.x / a.splits; b < a.B; b += gridDim.x / a.splits) {
    __syncthreads();
    forward_core<D>(a, c, b, Og);
    if (a.probs)
      for (int e = tid; e < O * N; e += NT)
        a.probs[((size_t)b * a.O + o0) * N + e] = c.S[(e / N) * NS + (e % N)];
    const int C4 = C / 4;
    for (int e = tid; e < O * C4; e += NT) {  // out = P V': four columns per thread
      const int o = e / C4, c4 = 4 * (e - o * C4);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int m = 0; m < N; ++m) fma4(acc, c.S[o * NS + m], ld4s(c.V + m * CS + c4));
      st4s(a.out + ((size_t)b * a.O + o0 + o) * C + c4, acc);
    }
  }
}

template <int D>
__global__ __launch_bounds__(NT) void sa_bwd_kernel(SaArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TS = D + 4;
  const int Og = a.O / a.splits, grp = blockIdx.x % a.splits, o0 = grp * Og;
  Carve<D> c;
  carve<D>(a.N, Og, a.C, true, smem, &c);
  const int N = a.N, O = Og, C = a.C, CS = c.CS, NS = N + 1, tid = threadIdx.x;
  stage_common<D>(a, c, o0, Og);
  const size_t P = (size_t)a.O * C + 2 * (size_t)C * D + 2 * C;
  float *part = a.partial + blockIdx.x * P;
  // this workgroup's query rows of the dq block; the other groups' rows stay zero
  for (int e = tid; e < a.O * C; e += NT)
    if (e < o0 * C || e >= (o0 + Og) * C) part[e] = 0.f;
  float *p_gq = part + (size_t)o0 * C, *p_gwk = part + (size_t)a.O * C,
        *p_gbk = p_gwk + (size_t)C * D, *p_gwv = p_gbk + C, *p_gbv = p_gwv + (size_t)C * D;
  // h-gradient of this query group: gh is (splits, B, N, D), summed by the caller
  float *gh = a.gh + (size_t)grp * a.B * N * D;
  bool first = true;
  for (int b = blockIdx.x / a.splits; b < a.B; b += gridDim.x / a.splits) {
    __syncthreads();
    forward_core<D>(a, c, b, Og);  // recompute K', V', P
    {
      const float *go = a.gout + ((size_t)b * a.O + o0) * C;
      if (((size_t)go & 15) == 0) {
        const int C4 = C / 4;
        for (int e = tid; e < O * C4; e += NT) {
          const int o = e / C4, c4 = 4 * (e - o * C4);
          st4s(c.GO + o * CS + c4, ld4s(go + (size_t)o * C + c4));
        }
      } else {
#pragma unroll 4
        for (int e = tid; e < O * C; e += NT) c.GO[(e / C) * CS + (e % C)] = go[e];
      }
    }
    __syncthreads();
    if ((C & 15) == 0) {  // dL/dP: 4 lanes per entry
      for (int t = tid; t < ((O * N * 4 + NT - 1) / NT) * NT; t += NT) {
        const int e = t >> 2, part = t & 3;
        const bool ok = e < O * N;
        const int o = ok ? e / N : 0, m = ok ? e - o * N : 0;
        const float s = dot_quarter(c.GO + o * CS, c.V + m * CS, C, part);
        if (ok && part == 0) c.GS[o * NS + m] = s;
      }
    } else {
      for (int e = tid; e < O * N; e += NT) {
        const int o = e / N, m = e - o * N;
        c.GS[o * NS + m] = dot4(c.GO + o * CS, c.V + m * CS, C / 4);
      }
    }
    __syncthreads();
    for (int e = tid; e < ((O * 16 + NT - 1) / NT) * NT; e += NT) {  // softmax backward
      const int o = e >> 4, l = e & 15;
      float p[NMAX / 16], gp[NMAX / 16];
      float dot = 0.f;
#pragma unroll
      for (int k = 0; k < NMAX / 16; ++k) {
        const int m = l + 16 * k;
        const bool in = o < O && m < N;
        p[k] = in ? c.S[o * NS + m] : 0.f;
        gp[k] = in ? c.GS[o * NS + m] : 0.f;
        dot = fmaf(p[k], gp[k], dot);
      }
      dot = group_sum<16>(dot);
#pragma unroll
      for (int k = 0; k < NMAX / 16; ++k) {
        const int m = l + 16 * k;
        if (o < O && m < N) c.GS[o * NS + m] = p[k] * (gp[k] - dot) / a.sqrt_c;
      }
    }
    // (V' is dead once dL/dP is known: its tile now receives dL/dV')
    const int C4 = C / 4;  // (N x C) products below: four columns per thread
    for (int e = tid; e < N * C4; e += NT) {
      const int m = e / C4, c4 = 4 * (e - m * C4);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int o = 0; o < O; ++o) fma4(acc, c.S[o * NS + m], ld4s(c.GO + o * CS + c4));
      st4s(c.V + m * CS + c4, acc);
    }
    __syncthreads();
    for (int e = tid; e < O * C4; e += NT) {  // dq (batch-invariant query): partial sum
      const int o = e / C4, c4 = 4 * (e - o * C4);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int m = 0; m < N; ++m) fma4(acc, c.GS[o * NS + m], ld4s(c.K + m * CS + c4));
      float *dst = p_gq + (size_t)o * C + c4;
      if (!first) {
        const float4 old = ld4s(dst);
        acc.x += old.x, acc.y += old.y, acc.z += old.z, acc.w += old.w;
      }
      st4s(dst, acc);
    }
    __syncthreads();
    for (int e = tid; e < N * C4; e += NT) {  // dL/dK' into the K' tile
      const int m = e / C4, c4 = 4 * (e - m * C4);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int o = 0; o < O; ++o) fma4(acc, c.GS[o * NS + m], ld4s(c.q + o * CS + c4));
      st4s(c.K + m * CS + c4, acc);
    }
    __syncthreads();
    // dh = dK' Wk2 + dV' Wvo: 8 lanes per entry, each an eighth of the C-long sums
    for (int t = tid; t < ((N * D * 8 + NT - 1) / NT) * NT; t += NT) {
      const int e = t >> 3, part = t & 7;
      const bool ok = e < N * D;
      const int m = ok ? e / D : 0, j = ok ? e - m * D : 0;
      float a0 = 0.f, a1 = 0.f;  // lane `part` takes cc = part, part + 8, ... (interleaved)
#pragma unroll 4
      for (int cc = part; cc < C; cc += 8) {
        a0 = fmaf(c.K[m * CS + cc], c.wk[cc * TS + j], a0);
        a1 = fmaf(c.V[m * CS + cc], c.wv[cc * TS + j], a1);
      }
      float sum = a0 + a1;
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      sum += __shfl_xor(sum, 4, 64);
      if (ok && part == 0) gh[(size_t)b * N * D + e] = sum;
    }
    constexpr int D4 = D / 4;
    for (int e = tid; e < C * D4; e += NT) {  // dWk2, dWvo (+ biases): four columns per thread
      const int cc = e / D4, j4 = 4 * (e - cc * D4);
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
      float b0 = 0.f, b1 = 0.f;
      for (int m = 0; m < N; ++m) {
        const float gk = c.K[m * CS + cc], gv = c.V[m * CS + cc];
        const float4 hv = ld4s(c.h + m * TS + j4);
        fma4(a0, gk, hv);
        fma4(a1, gv, hv);
        b0 += gk;
        b1 += gv;
      }
      float *dk = p_gwk + (size_t)cc * D + j4, *dv = p_gwv + (size_t)cc * D + j4;
      if (!first) {
        const float4 ok_ = ld4s(dk), ov = ld4s(dv);
        a0.x += ok_.x, a0.y += ok_.y, a0.z += ok_.z, a0.w += ok_.w;
        a1.x += ov.x, a1.y += ov.y, a1.z += ov.z, a1.w += ov.w;
      }
      st4s(dk, a0);
      st4s(dv, a1);
      if (j4 == 0) {
        p_gbk[cc] = first ? b0 : p_gbk[cc] + b0;
        p_gbv[cc] = first ? b1 : p_gbv[cc] + b1;
      }
    }
    first = false;
  }
}

template <int D>
size_t lds_bytes(int N, int O, int C, bool bwd) {
  return carve<D>(N, O, C, bwd, nullptr, nullptr) * sizeof(float);
}

int check(const SaArgs &a, int D) {
  if (a.B <= 0 || a.N <= 0 || a.O <= 0 || a.C <= 0) return SCAE_ERR_BAD_ARG;
  if (a.N > NMAX || a.O > NMAX || (a.C & 7) || (D != 8 && D != 16 && D != 32))
    return SCAE_ERR_UNSUPPORTED;
  return SCAE_OK;
}

// query groups per set: the smallest divisor of O that gives >= 256 workgroups (one per CU:
// the LDS tiles allow one resident workgroup per CU, so more groups would queue)
int sa_splits(int B, int O) {
  int s = 1;
  for (int d = 1; d <= O && d <= 4; ++d)
    if (O % d == 0) {
      s = d;
      if ((long)B * d >= 256) break;
    }
  return s;
}

template <int D>
int launch(SaArgs a, bool bwd, hipStream_t st) {
  a.splits = sa_splits(a.B, a.O);
  const size_t lds = lds_bytes<D>(a.N, a.O / a.splits, a.C, bwd);
  if (lds > 160 * 1024) return SCAE_ERR_UNSUPPORTED;
  const void *fn = bwd ? reinterpret_cast<const void *>(sa_bwd_kernel<D>)
                       : reinterpret_cast<const void *>(sa_fwd_kernel<D>);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int grid = scae_seed_attention_grid(a.B, a.O);
  if (bwd)
    scae::launch(sa_bwd_kernel<D>, dim3(grid), dim3(NT), lds, st, a);
  else
    scae::launch(sa_fwd_kernel<D>, dim3(grid), dim3(NT), lds, st, a);
  return scae_launch_status();
}
}  // namespace

extern "C" int scae_seed_attention_splits(int B, int O) {
  return B > 0 && O > 0 ? sa_splits(B, O) : 0;
}
extern "C" int scae_seed_attention_grid(int B, int O) {
  if (B <= 0 || O <= 0) return 0;
  return (B < 512 ? B : 512) * sa_splits(B, O);
}

extern "C" int scae_seed_attention_supported(int N, int O, int D, int C) {
  if (N <= 0 || O <= 0 || N > NMAX || O > NMAX || C <= 0 || (C & 7)) return 0;
  size_t need;
  switch (D) {
    case 8: need = lds_bytes<8>(N, O, C, true); break;
    case 16: need = lds_bytes<16>(N, O, C, true); break;
    case 32: need = lds_bytes<32>(N, O, C, true); break;
    default: return 0;
  }
  return need <= 160 * 1024 ? 1 : 0;
}

extern "C" int scae_seed_attention_fwd_f32(const float *h, const float *q, const float *wk,
                                           const float *bk, const float *wv, const float *bv,
                                           const float *presence, float *out, float *probs,
                                           int B, int N, int O, int D, int C, void *stream) {
  SCAE_REQUIRE(h && q && wk && bk && wv && bv && out);
  SaArgs a{h, q, wk, bk, wv, bv, presence, out, probs, nullptr, nullptr, nullptr,
           B, N, O, C, sqrtf((float)C)};
  int rc = check(a, D);
  if (rc) return rc;
  switch (D) {
    case 8: return launch<8>(a, false, (hipStream_t)stream);
    case 16: return launch<16>(a, false, (hipStream_t)stream);
    default: return launch<32>(a, false, (hipStream_t)stream);
  }
}

extern "C" int scae_seed_attention_bwd_f32(const float *h, const float *q, const float *wk,
                                           const float *bk, const float *wv, const float *bv,
                                           const float *presence, const float *gout, float *gh,
                                           float *partial, int B, int N, int O, int D, int C,
                                           void *stream) {
  SCAE_REQUIRE(h && q && wk && bk && wv && bv && gout && gh && partial);
  SaArgs a{h, q, wk, bk, wv, bv, presence, nullptr, nullptr, gout, gh, partial,
           B, N, O, C, sqrtf((float)C)};
  int rc = check(a, D);
  if (rc) return rc;
  switch (D) {
    case 8: return launch<8>(a, true, (hipStream_t)stream);
    case 16: return launch<16>(a, true, (hipStream_t)stream);
    default: return launch<32>(a, true, (hipStream_t)stream);
  }
}
