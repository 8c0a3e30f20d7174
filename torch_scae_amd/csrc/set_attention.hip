// K2 -- masked QKV attention of the set-transformer object encoder, gfx950.
// Replaces set_transformer.py:24-47 (bmm, in-place presence mask, softmax,
// bmm) and its autograd backward.
//
// One workgroup (4 waves) owns one (head*batch) problem: sets are small
// (N, M <= 64 capsules) so Q, K, V chunks, the probabilities P and dS all
// live in LDS, and both contractions of each pass (forward QK^T, PV;
// backward dO V^T, P^T dO, dS K, dS^T Q) run on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact f32, == an fmaf chain, which keeps the
// 1e32 presence mask arithmetic identical to the reference's fp32 sequence).
// The feature dimension (16 in the SABs, 256 in the output attention) is
// walked in 64-wide chunks so the LDS footprint is independent of d.
//
// MFMA 16x16x4 f32 lane maps (cdna_hip_programming.md section 3):
//   A[i][k]: lane l holds i = l&15, k = l>>4      B[k][j]: k = l>>4, j = l&15
//   C[i][j]: lane l, reg r holds i = (l>>4)*4 + r, j = l&15
#include "common.h"

namespace scae_attn_big {   // set_attention_big.hip: sets beyond SCAE_ATTN_MAX_SET
int fwd(const float *q, const float *k, const float *v, const float *presence, float *out,
        float *probs, int HB, int N, int M, int dk, int dv, float sqrt_dk, hipStream_t st);
int bwd(const float *q, const float *k, const float *v, const float *probs, const float *gout,
        float *gq, float *gk, float *gv, float *gpresence, int HB, int N, int M, int dk, int dv,
        float sqrt_dk, hipStream_t st);
}  // namespace scae_attn_big

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 256;
constexpr int DC = 64;  // feature chunk
constexpr int LD = 66;  // LDS row stride (floats): 66 = 2 mod 32 keeps the
                        // row-per-lane fragment reads bank-conflict free
constexpr int MAXT = SCAE_ATTN_MAX_SET / 16;  // 4 tiles per set dimension

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// rows x DC chunk of a (rows_valid x d) row-major matrix -> LDS, zero padded
__device__ __forceinline__ void stage(float *dst, const float *src, int rows_pad,
                                      int rows_valid, int d, int c0, int dc) {
  for (int i = threadIdx.x; i < rows_pad * DC; i += NT) {
    const int n = i / DC, c = i - n * DC;
    dst[n * LD + c] = (n < rows_valid && c < dc) ? src[(size_t)n * d + c0 + c] : 0.f;
  }
}

__device__ __forceinline__ float group16_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64));
  v = fmaxf(v, __shfl_xor(v, 2, 64));
  v = fmaxf(v, __shfl_xor(v, 4, 64));
  v = fmaxf(v, __shfl_xor(v, 8, 64));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

__global__ __launch_bounds__(NT) void attn_fwd_kernel(
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const float *__restrict__ presence, float *__restrict__ out, float *__restrict__ probs,
    int N, int M, int dk, int dv, float sqrt_dk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int hb = blockIdx.x, tid = threadIdx.x;
  const int wid = tid >> 6, lane = tid & 63, r = lane & 15, qd = lane >> 4;
  const int Npad = (N + 15) & ~15, Mpad = (M + 15) & ~15;
  const int nrt = Npad / 16, nct = Mpad / 16;
  float *Qs = smem;             // [Npad][LD]
  float *Ks = Qs + Npad * LD;   // [Mpad][LD]  (K chunks, then V chunks)
  float *Ps = Ks + Mpad * LD;   // [Npad][LD]
  const float *qb = q + (size_t)hb * N * dk;
  const float *kb = k + (size_t)hb * M * dk;
  const float *vb = v + (size_t)hb * M * dv;

  // ---- S = Q K^T -------------------------------------------------------
  f32x4 acc[MAXT];
#pragma unroll
  for (int j = 0; j < MAXT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < dk; c0 += DC) {
    const int dc = min(DC, dk - c0), dcp = (dc + 3) & ~3;
    __syncthreads();
    stage(Qs, qb, Npad, N, dk, c0, dc);
    stage(Ks, kb, Mpad, M, dk, c0, dc);
    __syncthreads();
    if (wid < nrt) {
      for (int kk = 0; kk < dcp; kk += 4) {
        const float a = Qs[(16 * wid + r) * LD + kk + qd];
#pragma unroll
        for (int j = 0; j < MAXT; ++j)
          if (j < nct) acc[j] = mfma4(a, Ks[(16 * j + r) * LD + kk + qd], acc[j]);
      }
    }
  }

  // ---- presence mask, scale, softmax (set_transformer.py:42-43) ---------
  if (wid < nrt) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = 16 * wid + qd * 4 + reg;
      float sv[MAXT];
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        const int col = 16 * j + r;
        float s = -INFINITY;
        if (j < nct && col < M) {
          s = acc[j][reg];
          if (presence) s = s - (1.f - presence[(size_t)hb * M + col]) * 1e32f;
          s = s / sqrt_dk;
        }
        sv[j] = s;
        m = fmaxf(m, s);
      }
      m = group16_max(m);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        sv[j] = (sv[j] == -INFINITY) ? 0.f : expf(sv[j] - m);
        sum += sv[j];
      }
      sum = group16_sum(sum);
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        const int col = 16 * j + r;
        if (j < nct) {
          const float p = sv[j] / sum;
          Ps[row * LD + col] = p;
          if (row < N && col < M) probs[((size_t)hb * N + row) * M + col] = p;
        }
      }
    }
  }

  // ---- O = P V ----------------------------------------------------------
  for (int c0 = 0; c0 < dv; c0 += DC) {
    const int dc = min(DC, dv - c0);
    const int nvt = (dc + 15) / 16;
    __syncthreads();
    stage(Ks, vb, Mpad, M, dv, c0, dc);
    __syncthreads();
    if (wid < nrt) {
      f32x4 o[DC / 16];
#pragma unroll
      for (int j = 0; j < DC / 16; ++j) o[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int kk = 0; kk < Mpad; kk += 4) {
        const float a = Ps[(16 * wid + r) * LD + kk + qd];
#pragma unroll
        for (int j = 0; j < DC / 16; ++j)
          if (j < nvt) o[j] = mfma4(a, Ks[(kk + qd) * LD + 16 * j + r], o[j]);
      }
#pragma unroll
      for (int j = 0; j < DC / 16; ++j) {
        const int col = 16 * j + r;
        if (j < nvt && col < dc) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * wid + qd * 4 + reg;
            if (row < N) out[((size_t)hb * N + row) * dv + c0 + col] = o[j][reg];
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// bf16 forward (BASELINE.json configs[2]: "bf16 ... MFMA attention path"): the same
// decomposition with both contractions on v_mfma_f32_16x16x16_bf16 (bf16 operands,
// fp32 accumulate; 4x the K per instruction and half the LDS bytes of the fp32
// kernel).  q, k, v arrive as bf16 -- what torch.autocast hands the attention after
// the bf16 q/k/v projections -- the mask / scale / softmax stay fp32 (the 1e32 mask
// needs the exponent range and the reference order of operations), the
// probabilities are rounded to bf16 only as the A operand of P V and are kept in
// fp32 for the backward pass, which runs the fp32 kernel above on upcast operands.
//
// MFMA 16x16x16 bf16 lane maps: A[i][k]: lane l holds i = l&15, k = 4*(l>>4) + 0..3
// (4 consecutive k = 8 bytes); B[k][j]: k = 4*(l>>4) + 0..3, j = l&15; C as above.
// ---------------------------------------------------------------------------
typedef short bf16x4 __attribute__((ext_vector_type(4)));
constexpr int LDH = DC + 8;  // LDS row stride in bf16 units: 144 B = 4 mod 32 banks... rows
                             // of a fragment read (8 B per lane) spread over all banks

__device__ __forceinline__ unsigned short f2bf(float f) {  // round to nearest even
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ bf16x4 ld_bf4(const unsigned short *p) {
  return *reinterpret_cast<const bf16x4 *>(p);
}

// rows x DC chunk of a (rows_valid x d) row-major bf16 matrix -> LDS, zero padded;
// TR: stored transposed, dst[c][n] (the B operand of P V wants 4 consecutive rows)
template <bool TR>
__device__ __forceinline__ void stage_bf(unsigned short *dst, const unsigned short *src,
                                         int rows_pad, int rows_valid, int d, int c0, int dc) {
  for (int i = threadIdx.x; i < rows_pad * DC; i += NT) {
    const int n = i / DC, c = i - n * DC;
    const unsigned short v = (n < rows_valid && c < dc) ? src[(size_t)n * d + c0 + c] : 0;
    if (TR)
      dst[c * LDH + n] = v;
    else
      dst[n * LDH + c] = v;
  }
}

__global__ __launch_bounds__(NT) void attn_fwd_bf16_kernel(
    const unsigned short *__restrict__ q, const unsigned short *__restrict__ k,
    const unsigned short *__restrict__ v, const float *__restrict__ presence,
    unsigned short *__restrict__ out, float *__restrict__ probs, int N, int M, int dk, int dv,
    float sqrt_dk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int hb = blockIdx.x, tid = threadIdx.x;
  const int wid = tid >> 6, lane = tid & 63, r = lane & 15, qd = lane >> 4;
  const int Npad = (N + 15) & ~15, Mpad = (M + 15) & ~15;
  const int nrt = Npad / 16, nct = Mpad / 16;
  unsigned short *Qs = reinterpret_cast<unsigned short *>(smem);  // [Npad][LDH]
  unsigned short *Ks = Qs + SCAE_ATTN_MAX_SET * LDH;  // [Mpad][LDH] K chunks, then V^T chunks [DC][LDH]
  unsigned short *Ps = Ks + SCAE_ATTN_MAX_SET * LDH;  // [Npad][LDH] probabilities (bf16)
  const unsigned short *qb = q + (size_t)hb * N * dk;
  const unsigned short *kb = k + (size_t)hb * M * dk;
  const unsigned short *vb = v + (size_t)hb * M * dv;

  // ---- S = Q K^T -------------------------------------------------------
  f32x4 acc[MAXT];
#pragma unroll
  for (int j = 0; j < MAXT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < dk; c0 += DC) {
    const int dc = min(DC, dk - c0), dcp = (dc + 15) & ~15;
    __syncthreads();
    stage_bf<false>(Qs, qb, Npad, N, dk, c0, dc);
    stage_bf<false>(Ks, kb, Mpad, M, dk, c0, dc);
    __syncthreads();
    if (wid < nrt) {
      for (int kk = 0; kk < dcp; kk += 16) {
        const bf16x4 a = ld_bf4(Qs + (16 * wid + r) * LDH + kk + 4 * qd);
#pragma unroll
        for (int j = 0; j < MAXT; ++j)
          if (j < nct)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(
                a, ld_bf4(Ks + (16 * j + r) * LDH + kk + 4 * qd), acc[j], 0, 0, 0);
      }
    }
  }

  // ---- presence mask, scale, softmax (set_transformer.py:42-43), fp32 ---
  if (wid < nrt) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = 16 * wid + qd * 4 + reg;
      float sv[MAXT];
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        const int col = 16 * j + r;
        float s = -INFINITY;
        if (j < nct && col < M) {
          s = acc[j][reg];
          if (presence) s = s - (1.f - presence[(size_t)hb * M + col]) * 1e32f;
          s = s / sqrt_dk;
        }
        sv[j] = s;
        m = fmaxf(m, s);
      }
      m = group16_max(m);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        sv[j] = (sv[j] == -INFINITY) ? 0.f : expf(sv[j] - m);
        sum += sv[j];
      }
      sum = group16_sum(sum);
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        const int col = 16 * j + r;
        if (j < nct) {
          const float p = sv[j] / sum;
          Ps[row * LDH + col] = f2bf(p);
          if (row < N && col < M) probs[((size_t)hb * N + row) * M + col] = p;
        }
      }
    }
  }

  // ---- O = P V ----------------------------------------------------------
  for (int c0 = 0; c0 < dv; c0 += DC) {
    const int dc = min(DC, dv - c0);
    const int nvt = (dc + 15) / 16;
    __syncthreads();
    stage_bf<true>(Ks, vb, Mpad, M, dv, c0, dc);  // Ks[d][m]
    __syncthreads();
    if (wid < nrt) {
      f32x4 o[DC / 16];
#pragma unroll
      for (int j = 0; j < DC / 16; ++j) o[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int kk = 0; kk < Mpad; kk += 16) {
        const bf16x4 a = ld_bf4(Ps + (16 * wid + r) * LDH + kk + 4 * qd);
#pragma unroll
        for (int j = 0; j < DC / 16; ++j)
          if (j < nvt)
            o[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(
                a, ld_bf4(Ks + (16 * j + r) * LDH + kk + 4 * qd), o[j], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < DC / 16; ++j) {
        const int col = 16 * j + r;
        if (j < nvt && col < dc) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * wid + qd * 4 + reg;
            if (row < N) out[((size_t)hb * N + row) * dv + c0 + col] = f2bf(o[j][reg]);
          }
        }
      }
    }
  }
}

// C[rows x cols tile (ti, tj)] = sum_kk A(ti, kk) * B(kk, tj) with both
// operands in LDS.  TA: A is read transposed (A[i][k] = As[k][i]).
template <bool TA>
__device__ __forceinline__ f32x4 lds_tile(const float *As, const float *Bs, int ti, int tj,
                                          int kdim, int r, int qd) {
  f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int kk = 0; kk < kdim; kk += 4) {
    const float a = TA ? As[(kk + qd) * LD + 16 * ti + r] : As[(16 * ti + r) * LD + kk + qd];
    const float b = Bs[(kk + qd) * LD + 16 * tj + r];
    c = mfma4(a, b, c);
  }
  return c;
}

__device__ __forceinline__ void store_tile(float *dst, f32x4 c, int ti, int tj, int rows,
                                           int d, int c0, int dc, int r, int qd) {
  const int col = 16 * tj + r;
  if (col >= dc) return;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = 16 * ti + qd * 4 + reg;
    if (row < rows) dst[(size_t)row * d + c0 + col] = c[reg];
  }
}

__global__ __launch_bounds__(NT) void attn_bwd_kernel(
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const float *__restrict__ probs, const float *__restrict__ gout, float *__restrict__ gq,
    float *__restrict__ gk, float *__restrict__ gv, float *__restrict__ gpresence, int N,
    int M, int dk, int dv, float sqrt_dk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int hb = blockIdx.x, tid = threadIdx.x;
  const int wid = tid >> 6, lane = tid & 63, r = lane & 15, qd = lane >> 4;
  const int Npad = (N + 15) & ~15, Mpad = (M + 15) & ~15;
  const int nrt = Npad / 16, nct = Mpad / 16;
  float *As = smem;             // [Npad][LD]  dO chunk / Q chunk
  float *Bs = As + Npad * LD;   // [Mpad][LD]  V chunk / K chunk
  float *Ps = Bs + Mpad * LD;   // [Npad][LD]  P
  float *Ds = Ps + Npad * LD;   // [Npad][LD]  dS / sqrt_dk
  const float *qb = q + (size_t)hb * N * dk;
  const float *kb = k + (size_t)hb * M * dk;
  const float *vb = v + (size_t)hb * M * dv;
  const float *gob = gout + (size_t)hb * N * dv;

  // ---- dP = dO V^T ------------------------------------------------------
  f32x4 acc[MAXT];
#pragma unroll
  for (int j = 0; j < MAXT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < dv; c0 += DC) {
    const int dc = min(DC, dv - c0), dcp = (dc + 3) & ~3;
    __syncthreads();
    stage(As, gob, Npad, N, dv, c0, dc);
    stage(Bs, vb, Mpad, M, dv, c0, dc);
    __syncthreads();
    if (wid < nrt) {
      for (int kk = 0; kk < dcp; kk += 4) {
        const float a = As[(16 * wid + r) * LD + kk + qd];
#pragma unroll
        for (int j = 0; j < MAXT; ++j)
          if (j < nct) acc[j] = mfma4(a, Bs[(16 * j + r) * LD + kk + qd], acc[j]);
      }
    }
  }
  // ---- softmax backward; dS_raw = P (dP - sum(P dP)) / sqrt_dk ------------
  if (wid < nrt) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = 16 * wid + qd * 4 + reg;
      float pv[MAXT];
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        const int col = 16 * j + r;
        pv[j] = (j < nct && row < N && col < M) ? probs[((size_t)hb * N + row) * M + col] : 0.f;
        t += pv[j] * acc[j][reg];
      }
      t = group16_sum(t);
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        const int col = 16 * j + r;
        if (j < nct) {
          Ps[row * LD + col] = pv[j];
          Ds[row * LD + col] = pv[j] * (acc[j][reg] - t) / sqrt_dk;
        }
      }
    }
  }
  __syncthreads();
  if (gpresence && tid < M) {  // routing -= (1-p)*1e32  =>  d/dp = +1e32
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += Ds[n * LD + tid];
    gpresence[(size_t)hb * M + tid] = s * 1e32f;
  }

  // ---- dV = P^T dO ------------------------------------------------------
  for (int c0 = 0; c0 < dv; c0 += DC) {
    const int dc = min(DC, dv - c0);
    const int ntj = (dc + 15) / 16;
    __syncthreads();
    stage(As, gob, Npad, N, dv, c0, dc);
    __syncthreads();
    for (int t = wid; t < nct * ntj; t += NT / 64) {
      const int ti = t / ntj, tj = t - ti * ntj;
      const f32x4 c = lds_tile<true>(Ps, As, ti, tj, Npad, r, qd);
      store_tile(gv + (size_t)hb * M * dv, c, ti, tj, M, dv, c0, dc, r, qd);
    }
  }
  // ---- dQ = dS K,  dK = dS^T Q -------------------------------------------
  for (int c0 = 0; c0 < dk; c0 += DC) {
    const int dc = min(DC, dk - c0);
    const int ntj = (dc + 15) / 16;
    __syncthreads();
    stage(As, qb, Npad, N, dk, c0, dc);
    stage(Bs, kb, Mpad, M, dk, c0, dc);
    __syncthreads();
    for (int t = wid; t < nrt * ntj; t += NT / 64) {
      const int ti = t / ntj, tj = t - ti * ntj;
      const f32x4 c = lds_tile<false>(Ds, Bs, ti, tj, Mpad, r, qd);
      store_tile(gq + (size_t)hb * N * dk, c, ti, tj, N, dk, c0, dc, r, qd);
    }
    for (int t = wid; t < nct * ntj; t += NT / 64) {
      const int ti = t / ntj, tj = t - ti * ntj;
      const f32x4 c = lds_tile<true>(Ds, As, ti, tj, Npad, r, qd);
      store_tile(gk + (size_t)hb * M * dk, c, ti, tj, M, dk, c0, dc, r, qd);
    }
  }
}

int check_attn(int HB, int N, int M, int dk, int dv) {
  if (HB <= 0 || N <= 0 || M <= 0 || dk <= 0 || dv <= 0) return SCAE_ERR_BAD_ARG;
  return SCAE_OK;
}
bool big_set(int N, int M) { return N > SCAE_ATTN_MAX_SET || M > SCAE_ATTN_MAX_SET; }

int raise_lds(const void *kernel, size_t bytes) {
  if (bytes > 48 * 1024) {
    hipError_t e =
        hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
  }
  return SCAE_OK;
}

}  // namespace

extern "C" int scae_qkv_attention_fwd_f32(const float *q, const float *k, const float *v,
                                          const float *presence, float *out, float *probs,
                                          int HB, int N, int M, int dk, int dv,
                                          float sqrt_dk, void *stream) {
  int rc = check_attn(HB, N, M, dk, dv);
  if (rc) return rc;
  SCAE_REQUIRE(q && k && v && out && probs && sqrt_dk > 0.f);
  if (big_set(N, M))   // beyond the matrix-core kernels' tiles: the general form
    return scae_attn_big::fwd(q, k, v, presence, out, probs, HB, N, M, dk, dv, sqrt_dk,
                              (hipStream_t)stream);
  const int Npad = (N + 15) & ~15, Mpad = (M + 15) & ~15;
  const size_t lds = sizeof(float) * (size_t)(2 * Npad + Mpad) * LD;
  rc = raise_lds(reinterpret_cast<const void *>(attn_fwd_kernel), lds);
  if (rc) return rc;
  scae::launch(attn_fwd_kernel, dim3(HB), dim3(NT), lds, (hipStream_t)stream, q, k, v,
                     presence, out, probs, N, M, dk, dv, sqrt_dk);
  return scae_launch_status();
}

extern "C" int scae_qkv_attention_fwd_bf16(const uint16_t *q, const uint16_t *k,
                                           const uint16_t *v, const float *presence,
                                           uint16_t *out, float *probs, int HB, int N, int M,
                                           int dk, int dv, float sqrt_dk, void *stream) {
  int rc = check_attn(HB, N, M, dk, dv);
  if (rc) return rc;
  if (big_set(N, M)) return SCAE_ERR_UNSUPPORTED;   // (the bf16 form has the tile limits)
  SCAE_REQUIRE(q && k && v && out && probs && sqrt_dk > 0.f);
  const size_t lds = sizeof(unsigned short) * 3 * (size_t)SCAE_ATTN_MAX_SET * LDH;
  scae::launch(attn_fwd_bf16_kernel, dim3(HB), dim3(NT), lds, (hipStream_t)stream, q, k,
                     v, presence, out, probs, N, M, dk, dv, sqrt_dk);
  return scae_launch_status();
}

extern "C" int scae_qkv_attention_bwd_f32(const float *q, const float *k, const float *v,
                                          const float *probs, const float *gout, float *gq,
                                          float *gk, float *gv, float *gpresence, int HB,
                                          int N, int M, int dk, int dv, float sqrt_dk,
                                          void *stream) {
  int rc = check_attn(HB, N, M, dk, dv);
  if (rc) return rc;
  SCAE_REQUIRE(q && k && v && probs && gout && gq && gk && gv && sqrt_dk > 0.f);
  if (big_set(N, M))
    return scae_attn_big::bwd(q, k, v, probs, gout, gq, gk, gv, gpresence, HB, N, M, dk, dv,
                              sqrt_dk, (hipStream_t)stream);
  const int Npad = (N + 15) & ~15, Mpad = (M + 15) & ~15;
  const size_t lds = sizeof(float) * (size_t)(3 * Npad + Mpad) * LD;
  rc = raise_lds(reinterpret_cast<const void *>(attn_bwd_kernel), lds);
  if (rc) return rc;
  scae::launch(attn_bwd_kernel, dim3(HB), dim3(NT), lds, (hipStream_t)stream, q, k, v,
                     probs, gout, gq, gk, gv, gpresence, N, M, dk, dv, sqrt_dk);
  return scae_launch_status();
}
