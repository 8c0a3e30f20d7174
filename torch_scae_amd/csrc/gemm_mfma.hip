// Batched fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32) with
// fused epilogues, for the O independent per-capsule MLPs of CapsuleLayer
// (object_decoder.py:86-107, :137-158) that the reference evaluates as a
// Python loop of 4*O tiny GEMMs:
//     C[g] = epilogue( A[g] (M x K) * B[g]^T (N x K) )
// Each operand may be k-contiguous ("K") or k-strided ("T"), which covers the
// forward (x W^T), the input gradient (g W) and the weight gradient (g^T x)
// without materialising a transpose.  Epilogue: + bias[n], ReLU, and a ReLU
// gate (multiply by mask[m][n] > 0) so that the backward pass hands the next
// layer the gradient w.r.t. its pre-activation directly.
//
// 64 x 64 output tile per workgroup (4 waves as 2 x 2, each 32 x 32 = 2 x 2
// MFMA tiles), K walked in 32-wide chunks through LDS.  LDS row strides are
// chosen per layout so that the row-per-lane fragment reads are bank-conflict
// free (k-contiguous tiles: stride 34 = 2 mod 32; k-strided tiles: stride 80 =
// 16 mod 32).  fp32 MFMA is exact fp32 (an fmaf chain), so results match a
// plain fp32 GEMM to re-association.
#include "common.h"

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 256;
constexpr int BM = 64, BN = 64, BK = 32;
constexpr int LDK = BK + 2;    // k-contiguous tile [rows][LDK]
constexpr int LDT = BM + 16;   // k-strided tile   [BK][LDT]

struct GemmArgs {
  const float *A, *B, *bias, *mask;
  float *C;
  long a_batch, b_batch, c_batch, bias_batch, mask_batch;
  int lda, ldb, ldc, bias_ld, ldmask;
  int M, N, K, relu;
};

// One (64 x 32) operand tile = 512 quads of 4 floats, 2 per thread, fetched
// into registers (so the next chunk's global loads overlap this chunk's MFMAs)
// and then written to LDS.  KC: X(row, k) = X[row*ld + k] (k-contiguous, quads
// along k) -> tile[row][LDK]; else X(row, k) = X[k*ld + row] (quads along row)
// -> tile[k][LDT].  `vec`: 16-byte global loads are legal (ld % 4 == 0 and an
// aligned base); otherwise 4-byte loads.
struct Quad2 {
  float4 v[2];
};

template <bool KC>
__device__ __forceinline__ Quad2 fetch(const float *X, int ld, int row0, int rows, int k0,
                                       int K, bool vec) {
  Quad2 q;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = threadIdx.x + NT * i;
    int row, k;  // first element of the quad
    if (KC) {
      row = row0 + id / (BK / 4);
      k = k0 + 4 * (id % (BK / 4));
    } else {
      k = k0 + id / (BM / 4);
      row = row0 + 4 * (id % (BM / 4));
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t base = KC ? (size_t)row * ld + k : (size_t)k * ld + row;
    const int lim = KC ? K - k : rows - row;       // valid elements along the quad
    const bool other = KC ? row < rows : k < K;    // the quad's other coordinate
    if (other && lim >= 4 && vec) {
      v = *reinterpret_cast<const float4 *>(X + base);
    } else if (other && lim > 0) {
      v.x = X[base];
      if (lim > 1) v.y = X[base + 1];
      if (lim > 2) v.z = X[base + 2];
      if (lim > 3) v.w = X[base + 3];
    }
    q.v[i] = v;
  }
  return q;
}

template <bool KC>
__device__ __forceinline__ void deposit(float *tile, const Quad2 &q) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = threadIdx.x + NT * i;
    if (KC) {
      float *p = tile + (id / (BK / 4)) * LDK + 4 * (id % (BK / 4));  // 8-byte aligned
      *reinterpret_cast<float2 *>(p) = make_float2(q.v[i].x, q.v[i].y);
      *reinterpret_cast<float2 *>(p + 2) = make_float2(q.v[i].z, q.v[i].w);
    } else {
      *reinterpret_cast<float4 *>(tile + (id / (BM / 4)) * LDT + 4 * (id % (BM / 4))) = q.v[i];
    }
  }
}

template <bool AK, bool BKC>
__global__ __launch_bounds__(NT) void gemm_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[BK * LDT > BM * LDK ? BK * LDT : BM * LDK];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDT > BN * LDK ? BK * LDT : BN * LDK];
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;  // wave position in the 2 x 2 grid
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, z = blockIdx.z;
  const float *A = g.A + z * g.a_batch, *B = g.B + z * g.b_batch;
  const bool avec = (g.lda & 3) == 0 && (g.a_batch & 3) == 0 && ((size_t)g.A & 15) == 0;
  const bool bvec = (g.ldb & 3) == 0 && (g.b_batch & 3) == 0 && ((size_t)g.B & 15) == 0;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  Quad2 ra = fetch<AK>(A, g.lda, m0, g.M, 0, g.K, avec);
  Quad2 rb = fetch<BKC>(B, g.ldb, n0, g.N, 0, g.K, bvec);
  for (int k0 = 0; k0 < g.K; k0 += BK) {
    __syncthreads();  // the previous chunk's fragment reads are done
    deposit<AK>(As, ra);
    deposit<BKC>(Bs, rb);
    __syncthreads();
    if (k0 + BK < g.K) {  // prefetch: in flight while the MFMAs below run
      ra = fetch<AK>(A, g.lda, m0, g.M, k0 + BK, g.K, avec);
      rb = fetch<BKC>(B, g.ldb, n0, g.N, k0 + BK, g.K, bvec);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 32 * wm + 16 * i + r;
        a[i] = AK ? As[row * LDK + kk + q] : As[(kk + q) * LDT + row];
        const int col = 32 * wn + 16 * i + r;
        b[i] = BKC ? Bs[col * LDK + kk + q] : Bs[(kk + q) * LDT + col];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // epilogue: C layout of the 16x16 tile: row = q*4 + reg, col = r
  float *C = g.C + z * g.c_batch;
  const float *bias = g.bias ? g.bias + z * g.bias_batch : nullptr;
  const float *mask = g.mask ? g.mask + z * g.mask_batch : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + 32 * wn + 16 * j + r;
      if (n >= g.N) continue;
      const float bv = bias ? bias[(size_t)n * g.bias_ld] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int m = m0 + 32 * wm + 16 * i + q * 4 + reg;
        if (m >= g.M) continue;
        float v = acc[i][j][reg] + bv;
        if (g.relu) v = fmaxf(v, 0.f);
        if (mask && !(mask[(size_t)m * g.ldmask + n] > 0.f)) v = 0.f;
        C[(size_t)m * g.ldc + n] = v;
      }
    }
}
}  // namespace

extern "C" int scae_gemm_f32(const float *A, const float *B, float *C, const float *bias,
                             const float *mask, int batch, int M, int N, int K, int a_kcontig,
                             int lda, int64_t a_batch, int b_kcontig, int ldb, int64_t b_batch,
                             int ldc, int64_t c_batch, int bias_ld, int64_t bias_batch,
                             int ldmask, int64_t mask_batch, int relu, void *stream) {
  SCAE_REQUIRE(A && B && C && batch > 0 && M > 0 && N > 0 && K > 0);
  GemmArgs g{A, B, bias, mask, C, (long)a_batch, (long)b_batch, (long)c_batch,
             (long)bias_batch, (long)mask_batch, lda, ldb, ldc, bias_ld, ldmask, M, N, K, relu};
  const dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch);
  hipStream_t st = (hipStream_t)stream;
  if (a_kcontig && b_kcontig)
    hipLaunchKernelGGL((gemm_kernel<true, true>), grid, dim3(NT), 0, st, g);
  else if (a_kcontig)
    hipLaunchKernelGGL((gemm_kernel<true, false>), grid, dim3(NT), 0, st, g);
  else if (b_kcontig)
    hipLaunchKernelGGL((gemm_kernel<false, true>), grid, dim3(NT), 0, st, g);
  else
    hipLaunchKernelGGL((gemm_kernel<false, false>), grid, dim3(NT), 0, st, g);
  return scae_launch_status();
}
