// Batched fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32) with
// fused epilogues, for the O independent per-capsule MLPs of CapsuleLayer
// (object_decoder.py:86-107, :137-158) that the reference evaluates as a
// Python loop of 4*O tiny GEMMs, and for the 1x1 attention convolution of the
// part-capsule encoder (part_encoder.py:71-73):
//     C[g] = epilogue( A[g] (M x K) * B[g]^T (N x K) )
// Each operand may be k-contiguous ("K") or k-strided ("T"), which covers the
// forward (x W^T), the input gradient (g W) and the weight gradient (g^T x)
// without materialising a transpose.  Epilogue: + bias[n], ReLU, and a ReLU
// gate (multiply by mask[m][n] > 0) so that the backward pass hands the next
// layer the gradient w.r.t. its pre-activation directly.  With a k-strided A
// the kernel can also emit asum[m] = sum_k A(m, k) -- the bias gradient of a
// weight-gradient GEMM -- from the operand registers it stages anyway (written
// with stride asum_ld, so it can land in a column of the weight gradient).
//
// Tile loop: mfma_tile.h (64 x 64 tiles, or 32 x 32 split-K tiles when the
// former would give the 256 CUs too few workgroups).  fp32 MFMA is exact fp32
// (an fmaf chain), so results match a plain fp32 GEMM to re-association.
#include "mfma_tile.h"

namespace {
using namespace scae_tile;

struct GemmArgs {
  const float *A, *B, *bias, *mask;
  float *C, *asum, *craw;
  long a_batch, b_batch, c_batch, bias_batch, mask_batch, asum_batch;
  int lda, ldb, ldc, bias_ld, ldmask;
  int M, N, K, relu, asum_ld;
};

// One (T x 32) operand tile = T*8 quads of 4 floats fetched into registers (so
// the next chunk's global loads overlap this chunk's MFMAs).  KC: X(row, k) =
// X[row*ld + k] (quads along k); else X(row, k) = X[k*ld + row] (quads along
// row).  `vec`: 16-byte global loads are legal (ld % 4 == 0 and an aligned
// base); otherwise 4-byte loads.
template <int SK, bool KC>
__device__ __forceinline__ void fetch(Quads<Tile<SK>::NQ> &q, const float *X, int ld, int row0,
                                      int rows, int k0, int K, bool vec, bool inside) {
  constexpr int T = Tile<SK>::T;
  if (inside) {   // workgroup-uniform: the whole tile chunk is in range and 16-byte loadable
#pragma unroll
    for (int i = 0; i < Tile<SK>::NQ; ++i) {
      const int id = threadIdx.x + NT * i;
      if (KC) {
        q.v[i] = ld4(X + (size_t)(row0 + id / QPR) * ld + k0 + 4 * (id % QPR));
      } else {
        int kq, rq;
        kstr_pos<SK, T>(threadIdx.x, i, kq, rq);
        q.v[i] = ld4(X + (size_t)(k0 + kq) * ld + row0 + rq);
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < Tile<SK>::NQ; ++i) {
    const int id = threadIdx.x + NT * i;
    int row, k;  // first element of the quad
    if (KC) {
      row = row0 + id / QPR;
      k = k0 + 4 * (id % QPR);
    } else {
      int kq, rq;
      kstr_pos<SK, T>(threadIdx.x, i, kq, rq);
      k = k0 + kq;
      row = row0 + rq;
    }
    float4 v = zero4();
    const size_t base = KC ? (size_t)row * ld + k : (size_t)k * ld + row;
    const int lim = KC ? K - k : rows - row;     // valid elements along the quad
    const bool other = KC ? row < rows : k < K;  // the quad's other coordinate
    if (other && lim >= 4 && vec) {
      v = ld4(X + base);
    } else if (other && lim > 0) {
      v.x = X[base];
      if (lim > 1) v.y = X[base + 1];
      if (lim > 2) v.z = X[base + 2];
      if (lim > 3) v.w = X[base + 3];
    }
    q.v[i] = v;
  }
}

// one output tile of problem `g`, batch element z (workgroup-uniform arguments)
template <int SK, bool AK, bool BKC>
__device__ __forceinline__ void gemm_tile(const GemmArgs &g, float *smem, int z, int bx, int by) {
  using TL = Tile<SK>;
  constexpr int T = TL::T, NQ = TL::NQ;
  float *As = smem, *Bs = smem + TL::OPER;
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int m0 = by * T, n0 = bx * T;
  if (m0 >= g.M || n0 >= g.N) return;
  const float *A = g.A + z * g.a_batch, *B = g.B + z * g.b_batch;
  const bool avec = (g.lda & 3) == 0 && (g.a_batch & 3) == 0 && ((size_t)g.A & 15) == 0;
  const bool bvec = (g.ldb & 3) == 0 && (g.b_batch & 3) == 0 && ((size_t)g.B & 15) == 0;
  // tiles whose rows and k chunks all lie inside the operands load without per-quad guards
  const bool ain = avec && m0 + T <= g.M && g.K % BK == 0;
  const bool bin = bvec && n0 + T <= g.N && g.K % BK == 0;
  const bool want_asum = !AK && g.asum && bx == 0;  // workgroup-uniform
  float4 asum = zero4();
  typename TL::Acc acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = typename TL::Acc{};

  tile_mainloop<STAGES, SK, AK, BKC>(
      (g.K + BK - 1) / BK, As, Bs, acc, wid, r, q,
      [&](int c, Quads<NQ> &ra, Quads<NQ> &rb) {
        fetch<SK, AK>(ra, A, g.lda, m0, g.M, c * BK, g.K, avec, ain);
        fetch<SK, BKC>(rb, B, g.ldb, n0, g.N, c * BK, g.K, bvec, bin);
      },
      [&](const Quads<NQ> &ra) {
        if (want_asum) {
#pragma unroll
          for (int i = 0; i < NQ; ++i)
            asum.x += ra.v[i].x, asum.y += ra.v[i].y, asum.z += ra.v[i].z, asum.w += ra.v[i].w;
        }
      });
  float *C = g.C + z * g.c_batch;
  float *craw = g.craw ? g.craw + z * g.c_batch : nullptr;  // pre-gate copy (layout of C)
  const float *bias = g.bias ? g.bias + z * g.bias_batch : nullptr;
  const float *mask = g.mask ? g.mask + z * g.mask_batch : nullptr;
  const bool cvec = (g.ldc & 3) == 0 && (g.c_batch & 3) == 0 && ((size_t)g.C & 15) == 0 &&
                    (!g.craw || ((size_t)g.craw & 15) == 0);  // craw shares C's ldc / batch stride
  // (workgroup-uniform) the tile lies inside C and its rows take 16-byte accesses: straight-line
  // vector epilogue; otherwise the per-element form with its guards
  const bool cin = cvec && m0 + T <= g.M && n0 + T <= g.N &&
                   (!mask || ((g.ldmask & 3) == 0 && (g.mask_batch & 3) == 0 &&
                              ((size_t)g.mask & 15) == 0));
  if (cin) {
    tile_epilogue<SK>(smem, acc, wid, r, q, [&](int row, int col, float4 v) {
      const int m = m0 + row, n = n0 + col;
      if (bias) {
        const float *bp = bias + (size_t)n * g.bias_ld;
        v.x += bp[0], v.y += bp[g.bias_ld], v.z += bp[2 * (size_t)g.bias_ld],
            v.w += bp[3 * (size_t)g.bias_ld];
      }
      if (g.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      if (craw) *reinterpret_cast<float4 *>(craw + (size_t)m * g.ldc + n) = v;
      if (mask) {
        const float4 mk = ld4(mask + (size_t)m * g.ldmask + n);
        v = make_float4(mk.x > 0.f ? v.x : 0.f, mk.y > 0.f ? v.y : 0.f, mk.z > 0.f ? v.z : 0.f,
                        mk.w > 0.f ? v.w : 0.f);
      }
      *reinterpret_cast<float4 *>(C + (size_t)m * g.ldc + n) = v;
    });
  } else {
  tile_epilogue<SK>(smem, acc, wid, r, q, [&](int row, int col, float4 v4) {
    const int m = m0 + row, n = n0 + col;
    if (m >= g.M || n >= g.N) return;
    float v[4] = {v4.x, v4.y, v4.z, v4.w};
    const int cnt = min(4, g.N - n);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (e >= cnt) break;
      if (bias) v[e] += bias[(size_t)(n + e) * g.bias_ld];
      if (g.relu) v[e] = fmaxf(v[e], 0.f);
      if (craw) craw[(size_t)m * g.ldc + n + e] = v[e];
      if (mask && !(mask[(size_t)m * g.ldmask + n + e] > 0.f)) v[e] = 0.f;
    }
    float *dst = C + (size_t)m * g.ldc + n;
    if (cnt == 4 && cvec) {
      *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < cnt) dst[e] = v[e];
    }
  });
  }
  if (want_asum) {  // threads tid % (T/4) stage the same 4 rows of A
    __syncthreads();
    reinterpret_cast<float4 *>(smem)[tid] = asum;
    __syncthreads();
    if (tid < T && m0 + tid < g.M) {
      float sum = 0.f;
      for (int j = 0; j < NT / (T / 4); ++j) sum += smem[4 * (tid / 4 + (T / 4) * j) + (tid & 3)];
      g.asum[z * g.asum_batch + (size_t)(m0 + tid) * g.asum_ld] = sum;
    }
  }
}

#ifndef SCAE_DEVICE_ONLY   // (seed_bwd_gemm.hip includes this file for its device code)
template <int SK, bool AK, bool BKC>
__global__ __launch_bounds__(NT) void gemm_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[Tile<SK>::SMEM];
  gemm_tile<SK, AK, BKC>(g, smem, blockIdx.z, blockIdx.x, blockIdx.y);
}

// Two independent problems in one launch (blockIdx.z < nz0: the first): the
// weight-gradient and data-gradient GEMMs of a layer both only wait for the
// same incoming gradient, and each is too small to fill the device.
struct GemmPair {
  GemmArgs g[2];
  int nz0, layout[2];  // layout = 2*a_kcontig + b_kcontig
};
template <int SK>
__global__ __launch_bounds__(NT) void gemm_pair_kernel(GemmPair p) {
  __shared__ __attribute__((aligned(16))) float smem[Tile<SK>::SMEM];
  const int which = (int)blockIdx.z >= p.nz0, z = blockIdx.z - (which ? p.nz0 : 0);
  const GemmArgs &g = p.g[which];
  switch (p.layout[which]) {  // workgroup-uniform
    case 3: gemm_tile<SK, true, true>(g, smem, z, blockIdx.x, blockIdx.y); break;
    case 2: gemm_tile<SK, true, false>(g, smem, z, blockIdx.x, blockIdx.y); break;
    case 1: gemm_tile<SK, false, true>(g, smem, z, blockIdx.x, blockIdx.y); break;
    default: gemm_tile<SK, false, false>(g, smem, z, blockIdx.x, blockIdx.y); break;
  }
}
#endif

// Up to four independent problems in one launch: a 1-D grid of exactly the tiles that exist
// (problem p owns the block range [first[p], first[p + 1]): its (batch, tile row, tile
// column) in that order) -- a common 3-D grid sized for the largest problem dispatches
// mostly empty workgroups when the problems differ in size (3.7 k of 5.4 k for the four
// weight-gradient GEMMs of the capsule MLPs).
struct GemmMulti {
  GemmArgs g[4];
  int first[5], tx[4], ty[4], layout[4], n;
};
// tile `blk` of the problem list (seed_bwd_gemm.hip runs these as the tail of another launch)
template <int SK>
__device__ __forceinline__ void gemm_multi_body(const GemmMulti &p, float *smem, int blk) {
  int which = 0;
  while (which + 1 < p.n && blk >= p.first[which + 1]) ++which;
  const int id = blk - p.first[which], per = p.tx[which] * p.ty[which];
  const int z = id / per, rem = id - z * per, by = rem / p.tx[which], bx = rem - by * p.tx[which];
  const GemmArgs &g = p.g[which];
  switch (p.layout[which]) {  // workgroup-uniform
    case 3: gemm_tile<SK, true, true>(g, smem, z, bx, by); break;
    case 2: gemm_tile<SK, true, false>(g, smem, z, bx, by); break;
    case 1: gemm_tile<SK, false, true>(g, smem, z, bx, by); break;
    default: gemm_tile<SK, false, false>(g, smem, z, bx, by); break;
  }
}
#ifndef SCAE_DEVICE_ONLY
template <int SK>
__global__ __launch_bounds__(NT) void gemm_multi_kernel(GemmMulti p) {
  __shared__ __attribute__((aligned(16))) float smem[Tile<SK>::SMEM];
  gemm_multi_body<SK>(p, smem, blockIdx.x);
}
#endif

// fewer 64 x 64 tiles than this: 32 x 32 split-K tiles (4x the workgroups)
#ifndef SCAE_GEMM_SK_BELOW
#define SCAE_GEMM_SK_BELOW 1024
#endif
constexpr long kSplitKBelow = SCAE_GEMM_SK_BELOW;

int fill_args(GemmArgs &g, const scae_gemm_desc *d) {
  if (!d || !d->A || !d->B || !d->C || d->batch <= 0 || d->M <= 0 || d->N <= 0 || d->K <= 0)
    return SCAE_ERR_BAD_ARG;
  if (d->asum && d->a_kcontig) return SCAE_ERR_UNSUPPORTED;
  g = GemmArgs{d->A, d->B, d->bias, d->mask, d->C, d->asum, d->c_nomask, (long)d->a_batch, (long)d->b_batch,
               (long)d->c_batch, (long)d->bias_batch, (long)d->mask_batch, (long)d->asum_batch,
               d->lda, d->ldb, d->ldc, d->bias_ld, d->ldmask, d->M, d->N, d->K, d->relu,
               d->asum_ld > 0 ? d->asum_ld : 1};
  return SCAE_OK;
}

// the 1-D tile grid of up to four problems; T: 32 (split-K tiles) or 64 -- or, asked for
// bf16 operands, 128 when every problem is large enough for those tiles (bf16_shape)
#ifndef SCAE_BF16_MIN_SIDE
#define SCAE_BF16_MIN_SIDE 32
#endif
static bool bf16_shape(int M, int N) {
  return M >= SCAE_BF16_MIN_SIDE && N >= SCAE_BF16_MIN_SIDE;
}
int plan_multi(GemmMulti &p, int &T, const scae_gemm_desc *descs, int n, bool bf16 = false) {
  if (!descs || n < 1 || n > 4) return SCAE_ERR_BAD_ARG;
  p = GemmMulti{};
  p.n = n;
  long tiles64 = 0;
  for (int i = 0; i < n; ++i) {
    int rc = fill_args(p.g[i], descs + i);
    if (rc) return rc;
    p.layout[i] = 2 * (descs[i].a_kcontig != 0) + (descs[i].b_kcontig != 0);
    tiles64 += (long)((descs[i].N + 63) / 64) * ((descs[i].M + 63) / 64) * descs[i].batch;
  }
  T = tiles64 < kSplitKBelow ? 32 : 64;
  if (bf16) {
    bool all = true;
    for (int i = 0; i < n; ++i) all = all && bf16_shape(descs[i].M, descs[i].N);
    if (all) T = 128;
  }
  for (int i = 0; i < n; ++i) {
    p.tx[i] = (descs[i].N + T - 1) / T, p.ty[i] = (descs[i].M + T - 1) / T;
    p.first[i + 1] = p.first[i] + p.tx[i] * p.ty[i] * descs[i].batch;
  }
  return SCAE_OK;
}

#ifndef SCAE_DEVICE_ONLY
template <int SK>
void launch(const GemmArgs &g, int batch, bool ak, bool bk, hipStream_t st) {
  constexpr int T = Tile<SK>::T;
  const dim3 grid((g.N + T - 1) / T, (g.M + T - 1) / T, batch);
  if (ak && bk)
    scae::launch((gemm_kernel<SK, true, true>), grid, dim3(NT), 0, st, g);
  else if (ak)
    scae::launch((gemm_kernel<SK, true, false>), grid, dim3(NT), 0, st, g);
  else if (bk)
    scae::launch((gemm_kernel<SK, false, true>), grid, dim3(NT), 0, st, g);
  else
    scae::launch((gemm_kernel<SK, false, false>), grid, dim3(NT), 0, st, g);
}
#endif
}  // namespace

#ifndef SCAE_DEVICE_ONLY

// bf16 operands (MODE 3, 128 x 128 tiles) unless a side is so short that most of a tile
// would be padding (rows / columns past the problem are not loaded, only multiplied):
// bf16_shape above

static int gemm_impl(const float *A, const float *B, float *C, const float *bias,
                     const float *mask, float *asum, int batch, int M, int N, int K,
                     int a_kcontig, int lda, int64_t a_batch, int b_kcontig, int ldb,
                     int64_t b_batch, int ldc, int64_t c_batch, int bias_ld, int64_t bias_batch,
                     int ldmask, int64_t mask_batch, int64_t asum_batch, int asum_ld, int relu,
                     bool bf16, void *stream) {
  SCAE_REQUIRE(A && B && C && batch > 0 && M > 0 && N > 0 && K > 0);
  if (asum && a_kcontig) return SCAE_ERR_UNSUPPORTED;
  GemmArgs g{A, B, bias, mask, C, asum, nullptr, (long)a_batch, (long)b_batch, (long)c_batch,
             (long)bias_batch, (long)mask_batch, (long)asum_batch, lda, ldb, ldc, bias_ld,
             ldmask, M, N, K, relu, asum_ld > 0 ? asum_ld : 1};
  const long tiles64 = (long)((N + 63) / 64) * ((M + 63) / 64) * batch;
  if (bf16 && bf16_shape(M, N))
    launch<3>(g, batch, a_kcontig, b_kcontig, (hipStream_t)stream);
  else if (tiles64 < kSplitKBelow)
    launch<1>(g, batch, a_kcontig, b_kcontig, (hipStream_t)stream);
  else
    launch<0>(g, batch, a_kcontig, b_kcontig, (hipStream_t)stream);
  return scae_launch_status();
}

extern "C" int scae_gemm_f32(const float *A, const float *B, float *C, const float *bias,
                             const float *mask, float *asum, int batch, int M, int N, int K,
                             int a_kcontig, int lda, int64_t a_batch, int b_kcontig, int ldb,
                             int64_t b_batch, int ldc, int64_t c_batch, int bias_ld,
                             int64_t bias_batch, int ldmask, int64_t mask_batch,
                             int64_t asum_batch, int asum_ld, int relu, void *stream) {
  return gemm_impl(A, B, C, bias, mask, asum, batch, M, N, K, a_kcontig, lda, a_batch, b_kcontig,
                   ldb, b_batch, ldc, c_batch, bias_ld, bias_batch, ldmask, mask_batch,
                   asum_batch, asum_ld, relu, false, stream);
}
extern "C" int scae_gemm_bf16(const float *A, const float *B, float *C, const float *bias,
                              const float *mask, float *asum, int batch, int M, int N, int K,
                              int a_kcontig, int lda, int64_t a_batch, int b_kcontig, int ldb,
                              int64_t b_batch, int ldc, int64_t c_batch, int bias_ld,
                              int64_t bias_batch, int ldmask, int64_t mask_batch,
                              int64_t asum_batch, int asum_ld, int relu, void *stream) {
  return gemm_impl(A, B, C, bias, mask, asum, batch, M, N, K, a_kcontig, lda, a_batch, b_kcontig,
                   ldb, b_batch, ldc, c_batch, bias_ld, bias_batch, ldmask, mask_batch,
                   asum_batch, asum_ld, relu, true, stream);
}

// gemm_ksplit.hip: small problem lists with long K on wave-private pipelines, or UNSUPPORTED
int scae_gemm_x6k_try(const scae_gemm_desc *descs, int n, void *stream);

static int gemm_multi_impl(const scae_gemm_desc *descs, int n, void *stream, bool bf16 = false) {
  SCAE_REQUIRE(descs && n >= 1 && n <= 4);
  if (!bf16) {
    const int rc = scae_gemm_x6k_try(descs, n, stream);
    if (rc != SCAE_ERR_UNSUPPORTED) return rc;
  }
  GemmMulti p;
  int T;
  int rc = plan_multi(p, T, descs, n, bf16);
  if (rc) return rc;
  if (T == 128)
    scae::launch(gemm_multi_kernel<3>, dim3(p.first[n]), dim3(NT), 0, (hipStream_t)stream, p);
  else if (T == 32)
    scae::launch(gemm_multi_kernel<1>, dim3(p.first[n]), dim3(NT), 0, (hipStream_t)stream, p);
  else
    scae::launch(gemm_multi_kernel<0>, dim3(p.first[n]), dim3(NT), 0, (hipStream_t)stream, p);
  return scae_launch_status();
}

static int gemm_pair_impl(const scae_gemm_desc *first, const scae_gemm_desc *second, bool bf16,
                          void *stream) {
  SCAE_REQUIRE(first && second);
  if (!(bf16 && bf16_shape(first->M, first->N) && bf16_shape(second->M, second->N))) {
    const scae_gemm_desc both[2] = {*first, *second};
    return gemm_multi_impl(both, 2, stream);   // exactly the tiles that exist
  }
  GemmPair p;
  int rc = fill_args(p.g[0], first);
  if (rc) return rc;
  rc = fill_args(p.g[1], second);
  if (rc) return rc;
  p.nz0 = first->batch;
  p.layout[0] = 2 * (first->a_kcontig != 0) + (first->b_kcontig != 0);
  p.layout[1] = 2 * (second->a_kcontig != 0) + (second->b_kcontig != 0);
  const int M = first->M > second->M ? first->M : second->M;
  const int N = first->N > second->N ? first->N : second->N;
  scae::launch(gemm_pair_kernel<3>, dim3((N + 127) / 128, (M + 127) / 128,
                                               first->batch + second->batch),
                     dim3(NT), 0, (hipStream_t)stream, p);
  return scae_launch_status();
}

extern "C" int scae_gemm_pair_f32(const scae_gemm_desc *first, const scae_gemm_desc *second,
                                  void *stream) {
  return gemm_pair_impl(first, second, false, stream);
}
extern "C" int scae_gemm_pair_bf16(const scae_gemm_desc *first, const scae_gemm_desc *second,
                                   void *stream) {
  return gemm_pair_impl(first, second, true, stream);
}

extern "C" int scae_gemm_multi_f32(const scae_gemm_desc *descs, int n, void *stream) {
  return gemm_multi_impl(descs, n, stream);
}
extern "C" int scae_gemm_multi_bf16(const scae_gemm_desc *descs, int n, void *stream) {
  return gemm_multi_impl(descs, n, stream, true);
}
#endif  // SCAE_DEVICE_ONLY
