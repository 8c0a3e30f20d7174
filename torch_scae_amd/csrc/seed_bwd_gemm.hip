// The output attention's backward (K2c, seed_attention_wave.hip) and the weight-gradient
// GEMMs of the capsule MLPs (K7, gemm_mfma.hip) in ONE launch.
//
// Both only wait for the data-gradient chain of the capsule MLPs (set_transformer.py:212-223
// backward needs the gradient of the object encoding; object_decoder.py:137-158 backward
// needs the pre-activation gradients), neither reads what the other writes, and kernels do
// not overlap on this stack.  Alone, the attention backward is one workgroup per set -- 128
// workgroups at B = 128, each a 15 us dependent chain on a quarter of the CUs -- and the four
// GEMMs are ~1.4 k short 32 x 32 split-K tiles that would fill the rest.  Here the
// attention workgroups are the head of the grid and the GEMM tiles its tail: 15.6 + 17.3 us
// as two launches.  Launch-uniform resources are the attention's (160 VGPRs, 47 KB of LDS
// at N, O <= 32 -- the only shape merged): the GEMM tiles (108 VGPRs, 18 KB on their own)
// still fit three workgroups per CU.
//
// Both device bodies come from their own files, compiled here without their kernels and
// entry points (SCAE_DEVICE_ONLY); each sits in a namespace of its own because the two tile
// vocabularies (scae_wave / scae_tile) share names.
#include <algorithm>

#include "mfma_tile.h"
#include "wave_mfma.h"

#define SCAE_DEVICE_ONLY
namespace scae_saw {
#include "seed_attention_wave.hip"
}
namespace scae_gemm {
#include "gemm_mfma.hip"
}
#undef SCAE_DEVICE_ONLY

namespace {
constexpr int NTH = 256;
static_assert(scae_tile::NT == NTH, "one block size for both parts");

template <int NT, bool BF>
__global__ __launch_bounds__(NTH) void saw_bwd_gemm_kernel(scae_saw::SwArgs a, int n_saw,
                                                           scae_gemm::GemmMulti p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < n_saw)   // (the long dependent chains first)
    scae_saw::saw_bwd_body<NT, BF>(a, smem, blockIdx.x, n_saw);
  else
    scae_gemm::gemm_multi_body<1>(p, smem, (int)blockIdx.x - n_saw);
}

template <int NT, bool BF>
int launch(const scae_saw::SwArgs &a, const scae_gemm::GemmMulti &p, hipStream_t st) {
  const size_t lds = sizeof(float) * std::max<size_t>(scae_saw::Geo<NT>::LDS_FLOATS,
                                                      scae_tile::Tile<1>::SMEM);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void *>(saw_bwd_gemm_kernel<NT, BF>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int n_saw = a.B < 512 ? a.B : 512;   // = scae_seed_attention_mfma_rows(B)
  scae::launch((saw_bwd_gemm_kernel<NT, BF>), dim3(n_saw + p.first[p.n]), dim3(NTH), lds,
                     st, a, n_saw, p);
  return scae_launch_status();
}

int impl(const float *h, const float *q, const float *wk, const float *wv,
         const float *presence, const float *gout, float *gh, float *partial, int B, int N,
         int O, int C, const scae_gemm_desc *descs, int n, int bf16, void *stream) {
  SCAE_REQUIRE(h && q && wk && wv && gout && gh && partial && descs);
  scae_saw::SwArgs a{h,  q,       wk, wv, nullptr, presence, nullptr, gout,
                     gh, partial, B,  N,  O,       C,        1.f / sqrtf((float)C), bf16};
  int rc = scae_saw::check(a);
  if (rc) return rc;
  scae_gemm::GemmMulti p;
  int T;
  rc = scae_gemm::plan_multi(p, T, descs, n);
  if (rc) return rc;
  if (T != 32) return SCAE_ERR_UNSUPPORTED;   // (large problems fill the device on their own)
  // sets of more than 32 elements: the attention's 4-tile form holds 268 VGPRs -- one GEMM
  // workgroup per SIMD instead of four; the two launches are faster apart
  if (N > 32 || O > 32) return SCAE_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  return bf16 ? launch<2, true>(a, p, st) : launch<2, false>(a, p, st);
}
}  // namespace

extern "C" int scae_seed_attention_mfma_bwd_gemm_f32(
    const float *h, const float *q, const float *wk, const float *wv, const float *presence,
    const float *gout, float *gh, float *partial, int B, int N, int O, int C,
    const scae_gemm_desc *descs, int n, void *stream) {
  return impl(h, q, wk, wv, presence, gout, gh, partial, B, N, O, C, descs, n, 0, stream);
}
extern "C" int scae_seed_attention_mfma_bwd_gemm_bf16(
    const float *h, const float *q, const float *wk, const float *wv, const float *presence,
    const float *gout, float *gh, float *partial, int B, int N, int O, int C,
    const scae_gemm_desc *descs, int n, void *stream) {
  return impl(h, q, wk, wv, presence, gout, gh, partial, B, N, O, C, descs, n, 1, stream);
}
