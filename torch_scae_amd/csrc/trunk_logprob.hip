// Two independent forward kernels in ONE launch: the object encoder's fused trunk (K2b,
// set_encoder_wave.hip: one wave per 16-row tile of a set -- 256 waves on 1024 SIMDs at
// cfg-2, a chain of dependent MFMA / LDS latencies) and the part decoder's likelihood (K1,
// render_gmm_wave_dev.h: VALU work that fills the chip).  Both only need the part
// encoder's outputs, kernels do not overlap on this stack (forked graphs serialise,
// DESIGN.md section 5), so the likelihood's workgroups ride in the trunk's launch as a
// second block range: 14.3 + 13.2 us alone, 19.5 together.
//   blocks [0, n_trunk)            : trunk workgroup (threads 0 .. 64 NT - 1; the other waves
//                                    of the workgroup exit at once)
//   blocks [n_trunk, + tiles * B)  : likelihood workgroup (tile, image)
// Register allocation is the maximum of the two bodies (the trunk's 128 VGPRs: 4 waves per
// SIMD), dynamic LDS the maximum of the two (the trunk's tiles lie over its dead input rows
// here, stw_fwd_body<.., ALIAS>: 28 KB; the likelihood's planes 38 KB at cfg-2).  What
// decides the launch's time is whether ALL its workgroups are resident at once: a CU deals a
// workgroup's waves to its SIMDs from the first one on, so at 4 waves per SIMD it holds two
// 5- or 7-wave workgroups but four 4-wave ones (tools/probes/lds_residency.cpp); the
// likelihood's tiling (render_gmm.hip, lp_tiling) takes 4-wave workgroups where four also
// fit the LDS -- with 7-wave ones a quarter of them ran in a second round (tools/tl_prof.py).
#include "common.h"
#define SCAE_DEVICE_ONLY
#include "set_encoder_wave.hip"
#undef SCAE_DEVICE_ONLY
#include "render_gmm_wave_dev.h"

#ifdef SCAE_TL_PROF   // start / end stamp (s_memrealtime, 100 MHz) of every workgroup
__device__ unsigned long long g_tl_prof[2048][2];
extern "C" int scae_debug_tl_prof(unsigned long long *out, int n) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl_prof), (size_t)n * 16);
}
#define TL_STAMP(i)                                                                       \
  do {                                                                                    \
    if (threadIdx.x == 0 && blockIdx.x < 2048)                                            \
      g_tl_prof[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime();                        \
  } while (0)
#else
#define TL_STAMP(i)
#endif

namespace {
template <int NT, int C, bool BF>
__global__ __launch_bounds__(1024) void trunk_logprob_kernel(
    scae_st::StArgs a, int n_trunk, scae_decoder_desc d, const float *__restrict__ x,
    float *__restrict__ lse_post, float *__restrict__ lse_prior, int ppb, int tiles,
    float *__restrict__ tile_sums) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  TL_STAMP(0);
  if ((int)blockIdx.x < n_trunk) {   // (workgroup-uniform)
    if (threadIdx.x >= 64 * NT) return;   // whole waves
    scae_st::stw_fwd_body<NT, BF, true>(a, smem, blockIdx.x, n_trunk);
    TL_STAMP(1);
    return;
  }
  const int id = (int)blockIdx.x - n_trunk, b = id / tiles, tile = id - b * tiles;
  scae_k1::logprob_wave_body<C>(d, x, nullptr, lse_post, lse_prior, ppb, tile_sums, smem, tile, b,
                                tiles, blockDim.x);
  TL_STAMP(1);
}

template <int NT, int C, bool BF>
int launch(const scae_st::StArgs &a, int n_trunk, const scae_decoder_desc *d,
           const scae_k1::LpTiling &t, const float *x, float *tile_sums, float *lse_post,
           float *lse_prior, hipStream_t st) {
  const size_t lds_t = scae_st::stw_fwd_lds_floats<NT>(a.Din) * sizeof(float);
  const size_t lds_l = scae_k1::logprob_wave_lds(d);
  const size_t lds = lds_t > lds_l ? lds_t : lds_l;
  if (lds > 160 * 1024) return SCAE_ERR_UNSUPPORTED;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(trunk_logprob_kernel<NT, C, BF>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int threads = t.ppb > 64 * NT ? t.ppb : 64 * NT;
  scae::launch((trunk_logprob_kernel<NT, C, BF>), dim3(n_trunk + t.tiles * d->B), dim3(threads),
                     lds, st, a, n_trunk, *d, x, lse_post, lse_prior, t.ppb, t.tiles, tile_sums);
  return scae_launch_status();
}
}  // namespace

namespace scae_fused {
// true when the shared launch covers this pair of problems (the trunk on the matrix-core
// kernels -- fp32 or bf16 attention products --, wave-form likelihood with 1 or 3 channels, the likelihood's workgroup at least as
// wide as the trunk's)
bool trunk_logprob_supported(const scae_st::StArgs &a, int Dh, const scae_decoder_desc *d) {
  if (!scae_st::wave_supported(a, Dh)) return false;   // (fp32 and bf16 attention products alike)
  // Sharing pays while the whole grid is resident at once: at B = 1024 (cfg-3) the shared
  // launch took 226 us against 36 + 145 apart (profiles/r04, the likelihood's 7000 workgroups
  // at the trunk's register / LDS allocation)
  if (d->B > 512) return false;
  if (d->C != 1 && d->C != 3) return false;
  const scae_k1::LpTiling t = scae_k1::lp_tiling(d);
  const int nt = (a.N + 15) / 16;
  return t.wave && t.ppb >= 64 * nt && t.ppb <= 1024;
}

int trunk_logprob_launch(const scae_st::StArgs &a, int n_trunk, const scae_decoder_desc *d,
                         const float *x, float *tile_sums, float *lse_post, float *lse_prior,
                         hipStream_t st) {
  const scae_k1::LpTiling t = scae_k1::lp_tiling(d);
  const int nt = (a.N + 15) / 16;
#define SCAE_TL2(NTV, BFV)                                                                        \
  return d->C == 1 ? launch<NTV, 1, BFV>(a, n_trunk, d, t, x, tile_sums, lse_post, lse_prior, st) \
                   : launch<NTV, 3, BFV>(a, n_trunk, d, t, x, tile_sums, lse_post, lse_prior, st)
#define SCAE_TL(NTV)                 \
  if (a.bf16_attention) {            \
    SCAE_TL2(NTV, true);             \
  } else {                           \
    SCAE_TL2(NTV, false);            \
  }
  switch (nt) {
    case 1: SCAE_TL(1);
    case 2: SCAE_TL(2);
    case 3: SCAE_TL(3);
    default: SCAE_TL(4);
  }
#undef SCAE_TL
#undef SCAE_TL2
}
}  // namespace scae_fused
