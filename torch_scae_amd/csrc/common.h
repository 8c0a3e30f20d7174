// Shared device helpers for the SCAE gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <tuple>
#include <type_traits>
#include <utility>

#include "scae_hip.h"

#define SCAE_WAVE 64

#define SCAE_REQUIRE(cond)            \
  do {                                \
    if (!(cond)) return SCAE_ERR_BAD_ARG; \
  } while (0)

// ---- launches, and the list a training step replays --------------------------------------
// Every kernel of this library is launched through scae::launch (the arguments of
// hipLaunchKernelGGL).  While a launch list is recording on the launch's stream
// (scae_launch_list_begin(stream) / _end, abi.hip) each SUCCESSFUL launch is also appended to
// it -- kernel, grid, block, LDS bytes and a copy of its arguments -- and
// scae_launch_list_run re-issues the list on a stream: what a captured HIP graph of the same
// launches does, without the ~8.6 us the end of a graph launch costs on the device
//  and without the launchers' host-side planning.  The list
// holds pointers, not buffers: whoever replays it keeps them alive.
namespace scae_rec {
bool recording();   // (a relaxed counter: no cost when nothing records)
void append(const void *fn, dim3 grid, dim3 block, size_t lds, hipStream_t st,
            void *const *args, const size_t *sizes, int n);
void note_launch_error(int e);   // per host thread
int take_launch_error();
}  // namespace scae_rec

// Returns the hipError_t of the launch(es) this thread has just enqueued (0 = success): what
// hipLaunchKernel itself returned to scae::launch, else the runtime's sticky last error.
static inline int scae_launch_status() {
  const int own = scae_rec::take_launch_error();
  hipError_t e = hipGetLastError();
  if (own) return own;
  return e == hipSuccess ? SCAE_OK : (int)e;
}

namespace scae {
template <class... KA, size_t... I, class... A>
inline void launch_impl(void (*kernel)(KA...), std::index_sequence<I...>, dim3 grid, dim3 block,
                        size_t lds, hipStream_t st, A &&...args) {
  static_assert(sizeof...(KA) == sizeof...(A), "one argument per kernel parameter");
  std::tuple<std::remove_cv_t<KA>...> held{static_cast<std::remove_cv_t<KA>>(args)...};
  void *ptrs[sizeof...(KA) + 1] = {const_cast<void *>(static_cast<const void *>(&std::get<I>(held)))...};
  const hipError_t e =
      hipLaunchKernel(reinterpret_cast<const void *>(kernel), grid, block, ptrs, lds, st);
  if (e != hipSuccess) {   // (a launcher that issues several kernels reports the first failure)
    scae_rec::note_launch_error((int)e);
    return;                // never recorded: a replay must not re-issue a launch that failed
  }
  if (scae_rec::recording()) {
    const size_t sizes[sizeof...(KA) + 1] = {sizeof(std::remove_cv_t<KA>)...};
    scae_rec::append(reinterpret_cast<const void *>(kernel), grid, block, lds, st, ptrs, sizes,
                     (int)sizeof...(KA));
  }
}
template <class... KA, class... A>
inline void launch(void (*kernel)(KA...), dim3 grid, dim3 block, size_t lds, hipStream_t st,
                   A &&...args) {
  launch_impl(kernel, std::index_sequence_for<KA...>{}, grid, block, lds, st,
              std::forward<A>(args)...);
}

constexpr float kHalfLog2Pi = 0.91893853320467274178f;
constexpr float kLogSafeEps = 1e-16f;   // math_ops.py:18
constexpr float kLogSafeFloor = -1e8f;  // math_ops.py:21

__device__ __forceinline__ float log_safe(float x) {
  return x < kLogSafeEps ? kLogSafeFloor : logf(x);
}
// d log_safe / dx: the reference's torch.where routes zero grad to the masked
// branch (log(1)=0 path), so the derivative is 0 below eps.
__device__ __forceinline__ float log_safe_grad(float x) {
  return x < kLogSafeEps ? 0.f : 1.f / x;
}
__device__ __forceinline__ float sigmoidf_(float x) {
  return 1.f / (1.f + expf(-x));
}
// F.softplus with the default threshold 20 (torch semantics).
__device__ __forceinline__ float softplusf_(float x) {
  return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float softplus_grad(float x) {
  return x > 20.f ? 1.f : sigmoidf_(x);
}

// Wave / 16-lane-row reductions without LDS round trips: quad butterflies and the
// half-row / row mirrors are DPP operand modifiers (VALU speed; a __shfl_xor is a
// ds_bpermute, ~100 cycles of dependent latency each), the four row sums of a wave meet
// through v_readlane.  Every lane gets the result; fixed summation order.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// lane ^ 1 / lane ^ 2 within a quad (what __shfl_xor(v, 1 | 2) returns, without the LDS trip)
__device__ __forceinline__ float xor1_f(float v) { return dpp_f<0xB1>(v); }
__device__ __forceinline__ float xor2_f(float v) { return dpp_f<0x4E>(v); }
__device__ __forceinline__ int row_min16(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true));
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true));
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true));
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true));
  return v;
}
__device__ __forceinline__ float row_sum16(float v) {
  v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);   // row_half_mirror
  v += dpp_f<0x140>(v);   // row_mirror
  return v;
}
__device__ __forceinline__ float row_max16(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  return v;
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row_sum16(v);
  return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = row_max16(v);
  return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)),
               fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}

// Sum `N` per-thread values over a block of NT threads (NT multiple of 64).
// `red` is LDS scratch of at least N * (NT/64) floats.  Result valid in
// thread 0 (returned in vals[]).  Contains __syncthreads().
template <int N, int NT>
__device__ __forceinline__ void block_sum(float (&vals)[N], float *red) {
  constexpr int NW = NT / 64;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) vals[i] = wave_sum(vals[i]);
  if (NW > 1) {
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < N; ++i) red[i * NW + wid] = vals[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        float s = 0.f;
        for (int w = 0; w < NW; ++w) s += red[i * NW + w];
        vals[i] = s;
      }
    }
    __syncthreads();
  }
}

// Online log-sum-exp accumulator.  m starts at a large finite negative so
// that (m - m_new) never evaluates inf - inf.
struct Lse {
  float m, s;
  __device__ __forceinline__ void init() {
    m = -3.0e38f;
    s = 0.f;
  }
  __device__ __forceinline__ void add(float v) {
    float mn = fmaxf(m, v);
    s = s * __expf(m - mn) + __expf(v - mn);
    m = mn;
  }
  __device__ __forceinline__ void merge(float m2, float s2) {
    float mn = fmaxf(m, m2);
    s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
    m = mn;
  }
  __device__ __forceinline__ float value() const { return m + __logf(s); }
};

}  // namespace scae
