// Do kernels of two HIP streams overlap on this stack when nothing but the streams orders them?
// (plain HIP, no torch, no graphs).  A kernel = `wgs` workgroups of 256 threads that each spin
// for `us` microseconds on s_memrealtime (100 MHz): it occupies wgs places and nothing else.
//   build: hipcc --offload-arch=gfx950 -O2 tools/probes/stream_overlap.cpp -o tools/probes/stream_overlap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(int ticks, unsigned long long *sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t = t0;
  while (t - t0 < (unsigned long long)ticks) t = __builtin_amdgcn_s_memrealtime();
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = t;
}

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("%s -> %s\n", #x, hipGetErrorString(e_));                         \
      return 1;                                                                \
    }                                                                          \
  } while (0)

static double now_us() {
  return std::chrono::duration<double, std::micro>(
             std::chrono::steady_clock::now().time_since_epoch()).count();
}

// n_a kernels of (wgs_a, us_a) on sa; n_b of (wgs_b, us_b) on sb; both start after a common
// event; returns wall microseconds (device idle before and after)
static int run(hipStream_t sa, hipStream_t sb, int n_a, int wgs_a, int us_a, int n_b, int wgs_b,
               int us_b, double *out) {
  CK(hipDeviceSynchronize());
  const double t0 = now_us();
  for (int i = 0; i < (n_a > n_b ? n_a : n_b); ++i) {   // interleaved host issue
    if (i < n_a) hipLaunchKernelGGL(spin, dim3(wgs_a), dim3(256), 0, sa, us_a * 100, nullptr);
    if (i < n_b) hipLaunchKernelGGL(spin, dim3(wgs_b), dim3(256), 0, sb, us_b * 100, nullptr);
  }
  CK(hipStreamSynchronize(sa));
  CK(hipStreamSynchronize(sb));
  *out = now_us() - t0;
  return 0;
}

int main() {
  hipStream_t s0, s1, n0, n1, p0, p1;
  CK(hipStreamCreate(&s0));
  CK(hipStreamCreate(&s1));
  CK(hipStreamCreateWithFlags(&n0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&n1, hipStreamNonBlocking));
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&p0, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&p1, hipStreamNonBlocking, lo));
  printf("priority range %d .. %d\n", lo, hi);
  // CU-masked pair: the first 64 CUs / the other 192 (8 x 32-bit words = 256 CUs)
  hipStream_t m0 = nullptr, m1 = nullptr;
  {
    std::vector<uint32_t> a(8, 0), b(8, 0xffffffffu);
    a[0] = a[1] = 0xffffffffu, b[0] = b[1] = 0;
    hipError_t e0 = hipExtStreamCreateWithCUMask(&m0, 8, a.data());
    hipError_t e1 = hipExtStreamCreateWithCUMask(&m1, 8, b.data());
    printf("cu-mask streams: %s / %s\n", hipGetErrorString(e0), hipGetErrorString(e1));
    if (e0 != hipSuccess || e1 != hipSuccess) m0 = m1 = nullptr;
  }
  struct Pair { const char *name; hipStream_t a, b; } pairs[] = {
      {"same stream        ", s0, s0}, {"hipStreamCreate x2 ", s0, s1},
      {"non-blocking x2    ", n0, n1}, {"priorities hi / lo ", p0, p1},
      {"cu masks 64 / 192  ", m0, m1}};
  struct Case { const char *name; int n_a, wgs_a, us_a, n_b, wgs_b, us_b; } cases[] = {
      {"1 x (64 wg, 200 us) || 1 x (64 wg, 200 us)", 1, 64, 200, 1, 64, 200},
      {"4 x (64 wg, 50 us)  || 4 x (64 wg, 50 us) ", 4, 64, 50, 4, 64, 50},
      {"4 x (2048 wg, 20 us) || 10 x (128 wg, 20 us)", 4, 2048, 20, 10, 128, 20},
      {"3 x (3200 wg, 15 us) || 6 x (128 wg, 15 us)", 3, 3200, 15, 6, 128, 15}};
  for (auto &c : cases) {
    printf("%s\n", c.name);
    for (auto &p : pairs) {
      if (!p.a) continue;
      double best = 1e30, t;
      for (int rep = 0; rep < 5; ++rep) {
        if (run(p.a, p.b, c.n_a, c.wgs_a, c.us_a, c.n_b, c.wgs_b, c.us_b, &t)) return 1;
        if (t < best) best = t;
      }
      printf("   %s  %8.1f us\n", p.name, best);
    }
  }
  // the same through events on the device: A on sa; B on sb after an event on sa's start
  return 0;
}
