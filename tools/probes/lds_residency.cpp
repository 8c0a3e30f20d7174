// How many workgroups of a kernel with L bytes of dynamic LDS does an MI355X CU hold at once?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/lds_residency.cpp -o tools/probes/lds_residency   (git-ignored; travels with gpurun)
// 2048 workgroups (8 per CU) of `waves` waves spin for ~20 us and stamp their start; those that
// start within the first 5 us are the first round: residency = that count / 256 CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void spin(unsigned long long *st, int ticks) {
  extern __shared__ float lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) st[blockIdx.x] = t0;
  lds[threadIdx.x] = (float)t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
  if (lds[threadIdx.x] == 1.f) st[0] = 0;
}
int main() {
  const int G = 2048;
  unsigned long long *d;
  hipMalloc(&d, G * 8);
  std::vector<unsigned long long> h(G);
  hipFuncSetAttribute(reinterpret_cast<const void *>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int waves : {1, 5, 7}) {
    for (int kb : {8, 16, 20, 24, 26, 28, 32, 36, 40, 42, 44, 48, 52, 56, 64, 72, 80, 96, 128, 160}) {
      hipMemset(d, 0, G * 8);
      hipLaunchKernelGGL(spin, dim3(G), dim3(64 * waves), kb * 1024, 0, d, 2000);
      if (hipDeviceSynchronize() != hipSuccess) { printf("waves %d lds %d KB: launch failed\n", waves, kb); continue; }
      hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
      const unsigned long long t0 = *std::min_element(h.begin(), h.end());
      int first = 0;
      for (auto t : h) first += (t - t0) < 500;
      printf("waves/WG %d  LDS %3d KB: %4d of %d workgroups in the first round = %.2f per CU\n", waves, kb, first, G, first / 256.0);
    }
  }
  return 0;
}
