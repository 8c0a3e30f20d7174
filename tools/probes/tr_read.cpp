// What ds_read_b64_tr_b16 returns for the address pattern the bf16 weight gradient wants:
// an LDS image [k = pixel][channel] with a row stride of RS bytes; lane t of each 16-lane group
// g supplies the address of row (t >> 2), channels c0(g) + 4 (t & 3) .. + 3.  Expected: lane t
// receives channel c0(g) + t of rows 0 .. 3 (a 4 x 16 block, transposed).
//   build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read.cpp -o tools/probes/tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short *out, int rs_elems) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, t = l & 15, g = l >> 4;
  const int elem = (t >> 2) * rs_elems + 16 * g + 4 * (t & 3);
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)(lds + elem));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short *d, h[256];
  hipMalloc(&d, sizeof(h));
  for (int rs : {16, 64, 128}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, rs);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) bad += h[l * 4 + j] != j * rs + 16 * (l >> 4) + (l & 15);
    printf("row stride %3d elements: %s   lane 0: %d %d %d %d  lane 5: %d %d %d %d  lane 17: %d %d %d %d\n", rs,
           bad ? "DIFFERENT" : "as expected (lane t <- column t of the 4 x 16 block)", h[0], h[1], h[2],
           h[3], h[20], h[21], h[22], h[23], h[68], h[69], h[70], h[71]);
  }
  return 0;
}
