// Argument block shared by the two implementations of the fused set-transformer trunk
// (set_encoder.hip: one 512-thread workgroup per set, any N <= 64 / D in {8,16,32};
//  set_encoder_wave.hip: one wavefront per set on the matrix cores, D = 16, N <= 32).
#pragma once
#include "common.h"

namespace scae_st {
constexpr int NMAX = 64;   // max set size
constexpr int MAXSEG = 4;  // input given as up to 4 column segments
constexpr float kLnEps = 1e-5f;

struct Seg {
  const float *ptr;  // element (b, n, j) at ptr[b*bs + n*rs + j]
  float *grad;       // nullable, contiguous (B, N, width)
  int width, rs;
  long bs;
};

struct StArgs {
  Seg seg[MAXSEG];
  int nseg;
  const float *presence;  // (B, N) nullable
  const float *params;    // packed, layout in set_encoder.hip (Layout<D>)
  float *z;               // (B, N, Dout)
  float *hsave;           // (B, L+1, N, D): input of every block + trunk output
  const float *gz;        // bwd: (B, N, Dout)
  float *pg_partial;      // bwd: (gridDim.x, P) per-workgroup parameter grads
  int B, N, Din, Dout, L, layer_norm;
  float sqrt_d;
  int bf16_attention;     // wave kernels only: bf16 operands for the attention products
};

// set_encoder_wave.hip: whether it covers the problem, and its launches (grid rows of
// pg_partial = scae_set_encoder_grid(B), as for the workgroup-per-set kernels)
bool wave_supported(const StArgs &a, int D);
int wave_launch(const StArgs &a, bool bwd, int grid, hipStream_t st);
}  // namespace scae_st
