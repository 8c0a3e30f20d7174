// U[0,1) noise for the presence logits (part_encoder.py:106, object_decoder.py:201:
// torch.rand_like in the reference) from a counter-based Philox4x32-10 generator
// whose state (seed, launch offset) lives in DEVICE memory and is advanced by the
// kernel itself: a captured HIP graph replays fresh draws without the two
// host-driven seed/offset fill launches torch's generator adds to every replay.
#include "common.h"
#include "noise_dev.h"

namespace {
__global__ __launch_bounds__(scae_noise::NT) void uniform_kernel(float *__restrict__ out, int64_t n,
                                                                 uint64_t *__restrict__ state) {
  scae_noise::uniform_block(out, n, state, blockIdx.x, gridDim.x);
}
}  // namespace

extern "C" int scae_uniform_f32(float *out, int64_t n, uint64_t *state, void *stream) {
  SCAE_REQUIRE(out && state && n > 0);
  scae::launch(uniform_kernel, dim3(scae_noise::blocks_for(n)), dim3(scae_noise::NT), 0,
                     (hipStream_t)stream, out, n, state);
  return scae_launch_status();
}
