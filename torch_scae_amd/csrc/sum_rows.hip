// Column sums of a (rows x cols) matrix of per-workgroup / per-image partial
// gradients, scattered to up to 8 contiguous destinations -- the second half
// of every "partials, then sum" backward kernel of this library (K1, K2b, K2c,
// K3, K8, K9, K10).  Replaces a generic ATen reduction plus one strided-slice
// copy per parameter with one launch; fixed summation order (bit-reproducible).
#include "sum_rows_dev.h"

namespace {
using namespace scae_sums;
__global__ __launch_bounds__(NT) void sum_rows_kernel(Jobs jobs) {
  __shared__ float red[NT];
  sum_block(jobs, blockIdx.x, red, [](float *dst, float v) { *dst = v; });
}
struct SumJobs {
  scae_scaled_sum j[8];
};
__global__ __launch_bounds__(1024) void scaled_sums_kernel(SumJobs jobs) {
  __shared__ float red[16];
  const scae_scaled_sum &job = jobs.j[blockIdx.x];
  float v[1] = {0.f};
  for (int64_t i = threadIdx.x; i < job.n; i += 1024) v[0] += job.src[i];
  scae::block_sum<1, 1024>(v, red);
  if (threadIdx.x == 0) job.dst[0] = v[0] * job.scale;
}
}  // namespace

extern "C" int scae_scaled_sums_f32(const scae_scaled_sum *jobs, int n_jobs, void *stream) {
  SCAE_REQUIRE(jobs && n_jobs > 0 && n_jobs <= 8);
  SumJobs sj;
  for (int i = 0; i < n_jobs; ++i) {
    sj.j[i] = jobs[i];
    SCAE_REQUIRE(sj.j[i].src && sj.j[i].dst && sj.j[i].n > 0);
  }
  scae::launch(scaled_sums_kernel, dim3(n_jobs), dim3(1024), 0, (hipStream_t)stream, sj);
  return scae_launch_status();
}

extern "C" int scae_sum_rows_multi_f32(const scae_sum_job *jobs, int n_jobs, void *stream) {
  Jobs js;
  const int blocks = fill_jobs(js, jobs, n_jobs);
  SCAE_REQUIRE(blocks > 0);
  scae::launch(sum_rows_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, js);
  return scae_launch_status();
}

extern "C" int scae_sum_rows_f32(const float *src, int64_t rows, int64_t cols,
                                 const scae_sum_segment *segments, int n_segments,
                                 void *stream) {
  const scae_sum_job job{src, rows, cols, segments, n_segments};
  return scae_sum_rows_multi_f32(&job, 1, stream);
}
