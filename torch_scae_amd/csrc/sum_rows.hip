// Column sums of a (rows x cols) matrix of per-workgroup / per-image partial
// gradients, scattered to up to 8 contiguous destinations -- the second half
// of every "partials, then sum" backward kernel of this library (K1, K2b, K2c,
// K3, K8, K9, K10).  Replaces a generic ATen reduction plus one strided-slice
// copy per parameter with one launch; fixed summation order (bit-reproducible).
#include "common.h"

namespace {
constexpr int NT = 256;
constexpr int MAXJOBS = 16;  // 16 x 224 B of kernel arguments
struct Seg {       // scae_sum_segment with 32-bit columns
  float *dst;
  int begin, end, period;
};
struct Job {
  const float *src;
  int rows, cols;
  int py;           // row parts per column (1 | 4 | 16 | 64); NT / py columns per workgroup
  int first_block;  // first workgroup of this job
  int n;
  Seg s[8];
};
struct Jobs {
  Job j[MAXJOBS];
  int n;
};

// workgroup = (256 / py columns) x py row-parts; parts meet in LDS
__global__ __launch_bounds__(NT) void sum_rows_kernel(Jobs jobs) {
  __shared__ float red[NT];
  int ji = 0;
  while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.j[ji + 1].first_block) ++ji;
  const Job &job = jobs.j[ji];
  const float *__restrict__ src = job.src;
  const long rows = job.rows, cols = job.cols;
  const int PY = job.py, CX = NT / PY;
  const int cx = threadIdx.x % CX, py = threadIdx.x / CX;
  const long j = (long)((int)blockIdx.x - job.first_block) * CX + cx;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (j < cols) {  // eight loads in flight per thread: the kernel is latency bound
    const long per = (rows + PY - 1) / PY, r0 = py * per, r1 = min(rows, r0 + per);
    // (the tail batch too: a row part of 5 or 6 rows -- the 22 split partials of a convolution
    // weight gradient over 4 parts -- used to be a chain of dependent round trips)
    for (long r = r0; r < r1; r += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = r + u < r1 ? src[(r + u) * cols + j] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += v[u];
    }
  }
  red[py * CX + cx] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (py != 0 || j >= cols) return;
  float tot = 0.f;
  for (int p = 0; p < PY; ++p) tot += red[p * CX + cx];
  for (int i = 0; i < job.n; ++i) {
    const Seg &g = job.s[i];
    if (g.period > 0) {  // the same column window of every period-wide block
      const long blk = j / g.period, c = j - blk * g.period;
      if (c >= g.begin && c < g.end) g.dst[blk * (g.end - g.begin) + c - g.begin] = tot;
    } else if (j >= g.begin && j < g.end) {
      if (g.period < 0) {  // the window is an (n x W) matrix: write its transpose
        const long W = -g.period, l = j - g.begin;
        g.dst[(l % W) * ((g.end - g.begin) / W) + l / W] = tot;
      } else {
        g.dst[j - g.begin] = tot;
      }
    }
  }
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
  hipLaunchKernelGGL(scaled_sums_kernel, dim3(n_jobs), dim3(1024), 0, (hipStream_t)stream, sj);
  return scae_launch_status();
}

extern "C" int scae_sum_rows_multi_f32(const scae_sum_job *jobs, int n_jobs, void *stream) {
  SCAE_REQUIRE(jobs && n_jobs > 0 && n_jobs <= MAXJOBS);
  Jobs js;
  js.n = n_jobs;
  int blocks = 0;
  for (int k = 0; k < n_jobs; ++k) {
    const scae_sum_job &in = jobs[k];
    SCAE_REQUIRE(in.src && in.segments && in.rows > 0 && in.cols > 0 && in.n_segments > 0 &&
                 in.n_segments <= 8 && in.rows < (1ll << 31) && in.cols < (1ll << 31));
    Job &job = js.j[k];
    job.src = in.src, job.rows = (int)in.rows, job.cols = (int)in.cols, job.n = in.n_segments;
    for (int i = 0; i < in.n_segments; ++i) {
      const scae_sum_segment &g = in.segments[i];
      job.s[i] = Seg{g.dst, (int)g.begin, (int)g.end, (int)g.period};
      SCAE_REQUIRE(g.dst && g.begin >= 0 && g.begin < g.end &&
                   g.end <= (g.period > 0 ? g.period : in.cols));
      SCAE_REQUIRE(g.period >= 0 || (g.end - g.begin) % -g.period == 0);
    }
    // few rows: a thread per column; tall and skinny: many row parts per column
#ifndef SCAE_SUMROWS_MID
#define SCAE_SUMROWS_MID 4
#endif
    job.py = in.rows <= 16 ? 1
             : (in.cols <= 8 && in.rows > 256) ? 64
             : (in.rows <= 128 || in.cols >= 16384) ? SCAE_SUMROWS_MID : 16;
    const int cx = NT / job.py;
    job.first_block = blocks;
    blocks += (int)((in.cols + cx - 1) / cx);
  }
  hipLaunchKernelGGL(sum_rows_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, js);
  return scae_launch_status();
}

extern "C" int scae_sum_rows_f32(const float *src, int64_t rows, int64_t cols,
                                 const scae_sum_segment *segments, int n_segments,
                                 void *stream) {
  const scae_sum_job job{src, rows, cols, segments, n_segments};
  return scae_sum_rows_multi_f32(&job, 1, stream);
}
