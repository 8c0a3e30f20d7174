// Column sums of a (rows x cols) matrix of per-workgroup / per-image partial
// gradients, scattered to up to 8 contiguous destinations -- the second half
// of every "partials, then sum" backward kernel of this library (K1, K2b, K2c,
// K3, K8, K9, K10).  Replaces a generic ATen reduction plus one strided-slice
// copy per parameter with one launch; fixed summation order (bit-reproducible).
#include "common.h"

namespace {
constexpr int NT = 256;
struct Segs {
  scae_sum_segment s[8];
  int n;
};

// workgroup = (256 / PY columns) x PY row-parts; parts meet in LDS
template <int PY>
__global__ __launch_bounds__(NT) void sum_rows_kernel(const float *__restrict__ src, long rows,
                                                      long cols, Segs segs) {
  constexpr int CX = NT / PY;
  __shared__ float red[PY][CX];
  const int cx = threadIdx.x % CX, py = threadIdx.x / CX;
  const long j = (long)blockIdx.x * CX + cx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (j < cols) {
    const long per = (rows + PY - 1) / PY, r0 = py * per, r1 = min(rows, r0 + per);
    long r = r0;
    for (; r + 4 <= r1; r += 4) {
      s0 += src[r * cols + j];
      s1 += src[(r + 1) * cols + j];
      s2 += src[(r + 2) * cols + j];
      s3 += src[(r + 3) * cols + j];
    }
    for (; r < r1; ++r) s0 += src[r * cols + j];
  }
  red[py][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (py != 0 || j >= cols) return;
  float tot = 0.f;
#pragma unroll
  for (int p = 0; p < PY; ++p) tot += red[p][cx];
  for (int i = 0; i < segs.n; ++i) {
    const scae_sum_segment &g = segs.s[i];
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

extern "C" int scae_sum_rows_f32(const float *src, int64_t rows, int64_t cols,
                                 const scae_sum_segment *segments, int n_segments,
                                 void *stream) {
  SCAE_REQUIRE(src && segments && rows > 0 && cols > 0 && n_segments > 0 && n_segments <= 8);
  Segs segs;
  segs.n = n_segments;
  for (int i = 0; i < n_segments; ++i) {
    segs.s[i] = segments[i];
    SCAE_REQUIRE(segs.s[i].dst && segs.s[i].begin >= 0 && segs.s[i].begin < segs.s[i].end &&
                 segs.s[i].end <= (segs.s[i].period > 0 ? segs.s[i].period : cols));
    SCAE_REQUIRE(segs.s[i].period >= 0 ||
                 (segs.s[i].end - segs.s[i].begin) % -segs.s[i].period == 0);
  }
  hipStream_t st = (hipStream_t)stream;
  if (rows <= 16) {
    hipLaunchKernelGGL(sum_rows_kernel<1>, dim3((unsigned)((cols + 255) / 256)), dim3(NT), 0, st,
                       src, (long)rows, (long)cols, segs);
  } else if (rows <= 128 || cols >= 16384) {
    hipLaunchKernelGGL(sum_rows_kernel<4>, dim3((unsigned)((cols + 63) / 64)), dim3(NT), 0, st,
                       src, (long)rows, (long)cols, segs);
  } else {
    hipLaunchKernelGGL(sum_rows_kernel<16>, dim3((unsigned)((cols + 15) / 16)), dim3(NT), 0, st,
                       src, (long)rows, (long)cols, segs);
  }
  return scae_launch_status();
}
