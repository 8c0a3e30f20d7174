// Device code of the column-sum kernel (sum_rows.hip) -- also the head of the fused
// "last sums + RMSprop" launch of a training step (optimizer.hip).
#pragma once
#include "common.h"

namespace scae_sums {
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

// workgroup `blk` = (256 / py columns) x py row-parts of its job; parts meet in LDS (`red`:
// NT floats); put(address, value) stores a finished sum
template <class Put>
__device__ __forceinline__ void sum_block(const Jobs &jobs, int blk, float *red, Put put) {
  int ji = 0;
  while (ji + 1 < jobs.n && blk >= jobs.j[ji + 1].first_block) ++ji;
  const Job &job = jobs.j[ji];
  const float *__restrict__ src = job.src;
  const long rows = job.rows, cols = job.cols;
  const int PY = job.py, CX = NT / PY;
  const int cx = threadIdx.x % CX, py = threadIdx.x / CX;
  const long j = (long)(blk - job.first_block) * CX + cx;
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
      const long b = j / g.period, c = j - b * g.period;
      if (c >= g.begin && c < g.end) put(g.dst + b * (g.end - g.begin) + c - g.begin, tot);
    } else if (j >= g.begin && j < g.end) {
      if (g.period < 0) {  // the window is an (n x W) matrix: write its transpose
        const long W = -g.period, l = j - g.begin;
        put(g.dst + (l % W) * ((g.end - g.begin) / W) + l / W, tot);
      } else {
        put(g.dst + j - g.begin, tot);
      }
    }
  }
}

// host: scae_sum_job[] -> Jobs; returns the workgroup count or < 0 (bad argument)
inline int fill_jobs(Jobs &js, const scae_sum_job *jobs, int n_jobs) {
  if (!(jobs && n_jobs > 0 && n_jobs <= MAXJOBS)) return -1;
  js.n = n_jobs;
  int blocks = 0;
  for (int k = 0; k < n_jobs; ++k) {
    const scae_sum_job &in = jobs[k];
    if (!(in.src && in.segments && in.rows > 0 && in.cols > 0 && in.n_segments > 0 &&
          in.n_segments <= 8 && in.rows < (1ll << 31) && in.cols < (1ll << 31)))
      return -1;
    Job &job = js.j[k];
    job.src = in.src, job.rows = (int)in.rows, job.cols = (int)in.cols, job.n = in.n_segments;
    for (int i = 0; i < in.n_segments; ++i) {
      const scae_sum_segment &g = in.segments[i];
      job.s[i] = Seg{g.dst, (int)g.begin, (int)g.end, (int)g.period};
      if (!(g.dst && g.begin >= 0 && g.begin < g.end &&
            g.end <= (g.period > 0 ? g.period : in.cols)))
        return -1;
      if (!(g.period >= 0 || (g.end - g.begin) % -g.period == 0)) return -1;
      // a periodic window over a ragged last period has no well-defined destination range
      // (rmsprop_sums_kernel derives the elements a sum workgroup owns from cols / period)
      if (g.period > 0 && in.cols % g.period != 0) return -1;
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
  return blocks;
}
}  // namespace scae_sums
