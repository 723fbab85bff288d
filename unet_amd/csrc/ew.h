// Shared pieces of the bandwidth-bound element-wise kernels: thread mapping (8 channels = 16 bytes per
// thread, lanes along the channel axis), per-workgroup combination of per-channel sums, and the reduction
// of per-workgroup partial rows.
#pragma once
#include "common.h"

static inline int ew_rows_per_iter(int C) { return 256 / (C / 8); }
static inline int ew_blocks(long long Mp, int C) {
  const int rpi = ew_rows_per_iter(C);
  long long b = (Mp + rpi - 1) / rpi;
  if (b > 2048) b = 2048;
  return (int)b;
}
// Workgroups of a kernel that ends in block_combine_store: the combine (an LDS round trip and NSTAT*C floats of
// partial output per workgroup) is paid once per workgroup, so every workgroup gets at least eight row groups.
static inline int ew_stat_blocks(long long Mp, int C) {
  const int rpi = ew_rows_per_iter(C);
  const long long iters = (Mp + rpi - 1) / rpi;
  long long b = (iters + 7) / 8;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}
#define EW_CHECK(name, C)                                                                              \
  UH_REQUIRE((C) > 0 && (C) % 8 == 0 && (C) / 8 <= 256, name ": C=%d must be a multiple of 8, at most 2048", (C))

// block-level combine of per-thread sums: threads with the same channel chunk cc = tid % cpr
// hold sums for channels cc*8..cc*8+7; result row layout [NSTAT][C].
template <int NSTAT>
__device__ __forceinline__ void block_combine_store(float (&acc)[NSTAT][8], int C, int cpr, int rpi, float* lds,
                                                    float* out_row) {
  const int tid = threadIdx.x;
  const int rr = tid / cpr, cc = tid % cpr;
  const bool active = rr < rpi;
  // lds[rr][stat][C]
  if (active) {
#pragma unroll
    for (int s = 0; s < NSTAT; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) lds[(rr * NSTAT + s) * C + cc * 8 + j] = acc[s][j];
  }
  __syncthreads();
  const int ncol = NSTAT * C;
  if (ncol <= 128 && rpi >= 16) {
    // two levels: 256/ncol threads per column add a slice of the rows each, then one thread per column adds the
    // slices (one level would be rpi additions one after the other by ncol threads: 256 deep when C = 8)
    const int nseg = 256 / ncol, col = tid % ncol, seg = tid / ncol;
    float v = 0.f;
    if (seg < nseg)
      for (int r = seg; r < rpi; r += nseg) v += lds[r * ncol + col];
    __syncthreads();
    if (seg < nseg) lds[seg * ncol + col] = v;
    __syncthreads();
    if (tid < ncol) {
      float t = 0.f;
      for (int q = 0; q < nseg; ++q) t += lds[q * ncol + tid];
      out_row[tid] = t;
    }
    return;
  }
  for (int i = tid; i < ncol; i += 256) {
    float v = 0.f;
    for (int r = 0; r < rpi; ++r) v += lds[r * ncol + i];
    out_row[i] = v;
  }
}

// ---- partial-row reductions: partial[P][ncols] -> per-column double sums ------------------------
// one workgroup of 1024 threads handles 8 channels; thread (c8 = t&7, slot = t>>3) walks rows.
template <int NSTAT>
__device__ __forceinline__ void reduce_partials(const float* __restrict__ partial, int P, int C, int c0,
                                                double (&out)[NSTAT], double* lds /*[128][NSTAT][8]*/) {
  const int t = threadIdx.x, c8 = t & 7, slot = t >> 3;
  double s[NSTAT];
#pragma unroll
  for (int k = 0; k < NSTAT; ++k) s[k] = 0.0;
  if (c0 + c8 < C) {
    // (Measured: fetching eight rows before adding any leaves the kernel's time unchanged, so the chain of loads is
    // not what it waits for.  The launches that take 20-28 us instead of 5 are those right behind a layer that wrote
    // hundreds of MB, with 8 rows per thread as with 33.)
    for (int p = slot; p < P; p += 128) {
#pragma unroll
      for (int k = 0; k < NSTAT; ++k) s[k] += (double)partial[((long long)p * NSTAT + k) * C + c0 + c8];
    }
  }
#pragma unroll
  for (int k = 0; k < NSTAT; ++k) lds[(slot * NSTAT + k) * 8 + c8] = s[k];
  __syncthreads();
  for (int stride = 64; stride > 0; stride >>= 1) {
    if (slot < stride) {
#pragma unroll
      for (int k = 0; k < NSTAT; ++k) lds[(slot * NSTAT + k) * 8 + c8] += lds[((slot + stride) * NSTAT + k) * 8 + c8];
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < NSTAT; ++k) out[k] = lds[k * 8 + c8];
}

