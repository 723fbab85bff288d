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
// device-scope stores / loads that go past this XCD's L2 (each of the eight XCDs has its own, and a plain store stays
// there until the kernel ends): what a workgroup on another XCD may read within the same launch is written and read
// with these, with no cache-wide write-back or invalidate
__device__ __forceinline__ void st_dev(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_dev(const float* p) {
  return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_dev(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_dev(const double* p) {
  return __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// COHERENT: the row is written with device-scope stores (another XCD's workgroup reads it within this launch)
template <int NSTAT, bool COHERENT = false>
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
      if (COHERENT) st_dev(out_row + tid, t);
      else out_row[tid] = t;
    }
    return;
  }
  for (int i = tid; i < ncol; i += 256) {
    float v = 0.f;
    for (int r = 0; r < rpi; ++r) v += lds[r * ncol + i];
    if (COHERENT) st_dev(out_row + i, v);
    else out_row[i] = v;
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
    // Tables of thousands of rows (the strip-walk kernels of lunext.hip write one per strip and column block: 8683 at
    // 8 channels, 256 x 256, batch 32) make this a chain of 68 round trips per thread: eight rows are fetched before
    // any is added (same order of additions), 24 -> 9 us there.
    const float* src = partial + c0 + c8;
    int p = slot;
    // (only for such tables: with the U-Net's <= 1024 rows the plain loop is the faster one, 12.80 against 12.82 ms
    //  per step)
    for (; P >= 2048 && p + 7 * 128 < P; p += 8 * 128) {
      float v[8][NSTAT];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < NSTAT; ++k) v[u][k] = src[((long long)(p + u * 128) * NSTAT + k) * C];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < NSTAT; ++k) s[k] += (double)v[u][k];
    }
    for (; p < P; p += 128) {
#pragma unroll
      for (int k = 0; k < NSTAT; ++k) s[k] += (double)src[((long long)p * NSTAT + k) * C];
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


// ---- finalisation inside the kernel that makes the partial rows -------------------------------------------------------
// The kernels that end in block_combine_store leave one partial row per workgroup and used to be followed by a
// one-workgroup-per-8-channels launch that adds the rows up (5-20 us, plus the gap between two dependent launches).
// Instead the LAST workgroups to finish do it: workgroups form groups of FIN_GS; the last of a group to arrive adds the
// group's rows (fp64, in row order) into a group row, the last group to be finished adds the group rows (in group order)
// and evaluates the per-channel result.  Who does the adding depends on timing, what is added in which order does not:
// results are run-independent.  Rows and group rows are written and read with device-scope accesses (st_dev / ld_dev),
// a workgroup counts itself in only after its own stores have been acknowledged (s_waitcnt vmcnt(0)), and arrival is
// counted with relaxed device-scope atomics.  (A release / acquire fence pair instead writes back and invalidates the
// whole L2 of the XCD once per workgroup: measured on the training steps, 13.1 -> 17.8 ms and 5.2 -> 10.1 ms.)
// `counters` (1 + number of groups, zero before the launch) are left zero.
#define FIN_GS 32
struct FinWs {
  unsigned* counters;     // null: no finalisation in this launch (the caller runs the separate one)
  double* grp;            // [groups][NSTAT * C]
};
static inline int fin_groups(int P) { return (P + FIN_GS - 1) / FIN_GS; }
#define FIN_MAX_COUNTERS 1024      // a caller-side constant: 1 + groups never exceeds it for the grids used here
// workspace of such a launch: the group rows (fp64), then the partial rows (fp32)
static inline size_t fin_ws_bytes(int parts, int C, int nstat) {
  if (parts <= 0 || C <= 0 || nstat <= 0) return 0;
  return (size_t)fin_groups(parts) * nstat * C * sizeof(double) + (size_t)parts * nstat * C * sizeof(float);
}
static inline bool fin_split(int parts, int C, int nstat, void* ws, size_t ws_bytes, unsigned* counters, FinWs& out,
                             float*& partial) {
  if (!ws || !counters || ws_bytes < fin_ws_bytes(parts, C, nstat) || ((uintptr_t)ws & 7) ||
      1 + fin_groups(parts) > FIN_MAX_COUNTERS)
    return false;
  out.counters = counters;
  out.grp = (double*)ws;
  partial = (float*)((char*)ws + (size_t)fin_groups(parts) * nstat * C * sizeof(double));
  return true;
}

__device__ __forceinline__ unsigned fin_count_in(unsigned* c) {
  return __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the workgroup's row of `partial` must have been written with block_combine_store<NSTAT, true>
template <int NSTAT, class F>
__device__ __forceinline__ void last_arriver_finish(const float* partial, int C, const FinWs& ws, F&& fin) {
  __shared__ int fin_flag;
  const int tid = threadIdx.x, P = gridDim.x, g = blockIdx.x / FIN_GS, G = (P + FIN_GS - 1) / FIN_GS;
  const int ncol = NSTAT * C;
  const int first = g * FIN_GS, members = (P - first < FIN_GS) ? P - first : FIN_GS;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's part of the row has reached the device's memory side
  __syncthreads();                                   // ... and everybody else's
  if (tid == 0) fin_flag = (fin_count_in(&ws.counters[1 + g]) == (unsigned)members - 1);
  __syncthreads();
  if (!fin_flag) return;
  for (int col = tid; col < ncol; col += blockDim.x) {
    // all of the group's rows requested before any is added: FIN_GS loads in flight, not FIN_GS round trips in a row
    float v[FIN_GS];
#pragma unroll
    for (int r = 0; r < FIN_GS; ++r) v[r] = ld_dev(partial + (long long)(first + (r < members ? r : 0)) * ncol + col);
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < FIN_GS; ++r) s += (r < members) ? (double)v[r] : 0.0;
    st_dev(ws.grp + (long long)g * ncol + col, s);
  }
  if (tid == 0) __hip_atomic_store(&ws.counters[1 + g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) fin_flag = (fin_count_in(&ws.counters[0]) == (unsigned)G - 1);
  __syncthreads();
  if (!fin_flag) return;
  for (int c = tid; c < C; c += blockDim.x) {
    double s[NSTAT];
#pragma unroll
    for (int k = 0; k < NSTAT; ++k) s[k] = 0.0;
    for (int q0 = 0; q0 < G; q0 += 16) {                  // sixteen group rows in flight at a time
      double v[16][NSTAT];
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int k = 0; k < NSTAT; ++k) v[i][k] = ld_dev(ws.grp + (long long)(q0 + i < G ? q0 + i : 0) * ncol + k * C + c);
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int k = 0; k < NSTAT; ++k) s[k] += (q0 + i < G) ? v[i][k] : 0.0;
    }
    fin(c, s);
  }
  if (tid == 0) __hip_atomic_store(&ws.counters[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// what bn_finalize_kernel computes per channel from (sum y, sum y^2)
struct BnFwdFin {
  FinWs ws;
  double count;
  const float *gamma, *beta;
  float eps, momentum;
  float *mean, *invstd, *scale, *shift, *rm, *rv;
  long long* nbt;
  int c_real;
  __device__ __forceinline__ void operator()(int c, const double* s) const {
    const double m = s[0] / count;
    double var = s[1] / count - m * m;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    const bool real = c < c_real;
    const double sc = real ? (double)gamma[c] * is : 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)is;
    scale[c] = (float)sc;
    shift[c] = real ? (float)((double)beta[c] - m * sc) : 0.f;
    if (rm && real) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      rm[c] = (float)((1.0 - momentum) * rm[c] + momentum * m);
      rv[c] = (float)((1.0 - momentum) * rv[c] + momentum * unbiased);
    }
    if (nbt && c == 0) *nbt += 1;
  }
};

// what bn_bwd_finalize_kernel computes per channel from (sum d, sum d * xhat)
struct BnBwdFin {
  FinWs ws;
  double count;
  float *dgamma, *dbeta, *coef;
  int c_real, C;
  __device__ __forceinline__ void operator()(int c, const double* s) const {
    if (c < c_real) {
      dbeta[c] = (float)s[0];
      dgamma[c] = (float)s[1];
    }
    coef[c] = (float)(s[0] / count);
    coef[C + c] = (float)(s[1] / count);
  }
};
