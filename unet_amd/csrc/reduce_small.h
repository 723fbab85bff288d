// The small reductions behind weight gradients and column sums (a table of partial rows -> a gradient tensor), as
// device functions over (block index, shared memory): the single-table launches of conv_wgrad.hip / bn_act.hip and the
// many-tables launch of reduce_multi.hip run the same bodies, so their results agree bit for bit where the thread
// geometry is the same.
#pragma once
#include "common.h"

// partial[split][n*on + t*ot + k] summed over the S splits -> grad[n*sn + t*st + k*sk].  SL threads share an output
// element and take every SL-th split (many splits of a small tensor would otherwise be one long serial loop per
// thread).  256 threads; red: SL * (256 / SL) floats.
template <int SL>
__device__ __forceinline__ void wgrad_reduce_body(long long blk, float* red, const float* __restrict__ partial, int S,
                                                  long long split_stride, int N, int T, int K, long long on,
                                                  long long ot, long long sn, long long st, long long sk,
                                                  float* __restrict__ grad, int accumulate) {
  constexpr int EPB = 256 / SL;
  const long long total = (long long)N * T * K;
  const int e = threadIdx.x % EPB, sl = threadIdx.x / EPB;
  const long long i = blk * EPB + e;
  int k = 0, t = 0, n = 0;
  float v = 0.f;
  if (i < total) {
    k = (int)(i % K);
    const long long r = i / K;
    t = (int)(r % T);
    n = (int)(r / T);
    const float* src = partial + n * on + t * ot + k;
    for (int sp = sl; sp < S; sp += SL) v += src[sp * split_stride];
  }
  red[sl * EPB + e] = v;
  __syncthreads();
  if (sl == 0 && i < total) {
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < SL; ++j) sum += red[j * EPB + e];
    const long long dst = n * sn + t * st + k * sk;
    grad[dst] = accumulate ? grad[dst] + sum : sum;
  }
}

// Transposed-conv weights, partial [tap][co][k] -> grad [k][co][tap] (T = 4): one block sums a
// (64 k) x (4 co) x (4 taps) piece and writes 64-byte runs instead of single floats one row apart.
// 256 threads; red: 4 * 64 * 17 floats.
__device__ __forceinline__ void wgrad_reduce_convt_body(int bx, int by, float* red, const float* __restrict__ partial,
                                                        int S, long long split_stride, int Cout, int K,
                                                        float* __restrict__ grad, int accumulate) {
  const int k0 = bx * 64, co0 = by * 4;
  const int kk = threadIdx.x & 63, sl = threadIdx.x >> 6;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  for (int sp = sl; sp < S; sp += 4) {
    const float* q = partial + sp * split_stride + k0 + kk;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[c * 4 + t] += q[((long long)t * Cout + co0 + c) * K];
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) red[(sl * 64 + kk) * 17 + j] = acc[j];
  __syncthreads();
  for (int o = threadIdx.x; o < 64 * 16; o += 256) {
    const int r = o >> 4, j = o & 15;
    const float v = (red[(0 * 64 + r) * 17 + j] + red[(1 * 64 + r) * 17 + j]) +
                    (red[(2 * 64 + r) * 17 + j] + red[(3 * 64 + r) * 17 + j]);
    float* dst = grad + ((long long)(k0 + r) * Cout + co0) * 4 + j;
    *dst = accumulate ? *dst + v : v;
  }
}
#define UH_WGRAD_REDUCE_CONVT_FLOATS (4 * 64 * 17)

// which body uh_wgrad_reduce_strided runs for a shape it does not hand to the 3x3 kernels, and how many blocks
enum { UH_RED_GENERIC32 = 0, UH_RED_GENERIC4 = 1, UH_RED_CONVT = 2, UH_RED_OIHW9 = 3 };
static inline int wgrad_reduce_route(int S, int N, int T, int K, long long on, long long ot, long long sn, long long st,
                                     long long sk, long long* blocks) {
  if (T == 9 && K % 64 == 0 && on == 9LL * K && ot == K && sn == 9LL * K && st == 1 && sk == 9 && N <= 65535) return UH_RED_OIHW9;
  if (T == 4 && K % 64 == 0 && N % 4 == 0 && on == K && ot == (long long)N * K && sn == 4 && st == 1 &&
      sk == 4LL * N && N / 4 <= 65535) {
    *blocks = (long long)(K / 64) * (N / 4);
    return UH_RED_CONVT;
  }
  const long long total = (long long)N * T * K;
  if (S >= 64 && total <= 65536) {
    *blocks = (total + 7) / 8;
    return UH_RED_GENERIC32;
  }
  *blocks = (total + 63) / 64;
  return UH_RED_GENERIC4;
}

// out[i] = sum_p partial[p][cols[i]] for i < n_out (cols == nullptr: column i), 8 outputs and 32 row slots per block of
// 256 threads, double accumulators; lds: 256 doubles
__device__ __forceinline__ void colsum_gather_body(int blk, double* lds, const float* __restrict__ partial, int P, int C,
                                                   int n_out, const int* __restrict__ cols, float* __restrict__ out,
                                                   int accumulate) {
  const int i = blk * 8 + (threadIdx.x & 7), slot = threadIdx.x >> 3;
  double s = 0.0;
  if (i < n_out) {
    const int c = cols ? cols[i] : i;
    const float* src = partial + c;
    int p = slot;
    for (; p + 7 * 32 < P; p += 8 * 32) {                     // eight rows in flight, added in row order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(long long)(p + u * 32) * C];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; p < P; p += 32) s += (double)src[(long long)p * C];
  }
  lds[threadIdx.x] = s;
  __syncthreads();
  for (int stride = 16; stride > 0; stride >>= 1) {
    if (slot < stride) lds[threadIdx.x] += lds[threadIdx.x + stride * 8];
    __syncthreads();
  }
  if (threadIdx.x < 8 && i < n_out) {
    const float v = (float)lds[threadIdx.x];
    out[i] = accumulate ? out[i] + v : v;
  }
}
