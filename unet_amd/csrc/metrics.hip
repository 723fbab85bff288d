// Evaluation path: binarise fp32 logits (sigmoid, then > 0.5) and count the confusion matrix against a mask,
// per plane, without the predictions ever leaving the device (trainer/k_fold.py:289-290,371-380;
// test_helper/utils.py:130-131).  HBM-bound: 8 bytes per pixel read, 32 bytes per plane written.
#include "common.h"

// counts[plane][4] = tp, tn, fp, fn.  Integer adds commute, so the atomics leave the result run-independent.
template <bool VEC4>
__global__ __launch_bounds__(256) void confusion_kernel(long long n, const float* __restrict__ logits,
                                                        const float* __restrict__ mask,
                                                        unsigned long long* __restrict__ counts) {
  const long long plane = blockIdx.y;
  const float* lg = logits + plane * n;
  const float* mk = mask + plane * n;
  unsigned c[4] = {0u, 0u, 0u, 0u};
  auto tally = [&](float x, float m) {
    // the reference's rule, evaluated in fp32 like it: sigmoid(0) == 0.5 is background, NaN is background
    const bool p = 1.0f / (1.0f + expf(-x)) > 0.5f;
    const bool t1 = m == 1.0f, t0 = m == 0.0f;            // any other mask value falls into no class
    c[0] += (p && t1);
    c[1] += (!p && t0);
    c[2] += (p && t0);
    c[3] += (!p && t1);
  };
  const long long stride = (long long)gridDim.x * 256;
  if (VEC4) {
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const float4 x = reinterpret_cast<const float4*>(lg)[i];
      const float4 m = reinterpret_cast<const float4*>(mk)[i];
      tally(x.x, m.x); tally(x.y, m.y); tally(x.z, m.z); tally(x.w, m.w);
    }
  } else {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) tally(lg[i], mk[i]);
  }
  __shared__ unsigned red[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned v = c[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const unsigned long long v = (unsigned long long)red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] +
                                 red[3][threadIdx.x];
    if (v) atomicAdd(&counts[plane * 4 + threadIdx.x], v);
  }
}

extern "C" int uh_confusion_counts(int planes, long long n, const float* logits, const float* mask, long long* counts,
                                   uh_stream s) {
  UH_REQUIRE(planes > 0 && planes <= 65535 && n > 0, "uh_confusion_counts: planes=%d n=%lld", planes, n);
  UH_REQUIRE(n < (1ll << 40), "uh_confusion_counts: n=%lld too large", n);
  const bool vec4 = n % 4 == 0 && ((uintptr_t)logits | (uintptr_t)mask) % 16 == 0;
  // a thread tallies in 32 bits: keep its share far below 2^32; fill the chip when there are few planes
  long long per_plane = (n / (vec4 ? 4 : 1) + 256 * 16 - 1) / (256 * 16);
  const long long want = (2048 + planes - 1) / planes;
  if (per_plane > want) per_plane = want;
  if (per_plane < 1) per_plane = 1;
  const dim3 grid((unsigned)per_plane, (unsigned)planes);
  unsigned long long* c = reinterpret_cast<unsigned long long*>(counts);
  if (vec4)
    hipLaunchKernelGGL(confusion_kernel<true>, grid, dim3(256), 0, (hipStream_t)s, n, logits, mask, c);
  else
    hipLaunchKernelGGL(confusion_kernel<false>, grid, dim3(256), 0, (hipStream_t)s, n, logits, mask, c);
  return uh_check_launch("uh_confusion_counts");
}

// ---- nn.BCEWithLogitsLoss() (mean reduction; trainer/training_utils.py:105) forward and backward in one pass ------
//   loss = mean_i( max(x,0) - x t + log(1 + exp(-|x|)) ),   dlogits_i = grad_scale * (sigmoid(x_i) - t_i) / n
// partial[blocks] holds per-workgroup sums; the second kernel adds them in a fixed order.
static inline int bce_blocks(long long n) {
  long long b = (n + 256 * 8 - 1) / (256 * 8);
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}
extern "C" size_t uh_bce_logits_workspace(long long n) { return n > 0 ? (size_t)bce_blocks(n) * sizeof(float) : 0; }

__global__ __launch_bounds__(256) void bce_logits_kernel(long long n, const float* __restrict__ x,
                                                         const float* __restrict__ t, float gscale,
                                                         float* __restrict__ dx, float* __restrict__ partial) {
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = x[i], y = t[i];
    const float e = expf(-fabsf(v));
    acc += fmaxf(v, 0.f) - v * y + log1pf(e);
    const float sg = v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    if (dx) dx[i] = gscale * (sg - y);
  }
  __shared__ float red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void bce_logits_finish_kernel(int blocks, double inv_n, const float* __restrict__ partial,
                                                                float* __restrict__ loss) {
  __shared__ double red[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < blocks; i += 256) a += (double)partial[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(red[0] * inv_n);
}

extern "C" int uh_bce_logits(long long n, const float* logits, const float* targets, float grad_scale, float* loss,
                             float* dlogits, void* workspace, size_t workspace_bytes, uh_stream s) {
  UH_REQUIRE(n > 0 && logits && targets && loss && workspace, "uh_bce_logits: n=%lld", n);
  UH_REQUIRE(workspace_bytes >= uh_bce_logits_workspace(n), "uh_bce_logits: workspace of %zu bytes, need %zu",
             workspace_bytes, uh_bce_logits_workspace(n));
  const int blocks = bce_blocks(n);
  hipLaunchKernelGGL(bce_logits_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, n, logits, targets,
                     (float)((double)grad_scale / (double)n), dlogits, (float*)workspace);
  hipLaunchKernelGGL(bce_logits_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, blocks, 1.0 / (double)n,
                     (const float*)workspace, loss);
  return uh_check_launch("uh_bce_logits");
}
