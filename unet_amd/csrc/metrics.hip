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

// ---- input pipeline (trainer/training_utils.py:52-84) -------------------------------------------------------------
// The reference turns the whole data set into fp32 NCHW on the host (FloatTensor(images).permute(0, 3, 1, 2)) and
// lets three DataLoader workers collate shuffled batches of it.  Here the data set stays on the device as it was
// loaded -- NHWC, one byte per value when every value is k/255 for an integer k (what the loaders produce from 8-bit
// files: data_helpers/utils.py:58-62), fp32 NHWC otherwise -- and one launch gathers a batch by index, converts and
// transposes it to the fp32 NCHW tensor the model takes.  HBM-bound: C bytes (or 4C) read, 4C bytes written per pixel.
// out[i][c][p] = lut[data[idx[i]][p][c]]  (u8)   |   data[idx[i]][p][c]  (f32);   p = y*W + x
template <int C, bool U8>
__global__ __launch_bounds__(256) void gather_batch_kernel(int n_items, long long hw, const void* __restrict__ data,
                                                           const long long* __restrict__ idx,
                                                           const float* __restrict__ lut, float* __restrict__ out) {
  // one thread: four consecutive pixels of one image (hw % 4 == 0): 4*C input values -> C float4 stores
  const long long q = hw >> 2;
  const long long total = (long long)gridDim.y * q;
  (void)total;
  const int i = blockIdx.y;
  const long long item = idx[i];
  if (item < 0 || item >= n_items) return;                  // the host checks its own indices; never read outside
  for (long long p4 = (long long)blockIdx.x * 256 + threadIdx.x; p4 < q; p4 += (long long)gridDim.x * 256) {
    float v[4 * C];
    if (U8) {
      const uint32_t* src = (const uint32_t*)((const uint8_t*)data + (item * hw + p4 * 4) * C);
      uint32_t w[C];
#pragma unroll
      for (int k = 0; k < C; ++k) w[k] = src[k];
#pragma unroll
      for (int k = 0; k < 4 * C; ++k) v[k] = lut[(w[k >> 2] >> (8 * (k & 3))) & 0xffu];
    } else {
      const float4* src = (const float4*)((const float*)data + (item * hw + p4 * 4) * C);
#pragma unroll
      for (int k = 0; k < C; ++k) {
        const float4 t = src[k];
        v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c)
      *(float4*)(out + ((long long)i * C + c) * hw + p4 * 4) = float4{v[c], v[C + c], v[2 * C + c], v[3 * C + c]};
  }
}

// any channel count / plane size: one thread per output value
template <bool U8>
__global__ __launch_bounds__(256) void gather_batch_generic_kernel(int n_items, long long hw, int C,
                                                                   const void* __restrict__ data,
                                                                   const long long* __restrict__ idx,
                                                                   const float* __restrict__ lut,
                                                                   float* __restrict__ out) {
  const int i = blockIdx.y;
  const long long item = idx[i];
  if (item < 0 || item >= n_items) return;
  const long long per = hw * C;
  for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < per; o += (long long)gridDim.x * 256) {
    const long long c = o / hw, p = o - c * hw;
    const long long src = (item * hw + p) * C + c;
    out[(long long)i * per + o] = U8 ? lut[((const uint8_t*)data)[src]] : ((const float*)data)[src];
  }
}

template <bool U8>
static int gather_batch_launch(int b, int n_items, int H, int W, int C, const void* data, const long long* idx,
                               const float* lut, float* out, uh_stream s) {
  UH_REQUIRE(b > 0 && b <= 65535 && n_items > 0 && H > 0 && W > 0 && C > 0, "gather_batch: bad sizes b=%d n=%d %dx%dx%d",
             b, n_items, H, W, C);
  UH_REQUIRE(data && idx && out && (!U8 || lut), "gather_batch: null pointer");
  const long long hw = (long long)H * W;
  const bool fast = hw % 4 == 0 && C <= 4 && ((uintptr_t)data & 15) == 0 && ((uintptr_t)out & 15) == 0;
  long long work = fast ? hw / 4 : hw * C;
  long long bx = (work + 255) / 256;
  if (bx > 4096) bx = 4096;
  const dim3 grid((unsigned)bx, (unsigned)b);
  hipStream_t st = (hipStream_t)s;
#define UH_GATHER(CC) hipLaunchKernelGGL((gather_batch_kernel<CC, U8>), grid, dim3(256), 0, st, n_items, hw, data, idx, lut, out)
  if (fast && C == 1) UH_GATHER(1);
  else if (fast && C == 2) UH_GATHER(2);
  else if (fast && C == 3) UH_GATHER(3);
  else if (fast && C == 4) UH_GATHER(4);
  else
    hipLaunchKernelGGL((gather_batch_generic_kernel<U8>), grid, dim3(256), 0, st, n_items, hw, C, data, idx, lut, out);
#undef UH_GATHER
  return uh_check_launch("gather_batch");
}

extern "C" int uh_gather_batch_u8(int b, int n_items, int H, int W, int C, const uint8_t* data, const long long* idx,
                                  const float* lut, float* out, uh_stream s) {
  return gather_batch_launch<true>(b, n_items, H, W, C, data, idx, lut, out, s);
}
extern "C" int uh_gather_batch_f32(int b, int n_items, int H, int W, int C, const float* data, const long long* idx,
                                   float* out, uh_stream s) {
  return gather_batch_launch<false>(b, n_items, H, W, C, data, idx, nullptr, out, s);
}

// ---- the same loss for callers that cannot allocate ------------------------------------------------------------
// uh_bce_logits needs a workspace the caller allocates.  Code under HIP graph capture, or a serving loop that must
// not touch the allocator, has nowhere to get one; for it the per-workgroup sums live in `scratch`, an array of this
// code object (one copy per GPU, held until the process ends).  On the host the array's NAME is only a handle: the
// device address comes from hipGetSymbolAddress, once per device with that device current, and only that pointer is
// handed to the kernels.  Nothing has to be cleared between calls: a call writes partial[0 .. blocks) and reads exactly
// those.  All calls on one GPU share the array, so calls that overlap in time on one GPU (two streams) race.
#define UH_BCE_NOALLOC_PARTIALS 1024                       /* bce_blocks() never exceeds it */
#define UH_BCE_NOALLOC_MAX_N (1ll << 32)
#define UH_MAX_DEVICES 64
static __device__ __align__(256) float scratch[UH_BCE_NOALLOC_PARTIALS];

#include <mutex>
static std::mutex g_scratch_mutex;
static float* g_scratch_ptr[UH_MAX_DEVICES];               // device address of `scratch`, per device

extern "C" int uh_bce_logits_noalloc(long long n, const float* logits, const float* targets, float grad_scale,
                                     float* loss, float* dlogits, uh_stream s) {
  UH_REQUIRE(n > 0 && logits && targets && loss, "uh_bce_logits_noalloc: n=%lld or a null pointer", n);
  UH_REQUIRE(n <= UH_BCE_NOALLOC_MAX_N, "uh_bce_logits_noalloc: n=%lld exceeds the maximum of %lld elements", n,
             (long long)UH_BCE_NOALLOC_MAX_N);
  const int blocks = bce_blocks(n);
  UH_REQUIRE(blocks <= UH_BCE_NOALLOC_PARTIALS, "uh_bce_logits_noalloc: %d partial sums exceed the scratch array of %d",
             blocks, UH_BCE_NOALLOC_PARTIALS);
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= UH_MAX_DEVICES) {
    uh_set_error("uh_bce_logits_noalloc: no current device (or device index above %d)", UH_MAX_DEVICES - 1);
    return UH_ERR_HIP;
  }
  float* partial;
  {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {                             // first call on this device
      void* ptr = nullptr;
      const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
      if (e != hipSuccess || !ptr) {
        uh_set_error("uh_bce_logits_noalloc: hipGetSymbolAddress(scratch) failed on device %d: %s", dev,
                     hipGetErrorString(e));
        return UH_ERR_HIP;
      }
      g_scratch_ptr[dev] = (float*)ptr;
    }
    partial = g_scratch_ptr[dev];
  }
  hipLaunchKernelGGL(bce_logits_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, n, logits, targets,
                     (float)((double)grad_scale / (double)n), dlogits, partial);
  hipLaunchKernelGGL(bce_logits_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, blocks, 1.0 / (double)n,
                     (const float*)partial, loss);
  return uh_check_launch("uh_bce_logits_noalloc");
}

// ---- sigmoid(outputs).clamp(lo, hi) in front of the clDice loss (trainer/training_utils.py:146,162,201) ------------
// forward: probs = min(max(1 / (1 + exp(-x)), lo), hi); backward: dlogits = dprobs * s * (1 - s) where lo <= s <= hi
// (s the unclamped sigmoid; torch's clamp passes the gradient on the closed interval), else 0.  One pass each instead
// of two / four element-wise launches; fp32, 8 B per element forward, 12 B backward.
__global__ __launch_bounds__(256) void sigmoid_clamp_kernel(long long n, const float* __restrict__ x, float lo, float hi,
                                                            float* __restrict__ p) {
  const long long n4 = n >> 2, stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 o;
    o.x = fminf(fmaxf(1.0f / (1.0f + expf(-v.x)), lo), hi);
    o.y = fminf(fmaxf(1.0f / (1.0f + expf(-v.y)), lo), hi);
    o.z = fminf(fmaxf(1.0f / (1.0f + expf(-v.z)), lo), hi);
    o.w = fminf(fmaxf(1.0f / (1.0f + expf(-v.w)), lo), hi);
    reinterpret_cast<float4*>(p)[i] = o;
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    p[i] = fminf(fmaxf(1.0f / (1.0f + expf(-x[i])), lo), hi);
}

__global__ __launch_bounds__(256) void sigmoid_clamp_bwd_kernel(long long n, const float* __restrict__ x, float lo,
                                                                float hi, const float* __restrict__ dp,
                                                                float* __restrict__ dx) {
  auto one = [&](float xv, float g) {
    const float s = 1.0f / (1.0f + expf(-xv));
    return (s >= lo && s <= hi) ? g * ((1.0f - s) * s) : 0.f;       // torch: grad * (1 - y) * y
  };
  const long long n4 = n >> 2, stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 g = reinterpret_cast<const float4*>(dp)[i];
    float4 o;
    o.x = one(v.x, g.x); o.y = one(v.y, g.y); o.z = one(v.z, g.z); o.w = one(v.w, g.w);
    reinterpret_cast<float4*>(dx)[i] = o;
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dx[i] = one(x[i], dp[i]);
}

static int sc_blocks(long long n) {
  long long b = (n / 4 + 255) / 256;
  if (b > 4096) b = 4096;
  return (int)(b < 1 ? 1 : b);
}

extern "C" int uh_sigmoid_clamp(long long n, const float* logits, float lo, float hi, float* probs, uh_stream s) {
  UH_REQUIRE(n > 0 && logits && probs && lo <= hi, "sigmoid_clamp: bad arguments");
  UH_REQUIRE((((uintptr_t)logits | (uintptr_t)probs) & 15) == 0, "sigmoid_clamp: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(sigmoid_clamp_kernel, dim3(sc_blocks(n)), dim3(256), 0, (hipStream_t)s, n, logits, lo, hi, probs);
  return uh_check_launch("sigmoid_clamp");
}

extern "C" int uh_sigmoid_clamp_bwd(long long n, const float* logits, float lo, float hi, const float* dprobs,
                                    float* dlogits, uh_stream s) {
  UH_REQUIRE(n > 0 && logits && dprobs && dlogits && lo <= hi, "sigmoid_clamp_bwd: bad arguments");
  UH_REQUIRE((((uintptr_t)logits | (uintptr_t)dprobs | (uintptr_t)dlogits) & 15) == 0,
             "sigmoid_clamp_bwd: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(sigmoid_clamp_bwd_kernel, dim3(sc_blocks(n)), dim3(256), 0, (hipStream_t)s, n, logits, lo, hi,
                     dprobs, dlogits);
  return uh_check_launch("sigmoid_clamp_bwd");
}
