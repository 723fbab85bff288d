// Error plumbing, geometry helpers, weight packing, row masks, layout shuffles (space<->depth).
#include <stdarg.h>
#include <math.h>
#include <stdio.h>
#include <mutex>
#include <set>
#include <utility>
#include "common.h"

static thread_local char g_err[512] = "";

void uh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int uh_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    uh_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return UH_ERR_HIP;
  }
  return UH_OK;
}

void uh_set_max_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  if (done.insert(std::make_pair(kernel, dev)).second)
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

extern "C" const char* uh_last_error(void) { return g_err; }
extern "C" int uh_version(void) { return 1; }
// (W+1)+1 rows cover the 3x3 neighbourhood of the first/last row; 512+16 more cover the overhang
// of the last M tile (tiles are at most 512 rows, bands 16 rows longer) and of the last wgrad stage.
extern "C" int uh_guard_rows(int W) { return W + 2 + 512 + 32; }

__global__ void rowmask_kernel(int H, int W, long long Mp, uint8_t* mask) {
  const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= Mp) return;
  const int Wp = W + 1, Hp = H + 1;
  const int x = (int)(m % Wp);
  const int y = (int)((m / Wp) % Hp);
  mask[m] = (x < W && y < H) ? 1 : 0;
}

extern "C" int uh_rowmask_build(int B, int H, int W, uint8_t* rowmask, uh_stream s) {
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && rowmask, "rowmask_build: bad arguments");
  const long long Mp = uh_mp(B, H, W);
  hipLaunchKernelGGL(rowmask_kernel, dim3((unsigned)((Mp + 255) / 256)), dim3(256), 0, (hipStream_t)s, H, W, Mp, rowmask);
  return uh_check_launch("rowmask_build");
}

// dst[n*on + t*ot + k] = bf16(src[n*sn + tmap(t)*st + k*sk])
__global__ void pack_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int N, int T, int K, long long sn,
                            long long st, long long sk, long long on, long long ot, int reverse) {
  const long long total = (long long)N * T * K;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    const long long r = i / K;
    const int t = (int)(r % T);
    const int n = (int)(r / T);
    const int ts = reverse ? (T - 1 - t) : t;
    dst[n * on + t * ot + k] = f2bf(src[n * sn + ts * st + k * sk]);
  }
}

extern "C" int uh_pack_weights(const float* src, void* dst, int N, int T, int K, long long sn, long long st,
                               long long sk, long long on, long long ot, int reverse, uh_stream s) {
  UH_REQUIRE(src && dst && N > 0 && T > 0 && K > 0, "pack_weights: bad arguments");
  const long long total = (long long)N * T * K;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, src, (bf16*)dst, N, T, K, sn, st, sk, on,
                     ot, reverse);
  return uh_check_launch("pack_weights");
}

// Many repacks in one launch: blockIdx.y selects the descriptor (device memory), blockIdx.x strides over its
// work.  A training step repacks every MFMA weight operand of the network twice (forward and rotated for the
// input gradient); one launch per direction replaces ~20 launches.
// The repack is a transpose: in the source the taps of one (n, k) pair are contiguous, in the destination k is.
// Tensors with N and K multiples of 32 go through 32 x 32 x T tiles in LDS, read in source order (runs of 32*T
// floats) and written in destination order (runs of 32 bf16); anything else takes the element-wise loop.
// reverse & 2: the destination holds fp32 (small zero-padded parameter tables of vector kernels, made in the same launch).
__global__ __launch_bounds__(256) void pack_multi_kernel(const uh_pack_desc* __restrict__ descs) {
  __shared__ float tile[32 * 9 * 33];
  const uh_pack_desc d = descs[blockIdx.y];
  const int N = (int)d.N, T = (int)d.T, K = (int)d.K;
  const float* __restrict__ src = d.src;
  bf16* __restrict__ dst = (bf16*)d.dst_bf16;
  const bool rev = (d.reverse & 1) != 0, as_f32 = (d.reverse & 2) != 0;
  if (d.st == 1 && T <= 9 && N % 32 == 0 && K % 32 == 0 && !as_f32) {
    const int tk = K / 32, ntiles = (N / 32) * tk;
    const bool k_fast = d.sk < d.sn;               // which of n, k is the faster-varying source dimension
    const int per = 32 * T * 32;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
      const int n0 = (tl / tk) * 32, k0 = (tl % tk) * 32;
      __syncthreads();
      for (int i = threadIdx.x; i < per; i += 256) {
        const int ts = i % T;
        const int r = i / T;
        const int a = r & 31, bq = r >> 5;         // a: faster source dimension
        const int nn = k_fast ? bq : a, kk = k_fast ? a : bq;
        const int t = rev ? (T - 1 - ts) : ts;
        tile[(nn * T + t) * 33 + kk] = src[(n0 + nn) * d.sn + ts + (k0 + kk) * d.sk];
      }
      __syncthreads();
      for (int j = threadIdx.x; j < per; j += 256) {
        const int kk = j & 31;
        const int r = j >> 5;                      // nn * T + t
        const int t = r % T, nn = r / T;
        dst[(n0 + nn) * d.on + t * d.ot + k0 + kk] = f2bf(tile[r * 33 + kk]);
      }
    }
    return;
  }
  const long long total = (long long)N * T * K;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    const long long r = i / K;
    const int t = (int)(r % T);
    const int n = (int)(r / T);
    const int ts = rev ? (T - 1 - t) : t;
    const float v = src[n * d.sn + ts * d.st + k * d.sk];
    if (as_f32) ((float*)d.dst_bf16)[n * d.on + t * d.ot + k] = v;
    else dst[n * d.on + t * d.ot + k] = f2bf(v);
  }
}

extern "C" int uh_pack_weights_multi(const uh_pack_desc* descs_device, int count, int blocks_per_desc, uh_stream s) {
  UH_REQUIRE(descs_device && count > 0 && count <= 65535 && blocks_per_desc > 0, "pack_weights_multi: bad arguments");
  hipLaunchKernelGGL(pack_multi_kernel, dim3(blocks_per_desc, count), dim3(256), 0, (hipStream_t)s, descs_device);
  return uh_check_launch("pack_weights_multi");
}

// ---- space <-> depth for k2 s2 (transposed) convolutions -------------------------------------
// Small geometry (B,H,W); large geometry (B,2H,2W).  One thread moves 8 channels of one
// (small position, tap); every row of the destination, pads included, is written exactly once.
// The large geometry may also be (B, 2H+1, 2W+1): a k2 s2 convolution drops the last row / column of an odd-sized
// input, and their gradient is zero (bigH, bigW are the large geometry's own H and W).
// `add` (depth-to-space only): a tensor of the large geometry that is summed into the result (the gradient of a skip
// connection joining the gradient that comes back through a k2 s2 convolution).
template <bool TO_DEPTH>
__global__ void s2d_kernel(int B, int H, int W, int C, int bigH, int bigW, const char* __restrict__ in, int i_cstride,
                           int i_coff, char* __restrict__ out, int o_cstride, int o_coff,
                           const char* __restrict__ add, int a_cstride, int a_coff) {
  const unsigned Wp = W + 1, Hp = H + 1, oWp = bigW + 1, oHp = bigH + 1;
  const unsigned cpr = C / 8;
  const unsigned total = (unsigned)B * Hp * Wp * cpr;                    // the host checks that this fits
  // one thread moves the four taps of a (row, 8-channel chunk): one decode, four loads in flight
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned cc = i % cpr, ms = i / cpr;
    const unsigned x = ms % Wp, r = ms / Wp, y = r % Hp, b = r / Hp;
    const bool interior = (y < (unsigned)H && x < (unsigned)W);
    U4 v[4];
    long long mb[4];
    bool big_ok[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned Y = 2 * y + (t >> 1), X = 2 * x + (t & 1);
      big_ok[t] = (Y < oHp && X < oWp);
      mb[t] = ((long long)b * oHp + Y) * oWp + X;
      v[t] = U4{0u, 0u, 0u, 0u};
    }
    if (TO_DEPTH) {
      if (interior) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = *(const U4*)(in + (mb[t] * i_cstride + i_coff + cc * 8) * 2);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) *(U4*)(out + ((long long)ms * o_cstride + o_coff + t * C + cc * 8) * 2) = v[t];
    } else {
      if (interior) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = *(const U4*)(in + ((long long)ms * i_cstride + i_coff + t * C + cc * 8) * 2);
      }
      if (add) {                                       // pads of `add` are zero, so the sum's pads stay zero
        U4 av[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
          av[t] = big_ok[t] ? *(const U4*)(add + (mb[t] * a_cstride + a_coff + cc * 8) * 2) : U4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float f[8], h[8];
          unpack8(v[t], f);
          unpack8(av[t], h);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] += h[j];
          v[t] = pack8(f);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (big_ok[t]) *(U4*)(out + (mb[t] * o_cstride + o_coff + cc * 8) * 2) = v[t];
    }
  }
}

static int s2d_launch(bool to_depth, int B, int H, int W, int C, int bigH, int bigW, const void* in, int i_cstride,
                      int i_coff, void* out, int o_cstride, int o_coff, uh_stream s, const void* add = nullptr,
                      int a_cstride = 0, int a_coff = 0) {
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && in && out, "space/depth: bad arguments");
  UH_REQUIRE((bigH == 2 * H || bigH == 2 * H + 1) && (bigW == 2 * W || bigW == 2 * W + 1),
             "space/depth: large geometry %dx%d does not halve to %dx%d", bigH, bigW, H, W);
  UH_REQUIRE(i_cstride % 8 == 0 && i_coff % 8 == 0 && o_cstride % 8 == 0 && o_coff % 8 == 0, "space/depth: bad views");
  const long long total = uh_mp(B, H, W) * (C / 8);
  UH_REQUIRE(total < (1ll << 31), "space/depth: %lld items", total);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  if (to_depth)
    hipLaunchKernelGGL((s2d_kernel<true>), dim3(blocks), dim3(256), 0, (hipStream_t)s, B, H, W, C, bigH, bigW, (const char*)in,
                       i_cstride, i_coff, (char*)out, o_cstride, o_coff, (const char*)nullptr, 0, 0);
  else
    hipLaunchKernelGGL((s2d_kernel<false>), dim3(blocks), dim3(256), 0, (hipStream_t)s, B, H, W, C, bigH, bigW, (const char*)in,
                       i_cstride, i_coff, (char*)out, o_cstride, o_coff, (const char*)add, a_cstride, a_coff);
  return uh_check_launch("space/depth");
}

extern "C" int uh_space_to_depth2(int B, int H, int W, int C, const void* in, int i_cstride, int i_coff, void* out,
                                  int o_cstride, int o_coff, uh_stream s) {
  return s2d_launch(true, B, H, W, C, 2 * H, 2 * W, in, i_cstride, i_coff, out, o_cstride, o_coff, s);
}
extern "C" int uh_space_to_depth2_of(int B, int H, int W, int C, int bigH, int bigW, const void* in, int i_cstride,
                                     int i_coff, void* out, int o_cstride, int o_coff, uh_stream s) {
  return s2d_launch(true, B, H, W, C, bigH, bigW, in, i_cstride, i_coff, out, o_cstride, o_coff, s);
}
extern "C" int uh_depth_to_space2(int B, int H, int W, int C, const void* in, int i_cstride, int i_coff, void* out,
                                  int o_cstride, int o_coff, uh_stream s) {
  return s2d_launch(false, B, H, W, C, 2 * H, 2 * W, in, i_cstride, i_coff, out, o_cstride, o_coff, s);
}
extern "C" int uh_depth_to_space2_to(int B, int H, int W, int C, int bigH, int bigW, const void* in, int i_cstride,
                                     int i_coff, void* out, int o_cstride, int o_coff, uh_stream s) {
  return s2d_launch(false, B, H, W, C, bigH, bigW, in, i_cstride, i_coff, out, o_cstride, o_coff, s);
}
extern "C" int uh_depth_to_space2_add(int B, int H, int W, int C, int bigH, int bigW, const void* in, int i_cstride,
                                      int i_coff, const void* add, int a_cstride, int a_coff, void* out, int o_cstride,
                                      int o_coff, uh_stream s) {
  UH_REQUIRE(add && a_cstride % 8 == 0 && a_coff % 8 == 0, "depth_to_space2_add: bad `add` view");
  return s2d_launch(false, B, H, W, C, bigH, bigW, in, i_cstride, i_coff, out, o_cstride, o_coff, s, add, a_cstride,
                    a_coff);
}

// ---- Adam over one flat parameter buffer (torch.optim.Adam defaults, trainer/training_utils.py:107) ------------
// p, g, m, v: n floats each.  One pass: 16 B read + 12 B written per parameter; the 82 parameter tensors of
// the U-Net are views of one buffer, so this is one launch instead of a multi-tensor sweep.
// AMP = true: the gradient is divided by *grad_scale on the way in, and the whole launch does nothing when *found_inf
// is non-zero (what torch.amp.GradScaler.step asks of an optimizer that takes over unscaling and skipping).
template <bool AMP>
__global__ __launch_bounds__(256) void adam_kernel(long long n4, float4* __restrict__ p, const float4* __restrict__ g,
                                                   float4* __restrict__ m, float4* __restrict__ v, float b1, float b2,
                                                   float step_size, float inv_sqrt_bc2, float eps, float wd,
                                                   const float* __restrict__ grad_scale,
                                                   const float* __restrict__ found_inf,
                                                   const float* __restrict__ step_dev, float lr, float b1d, float b2d) {
  float inv_scale = 1.f;
  if constexpr (AMP) {
    if (found_inf && *found_inf != 0.f) return;
    if (grad_scale) inv_scale = (float)(1.0 / (double)*grad_scale);       // GradScaler's own reciprocal
    if (step_dev) {                        // the step count lives on the device (skipped steps are not counted)
      const double t = (double)*step_dev;
      step_size = (float)((double)lr / (1.0 - pow((double)b1d, t)));
      inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2d, t)));
    }
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
    float* pf = (float*)&pp;
    float* gf = (float*)&gg;
    float* mf = (float*)&mm;
    float* vf = (float*)&vv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gu = AMP ? gf[j] * inv_scale : gf[j];
      const float gr = gu + wd * pf[j];
      mf[j] = mf[j] + (1.f - b1) * (gr - mf[j]);                 // lerp, as torch's fused kernel
      vf[j] = b2 * vf[j] + (1.f - b2) * gr * gr;
      const float denom = sqrtf(vf[j]) * inv_sqrt_bc2 + eps;
      pf[j] -= step_size * (mf[j] / denom);
    }
    p[i] = pp;
    m[i] = mm;
    v[i] = vv;
  }
}

static int adam_launch(long long n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                       float eps, float weight_decay, int step, const float* grad_scale, const float* found_inf,
                       const float* step_dev, bool amp, uh_stream s) {
  UH_REQUIRE(n > 0 && n % 4 == 0 && p && g && m && v && step >= 1, "adam_step: bad arguments (n must be a multiple of 4)");
  UH_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const long long n4 = n / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (amp)
    hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, n4, (float4*)p,
                       (const float4*)g, (float4*)m, (float4*)v, beta1, beta2, (float)(lr / bc1),
                       (float)(1.0 / sqrt(bc2)), eps, weight_decay, grad_scale, found_inf, step_dev, lr, beta1, beta2);
  else
    hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, n4, (float4*)p,
                       (const float4*)g, (float4*)m, (float4*)v, beta1, beta2, (float)(lr / bc1),
                       (float)(1.0 / sqrt(bc2)), eps, weight_decay, nullptr, nullptr, nullptr, lr, beta1, beta2);
  return uh_check_launch("adam_step");
}

extern "C" int uh_adam_step(long long n, float* p, const float* g, float* m, float* v, float lr, float beta1,
                            float beta2, float eps, float weight_decay, int step, uh_stream s) {
  return adam_launch(n, p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, nullptr, nullptr, nullptr, false, s);
}

extern "C" int uh_adam_step_amp(long long n, float* p, const float* g, float* m, float* v, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int step, const float* grad_scale,
                                const float* found_inf, const float* step_dev, uh_stream s) {
  return adam_launch(n, p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale, found_inf, step_dev, true, s);
}
