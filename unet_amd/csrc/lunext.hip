// LUNeXt blocks (models/lunext.py of the reference) on gfx950: all bandwidth bound.
//   DepthwiseConvBlock  relu(BN(dwconv3x3(x)) + x)                            lunext.py:6-30
//   AttentionGate       sigmoid(W3(g+x))*x + sigmoid(W1 g)*tanh(W2 g)         lunext.py:76-122
// Tensors are padded-linear NHWC bf16 (common.h) with C a multiple of 8; the 3-channel full-resolution
// level is carried with its channels zero-padded to 8, so one 16-byte lane access is one pixel.
// The 1x1 convolutions of the gate at C >= 64 run on the MFMA GEMM (conv_gemm.hip); here are the depthwise
// stencil (forward, input gradient, weight gradient), BatchNorm+residual+ReLU and its backward, the gate's
// pointwise parts, and the fully fused 8-channel gate of the top level.
#include "ew.h"
#include <cstdlib>
#include <type_traits>

// ---- depthwise 3x3, pad 1 ------------------------------------------------------------------------------------
// The padded-linear tensor is one long grid of B*(H+1) rows of Wp = W+1 positions whose pad positions are zero, so a
// 3x3 stencil needs no image-border cases anywhere.  A thread owns one position x and 8 channels and walks DOWN a
// strip of DW_R rows with a 3x3 register window: each step loads only the three positions of the new bottom row
// (3 loads per output instead of 9; the loads of neighbouring threads overlap and hit in L1), and lanes run along
// channels, then x, so every access is coalesced for any C.  Workgroup (bx, by) = columns [bx*xt, (bx+1)*xt) of
// strip by, xt = 256 / (C/8).
// Rows per walk.  Measured on the LUNeXt training step: 16 rows 7.39 ms, 8 rows 7.51 ms, 4 rows 7.98 ms (shorter walks
// re-read their two halo rows more often), also for the launches that 16-row walks cut into only ~500 workgroups.
#define DW_R 16
struct DwGeo {
  int cpr, xt, cc;
  long long m;          // first position of this thread's walk
  int nr;               // rows to walk (0: idle thread)
};
// threads are dealt densely over (strip, x) pairs, xt = 256 / (C/8) pairs per workgroup
__device__ __forceinline__ DwGeo dw_geo(long long Mp, int C, int Wp) {
  DwGeo g;
  g.cpr = C / 8;
  g.xt = 256 / g.cpr;
  g.cc = threadIdx.x % g.cpr;
  const int xi = threadIdx.x / g.cpr;
  const long long rows = Mp / Wp;
  const long long q = (long long)blockIdx.x * g.xt + xi;
  const long long strip = q / Wp;
  const int x = (int)(q - strip * Wp);
  const long long r0 = strip * DW_R;
  g.m = r0 * Wp + x;
  const long long left = rows - r0;
  g.nr = (xi < g.xt && left > 0) ? (int)(left < DW_R ? left : DW_R) : 0;
  return g;
}
static inline dim3 dw_grid(int B, int H, int W, int C) {
  const int xt = 256 / (C / 8);
  const long long strips = ((long long)B * (H + 1) + DW_R - 1) / DW_R;
  return dim3((unsigned)((strips * (W + 1) + xt - 1) / xt));
}
// eight consecutive floats of a per-channel table (32-byte aligned: C is a multiple of 8)
__device__ __forceinline__ void dw_load8f(const float* __restrict__ p, float (&f)[8]) {
  const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[j] = a[j]; f[4 + j] = b[j]; }
}
// the 3x3 filters of channels c0 .. c0+7 from the module's own [C_real][1][3][3] fp32 weight (channel c at w + 9 c);
// channels >= c_real (zero padding of the 3-channel level) get zero filters
__device__ __forceinline__ void dw_load_filters(const float* __restrict__ wt, int c0, int c_real, float (&w)[9][8]) {
  if (c0 + 8 <= c_real) {
    float flat[72];
#pragma unroll
    for (int q = 0; q < 18; ++q) {
      const f32x4 v = *(const f32x4*)(wt + c0 * 9 + 4 * q);     // 288 bytes per thread: 16-byte aligned
#pragma unroll
      for (int e = 0; e < 4; ++e) flat[4 * q + e] = v[e];
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) w[t][j] = flat[j * 9 + t];
  } else {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) w[t][j] = (c0 + j < c_real) ? wt[(c0 + j) * 9 + t] : 0.f;
  }
}
__device__ __forceinline__ void dw_load8f_guarded(const float* __restrict__ p, int c0, int c_real, float (&f)[8]) {
  if (c0 + 8 <= c_real) {
    dw_load8f(p + c0, f);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (c0 + j < c_real) ? p[c0 + j] : 0.f;
  }
}
// positions m-1, m, m+1 of one row (p points at position m), 8 channels each, as floats
__device__ __forceinline__ void dw_load3(const char* __restrict__ p, long long cs2, float (&f)[3][8]) {
  unpack8(*(const U4*)(p - cs2), f[0]);
  unpack8(*(const U4*)p, f[1]);
  unpack8(*(const U4*)(p + cs2), f[2]);
}

// v = sum_t x[m+off(t)][c]*w[c][t] + bias[c];  train (RAW_STATS): out = fp16(v) + per-workgroup (sum v, sum v^2);
// eval (AFFINE_RELU): out = bf16(relu(v*scale + shift + x[m][c])).
template <int MODE>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(long long Mp, int C, int Wp, const char* __restrict__ x,
                                                         int x_cs, int x_co, const float* __restrict__ wt, int c_real,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const uint8_t* __restrict__ rowmask, char* __restrict__ out,
                                                         int o_cs, int o_co, float* __restrict__ partial, BnFwdFin fin) {
  extern __shared__ float lds[];
  const DwGeo g = dw_geo(Mp, C, Wp);
  float acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = 0.f;
  if (g.nr > 0) {
    float w[9][8], b[8], sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = sc[j] = sh[j] = 0.f;
    if (bias) dw_load8f_guarded(bias, g.cc * 8, c_real, b);
    if (MODE == UH_EPI_AFFINE_RELU) {
      dw_load8f(scale + g.cc * 8, sc);
      dw_load8f(shift + g.cc * 8, sh);
    }
    dw_load_filters(wt, g.cc * 8, c_real, w);
    const long long xs2 = (long long)x_cs * 2, os2 = (long long)o_cs * 2;
    const long long xrow = xs2 * Wp, orow = os2 * Wp;
    const char* xp = x + (g.m * x_cs + x_co + g.cc * 8) * 2;
    char* op = out + (g.m * o_cs + o_co + g.cc * 8) * 2;
    const uint8_t* mp = rowmask + g.m;
    float win[3][3][8];                 // [row slot][position][channel]
    dw_load3(xp - xrow, xs2, win[0]);
    dw_load3(xp, xs2, win[1]);
    // the row after the next one is in flight while a row is computed: with two or three waves per SIMD one row of
    // loads per thread does not cover the memory latency
    U4 ahead[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) ahead[k] = *(const U4*)(xp + xrow + (k - 1) * xs2);
#pragma unroll
    for (int r = 0; r < DW_R; ++r) {
      if (r >= g.nr) continue;          // (not break: the loop must unroll so that the window slots are static)
      // the rows rotate through the three slots: row r + k - 1 of the walk lives in slot (r + k) % 3
      float(&top)[3][8] = win[r % 3];
      float(&mid)[3][8] = win[(r + 1) % 3];
      float(&bot)[3][8] = win[(r + 2) % 3];
#pragma unroll
      for (int k = 0; k < 3; ++k) unpack8(ahead[k], bot[k]);
      if (r + 1 < g.nr) {
#pragma unroll
        for (int k = 0; k < 3; ++k) ahead[k] = *(const U4*)(xp + 2 * xrow + (k - 1) * xs2);
      }
      U4 o = {0u, 0u, 0u, 0u};
      if (*mp) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float t = b[j];
#pragma unroll
          for (int d = 0; d < 3; ++d) t = fmaf(top[d][j], w[d][j], t);        // taps in row-major order
#pragma unroll
          for (int d = 0; d < 3; ++d) t = fmaf(mid[d][j], w[3 + d][j], t);
#pragma unroll
          for (int d = 0; d < 3; ++d) t = fmaf(bot[d][j], w[6 + d][j], t);
          v[j] = t;
        }
        if (MODE == UH_EPI_AFFINE_RELU) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j] * sc[j] + sh[j] + mid[1][j], 0.f);
          o = pack8(v);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            acc[0][j] += v[j];
            acc[1][j] += v[j] * v[j];
          }
          o = pack8h(v);
        }
      }
      *(U4*)op = o;
      xp += xrow;
      op += orow;
      mp += Wp;
    }
  }
  if (MODE == UH_EPI_RAW_STATS) {
    if (fin.ws.counters) {                                  // BatchNorm scale / shift right here
      block_combine_store<2, true>(acc, C, g.cpr, g.xt, lds, partial + (long long)blockIdx.x * 2 * C);
      last_arriver_finish<2>(partial, C, fin.ws, fin);
    } else {
      block_combine_store<2>(acc, C, g.cpr, g.xt, lds, partial + (long long)blockIdx.x * 2 * C);
    }
  }
}

// the training forward as a per-wave strip walk (dwconv_fwd_strip_kernel, further down)
static bool dwf_use_strip(int C);
struct DwfApply;
static int dwconv_fwd_strip_launch(int B, int H, int W, int C, const void* x, int x_cstride, int x_coff, const float* w,
                                   int c_real, const float* bias, void* out, int o_cstride, int o_coff, float* partial,
                                   uh_stream s, const DwfApply* ap = nullptr);

extern "C" int uh_dw_parts(int B, int H, int W, int C) {
  if (!(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && C / 8 <= 256)) return 0;
  return (int)dw_grid(B, H, W, C).x;
}

extern "C" int uh_dwconv3x3_fwd(int B, int H, int W, int C, const void* x, int x_cstride, int x_coff,
                                const float* w, int c_real, const float* bias, int mode, const float* scale, const float* shift,
                                const uint8_t* rowmask, void* out, int o_cstride, int o_coff, float* partial,
                                uh_stream s) {
  EW_CHECK("dwconv3x3_fwd", C);
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && x && w && rowmask && out, "dwconv3x3_fwd: bad arguments");
  UH_REQUIRE(c_real > 0 && c_real <= C && ((uintptr_t)w & 15) == 0, "dwconv3x3_fwd: c_real=%d of C=%d, w 16-byte aligned", c_real, C);
  UH_REQUIRE(mode == UH_EPI_RAW_STATS || mode == UH_EPI_AFFINE_RELU, "dwconv3x3_fwd: bad mode %d", mode);
  UH_REQUIRE(mode != UH_EPI_RAW_STATS || partial, "dwconv3x3_fwd: RAW_STATS needs partial");
  UH_REQUIRE(mode != UH_EPI_AFFINE_RELU || (scale && shift), "dwconv3x3_fwd: AFFINE_RELU needs scale/shift");
  if (mode == UH_EPI_RAW_STATS && dwf_use_strip(C))
    return dwconv_fwd_strip_launch(B, H, W, C, x, x_cstride, x_coff, w, c_real, bias, out, o_cstride, o_coff, partial, s);
  const long long Mp = uh_mp(B, H, W);
  const int rpi = ew_rows_per_iter(C);
  BnFwdFin fin{};
  if (mode == UH_EPI_RAW_STATS)
    hipLaunchKernelGGL(dwconv_fwd_kernel<UH_EPI_RAW_STATS>, dw_grid(B, H, W, C), dim3(256), (size_t)rpi * 2 * C * 4,
                       (hipStream_t)s, Mp, C, W + 1, (const char*)x, x_cstride, x_coff, w, c_real, bias, scale, shift,
                       rowmask, (char*)out, o_cstride, o_coff, partial, fin);
  else
    hipLaunchKernelGGL(dwconv_fwd_kernel<UH_EPI_AFFINE_RELU>, dw_grid(B, H, W, C), dim3(256), 0, (hipStream_t)s, Mp, C,
                       W + 1, (const char*)x, x_cstride, x_coff, w, c_real, bias, scale, shift, rowmask, (char*)out,
                       o_cstride, o_coff, partial, fin);
  return uh_check_launch("dwconv3x3_fwd");
}

// The train-mode launch that also finalises its BatchNorm statistics (what uh_bn_finalize would do behind it), in the
// last workgroups to finish (ew.h: last_arriver_finish).  ws / counters: uh_fin_ws_bytes(uh_dw_parts(..), C, 2),
// uh_fin_counters().
extern "C" int uh_dwconv3x3_fwd_bn(int B, int H, int W, int C, const void* x, int x_cstride, int x_coff, const float* w,
                                   int c_real, const float* bias, const uint8_t* rowmask, void* out, int o_cstride,
                                   int o_coff, const float* gamma, const float* beta, float eps, float momentum,
                                   float* mean, float* invstd, float* scale, float* shift, float* running_mean,
                                   float* running_var, long long* nbt, void* ws, size_t ws_bytes, unsigned* counters,
                                   uh_stream s) {
  EW_CHECK("dwconv3x3_fwd_bn", C);
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && x && w && rowmask && out && gamma && beta && mean && invstd && scale && shift,
             "dwconv3x3_fwd_bn: bad arguments");
  UH_REQUIRE(c_real > 0 && c_real <= C && ((uintptr_t)w & 15) == 0, "dwconv3x3_fwd_bn: c_real=%d of C=%d, w 16-byte aligned", c_real, C);
  UH_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "dwconv3x3_fwd_bn: running stats must come in pairs");
  const dim3 grid = dw_grid(B, H, W, C);
  BnFwdFin fin{};
  float* partial = nullptr;
  UH_REQUIRE(fin_split((int)grid.x, C, 2, ws, ws_bytes, counters, fin.ws, partial),
             "dwconv3x3_fwd_bn: workspace of %zu bytes (8-byte aligned) and counters needed", fin_ws_bytes((int)grid.x, C, 2));
  fin.count = (double)B * H * W;
  fin.gamma = gamma;
  fin.beta = beta;
  fin.eps = eps;
  fin.momentum = momentum;
  fin.mean = mean;
  fin.invstd = invstd;
  fin.scale = scale;
  fin.shift = shift;
  fin.rm = running_mean;
  fin.rv = running_var;
  fin.nbt = nbt;
  fin.c_real = c_real;
  const long long Mp = uh_mp(B, H, W);
  const int rpi = ew_rows_per_iter(C);
  hipLaunchKernelGGL(dwconv_fwd_kernel<UH_EPI_RAW_STATS>, grid, dim3(256), (size_t)rpi * 2 * C * 4, (hipStream_t)s, Mp, C,
                     W + 1, (const char*)x, x_cstride, x_coff, w, c_real, bias, nullptr, nullptr, rowmask, (char*)out,
                     o_cstride, o_coff, partial, fin);
  return uh_check_launch("dwconv3x3_fwd_bn");
}

// ---- BatchNorm + residual + ReLU: a = relu(y*scale + shift + x) ---------------------------------------------------
__global__ __launch_bounds__(256) void bn_res_relu_apply_kernel(long long Mp, int C, const char* __restrict__ y,
                                                                int y_cs, int y_co, const char* __restrict__ x,
                                                                int x_cs, int x_co, const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const uint8_t* __restrict__ rowmask,
                                                                char* __restrict__ a, int a_cs, int a_co) {
  const int cpr = C / 8, rpi = 256 / cpr;
  const int rr = threadIdx.x / cpr, cc = threadIdx.x % cpr;
  if (rr >= rpi) return;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = scale[cc * 8 + j];
    sh[j] = shift[cc * 8 + j];
  }
  for (long long m = (long long)blockIdx.x * rpi + rr; m < Mp; m += (long long)gridDim.x * rpi) {
    U4 o = {0u, 0u, 0u, 0u};
    // loads issued before the mask is known (pad rows are readable): one memory latency per row, not two
    const U4 yr = *(const U4*)(y + (m * y_cs + y_co + cc * 8) * 2);
    const U4 xr = *(const U4*)(x + (m * x_cs + x_co + cc * 8) * 2);
    if (rowmask[m]) {
      float f[8], r[8];
      unpack8h(yr, f);
      unpack8(xr, r);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * sc[j] + sh[j] + r[j], 0.f);
      o = pack8(f);
    }
    *(U4*)(a + (m * a_cs + a_co + cc * 8) * 2) = o;
  }
}

extern "C" int uh_bn_res_relu_apply(long long Mp, int C, const void* y, int y_cstride, int y_coff, const void* x,
                                    int x_cstride, int x_coff, const float* scale, const float* shift,
                                    const uint8_t* rowmask, void* a, int a_cstride, int a_coff, uh_stream s) {
  EW_CHECK("bn_res_relu_apply", C);
  UH_REQUIRE(Mp > 0 && y && x && scale && shift && rowmask && a, "bn_res_relu_apply: bad arguments");
  hipLaunchKernelGGL(bn_res_relu_apply_kernel, dim3(ew_blocks(Mp, C)), dim3(256), 0, (hipStream_t)s, Mp, C,
                     (const char*)y, y_cstride, y_coff, (const char*)x, x_cstride, x_coff, scale, shift, rowmask,
                     (char*)a, a_cstride, a_coff);
  return uh_check_launch("bn_res_relu_apply");
}

// The same pass for a block whose output also feeds a 2x2 stride-2 convolution (lunext.py:139-154): every interior
// value is stored a second time where that convolution's GEMM reads it, depth[(b, y/2, x/2)][t*C + c] with
// t = 2 (y & 1) + (x & 1) -- what uh_space_to_depth2_of would copy there in a pass of its own.  Pad rows of `depth` are
// not written (they are zero in a buffer nothing else writes); a last odd row / column has no place in it.
__global__ __launch_bounds__(256) void bn_res_relu_apply_s2d_kernel(
    long long Mp, int C, int H, int W, const char* __restrict__ y, int y_cs, int y_co, const char* __restrict__ x,
    int x_cs, int x_co, const float* __restrict__ scale, const float* __restrict__ shift, char* __restrict__ a, int a_cs,
    int a_co, char* __restrict__ depth, int d_cs, int d_co) {
  const int cpr = C / 8, rpi = 256 / cpr;
  const int rr = threadIdx.x / cpr, cc = threadIdx.x % cpr;
  if (rr >= rpi) return;
  const unsigned Wp = W + 1, Hp = H + 1, Hs = H / 2, Ws = W / 2;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = scale[cc * 8 + j];
    sh[j] = shift[cc * 8 + j];
  }
  for (long long m = (long long)blockIdx.x * rpi + rr; m < Mp; m += (long long)gridDim.x * rpi) {
    const unsigned mu = (unsigned)m, px = mu % Wp, r = mu / Wp, py = r % Hp, b = r / Hp;
    U4 o = {0u, 0u, 0u, 0u};
    const U4 yr = *(const U4*)(y + (m * y_cs + y_co + cc * 8) * 2);
    const U4 xr = *(const U4*)(x + (m * x_cs + x_co + cc * 8) * 2);
    const bool interior = (py < (unsigned)H) & (px < (unsigned)W);
    if (interior) {
      float f[8], q[8];
      unpack8h(yr, f);
      unpack8(xr, q);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * sc[j] + sh[j] + q[j], 0.f);
      o = pack8(f);
    }
    *(U4*)(a + (m * a_cs + a_co + cc * 8) * 2) = o;
    if (interior & (py < 2 * Hs) & (px < 2 * Ws)) {
      const long long ms = ((long long)b * (Hs + 1) + (py >> 1)) * (Ws + 1) + (px >> 1);
      const int t = ((py & 1) << 1) | (px & 1);
      *(U4*)(depth + (ms * d_cs + d_co + t * C + cc * 8) * 2) = o;
    }
  }
}

extern "C" int uh_bn_res_relu_apply_s2d(int B, int H, int W, int C, const void* y, int y_cstride, int y_coff,
                                        const void* x, int x_cstride, int x_coff, const float* scale,
                                        const float* shift, void* a, int a_cstride, int a_coff, void* depth,
                                        int d_cstride, int d_coff, uh_stream s) {
  EW_CHECK("bn_res_relu_apply_s2d", C);
  UH_REQUIRE(B > 0 && H >= 2 && W >= 2 && y && x && scale && shift && a && depth, "bn_res_relu_apply_s2d: bad arguments");
  UH_REQUIRE(d_cstride % 8 == 0 && d_coff % 8 == 0 && d_cstride >= d_coff + 4 * C, "bn_res_relu_apply_s2d: bad depth view");
  const long long Mp = uh_mp(B, H, W);
  UH_REQUIRE(Mp < (1ll << 31), "bn_res_relu_apply_s2d: %lld positions", Mp);
  hipLaunchKernelGGL(bn_res_relu_apply_s2d_kernel, dim3(ew_blocks(Mp, C)), dim3(256), 0, (hipStream_t)s, Mp, C, H, W,
                     (const char*)y, y_cstride, y_coff, (const char*)x, x_cstride, x_coff, scale, shift, (char*)a,
                     a_cstride, a_coff, (char*)depth, d_cstride, d_coff);
  return uh_check_launch("bn_res_relu_apply_s2d");
}

// backward: d = da * (y*scale + shift + x > 0); partial = (sum d, sum d*xhat)
__global__ __launch_bounds__(256) void bn_res_relu_bwd_reduce_kernel(
    long long Mp, int C, const char* __restrict__ da, int da_cs, int da_co, const char* __restrict__ y, int y_cs,
    int y_co, const char* __restrict__ x, int x_cs, int x_co, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ scale, const float* __restrict__ shift,
    float* __restrict__ partial, BnBwdFin fin) {
  extern __shared__ float lds[];
  const int cpr = C / 8, rpi = 256 / cpr;
  const int rr = threadIdx.x / cpr, cc = threadIdx.x % cpr;
  float acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = 0.f;
  if (rr < rpi) {
    float sc[8], sh[8], mu[8], is[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = scale[cc * 8 + j];
      sh[j] = shift[cc * 8 + j];
      mu[j] = mean[cc * 8 + j];
      is[j] = invstd[cc * 8 + j];
    }
    for (long long m = (long long)blockIdx.x * rpi + rr; m < Mp; m += (long long)gridDim.x * rpi) {
      float g[8], v[8], r[8];
      unpack8(*(const U4*)(da + (m * da_cs + da_co + cc * 8) * 2), g);
      unpack8h(*(const U4*)(y + (m * y_cs + y_co + cc * 8) * 2), v);
      unpack8(*(const U4*)(x + (m * x_cs + x_co + cc * 8) * 2), r);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = (v[j] * sc[j] + sh[j] + r[j] > 0.f) ? g[j] : 0.f;
        acc[0][j] += d;
        acc[1][j] += d * (v[j] - mu[j]) * is[j];
      }
    }
  }
  if (fin.ws.counters) {
    block_combine_store<2, true>(acc, C, cpr, rpi, lds, partial + (long long)blockIdx.x * 2 * C);
    last_arriver_finish<2>(partial, C, fin.ws, fin);
  } else {
    block_combine_store<2>(acc, C, cpr, rpi, lds, partial + (long long)blockIdx.x * 2 * C);
  }
}

extern "C" int uh_bn_res_relu_bwd_reduce(long long Mp, int C, const void* da, int da_cstride, int da_coff,
                                         const void* y, int y_cstride, int y_coff, const void* x, int x_cstride,
                                         int x_coff, const float* mean, const float* invstd, const float* scale,
                                         const float* shift, float* partial, uh_stream s) {
  EW_CHECK("bn_res_relu_bwd_reduce", C);
  UH_REQUIRE(Mp > 0 && da && y && x && mean && invstd && scale && shift && partial, "bn_res_relu_bwd_reduce: bad arguments");
  const int rpi = ew_rows_per_iter(C);
  BnBwdFin fin{};
  hipLaunchKernelGGL(bn_res_relu_bwd_reduce_kernel, dim3(ew_stat_blocks(Mp, C)), dim3(256), (size_t)rpi * 2 * C * 4,
                     (hipStream_t)s, Mp, C, (const char*)da, da_cstride, da_coff, (const char*)y, y_cstride, y_coff,
                     (const char*)x, x_cstride, x_coff, mean, invstd, scale, shift, partial, fin);
  return uh_check_launch("bn_res_relu_bwd_reduce");
}

// the same launch followed, in its last workgroups, by what uh_bn_bwd_finalize does (ew.h: last_arriver_finish);
// ws / counters: uh_fin_ws_bytes(uh_bn_bwd_parts(Mp, C), C, 2), uh_fin_counters()
extern "C" int uh_bn_res_relu_bwd_reduce_fin(long long Mp, int C, const void* da, int da_cstride, int da_coff,
                                             const void* y, int y_cstride, int y_coff, const void* x, int x_cstride,
                                             int x_coff, const float* mean, const float* invstd, const float* scale,
                                             const float* shift, int c_real, long long count, float* dgamma,
                                             float* dbeta, float* coef, void* ws, size_t ws_bytes, unsigned* counters,
                                             uh_stream s) {
  EW_CHECK("bn_res_relu_bwd_reduce_fin", C);
  UH_REQUIRE(Mp > 0 && da && y && x && mean && invstd && scale && shift && dgamma && dbeta && coef && count > 0 &&
                 c_real > 0 && c_real <= C,
             "bn_res_relu_bwd_reduce_fin: bad arguments");
  const int parts = ew_stat_blocks(Mp, C);
  BnBwdFin fin{};
  float* partial = nullptr;
  UH_REQUIRE(fin_split(parts, C, 2, ws, ws_bytes, counters, fin.ws, partial),
             "bn_res_relu_bwd_reduce_fin: workspace of %zu bytes (8-byte aligned) and counters needed", fin_ws_bytes(parts, C, 2));
  fin.count = (double)count;
  fin.dgamma = dgamma;
  fin.dbeta = dbeta;
  fin.coef = coef;
  fin.c_real = c_real;
  fin.C = C;
  const int rpi = ew_rows_per_iter(C);
  hipLaunchKernelGGL(bn_res_relu_bwd_reduce_kernel, dim3(parts), dim3(256), (size_t)rpi * 2 * C * 4, (hipStream_t)s, Mp,
                     C, (const char*)da, da_cstride, da_coff, (const char*)y, y_cstride, y_coff, (const char*)x,
                     x_cstride, x_coff, mean, invstd, scale, shift, partial, fin);
  return uh_check_launch("bn_res_relu_bwd_reduce_fin");
}

// dy = scale*(d - coef0 - xhat*coef1)  (gradient of the depthwise conv output),  dr = d  (residual branch)
__global__ __launch_bounds__(256) void bn_res_relu_bwd_apply_kernel(
    long long Mp, int C, const char* __restrict__ da, int da_cs, int da_co, const char* __restrict__ y, int y_cs,
    int y_co, const char* __restrict__ x, int x_cs, int x_co, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ coef, const uint8_t* __restrict__ rowmask, char* __restrict__ dy, int dy_cs, int dy_co,
    char* __restrict__ dr, int dr_cs, int dr_co) {
  const int cpr = C / 8, rpi = 256 / cpr;
  const int rr = threadIdx.x / cpr, cc = threadIdx.x % cpr;
  if (rr >= rpi) return;
  float sc[8], sh[8], mu[8], is[8], c0[8], c1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cc * 8 + j;
    sc[j] = scale[c];
    sh[j] = shift[c];
    mu[j] = mean[c];
    is[j] = invstd[c];
    c0[j] = coef[c];
    c1[j] = coef[C + c];
  }
  for (long long m = (long long)blockIdx.x * rpi + rr; m < Mp; m += (long long)gridDim.x * rpi) {
    U4 o = {0u, 0u, 0u, 0u}, o2 = {0u, 0u, 0u, 0u};
    const U4 gr = *(const U4*)(da + (m * da_cs + da_co + cc * 8) * 2);      // before the mask is known, as above
    const U4 yr = *(const U4*)(y + (m * y_cs + y_co + cc * 8) * 2);
    const U4 xr = *(const U4*)(x + (m * x_cs + x_co + cc * 8) * 2);
    if (rowmask[m]) {
      float g[8], v[8], r[8];
      unpack8(gr, g);
      unpack8h(yr, v);
      unpack8(xr, r);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = (v[j] * sc[j] + sh[j] + r[j] > 0.f) ? g[j] : 0.f;
        const float xh = (v[j] - mu[j]) * is[j];
        r[j] = d;
        g[j] = sc[j] * (d - c0[j] - xh * c1[j]);
      }
      o = pack8(g);
      o2 = pack8(r);
    }
    *(U4*)(dy + (m * dy_cs + dy_co + cc * 8) * 2) = o;
    *(U4*)(dr + (m * dr_cs + dr_co + cc * 8) * 2) = o2;
  }
}

extern "C" int uh_bn_res_relu_bwd_apply(long long Mp, int C, const void* da, int da_cstride, int da_coff,
                                        const void* y, int y_cstride, int y_coff, const void* x, int x_cstride,
                                        int x_coff, const float* mean, const float* invstd, const float* scale,
                                        const float* shift, const float* coef, const uint8_t* rowmask, void* dy,
                                        int dy_cstride, int dy_coff, void* dr, int dr_cstride, int dr_coff,
                                        uh_stream s) {
  EW_CHECK("bn_res_relu_bwd_apply", C);
  UH_REQUIRE(Mp > 0 && da && y && x && mean && invstd && scale && shift && coef && rowmask && dy && dr,
             "bn_res_relu_bwd_apply: bad arguments");
  hipLaunchKernelGGL(bn_res_relu_bwd_apply_kernel, dim3(ew_blocks(Mp, C)), dim3(256), 0, (hipStream_t)s, Mp, C,
                     (const char*)da, da_cstride, da_coff, (const char*)y, y_cstride, y_coff, (const char*)x, x_cstride,
                     x_coff, mean, invstd, scale, shift, coef, rowmask, (char*)dy, dy_cstride, dy_coff, (char*)dr,
                     dr_cstride, dr_coff);
  return uh_check_launch("bn_res_relu_bwd_apply");
}

// dx[m][c] = sum_t dy[m - off(t)][c]*w[c][t] + dr[m][c]   (same strip walk as the forward kernel)
__global__ __launch_bounds__(256) void dwconv_dgrad_kernel(long long Mp, int C, int Wp, const char* __restrict__ dy,
                                                           int dy_cs, int dy_co, const float* __restrict__ wt, int c_real,
                                                           const char* __restrict__ dr, int dr_cs, int dr_co,
                                                           const uint8_t* __restrict__ rowmask,
                                                           char* __restrict__ dx, int dx_cs, int dx_co) {
  const DwGeo g = dw_geo(Mp, C, Wp);
  if (g.nr <= 0) return;
  float w[9][8];
  dw_load_filters(wt, g.cc * 8, c_real, w);
  const long long gs2 = (long long)dy_cs * 2, grow = gs2 * Wp;
  const long long rrow = (long long)dr_cs * 2 * Wp, orow = (long long)dx_cs * 2 * Wp;
  const char* gp = dy + (g.m * dy_cs + dy_co + g.cc * 8) * 2;
  const char* rp = dr ? dr + (g.m * dr_cs + dr_co + g.cc * 8) * 2 : nullptr;
  char* op = dx + (g.m * dx_cs + dx_co + g.cc * 8) * 2;
  const uint8_t* mp = rowmask + g.m;
  float win[3][3][8];
  dw_load3(gp - grow, gs2, win[0]);
  dw_load3(gp, gs2, win[1]);
#pragma unroll
  for (int r = 0; r < DW_R; ++r) {
    if (r >= g.nr) continue;            // (not break: the loop must unroll so that the window slots are static)
    float(&top)[3][8] = win[r % 3];
    float(&mid)[3][8] = win[(r + 1) % 3];
    float(&bot)[3][8] = win[(r + 2) % 3];
    dw_load3(gp + grow, gs2, bot);
    U4 o = {0u, 0u, 0u, 0u};
    if (*mp) {
      float v[8];
      if (rp) {
        unpack8(*(const U4*)rp, v);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
      }
      // tap t = (ky, kx) reads dy at m - off(t): the window mirrored in both directions
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int d = 0; d < 3; ++d) v[j] = fmaf(bot[2 - d][j], w[d][j], v[j]);     // taps in row-major order
#pragma unroll
        for (int d = 0; d < 3; ++d) v[j] = fmaf(mid[2 - d][j], w[3 + d][j], v[j]);
#pragma unroll
        for (int d = 0; d < 3; ++d) v[j] = fmaf(top[2 - d][j], w[6 + d][j], v[j]);
      }
      o = pack8(v);
    }
    *(U4*)op = o;
    gp += grow;
    if (rp) rp += rrow;
    op += orow;
    mp += Wp;
  }
}

extern "C" int uh_dwconv3x3_dgrad(int B, int H, int W, int C, const void* dy, int dy_cstride, int dy_coff,
                                  const float* w, int c_real, const void* dr, int dr_cstride, int dr_coff,
                                  const uint8_t* rowmask, void* dx, int dx_cstride, int dx_coff, uh_stream s) {
  EW_CHECK("dwconv3x3_dgrad", C);
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && dy && w && rowmask && dx, "dwconv3x3_dgrad: bad arguments");
  UH_REQUIRE(c_real > 0 && c_real <= C && ((uintptr_t)w & 15) == 0, "dwconv3x3_dgrad: c_real=%d of C=%d, w 16-byte aligned", c_real, C);
  const long long Mp = uh_mp(B, H, W);
  hipLaunchKernelGGL(dwconv_dgrad_kernel, dw_grid(B, H, W, C), dim3(256), 0, (hipStream_t)s, Mp, C, W + 1,
                     (const char*)dy, dy_cstride, dy_coff, w, c_real, (const char*)dr, dr_cstride, dr_coff, rowmask,
                     (char*)dx, dx_cstride, dx_coff);
  return uh_check_launch("dwconv3x3_dgrad");
}

// partial[blk][9][C]: sum_m dy[m][c] * x[m + off(t)][c]   (dy is zero on pad positions)
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(long long Mp, int C, int Wp, const char* __restrict__ dy,
                                                           int dy_cs, int dy_co, const char* __restrict__ x, int x_cs,
                                                           int x_co, float* __restrict__ partial) {
  extern __shared__ float lds[];
  const DwGeo g = dw_geo(Mp, C, Wp);
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
  if (g.nr > 0) {
    const long long xs2 = (long long)x_cs * 2, xrow = xs2 * Wp, grow = (long long)dy_cs * 2 * Wp;
    const char* xp = x + (g.m * x_cs + x_co + g.cc * 8) * 2;
    const char* gp = dy + (g.m * dy_cs + dy_co + g.cc * 8) * 2;
    float win[3][3][8];
    dw_load3(xp - xrow, xs2, win[0]);
    dw_load3(xp, xs2, win[1]);
#pragma unroll
    for (int r = 0; r < DW_R; ++r) {
      if (r >= g.nr) continue;          // (not break: the loop must unroll so that the window slots are static)
      float(&top)[3][8] = win[r % 3];
      float(&mid)[3][8] = win[(r + 1) % 3];
      float(&bot)[3][8] = win[(r + 2) % 3];
      dw_load3(xp + xrow, xs2, bot);
      float gq[8];
      unpack8(*(const U4*)gp, gq);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          acc[d][j] = fmaf(gq[j], top[d][j], acc[d][j]);
          acc[3 + d][j] = fmaf(gq[j], mid[d][j], acc[3 + d][j]);
          acc[6 + d][j] = fmaf(gq[j], bot[d][j], acc[6 + d][j]);
        }
      }
      xp += xrow;
      gp += grow;
    }
  }
  float* row = partial + (long long)blockIdx.x * 9 * C;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float one[1][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) one[0][j] = acc[t][j];
    __syncthreads();
    block_combine_store<1>(one, C, g.cpr, g.xt, lds, row + t * C);
  }
}

extern "C" int uh_dwconv3x3_wgrad(int B, int H, int W, int C, const void* dy, int dy_cstride, int dy_coff,
                                  const void* x, int x_cstride, int x_coff, float* partial, uh_stream s) {
  EW_CHECK("dwconv3x3_wgrad", C);
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && dy && x && partial, "dwconv3x3_wgrad: bad arguments");
  const long long Mp = uh_mp(B, H, W);
  const int rpi = ew_rows_per_iter(C);
  hipLaunchKernelGGL(dwconv_wgrad_kernel, dw_grid(B, H, W, C), dim3(256), (size_t)rpi * C * 4, (hipStream_t)s, Mp, C,
                     W + 1, (const char*)dy, dy_cstride, dy_coff, (const char*)x, x_cstride, x_coff, partial);
  return uh_check_launch("dwconv3x3_wgrad");
}

// ---- the whole backward of relu(BN(dwconv3x3(x)) + x) behind the BatchNorm sums, in one pass --------------------------
//   d     = da * (y*scale + shift + x > 0)            gradient at the sum = gradient of the residual branch
//   dy    = scale*(d - coef0 - xhat*coef1)            gradient of the depthwise conv output, 0 on pad positions
//   dx[m] = sum_t dy[m - off(t)]*w[t] + d[m]
//   dw[t] = sum_m dy[m - off(t)]*x[m]                 (= sum_m dy[m]*x[m + off(t)])
// Both sums run over the same mirrored 3x3 window of dy around m, which is never stored: a thread owns one position
// and FOUR channels (8-byte accesses: the window, the filters and the nine weight-gradient sums of eight channels
// would not fit the register file at a useful occupancy) and walks down a strip as the kernels above do, evaluating
// dy for the three positions of each new bottom row from da, y and x (neighbouring threads evaluate the same
// elements again; the loads hit in L1).  dy = scale*d - (y*cc + bp) with cc = scale*coef1*invstd,
// bp = scale*coef0 - cc*mean.
typedef uint2 U2;
__device__ __forceinline__ void unpack4(const U2& u, float (&f)[4]) {
  f[0] = __uint_as_float(u.x << 16);
  f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16);
  f[3] = __uint_as_float(u.y & 0xffff0000u);
}
__device__ __forceinline__ void unpack4h(const U2& u, float (&f)[4]) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  const h2 a = *reinterpret_cast<const h2*>(&u.x), b = *reinterpret_cast<const h2*>(&u.y);
  f[0] = (float)a[0];
  f[1] = (float)a[1];
  f[2] = (float)b[0];
  f[3] = (float)b[1];
}
static inline dim3 dwb_grid(int B, int H, int W, int C) {
  const int xt = 256 / (C / 4);
  const long long strips = ((long long)B * (H + 1) + DW_R - 1) / DW_R;
  return dim3((unsigned)((strips * (W + 1) + xt - 1) / xt));
}

struct DwbRaw {
  U2 g, v, r;
};

__global__ __launch_bounds__(256) void dwblock_bwd_kernel(
    long long Mp, int C, int Wp, int Hp, const char* __restrict__ da, int da_cs, int da_co, const char* __restrict__ y,
    int y_cs, int y_co, const char* __restrict__ x, int x_cs, int x_co, const float* __restrict__ wt, int c_real,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ coef, char* __restrict__ dx, int dx_cs, int dx_co,
    float* __restrict__ partial) {
  extern __shared__ float lds[];
  const int cpr = C / 4, xt = 256 / cpr;           // cpr a power of two <= 128: at least two positions per workgroup
  const int cc = threadIdx.x % cpr, xi = threadIdx.x / cpr;
  const int c0 = cc * 4;
  const long long rows = Mp / Wp;
  const long long q = (long long)blockIdx.x * xt + xi;
  const long long strip = q / Wp;
  const int x0 = (int)(q - strip * Wp);
  const long long r0 = strip * DW_R;
  const long long left = rows - r0;
  const int nr = left > 0 ? (int)(left < DW_R ? left : DW_R) : 0;
  // Every thread of the workgroup walks all DW_R rows (the row exchange below has a barrier per row).  Row e of the
  // walk is tensor row r0 - 1 + e (e = 0: the row above the strip ... DW_R + 1: the row below it); rows e <= lim exist,
  // later ones count as padding and are fetched from the last row that does (a thread past the end of the tensor:
  // position 0, nothing exists).
  const long long m0 = nr > 0 ? r0 * Wp + x0 : 0;
  const int lim = nr > 0 ? (int)(left < DW_R + 1 ? left : DW_R + 1) : -1;
  const bool edge_l = xi == 0, edge_r = xi == xt - 1, edge = edge_l || edge_r;
  const int hoff = edge_l ? -1 : 1;                // the column an edge thread evaluates for its missing neighbour
  const bool col_own = x0 < Wp - 1, col_halo = x0 + hoff >= 0 && x0 + hoff < Wp - 1;
  int ry = (int)((r0 + Hp - 1) % Hp);              // image row of walk row 0 (the row above tensor row 0 counts as padding)
  float acc[9][4], w[9][4], sc[4], sh[4], bp[4], cq[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + j;
    sc[j] = scale[c];
    sh[j] = shift[c];
    cq[j] = sc[j] * coef[C + c] * invstd[c];
    bp[j] = sc[j] * coef[c] - cq[j] * mean[c];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      acc[t][j] = 0.f;
      w[t][j] = (c < c_real) ? wt[c * 9 + t] : 0.f;
    }
  }
  const long long da_row = (long long)da_cs * 2 * Wp, y_row = (long long)y_cs * 2 * Wp, x_row = (long long)x_cs * 2 * Wp;
  const char* dap = da + ((m0 - Wp) * da_cs + da_co + c0) * 2;      // walk row 0; guard rows make it readable
  const char* yp = y + ((m0 - Wp) * y_cs + y_co + c0) * 2;
  const char* xp = x + ((m0 - Wp) * x_cs + x_co + c0) * 2;
  const long long da_h = (long long)hoff * da_cs * 2, y_h = (long long)hoff * y_cs * 2, x_h = (long long)hoff * x_cs * 2;
  int e_next = 0;                                    // the walk row the pointers are at
  auto fetch = [&](DwbRaw& own, DwbRaw& hal, bool& ok) {
    own.g = *(const U2*)dap;
    own.v = *(const U2*)yp;
    own.r = *(const U2*)xp;
    if (edge) {
      hal.g = *(const U2*)(dap + da_h);
      hal.v = *(const U2*)(yp + y_h);
      hal.r = *(const U2*)(xp + x_h);
    }
    ok = e_next <= lim && ry != Hp - 1;
    const bool step = e_next <= nr;                  // the row after this one can be read as well
    dap += step ? da_row : 0;
    yp += step ? y_row : 0;
    xp += step ? x_row : 0;
    ++e_next;
    ry = ry + 1 == Hp ? 0 : ry + 1;
  };
  auto eval = [&](const DwbRaw& t, bool keep, float (&dyv)[4], float (&dc)[4], float (&xc)[4]) {
    float gf[4], vf[4], xf[4];
    unpack4(t.g, gf);
    unpack4h(t.v, vf);
    unpack4(t.r, xf);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = (keep && vf[j] * sc[j] + sh[j] + xf[j] > 0.f) ? gf[j] : 0.f;
      dc[j] = d;
      xc[j] = keep ? xf[j] : 0.f;
      dyv[j] = keep ? fmaf(sc[j], d, -fmaf(vf[j], cq[j], bp[j])) : 0.f;
    }
  };
  // dy of this thread's column goes to slot xi + 1 of the row buffer, the edge threads' extra column to slot 0 / xt + 1;
  // after the barrier the neighbours' values are read back.  Two buffers alternate, so one barrier per row is enough.
  float win[3][3][4], dcen[3][4], xcen[3][4];       // [row slot][column][channel]
  auto exchange = [&](int e, const DwbRaw& own, const DwbRaw& hal, bool ok, float (&dst)[3][4], float (&dc)[4],
                      float (&xc)[4]) {
    float* buf = lds + (e & 1) * (xt + 2) * C + c0;
    eval(own, ok && col_own, dst[1], dc, xc);
    *(f32x4*)(buf + (xi + 1) * C) = f32x4{dst[1][0], dst[1][1], dst[1][2], dst[1][3]};
    if (edge) {
      float hv[4], t0[4], t1[4];
      eval(hal, ok && col_halo, hv, t0, t1);
      *(f32x4*)(buf + (edge_l ? 0 : xt + 1) * C) = f32x4{hv[0], hv[1], hv[2], hv[3]};
    }
    __syncthreads();
    const f32x4 l = *(const f32x4*)(buf + xi * C), r = *(const f32x4*)(buf + (xi + 2) * C);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dst[0][j] = l[j];
      dst[2][j] = r[j];
    }
  };
  DwbRaw own, hal = {{0u, 0u}, {0u, 0u}, {0u, 0u}}, own_n, hal_n = hal;
  bool ok, ok_n;
  fetch(own, hal, ok);
  fetch(own_n, hal_n, ok_n);
  exchange(0, own, hal, ok, win[0], dcen[0], xcen[0]);
  own = own_n; hal = hal_n; ok = ok_n;
  fetch(own_n, hal_n, ok_n);
  exchange(1, own, hal, ok, win[1], dcen[1], xcen[1]);
  bool okc[3] = {false, ok && col_own, false};      // is the centre of the row in slot k a real position
  char* op = dx + (m0 * dx_cs + dx_co + c0) * 2;
  const long long orow = (long long)dx_cs * 2 * Wp;
#pragma unroll
  for (int r = 0; r < DW_R; ++r) {
    float(&top)[3][4] = win[r % 3];
    float(&mid)[3][4] = win[(r + 1) % 3];
    float(&bot)[3][4] = win[(r + 2) % 3];
    own = own_n; hal = hal_n; ok = ok_n;
    if (r + 1 < DW_R) fetch(own_n, hal_n, ok_n);    // the row after the next one is in flight while this one is used
    exchange(r + 2, own, hal, ok, bot, dcen[(r + 2) % 3], xcen[(r + 2) % 3]);
    okc[(r + 2) % 3] = ok && col_own;
    const float(&dm)[4] = dcen[(r + 1) % 3];
    const float(&xm)[4] = xcen[(r + 1) % 3];         // 0 where the centre is padding or does not exist
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // summation order of dx: the row above, the centre term, this row, the row below (the order in which a walk
      // that meets each row once can add them; dwblock_bwd_strip_kernel does, and gives the same bits)
      float v = 0.f;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        v = fmaf(top[2 - d][j], w[6 + d][j], v);
        acc[6 + d][j] = fmaf(xm[j], top[2 - d][j], acc[6 + d][j]);
      }
      v += dm[j];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        v = fmaf(mid[2 - d][j], w[3 + d][j], v);
        acc[3 + d][j] = fmaf(xm[j], mid[2 - d][j], acc[3 + d][j]);
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        v = fmaf(bot[2 - d][j], w[d][j], v);
        acc[d][j] = fmaf(xm[j], bot[2 - d][j], acc[d][j]);
      }
      o[j] = okc[(r + 1) % 3] ? v : 0.f;             // (the window around a pad position holds real neighbours)
    }
    if (r < nr) {
      U2 ov;
      ov.x = pack2(o[0], o[1]);
      ov.y = pack2(o[2], o[3]);
      *(U2*)op = ov;
      op += orow;
    }
  }
  // lds[xi][9][C] -> partial[blk][9][C]
  __syncthreads();
  float* mine = lds + ((long long)xi * 9) * C + c0;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) mine[t * C + j] = acc[t][j];
  __syncthreads();
  float* row = partial + (long long)blockIdx.x * 9 * C;
  for (int i = threadIdx.x; i < 9 * C; i += 256) {
    float v = 0.f;
    for (int k = 0; k < xt; ++k) v += lds[(long long)k * 9 * C + i];
    row[i] = v;
  }
}

// ---- the same pass as a strip walk without LDS memory or barriers ----------------------------------------------------
// dwblock_bwd_kernel above exchanges dy between the columns of a workgroup through LDS, with a barrier per row, and
// holds a 3x3 window per thread (~220 registers): two workgroups per CU that wait for each other sixteen times per
// strip.  Here a WAVE owns 16 adjacent columns (14 core columns + one halo column on either side) of four
// (column block, channel group) pairs, four channels per lane, and the neighbouring columns come from other lanes of
// the wave through ds_bpermute (the LDS crossbar; no LDS memory), so waves never wait for one another.  The pair index
// is the FASTEST lane index (lane = 4 * column + pair): where the pairs are adjacent channel groups of one column
// block, four adjacent lanes read 32 contiguous bytes, which the texture unit merges into one access -- with the
// column fastest every lane is an access of its own and the L1 tag rate, not HBM, bounds the kernel (measured: 64
// accesses per load instruction, 47 M per launch at 64 channels, 128 x 128, batch 32).
// Each tensor row is met ONCE, when it arrives: its dy values finish the output row above (its bottom taps), extend
// the output row itself (centre term, middle taps) and start the output row below (top taps), so no window is kept,
// only two partial output rows.  The weight gradient pairs a dy row with the block input of the rows above, at and
// below it, which is why it runs one row behind the arrivals.  The four pairs of a wave belong to ONE strip: row
// counters, the image-row test and the address steps are then wave-uniform and run on the scalar unit.  Per-lane
// arithmetic and its order are those of dwblock_bwd_kernel: dx is bit-identical; the weight-gradient partial rows are
// one per (strip, column block) instead of one per workgroup.
#define DWS_CORE 14
#define DWS_PF 3      // a row's loads are issued this many steps before its use
// Two channels of a lane side by side: the multiply-adds below are written on pairs so that they are v_pk_fma_f32
// (two FMAs per lane and issue) whatever the compiler's own pairing of scalar code would have found.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 sel2(bool k, f2 a) { return k ? a : (f2){0.f, 0.f}; }
__device__ __forceinline__ float lane_read(int byte_addr, float v) {          // v of lane byte_addr / 4
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}
// Per-lane sums v[N][4] (N quantities x this lane's four channels; lanes that own nothing hold zeros) -> one row of
// table[rows][N][C].  The 16 columns of a pair are added by a butterfly that halves what a lane holds at every stage
// (lanes with the stage's bit clear keep the first half of their values and take the partner's first half, the others
// the second): 4N values cost about 4N lane reads instead of 16N, and the tail is a seventh of a wave's instructions
// otherwise.  Fixed order.  With two channel groups (C = 8) the wave's two column blocks are added as well, so that the
// table has one row per strip and PAIR of column blocks (half as many rows for the finalisation launch to add up, which
// is a chain of loads per row).
__host__ __device__ __forceinline__ int dws_rows_per_strip(int nsx, int ncg) { return ncg == 2 ? (nsx + 1) / 2 : nsx; }
template <int N>
__device__ __forceinline__ void dws_store_row(float (&v)[N][4], int lane, bool live, int ncg, int nsx, long long strip,
                                              int sx, int C, int c0, float* __restrict__ table) {
  if constexpr (N <= 2) {
    // few values: the plain tree (every value over the 16 columns, lanes 0..3 end up with the totals); the butterfly's
    // bookkeeping would cost the forward kernel its 128th register and with it a wave per SIMD
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = v[k][j];
#pragma unroll
        for (int sh8 = 32; sh8 >= 4; sh8 >>= 1) {
          const float o = lane_read((lane + sh8) * 4, a);
          a += (lane + sh8 < 64) ? o : 0.f;
        }
        const float o2 = lane_read((lane + 2) * 4, a);
        a += ncg == 2 ? o2 : 0.f;
        v[k][j] = a;
      }
    if (live & (lane < (ncg == 2 ? 2 : 4))) {
      const long long row = ncg == 2 ? strip * ((nsx + 1) / 2) + (sx >> 1) : strip * nsx + sx;
      float* dst = table + (row * N) * C + c0;
#pragma unroll
      for (int k = 0; k < N; ++k) *(float4*)(dst + (long long)k * C) = make_float4(v[k][0], v[k][1], v[k][2], v[k][3]);
    }
    return;
  }
  constexpr int N4 = N * 4, NP = (N4 + 15) / 16 * 16, F = NP / 16;
  float a[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) a[i] = i < N4 ? v[i / 4][i % 4] : 0.f;
  int base = 0;                                               // a[i] is the sum of value base + i
#pragma unroll
  for (int st = 0; st < 4; ++st) {                            // lane bits 5, 4, 3, 2: the column index
    const int half = NP >> (st + 1);
    const int bit = 5 - st;
    const bool up = (lane >> bit) & 1;
    const int partner = (lane ^ (1 << bit)) * 4;
#pragma unroll
    for (int i = 0; i < half; ++i) {
      const float send = up ? a[i] : a[half + i];
      const float keep = up ? a[half + i] : a[i];
      a[i] = keep + lane_read(partner, send);
    }
    base += up ? half : 0;
  }
  // (selects, not branches, up to here: with a branch in this tail the compiler once stopped pairing the callers'
  //  multiply-adds into packed instructions in their main loops)
#pragma unroll
  for (int i = 0; i < F; ++i) {
    const float o2 = lane_read((lane + 2) * 4, a[i]);         // lanes 0 and 1 of a column use it; lanes 2 and 3 exist
    a[i] += ncg == 2 ? o2 : 0.f;
  }
  const bool writer = live & ((lane & 3) < (ncg == 2 ? 2 : 4));
  const long long row = ncg == 2 ? strip * ((nsx + 1) / 2) + (sx >> 1) : strip * nsx + sx;
  float* dst = table + (row * N) * C + c0;
#pragma unroll
  for (int i = 0; i < F; ++i) {
    const int idx = base + i;
    if (writer & (idx < N4)) dst[(long long)(idx >> 2) * C + (idx & 3)] = a[i];
  }
}

struct DwsGeo {
  long long strips, waves;
  int nsx, ncg, wps, R;     // column blocks, channel groups, waves per strip, output rows per strip
};
static inline DwsGeo dws_geo_r(int B, int H, int W, int C, int R) {
  DwsGeo g;
  g.R = R;
  g.strips = ((long long)B * (H + 1) + R - 1) / R;
  g.nsx = (W + 1 + DWS_CORE - 1) / DWS_CORE;
  g.ncg = C / 4;
  g.wps = (g.nsx * g.ncg + 3) / 4;
  g.waves = g.strips * g.wps;
  return g;
}
// Strips of 18 rows: 21 rows are read for 18.  (Strips of 9 rows double the number of waves at the small levels and
// were measured slower there, 33.9 against 27.2 us at 256 channels, 32 x 32, batch 32: a wave's fixed cost outweighs
// the shorter walk.)
#define DWS_R 18
static inline DwsGeo dws_geo(int B, int H, int W, int C) { return dws_geo_r(B, H, W, C, DWS_R); }

// PREV: the block's input x is the output of another depthwise block (x = relu(bn(yq) + its input)) and dx is that
// block's upstream gradient.  Its BatchNorm-backward sums (sum d, sum d * xhat with d = dx where x > 0, what
// bn_res_relu_bwd_reduce_kernel makes from dx, yq and its input) are then taken here, from the dx values as they are
// stored and one more tensor read (yq), instead of a pass of their own over three tensors.
struct DwsPrev {
  const char* yq;
  int cs, co;
  const float *mean, *invstd;
  float* partial;       // [strips * column blocks][2][C]
};

// COLS: also the column sums of dx as stored (dx is the output gradient of the 2x2 stride-2 convolution in front of
// the block: its bias gradient, lunext.py:139-154) in colsum[strips * column blocks][C], for uh_colsum_finalize.
template <int R, bool PREV, bool COLS>
__global__ __launch_bounds__(256) void dwblock_bwd_strip_kernel(
    long long Mp, int C, int Wp, int Hp, long long waves, int nsx, int wps, const char* __restrict__ da, int da_cs,
    int da_co, const char* __restrict__ y, int y_cs, int y_co, const char* __restrict__ x, int x_cs, int x_co,
    const float* __restrict__ wt, int c_real, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ coef,
    char* __restrict__ dx, int dx_cs, int dx_co, float* __restrict__ partial, DwsPrev pv, float* __restrict__ colsum) {
  const int ncg = C / 4;
  const int lane = threadIdx.x & 63;
  const int xi = lane >> 2;
  // workgroups go to the eight XCDs in turn: each XCD gets a contiguous range of strips, so the halo rows and columns
  // that neighbouring waves share are found in its own L2 (the grid is a multiple of 8)
  const long long blk = (long long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const long long wv = blk * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wv >= waves) return;                                    // whole waves leave; nothing below waits for another wave
  const long long strip = wv / wps;
  const int q = (int)(wv % wps) * 4 + (lane & 3);
  const int left_lane = (lane - 4) * 4, right_lane = (lane + 4) * 4;          // byte addresses for ds_bpermute
  const bool live = q < nsx * ncg;
  const int qc = live ? q : 0;
  const int cg = qc % ncg, sx = qc / ncg;
  const int c0 = cg * 4;
  const long long rows = Mp / Wp;
  const long long r0 = strip * R;
  const int xcol = sx * DWS_CORE - 1 + xi;                    // this lane's column; -1 and >= Wp - 1 are never interior
  const bool core = live & (xi >= 1) & (xi <= DWS_CORE) & (xcol < Wp);  // columns this lane writes (pad column included)
  const bool col_in = live & (xcol >= 0) & (xcol < Wp - 1);   // an interior column of the images
  float acc[9][4], w[9][4], sc[4], sh[4], bp[4], cq[4];
  f2 sc2[2], sh2[2], bp2[2], cq2[2];
  // step e (0 .. R + 2) meets tensor row r0 - 1 + e.  The row before the first strip is a guard row (zeros, and
  // counts as a pad row); past the last row the walk stays on the first row after the end (a guard row as well) and
  // everything there counts as padding.  All of this is the same for the whole wave.
  const long long left_ll = rows - r0;
  const int left = (int)(left_ll < (1 << 20) ? left_ll : (1 << 20));       // rows from r0 to the end of the tensor
  const int nr = left < R ? left : R;
  const long long m0 = (r0 - 1) * Wp + xcol;
  const char* dap = da + (m0 * da_cs + da_co + c0) * 2;
  const char* yp = y + (m0 * y_cs + y_co + c0) * 2;
  const char* xp = x + (m0 * x_cs + x_co + c0) * 2;
  const long long da_row = (long long)da_cs * 2 * Wp, y_row = (long long)y_cs * 2 * Wp, x_row = (long long)x_cs * 2 * Wp;
  int ry_f = (int)((r0 + Hp - 1) % Hp);                       // image row of the row the next fetch reads
  int ry_u = ry_f;                                            // ... and of the row the next step uses
  auto fetch = [&](int e, DwbRaw& t) {
    t.g = *(const U2*)dap;
    t.v = *(const U2*)yp;
    t.r = *(const U2*)xp;
    const bool in = e <= left;
    dap += in ? da_row : 0;
    yp += in ? y_row : 0;
    xp += in ? x_row : 0;
  };
  // (pairs of channels, [h] = channels 2h and 2h + 1 of the lane's four: see fma2)
  auto eval = [&](const DwbRaw& t, bool keep, f2 (&dyv)[2], f2 (&dc)[2], f2 (&xc)[2]) {
    float gf[4], vf[4], xf[4];
    unpack4(t.g, gf);
    unpack4h(t.v, vf);
    unpack4(t.r, xf);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f2 g2 = {gf[2 * h], gf[2 * h + 1]}, v2 = {vf[2 * h], vf[2 * h + 1]}, x2 = {xf[2 * h], xf[2 * h + 1]};
      const f2 pre = fma2(v2, sc2[h], sh2[h]) + x2;
      const f2 d = {(keep & (pre.x > 0.f)) ? g2.x : 0.f, (keep & (pre.y > 0.f)) ? g2.y : 0.f};
      dc[h] = d;
      xc[h] = sel2(keep, x2);
      dyv[h] = sel2(keep, fma2(sc2[h], d, -fma2(v2, cq2[h], bp2[h])));
    }
  };
  DwbRaw raw[DWS_PF];
#pragma unroll
  for (int e = 0; e < DWS_PF; ++e) fetch(e, raw[e]);
  // yq of the row of step e + 1 is read at step e and used at step e + 3, when that row's dx is stored
  const char* yqp = PREV ? pv.yq + (((r0 * Wp + xcol) * pv.cs + pv.co + c0) * 2) : nullptr;
  const long long yq_row = PREV ? (long long)pv.cs * 2 * Wp : 0;
  U2 rawq[3] = {};
  float qs[2][4], qmu[4], qis[4], cs[1][4] = {{0.f, 0.f, 0.f, 0.f}};
  if constexpr (PREV) {
    const float4 m4 = *(const float4*)(pv.mean + c0), i4 = *(const float4*)(pv.invstd + c0);
    qmu[0] = m4.x; qmu[1] = m4.y; qmu[2] = m4.z; qmu[3] = m4.w;
    qis[0] = i4.x; qis[1] = i4.y; qis[2] = i4.z; qis[3] = i4.w;
#pragma unroll
    for (int j = 0; j < 4; ++j) qs[0][j] = qs[1][j] = 0.f;
  }
  // (the rows' first loads are on their way while the parameters are read)
  if (c0 + 4 <= c_real) {                                     // 36 consecutive floats: nine 16-byte loads
    const float4* wp4 = (const float4*)(wt + c0 * 9);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const float4 v4 = wp4[i];
      const float f[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) w[(i * 4 + k) % 9][(i * 4 + k) / 9] = f[k];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int c = c0 + j;
        const float wv9 = wt[(c < c_real ? c : c_real - 1) * 9 + t];
        w[t][j] = (c < c_real) ? wv9 : 0.f;
      }
  }
  {
    const float4 s4 = *(const float4*)(scale + c0), h4 = *(const float4*)(shift + c0), i4 = *(const float4*)(invstd + c0),
                 m4 = *(const float4*)(mean + c0), a4 = *(const float4*)(coef + c0), b4 = *(const float4*)(coef + C + c0);
    const float s_[4] = {s4.x, s4.y, s4.z, s4.w}, h_[4] = {h4.x, h4.y, h4.z, h4.w}, i_[4] = {i4.x, i4.y, i4.z, i4.w},
                m_[4] = {m4.x, m4.y, m4.z, m4.w}, a_[4] = {a4.x, a4.y, a4.z, a4.w}, b_[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sc[j] = s_[j];
      sh[j] = h_[j];
      cq[j] = sc[j] * b_[j] * i_[j];
      bp[j] = sc[j] * a_[j] - cq[j] * m_[j];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      sc2[h] = (f2){sc[2 * h], sc[2 * h + 1]};
      sh2[h] = (f2){sh[2 * h], sh[2 * h + 1]};
      cq2[h] = (f2){cq[2 * h], cq[2 * h + 1]};
      bp2[h] = (f2){bp[2 * h], bp[2 * h + 1]};
    }
  }
  f2 xr[3][2];                                                // block input of the rows of steps e - 2, e - 1, e (mod 3)
  f2 part_a[2], part_b[2];                                    // output rows e - 2 (top and middle taps in) and e - 1 (top taps in)
  f2 hc[2], hd[2], hl[2], hr[2];                              // row of step e - 1: dy and centre term at this column, dy left, dy right
  f2 acc2[9][2], w2[9][2];
  bool ok_a = false, ok_h = false;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f2 z = {0.f, 0.f};
    xr[0][h] = xr[1][h] = xr[2][h] = z;
    part_a[h] = part_b[h] = hc[h] = hd[h] = hl[h] = hr[h] = z;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      acc2[t][h] = z;
      w2[t][h] = (f2){w[t][2 * h], w[t][2 * h + 1]};
    }
  }
  char* op = dx + (((r0 - 3) * Wp + xcol) * dx_cs + dx_co + c0) * 2;         // step e finishes the row of step e - 2
  const long long orow = (long long)dx_cs * 2 * Wp;
  // Step e has two halves.  The first evaluates the row that arrives (step e's) and sends its dy to the neighbouring
  // lanes; the second works on the row of step e - 1, whose exchanged values have had a whole step to come back:
  // bottom taps finish output row e - 2, centre term and middle taps extend row e - 1, top taps start row e, and the
  // weight gradient pairs it with the block input of rows e - 2, e - 1 and e.  Rows that do not exist (before step 0,
  // after step R + 1) are rows of zeros.
  auto step = [&](auto ph, auto with_fetch, int e) {
    constexpr int S0 = decltype(ph)::value;                   // e mod 3
    constexpr int S1 = (S0 + 2) % 3, S2 = (S0 + 1) % 3;       // slots of steps e - 1 and e - 2
    const bool row_ok = (e <= left) & (ry_u != Hp - 1) & (e <= R + 1);
    ry_u = ry_u + 1 == Hp ? 0 : ry_u + 1;
    const bool ok = row_ok & col_in;
    f2 c[2], dc[2], xc[2], nl[2], nr2[2];
    eval(raw[S0], ok, c, dc, xc);
    if constexpr (decltype(with_fetch)::value) fetch(e + DWS_PF, raw[S0]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {                             // (the first and last column of the wave read some other
      nl[h] = (f2){lane_read(left_lane, c[h].x), lane_read(left_lane, c[h].y)};      // lane: their taps are incomplete
      nr2[h] = (f2){lane_read(right_lane, c[h].x), lane_read(right_lane, c[h].y)};   // anyway, see `core`)
    }
    const bool own_row = (e >= 1) & (e <= R);                 // only the strip's own rows add their block input
#pragma unroll
    for (int h = 0; h < 2; ++h) xr[S0][h] = sel2(core & own_row, xc[h]);
    // ---- second half: the row of step e - 1 (tap d of a row pairs with the column x + 1 - d)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      acc2[0][h] = fma2(xr[S2][h], hr[h], acc2[0][h]);
      acc2[1][h] = fma2(xr[S2][h], hc[h], acc2[1][h]);
      acc2[2][h] = fma2(xr[S2][h], hl[h], acc2[2][h]);
      acc2[3][h] = fma2(xr[S1][h], hr[h], acc2[3][h]);
      acc2[4][h] = fma2(xr[S1][h], hc[h], acc2[4][h]);
      acc2[5][h] = fma2(xr[S1][h], hl[h], acc2[5][h]);
      acc2[6][h] = fma2(xr[S0][h], hr[h], acc2[6][h]);
      acc2[7][h] = fma2(xr[S0][h], hc[h], acc2[7][h]);
      acc2[8][h] = fma2(xr[S0][h], hl[h], acc2[8][h]);
    }
    f2 o[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f2 v = part_a[h];                                       // the bottom taps finish output row e - 2
      v = fma2(hr[h], w2[0][h], v);
      v = fma2(hc[h], w2[1][h], v);
      v = fma2(hl[h], w2[2][h], v);
      o[h] = sel2(ok_a, v);                                   // (the window around a pad position holds real neighbours)
      v = part_b[h] + hd[h];                                  // centre term and middle taps: output row e - 1
      v = fma2(hr[h], w2[3][h], v);
      v = fma2(hc[h], w2[4][h], v);
      part_a[h] = fma2(hl[h], w2[5][h], v);
      f2 t = fma2(hr[h], w2[6][h], (f2){0.f, 0.f});           // top taps: output row e starts
      t = fma2(hc[h], w2[7][h], t);
      part_b[h] = fma2(hl[h], w2[8][h], t);
    }
    U2 ov;
    ov.x = pack2(o[0].x, o[0].y);
    ov.y = pack2(o[1].x, o[1].y);
    const bool st = core & (e >= 3) & (e - 2 <= nr);
    if (st) *(U2*)op = ov;
    op += orow;
    if constexpr (COLS) {
      float dr[4];
      unpack4(ov, dr);
#pragma unroll
      for (int j = 0; j < 4; ++j) cs[0][j] += st ? dr[j] : 0.f;
    }
    if constexpr (PREV) {
      // xr of the stored row is the block input where this lane stores and 0 elsewhere (other lanes' columns, other
      // strips' rows, padding): exactly the positions whose dx counts, once
      float dr[4], vq[4];
      unpack4(ov, dr);
      unpack4h(rawq[S2], vq);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = xr[S2][j >> 1][j & 1] > 0.f ? dr[j] : 0.f;
        qs[0][j] += d;
        qs[1][j] += d * (vq[j] - qmu[j]) * qis[j];
      }
      if constexpr (decltype(with_fetch)::value) {
        rawq[S2] = *(const U2*)yqp;
        yqp += (e + 1 < left) ? yq_row : 0;
      }
    }
    ok_a = ok_h;
    ok_h = ok;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      hc[h] = c[h];
      hd[h] = dc[h];
      hl[h] = nl[h];
      hr[h] = nr2[h];
    }
  };
  static_assert(R % 3 == 0 && DWS_PF == 3, "the rings are walked three steps at a time");
  using Yes = std::true_type;
  using No = std::false_type;
  // (the last loop pass fetches the row of step R + 2, which nothing uses: one row in R + 2 that the next strip
  //  reads anyway, for a loop body without a condition)
#pragma unroll 1
  for (int e = 0; e < R; e += 3) {
    step(std::integral_constant<int, 0>{}, Yes{}, e);
    step(std::integral_constant<int, 1>{}, Yes{}, e + 1);
    step(std::integral_constant<int, 2>{}, Yes{}, e + 2);
  }
  step(std::integral_constant<int, 0>{}, No{}, R);
  step(std::integral_constant<int, 1>{}, No{}, R + 1);
  step(std::integral_constant<int, 2>{}, No{}, R + 2);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      acc[t][2 * h] = acc2[t][h].x;
      acc[t][2 * h + 1] = acc2[t][h].y;
    }
  dws_store_row<9>(acc, lane, live, ncg, nsx, strip, sx, C, c0, partial);
  if constexpr (PREV) dws_store_row<2>(qs, lane, live, ncg, nsx, strip, sx, C, c0, pv.partial);
  if constexpr (COLS) dws_store_row<1>(cs, lane, live, ncg, nsx, strip, sx, C, c0, colsum);
}

// ---- the depthwise forward of a training step as the same kind of walk ------------------------------------------------
// y = fp16(dwconv3x3(x) + bias) and the BatchNorm sums (sum y, sum y^2) per (strip, column block).  A row of x is met
// once: it starts the output row below it (top taps), extends its own (middle taps) and finishes the one above (bottom
// taps) -- the taps of an output are added in dwconv_fwd_kernel's order (bias, then row-major), so y has the same bits.
// One 8-byte load per lane and row instead of three 16-byte ones; lanes, strips and the one-step delay of the exchanged
// values as in dwblock_bwd_strip_kernel.
// APPLY: the convolution's input is the output of the block before, a = relu(scale_q * yq + shift_q + x), which no
// launch has made yet: it is evaluated where the rows arrive (from yq and x, two loads instead of one), rounded to bf16
// as the stored tensor is, written out by the lanes and strips that own the position, and convolved -- what
// uh_bn_res_relu_apply followed by the plain walk computes, bit for bit, without reading a back.
struct DwfApply {
  const char* yq;
  int cs, co;
  const float *scale, *shift;
  char* a;
  int a_cs, a_co;
};

template <int R, bool APPLY>
__global__ __launch_bounds__(256) void dwconv_fwd_strip_kernel(
    long long Mp, int C, int Wp, int Hp, long long waves, int nsx, int wps, const char* __restrict__ x, int x_cs, int x_co,
    const float* __restrict__ wt, int c_real, const float* __restrict__ bias, char* __restrict__ out, int o_cs, int o_co,
    float* __restrict__ partial, DwfApply ap) {
  const int ncg = C / 4;
  const int lane = threadIdx.x & 63;
  const int xi = lane >> 2;
  const long long blk = (long long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const long long wv = blk * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wv >= waves) return;
  const long long strip = wv / wps;
  const int q = (int)(wv % wps) * 4 + (lane & 3);
  const int left_lane = (lane - 4) * 4, right_lane = (lane + 4) * 4;
  const bool live = q < nsx * ncg;
  const int qc = live ? q : 0;
  const int cg = qc % ncg, sx = qc / ncg;
  const int c0 = cg * 4;
  const long long rows = Mp / Wp;
  const long long r0 = strip * R;
  const int xcol = sx * DWS_CORE - 1 + xi;
  const bool core = live & (xi >= 1) & (xi <= DWS_CORE) & (xcol < Wp);
  const bool col_in = live & (xcol >= 0) & (xcol < Wp - 1);
  const long long left_ll = rows - r0;
  const int left = (int)(left_ll < (1 << 20) ? left_ll : (1 << 20));
  const int nr = left < R ? left : R;
  const char* xp = x + ((((r0 - 1) * Wp + xcol) * x_cs + x_co + c0) * 2);
  const long long x_row = (long long)x_cs * 2 * Wp;
  const char* yqp = APPLY ? ap.yq + ((((r0 - 1) * Wp + xcol) * ap.cs + ap.co + c0) * 2) : nullptr;
  const long long yq_row = APPLY ? (long long)ap.cs * 2 * Wp : 0;
  U2 rawq[DWS_PF] = {};
  auto fetch = [&](int e, int slot, U2 (&rx)[DWS_PF], U2 (&rq)[DWS_PF]) {
    rx[slot] = *(const U2*)xp;
    xp += (e <= left) ? x_row : 0;
    if constexpr (APPLY) {
      rq[slot] = *(const U2*)yqp;
      yqp += (e <= left) ? yq_row : 0;
    }
  };
  U2 raw[DWS_PF];
#pragma unroll
  for (int e = 0; e < DWS_PF; ++e) fetch(e, e, raw, rawq);
  float qsc[4], qsh[4];
  char* aptr = APPLY ? ap.a + ((((r0 - 1) * Wp + xcol) * ap.a_cs + ap.a_co + c0) * 2) : nullptr;
  const long long a_row = APPLY ? (long long)ap.a_cs * 2 * Wp : 0;
  if constexpr (APPLY) {
    const float4 s4 = *(const float4*)(ap.scale + c0), h4 = *(const float4*)(ap.shift + c0);
    qsc[0] = s4.x; qsc[1] = s4.y; qsc[2] = s4.z; qsc[3] = s4.w;
    qsh[0] = h4.x; qsh[1] = h4.y; qsh[2] = h4.z; qsh[3] = h4.w;
  }
  float w[9][4], b[4], acc[2][4];
  if (c0 + 4 <= c_real) {
    const float4* wp4 = (const float4*)(wt + c0 * 9);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const float4 v4 = wp4[i];
      const float f[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) w[(i * 4 + k) % 9][(i * 4 + k) / 9] = f[k];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int c = c0 + j;
        const float wv9 = wt[(c < c_real ? c : c_real - 1) * 9 + t];
        w[t][j] = (c < c_real) ? wv9 : 0.f;
      }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + j;
    const float bv = bias ? bias[c < c_real ? c : c_real - 1] : 0.f;
    b[j] = ((bias != nullptr) & (c < c_real)) ? bv : 0.f;
    acc[0][j] = acc[1][j] = 0.f;
  }
  int ry_u = (int)((r0 + Hp - 1) % Hp);                       // image row of the row the next step meets
  // pairs of channels from here on: [h] = channels 2h, 2h + 1 of the lane's four
  f2 w2[9][2], b2[2], acc2[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int t = 0; t < 9; ++t) w2[t][h] = (f2){w[t][2 * h], w[t][2 * h + 1]};
    b2[h] = (f2){b[2 * h], b[2 * h + 1]};
    acc2[0][h] = acc2[1][h] = (f2){0.f, 0.f};
  }
  f2 part_a[2], part_b[2], hc[2], hl[2], hr[2];               // output rows e - 2 and e - 1 so far; x of the row of step e - 1
  bool ok_a = false, ok_h = false;
#pragma unroll
  for (int h = 0; h < 2; ++h) part_a[h] = part_b[h] = hc[h] = hl[h] = hr[h] = (f2){0.f, 0.f};
  char* op = out + (((r0 - 3) * Wp + xcol) * o_cs + o_co + c0) * 2;          // step e finishes the row of step e - 2
  const long long orow = (long long)o_cs * 2 * Wp;
  auto step = [&](auto ph, auto with_fetch, int e) {
    constexpr int S0 = decltype(ph)::value;
    const bool ok = (e <= left) & (ry_u != Hp - 1) & (e <= R + 1) & col_in;
    ry_u = ry_u + 1 == Hp ? 0 : ry_u + 1;
    float c[4], nl[4], nr4[4];
    unpack4(raw[S0], c);
    if constexpr (APPLY) {
      float vq[4];
      unpack4h(rawq[S0], vq);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = fmaxf(vq[j] * qsc[j] + qsh[j] + c[j], 0.f);
        c[j] = ok ? t : 0.f;
      }
      U2 av;
      av.x = pack2(c[0], c[1]);
      av.y = pack2(c[2], c[3]);
      unpack4(av, c);                                         // the convolution sees the values as they are stored
      if (core & (e >= 1) & (e <= nr)) *(U2*)aptr = av;
      aptr += a_row;
    }
    if constexpr (decltype(with_fetch)::value) fetch(e + DWS_PF, S0, raw, rawq);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      nl[j] = lane_read(left_lane, c[j]);
      nr4[j] = lane_read(right_lane, c[j]);
    }
    f2 v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f2 t = part_a[h];                                       // the bottom taps finish output row e - 2
      t = fma2(hl[h], w2[6][h], t);
      t = fma2(hc[h], w2[7][h], t);
      t = fma2(hr[h], w2[8][h], t);
      v[h] = sel2(ok_a, t);
      t = part_b[h];                                          // middle taps: output row e - 1
      t = fma2(hl[h], w2[3][h], t);
      t = fma2(hc[h], w2[4][h], t);
      part_a[h] = fma2(hr[h], w2[5][h], t);
      t = fma2(hl[h], w2[0][h], b2[h]);                       // bias and top taps: output row e starts
      t = fma2(hc[h], w2[1][h], t);
      part_b[h] = fma2(hr[h], w2[2][h], t);
    }
    const bool st = core & (e >= 3) & (e - 2 <= nr);
    if (st) {
      U2 ov;
      ov.x = pack2h(v[0].x, v[0].y);
      ov.y = pack2h(v[1].x, v[1].y);
      *(U2*)op = ov;
    }
    op += orow;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f2 u = sel2(st, v[h]);
      acc2[0][h] += u;
      acc2[1][h] = fma2(u, u, acc2[1][h]);
      hc[h] = (f2){c[2 * h], c[2 * h + 1]};
      hl[h] = (f2){nl[2 * h], nl[2 * h + 1]};
      hr[h] = (f2){nr4[2 * h], nr4[2 * h + 1]};
    }
    ok_a = ok_h;
    ok_h = ok;
  };
  static_assert(R % 3 == 0 && DWS_PF == 3, "the load ring is walked three steps at a time");
  using Yes = std::true_type;
  using No = std::false_type;
#pragma unroll 1
  for (int e = 0; e < R; e += 3) {
    step(std::integral_constant<int, 0>{}, Yes{}, e);
    step(std::integral_constant<int, 1>{}, Yes{}, e + 1);
    step(std::integral_constant<int, 2>{}, Yes{}, e + 2);
  }
  step(std::integral_constant<int, 0>{}, No{}, R);
  step(std::integral_constant<int, 1>{}, No{}, R + 1);
  step(std::integral_constant<int, 2>{}, No{}, R + 2);
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      acc[k][2 * h] = acc2[k][h].x;
      acc[k][2 * h + 1] = acc2[k][h].y;
    }
  dws_store_row<2>(acc, lane, live, ncg, nsx, strip, sx, C, c0, partial);
}

// UNET_AMD_DWFWD_TILES=1: the training forward as dwconv_fwd_kernel (one thread per column, three loads per row)
static bool dwf_use_strip(int C) {
  const char* e = getenv("UNET_AMD_DWFWD_TILES");
  return (!e || atoi(e) == 0) && C >= 8 && C <= 512 && (C & (C - 1)) == 0;
}

// rows of the partial-sum table uh_dwconv3x3_fwd writes in RAW_STATS mode (for uh_bn_finalize)
extern "C" int uh_dwconv3x3_fwd_parts(int B, int H, int W, int C) {
  if (!(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && C / 8 <= 256)) return 0;
  if (dwf_use_strip(C)) {
    const DwsGeo g = dws_geo(B, H, W, C);
    return (int)(g.strips * dws_rows_per_strip(g.nsx, g.ncg));
  }
  return (int)dw_grid(B, H, W, C).x;
}

static int dwconv_fwd_strip_launch(int B, int H, int W, int C, const void* x, int x_cstride, int x_coff, const float* w,
                                   int c_real, const float* bias, void* out, int o_cstride, int o_coff, float* partial,
                                   uh_stream s, const DwfApply* ap) {
  UH_REQUIRE(x_cstride % 4 == 0 && x_coff % 4 == 0 && o_cstride % 4 == 0 && o_coff % 4 == 0, "dwconv3x3_fwd: bad views");
  const DwsGeo g = dws_geo(B, H, W, C);
  UH_REQUIRE(g.strips * g.nsx < (1ll << 31) && (g.waves + 3) / 4 + 8 < (1ll << 31), "dwconv3x3_fwd: %lld waves", g.waves);
  auto* kern = ap ? dwconv_fwd_strip_kernel<DWS_R, true> : dwconv_fwd_strip_kernel<DWS_R, false>;
  hipLaunchKernelGGL(kern, dim3((unsigned)(((g.waves + 3) / 4 + 7) / 8 * 8)), dim3(256), 0,
                     (hipStream_t)s, uh_mp(B, H, W), C, W + 1, H + 1, g.waves, g.nsx, g.wps, (const char*)x, x_cstride,
                     x_coff, w, c_real, bias, (char*)out, o_cstride, o_coff, partial, ap ? *ap : DwfApply{});
  return uh_check_launch("dwconv3x3_fwd");
}

// 1 when uh_dwconv3x3_fwd_applied can be called for C channels (the strip walk is in use), else 0
extern "C" int uh_dwconv3x3_fwd_takes_apply(int C) { return dwf_use_strip(C) ? 1 : 0; }

// The training forward of the SECOND block of a pair: its input a = relu(scale_q * yq + shift_q + xq) (the first block's
// BatchNorm + residual + ReLU, uh_bn_res_relu_apply) is made here, written to `a`, and convolved:
// out = fp16(dwconv3x3(a) + bias), partial[uh_dwconv3x3_fwd_parts][2][C] = (sum, sum of squares).  Same bits as the two
// launches it replaces.
extern "C" int uh_dwconv3x3_fwd_applied(int B, int H, int W, int C, const void* yq, int yq_cstride, int yq_coff,
                                        const void* xq, int xq_cstride, int xq_coff, const float* scale_q,
                                        const float* shift_q, void* a, int a_cstride, int a_coff, const float* w,
                                        int c_real, const float* bias, void* out, int o_cstride, int o_coff,
                                        float* partial, uh_stream s) {
  EW_CHECK("dwconv3x3_fwd_applied", C);
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && yq && xq && scale_q && shift_q && a && w && out && partial,
             "dwconv3x3_fwd_applied: bad arguments");
  UH_REQUIRE(dwf_use_strip(C), "dwconv3x3_fwd_applied: C=%d needs the strip walk (a power of two in [8, 512], "
             "UNET_AMD_DWFWD_TILES unset)", C);
  UH_REQUIRE(c_real > 0 && c_real <= C && ((uintptr_t)w & 15) == 0, "dwconv3x3_fwd_applied: c_real=%d of C=%d, w 16-byte aligned", c_real, C);
  UH_REQUIRE(yq_cstride % 4 == 0 && yq_coff % 4 == 0 && a_cstride % 4 == 0 && a_coff % 4 == 0 && a != xq && a != out,
             "dwconv3x3_fwd_applied: bad views");
  const DwfApply ap{(const char*)yq, yq_cstride, yq_coff, scale_q, shift_q, (char*)a, a_cstride, a_coff};
  return dwconv_fwd_strip_launch(B, H, W, C, xq, xq_cstride, xq_coff, w, c_real, bias, out, o_cstride, o_coff, partial, s, &ap);
}

// UNET_AMD_DWBLOCK_LDS=1: the workgroup kernel with the LDS exchange instead of the strip walk.  Read at every call
// (uh_dwblock_bwd_parts and the launch that follows it see the same value), so a test can compare the two in one process.
static bool dwb_use_strip() {
  const char* e = getenv("UNET_AMD_DWBLOCK_LDS");
  return !e || atoi(e) == 0;
}

extern "C" int uh_dwblock_bwd_parts(int B, int H, int W, int C) {
  if (!(B > 0 && H > 0 && W > 0 && C >= 8 && C <= 512 && (C & (C - 1)) == 0)) return 0;
  if (dwb_use_strip()) {
    const DwsGeo g = dws_geo(B, H, W, C);
    return (int)(g.strips * dws_rows_per_strip(g.nsx, g.ncg));
  }
  return (int)dwb_grid(B, H, W, C).x;
}

static int dwblock_bwd_launch(const char* what, int B, int H, int W, int C, const void* da, int da_cstride, int da_coff,
                              const void* y, int y_cstride, int y_coff, const void* x, int x_cstride, int x_coff,
                              const float* w, int c_real, const float* mean, const float* invstd, const float* scale,
                              const float* shift, const float* coef, void* dx, int dx_cstride, int dx_coff,
                              float* partial, const DwsPrev* pv, float* colsum, uh_stream s) {
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && da && y && x && w && mean && invstd && scale && shift && coef && dx && partial,
             "%s: bad arguments", what);
  UH_REQUIRE(C >= 8 && C <= 512 && (C & (C - 1)) == 0, "%s: C=%d must be a power of two in [8,512]", what, C);
  UH_REQUIRE(c_real > 0 && c_real <= C, "%s: c_real=%d of C=%d", what, c_real, C);
  UH_REQUIRE(da_cstride % 4 == 0 && da_coff % 4 == 0 && y_cstride % 4 == 0 && y_coff % 4 == 0 && x_cstride % 4 == 0 &&
                 x_coff % 4 == 0 && dx_cstride % 4 == 0 && dx_coff % 4 == 0, "%s: bad views", what);
  const long long Mp = uh_mp(B, H, W);
  if (dwb_use_strip()) {
    const DwsGeo g = dws_geo(B, H, W, C);
    UH_REQUIRE(g.strips * g.nsx < (1ll << 31) && (g.waves + 3) / 4 + 8 < (1ll << 31), "%s: %lld waves", what, g.waves);
    UH_REQUIRE(!(pv && colsum), "%s: one extra output at a time", what);
    auto* kern = pv ? dwblock_bwd_strip_kernel<DWS_R, true, false>
                    : colsum ? dwblock_bwd_strip_kernel<DWS_R, false, true> : dwblock_bwd_strip_kernel<DWS_R, false, false>;
    hipLaunchKernelGGL(kern, dim3((unsigned)(((g.waves + 3) / 4 + 7) / 8 * 8)), dim3(256), 0, (hipStream_t)s, Mp, C,
                       W + 1, H + 1, g.waves, g.nsx, g.wps, (const char*)da, da_cstride, da_coff, (const char*)y, y_cstride,
                       y_coff, (const char*)x, x_cstride, x_coff, w, c_real, mean, invstd, scale, shift, coef, (char*)dx,
                       dx_cstride, dx_coff, partial, pv ? *pv : DwsPrev{}, colsum);
    return uh_check_launch(what);
  }
  UH_REQUIRE(!pv && !colsum, "%s: only the strip walk takes further sums (UNET_AMD_DWBLOCK_LDS is set)", what);
  const int xt = 256 / (C / 4);
  hipLaunchKernelGGL(dwblock_bwd_kernel, dwb_grid(B, H, W, C), dim3(256), (size_t)xt * 9 * C * 4, (hipStream_t)s, Mp, C,
                     W + 1, H + 1, (const char*)da, da_cstride, da_coff, (const char*)y, y_cstride, y_coff,
                     (const char*)x, x_cstride, x_coff, w, c_real, mean, invstd, scale, shift, coef, (char*)dx,
                     dx_cstride, dx_coff, partial);
  return uh_check_launch(what);
}

extern "C" int uh_dwblock_bwd(int B, int H, int W, int C, const void* da, int da_cstride, int da_coff, const void* y,
                              int y_cstride, int y_coff, const void* x, int x_cstride, int x_coff, const float* w,
                              int c_real, const float* mean, const float* invstd, const float* scale,
                              const float* shift, const float* coef, void* dx, int dx_cstride, int dx_coff,
                              float* partial, uh_stream s) {
  return dwblock_bwd_launch("dwblock_bwd", B, H, W, C, da, da_cstride, da_coff, y, y_cstride, y_coff, x, x_cstride, x_coff,
                            w, c_real, mean, invstd, scale, shift, coef, dx, dx_cstride, dx_coff, partial, nullptr, nullptr, s);
}

// uh_dwblock_bwd that also leaves the column sums of dx as stored: colsum[uh_dwblock_bwd_parts][C], for
// uh_colsum_finalize(colsum, parts, C, ...).  Available when uh_dwblock_bwd_takes_prev() is 1.
extern "C" int uh_dwblock_bwd_colsum(int B, int H, int W, int C, const void* da, int da_cstride, int da_coff,
                                     const void* y, int y_cstride, int y_coff, const void* x, int x_cstride, int x_coff,
                                     const float* w, int c_real, const float* mean, const float* invstd,
                                     const float* scale, const float* shift, const float* coef, void* dx,
                                     int dx_cstride, int dx_coff, float* partial, float* colsum, uh_stream s) {
  UH_REQUIRE(colsum, "dwblock_bwd_colsum: bad arguments");
  return dwblock_bwd_launch("dwblock_bwd_colsum", B, H, W, C, da, da_cstride, da_coff, y, y_cstride, y_coff, x, x_cstride,
                            x_coff, w, c_real, mean, invstd, scale, shift, coef, dx, dx_cstride, dx_coff, partial, nullptr,
                            colsum, s);
}

// 1 when uh_dwblock_bwd_prev can be called (the strip walk is in use), else 0
extern "C" int uh_dwblock_bwd_takes_prev(void) { return dwb_use_strip() ? 1 : 0; }

// uh_dwblock_bwd for a block whose input x is the OUTPUT of another depthwise block (x = relu(bn(yq) + that block's
// input), nothing else reads it): also writes that block's BatchNorm-backward partial sums, the ones
// uh_bn_res_relu_bwd_reduce would make from dx, yq and its input, to prev_partial[uh_dwblock_bwd_parts][2][C]
// (uh_bn_bwd_finalize reduces them).  mean_q / invstd_q are that block's batch statistics.
extern "C" int uh_dwblock_bwd_prev(int B, int H, int W, int C, const void* da, int da_cstride, int da_coff, const void* y,
                                   int y_cstride, int y_coff, const void* x, int x_cstride, int x_coff, const float* w,
                                   int c_real, const float* mean, const float* invstd, const float* scale,
                                   const float* shift, const float* coef, void* dx, int dx_cstride, int dx_coff,
                                   float* partial, const void* yq, int yq_cstride, int yq_coff, const float* mean_q,
                                   const float* invstd_q, float* prev_partial, uh_stream s) {
  UH_REQUIRE(yq && mean_q && invstd_q && prev_partial && yq_cstride % 4 == 0 && yq_coff % 4 == 0 && yq_cstride >= yq_coff + C,
             "dwblock_bwd_prev: bad arguments");
  const DwsPrev pv{(const char*)yq, yq_cstride, yq_coff, mean_q, invstd_q, prev_partial};
  return dwblock_bwd_launch("dwblock_bwd_prev", B, H, W, C, da, da_cstride, da_coff, y, y_cstride, y_coff, x, x_cstride,
                            x_coff, w, c_real, mean, invstd, scale, shift, coef, dx, dx_cstride, dx_coff, partial, &pv, nullptr, s);
}

// partial[P][9][C] -> the module's own weight-gradient layout out[c_real][1][3][3] (padding channels dropped)
// A workgroup adds 32 adjacent columns of the table: 32 lanes read 128 contiguous bytes of a row, 32 row slots walk the
// rows, eight rows fetched before any is added (the strip-walk backward leaves thousands of rows: a chain of one load
// per row and 32-byte pieces took 20-45 us per launch).  Fixed order, double accumulators.
__device__ __forceinline__ void dw_wgrad_finalize_block(const float* __restrict__ partial, int P, int C, int c_real,
                                                        float* __restrict__ out, int blk, double* lds /*[DWFIN_LDS]*/) {
  const int ncol = 9 * C;
  const int t = threadIdx.x;
  if (ncol <= 128) {
    // few columns (the 8-channel level, whose tables are the longest): 8 columns and 128 row slots per workgroup, so
    // that the rows are spread over more threads (34 rows per thread instead of 136 at 4342 rows)
    const int c0 = blk * 8, c8 = t & 7, slot = t >> 3, col = c0 + c8;
    double s = 0.0;
    if (col < ncol) {
      const float* src = partial + col;
      int p = slot;
      for (; p + 7 * 128 < P; p += 8 * 128) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(long long)(p + u * 128) * ncol];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (double)v[u];
      }
      for (; p < P; p += 128) s += (double)src[(long long)p * ncol];
    }
    lds[slot * 8 + c8] = s;
    __syncthreads();
    for (int stride = 64; stride > 0; stride >>= 1) {
      if (slot < stride) lds[slot * 8 + c8] += lds[(slot + stride) * 8 + c8];
      __syncthreads();
    }
    if (t < 8 && c0 + t < ncol) {
      const int tap = (c0 + t) / C, c = (c0 + t) % C;
      if (c < c_real) out[c * 9 + tap] = (float)lds[t];
    }
    return;
  }
  const int c0 = blk * 32;
  const int cl = t & 31, slot = t >> 5;
  const int col = c0 + cl;
  double s = 0.0;
  if (col < ncol) {
    const float* src = partial + col;
    int p = slot;
    for (; p + 7 * 32 < P; p += 8 * 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(long long)(p + u * 32) * ncol];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; p < P; p += 32) s += (double)src[(long long)p * ncol];
  }
  lds[slot * 33 + cl] = s;
  __syncthreads();
  if (t < 32 && c0 + t < ncol) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) v += lds[k * 33 + t];
    const int tap = (c0 + t) / C, c = (c0 + t) % C;
    if (c < c_real) out[c * 9 + tap] = (float)v;
  }
}
#define DWFIN_LDS (32 * 33)
static inline int dw_wgrad_finalize_blocks(int C) { return 9 * C <= 128 ? (9 * C + 7) / 8 : (9 * C + 31) / 32; }

__global__ __launch_bounds__(1024) void dw_wgrad_finalize_kernel(const float* __restrict__ partial, int P, int C,
                                                                 int c_real, float* __restrict__ out) {
  __shared__ double lds[DWFIN_LDS];
  dw_wgrad_finalize_block(partial, P, C, c_real, out, blockIdx.x, lds);
}

extern "C" int uh_dw_wgrad_finalize(const float* partial, int P, int C, int c_real, float* out, uh_stream s) {
  UH_REQUIRE(partial && out && P > 0 && C > 0 && c_real > 0 && c_real <= C, "dw_wgrad_finalize: bad arguments");
  hipLaunchKernelGGL(dw_wgrad_finalize_kernel, dim3(dw_wgrad_finalize_blocks(C)), dim3(1024), 0, (hipStream_t)s, partial, P, C,
                     c_real, out);
  return uh_check_launch("dw_wgrad_finalize");
}

// the same for up to UH_DWFIN_MAX tables in one launch: the descriptors travel as kernel arguments
struct DwFinTable {
  uh_dwfin_desc d[UH_DWFIN_MAX];
  int first[UH_DWFIN_MAX + 1];      // first workgroup of each table
  int n;
};
__global__ __launch_bounds__(1024) void dw_wgrad_finalize_multi_kernel(DwFinTable t) {
  __shared__ double lds[DWFIN_LDS];
  int i = 0;
  while (i + 1 < t.n && (int)blockIdx.x >= t.first[i + 1]) ++i;            // (uniform over the workgroup)
  const uh_dwfin_desc& d = t.d[i];
  dw_wgrad_finalize_block(d.partial, d.parts, d.C, d.c_real, d.out, (int)blockIdx.x - t.first[i], lds);
}

extern "C" int uh_dw_wgrad_finalize_multi(const uh_dwfin_desc* descs, int count, uh_stream s) {
  UH_REQUIRE(descs && count > 0 && count <= UH_DWFIN_MAX, "dw_wgrad_finalize_multi: 1..%d tables, got %d", UH_DWFIN_MAX, count);
  DwFinTable t;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const uh_dwfin_desc& d = descs[i];
    UH_REQUIRE(d.partial && d.out && d.parts > 0 && d.C > 0 && d.c_real > 0 && d.c_real <= d.C,
               "dw_wgrad_finalize_multi: bad table %d", i);
    t.d[i] = d;
    t.first[i] = blocks;
    blocks += dw_wgrad_finalize_blocks(d.C);
  }
  t.first[count] = blocks;
  t.n = count;
  hipLaunchKernelGGL(dw_wgrad_finalize_multi_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)s, t);
  return uh_check_launch("dw_wgrad_finalize_multi");
}

// ---- attention gate, pointwise parts (C >= 64; z = [z1 | z2 | z3] pre-activations, 3C channels) ----------------------
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

__global__ __launch_bounds__(256) void gate_fwd_kernel(long long Mp, int C, const char* __restrict__ z, int z_cs,
                                                       int z_co, const char* __restrict__ x, int x_cs, int x_co,
                                                       char* __restrict__ out, int o_cs, int o_co) {
  const int cpr = C / 8, rpi = 256 / cpr;
  const int rr = threadIdx.x / cpr, cc = threadIdx.x % cpr;
  if (rr >= rpi) return;
  for (long long m = (long long)blockIdx.x * rpi + rr; m < Mp; m += (long long)gridDim.x * rpi) {
    float z1[8], z2[8], z3[8], xv[8];
    const char* zr = z + (m * z_cs + z_co + cc * 8) * 2;
    unpack8(*(const U4*)zr, z1);
    unpack8(*(const U4*)(zr + C * 2), z2);
    unpack8(*(const U4*)(zr + C * 4), z3);
    unpack8(*(const U4*)(x + (m * x_cs + x_co + cc * 8) * 2), xv);
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = sigmoidf_(z3[j]) * xv[j] + sigmoidf_(z1[j]) * tanhf(z2[j]);
    // pad rows: z holds zeros there (masked GEMM epilogue) and x is zero, so the result is 0.5*tanh(0) = 0
    *(U4*)(out + (m * o_cs + o_co + cc * 8) * 2) = pack8(xv);
  }
}

extern "C" int uh_gate_fwd(long long Mp, int C, const void* z, int z_cstride, int z_coff, const void* x, int x_cstride,
                           int x_coff, void* out, int o_cstride, int o_coff, uh_stream s) {
  EW_CHECK("gate_fwd", C);
  UH_REQUIRE(Mp > 0 && z && x && out && z_cstride >= z_coff + 3 * C, "gate_fwd: bad arguments");
  hipLaunchKernelGGL(gate_fwd_kernel, dim3(ew_blocks(Mp, C)), dim3(256), 0, (hipStream_t)s, Mp, C, (const char*)z,
                     z_cstride, z_coff, (const char*)x, x_cstride, x_coff, (char*)out, o_cstride, o_coff);
  return uh_check_launch("gate_fwd");
}

// dz1 = dout*c*b*(1-b), dz2 = dout*b*(1-c^2), dz3 = dout*x*a*(1-a), dxd = dout*a
// SUMS: also the column sums of dz as stored (the gradients of the gate's three biases, lunext.py:88-95), one
// partial row [3][C] per workgroup for uh_colsum_finalize: the pass that would read the 3C-channel tensor again is
// the largest of the step's column-sum launches (92 us at 64 channels, 128 x 128, batch 32).
template <bool SUMS>
__global__ __launch_bounds__(256) void gate_bwd_kernel(long long Mp, int C, const char* __restrict__ dout, int d_cs,
                                                       int d_co, const char* __restrict__ z, int z_cs, int z_co,
                                                       const char* __restrict__ x, int x_cs, int x_co,
                                                       const uint8_t* __restrict__ rowmask, char* __restrict__ dz,
                                                       int dz_cs, int dz_co, char* __restrict__ dxd, int dx_cs,
                                                       int dx_co, float* __restrict__ partial) {
  extern __shared__ float lds[];
  const int cpr = C / 8, rpi = 256 / cpr;
  const int rr = threadIdx.x / cpr, cc = threadIdx.x % cpr;
  float acc[3][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = acc[2][j] = 0.f;
  if (rr < rpi) {
    for (long long m = (long long)blockIdx.x * rpi + rr; m < Mp; m += (long long)gridDim.x * rpi) {
      float z1[8], z2[8], z3[8], xv[8], g[8], o1[8], o2[8], o3[8], ox[8];
      const bool on = rowmask[m];
      const char* zr = z + (m * z_cs + z_co + cc * 8) * 2;
      unpack8(*(const U4*)zr, z1);
      unpack8(*(const U4*)(zr + C * 2), z2);
      unpack8(*(const U4*)(zr + C * 4), z3);
      unpack8(*(const U4*)(x + (m * x_cs + x_co + cc * 8) * 2), xv);
      unpack8(*(const U4*)(dout + (m * d_cs + d_co + cc * 8) * 2), g);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float b = sigmoidf_(z1[j]), c = tanhf(z2[j]), a = sigmoidf_(z3[j]);
        const float d = on ? g[j] : 0.f;
        o1[j] = d * c * b * (1.f - b);
        o2[j] = d * b * (1.f - c * c);
        o3[j] = d * xv[j] * a * (1.f - a);
        ox[j] = d * a;
      }
      char* dr = dz + (m * dz_cs + dz_co + cc * 8) * 2;
      const U4 p1 = pack8(o1), p2 = pack8(o2), p3 = pack8(o3);
      *(U4*)dr = p1;
      *(U4*)(dr + C * 2) = p2;
      *(U4*)(dr + C * 4) = p3;
      *(U4*)(dxd + (m * dx_cs + dx_co + cc * 8) * 2) = pack8(ox);
      if (SUMS) {
        unpack8(p1, o1);
        unpack8(p2, o2);
        unpack8(p3, o3);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc[0][j] += o1[j];
          acc[1][j] += o2[j];
          acc[2][j] += o3[j];
        }
      }
    }
  }
  if (SUMS) block_combine_store<3>(acc, C, cpr, rpi, lds, partial + (long long)blockIdx.x * 3 * C);
}

extern "C" int uh_gate_bwd(long long Mp, int C, const void* dout, int d_cstride, int d_coff, const void* z,
                           int z_cstride, int z_coff, const void* x, int x_cstride, int x_coff,
                           const uint8_t* rowmask, void* dz, int dz_cstride, int dz_coff, void* dxd, int dx_cstride,
                           int dx_coff, uh_stream s) {
  EW_CHECK("gate_bwd", C);
  UH_REQUIRE(Mp > 0 && dout && z && x && rowmask && dz && dxd, "gate_bwd: bad arguments");
  hipLaunchKernelGGL(gate_bwd_kernel<false>, dim3(ew_blocks(Mp, C)), dim3(256), 0, (hipStream_t)s, Mp, C,
                     (const char*)dout, d_cstride, d_coff, (const char*)z, z_cstride, z_coff, (const char*)x, x_cstride,
                     x_coff, rowmask, (char*)dz, dz_cstride, dz_coff, (char*)dxd, dx_cstride, dx_coff, nullptr);
  return uh_check_launch("gate_bwd");
}

// uh_gate_bwd that also leaves the column sums of dz: partial[uh_gate_bwd_parts(Mp, C)][3 * C], for
// uh_colsum_finalize(partial, parts, 3 * C, ...)
extern "C" int uh_gate_bwd_parts(long long Mp, int C) { return (C > 0 && C % 8 == 0 && C / 8 <= 256) ? ew_stat_blocks(Mp, C) : 0; }

extern "C" int uh_gate_bwd_colsum(long long Mp, int C, const void* dout, int d_cstride, int d_coff, const void* z,
                                  int z_cstride, int z_coff, const void* x, int x_cstride, int x_coff,
                                  const uint8_t* rowmask, void* dz, int dz_cstride, int dz_coff, void* dxd,
                                  int dx_cstride, int dx_coff, float* partial, uh_stream s) {
  EW_CHECK("gate_bwd_colsum", C);
  UH_REQUIRE(Mp > 0 && dout && z && x && rowmask && dz && dxd && partial, "gate_bwd_colsum: bad arguments");
  const int rpi = ew_rows_per_iter(C);
  hipLaunchKernelGGL(gate_bwd_kernel<true>, dim3(ew_stat_blocks(Mp, C)), dim3(256), (size_t)rpi * 3 * C * 4,
                     (hipStream_t)s, Mp, C, (const char*)dout, d_cstride, d_coff, (const char*)z, z_cstride, z_coff,
                     (const char*)x, x_cstride, x_coff, rowmask, (char*)dz, dz_cstride, dz_coff, (char*)dxd, dx_cstride,
                     dx_coff, partial);
  return uh_check_launch("gate_bwd_colsum");
}

// out = a + b (+ c)
__global__ __launch_bounds__(256) void add_kernel(long long Mp, int C, const char* __restrict__ a, int a_cs, int a_co,
                                                  const char* __restrict__ b, int b_cs, int b_co,
                                                  const char* __restrict__ c, int c_cs, int c_co,
                                                  char* __restrict__ out, int o_cs, int o_co) {
  const int cpr = C / 8, rpi = 256 / cpr;
  const int rr = threadIdx.x / cpr, cc = threadIdx.x % cpr;
  if (rr >= rpi) return;
  for (long long m = (long long)blockIdx.x * rpi + rr; m < Mp; m += (long long)gridDim.x * rpi) {
    float u[8], v[8];
    unpack8(*(const U4*)(a + (m * a_cs + a_co + cc * 8) * 2), u);
    unpack8(*(const U4*)(b + (m * b_cs + b_co + cc * 8) * 2), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) u[j] += v[j];
    if (c) {
      unpack8(*(const U4*)(c + (m * c_cs + c_co + cc * 8) * 2), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) u[j] += v[j];
    }
    *(U4*)(out + (m * o_cs + o_co + cc * 8) * 2) = pack8(u);
  }
}

extern "C" int uh_add(long long Mp, int C, const void* a, int a_cstride, int a_coff, const void* b, int b_cstride,
                      int b_coff, const void* c, int c_cstride, int c_coff, void* out, int o_cstride, int o_coff,
                      uh_stream s) {
  EW_CHECK("add", C);
  UH_REQUIRE(Mp > 0 && a && b && out, "add: bad arguments");
  hipLaunchKernelGGL(add_kernel, dim3(ew_blocks(Mp, C)), dim3(256), 0, (hipStream_t)s, Mp, C, (const char*)a, a_cstride,
                     a_coff, (const char*)b, b_cstride, b_coff, (const char*)c, c_cstride, c_coff, (char*)out,
                     o_cstride, o_coff);
  return uh_check_launch("add");
}

// ---- the 8-channel gate of the top level, fully fused -------------------------------------------------------------------
// wts: W1[8][8] | W2[8][8] | W3[8][8] | b1[8] | b2[8] | b3[8]   (fp32, [co][ci], zero padded beyond the real channels)
#define G8_W 216
// CR = real channels (the rest of the 8 are zero padding: zero activations, zero weights, so their terms vanish and
// are not computed: at the 3-channel level that is 9 of 64 multiply-adds per matrix)
template <int CR>
__global__ __launch_bounds__(256) void gate8_fwd_kernel(long long Mp, const char* __restrict__ g, int g_cs, int g_co,
                                                        const char* __restrict__ x, int x_cs, int x_co,
                                                        const float* __restrict__ wts,
                                                        const uint8_t* __restrict__ rowmask, char* __restrict__ out,
                                                        int o_cs, int o_co) {
  __shared__ float w[G8_W];
  for (int i = threadIdx.x; i < G8_W; i += 256) w[i] = wts[i];
  __syncthreads();
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < Mp; m += (long long)gridDim.x * 256) {
    float gv[8], xv[8], o[8];
    if (!rowmask[m]) {   // the biases make the gate non-zero at zero input: pad rows must be written as zero
      *(U4*)(out + (m * o_cs + o_co) * 2) = U4{0u, 0u, 0u, 0u};
      continue;
    }
    unpack8(*(const U4*)(g + (m * g_cs + g_co) * 2), gv);
    unpack8(*(const U4*)(x + (m * x_cs + x_co) * 2), xv);
#pragma unroll
    for (int co = 0; co < 8; ++co) o[co] = 0.f;
#pragma unroll
    for (int co = 0; co < CR; ++co) {
      float z1 = w[192 + co], z2 = w[200 + co], z3 = w[208 + co];
#pragma unroll
      for (int ci = 0; ci < CR; ++ci) {
        z1 += w[co * 8 + ci] * gv[ci];
        z2 += w[64 + co * 8 + ci] * gv[ci];
        z3 += w[128 + co * 8 + ci] * (gv[ci] + xv[ci]);
      }
      o[co] = sigmoidf_(z3) * xv[co] + sigmoidf_(z1) * tanhf(z2);
    }
    // (a padding channel: x = 0 kills the first term, and the second is sigmoid(b1)*tanh(b2) = 0.5*tanh(0) = 0)
    *(U4*)(out + (m * o_cs + o_co) * 2) = pack8(o);
  }
}

#define UH_GATE8_DISPATCH(c_real, LAUNCH) \
  switch (c_real) {                       \
    case 1: LAUNCH(1); break;             \
    case 2: LAUNCH(2); break;             \
    case 3: LAUNCH(3); break;             \
    case 4: LAUNCH(4); break;             \
    default: LAUNCH(8); break;            \
  }

extern "C" int uh_gate8_fwd(long long Mp, int c_real, const void* g, int g_cstride, int g_coff, const void* x,
                            int x_cstride, int x_coff, const float* wts, const uint8_t* rowmask, void* out,
                            int o_cstride, int o_coff, uh_stream s) {
  UH_REQUIRE(Mp > 0 && g && x && wts && rowmask && out, "gate8_fwd: bad arguments");
  UH_REQUIRE(c_real >= 1 && c_real <= 8, "gate8_fwd: c_real=%d", c_real);
  long long blocks = (Mp + 255) / 256;
  if (blocks > 2048) blocks = 2048;
#define UH_L(CR)                                                                                                       \
  hipLaunchKernelGGL(gate8_fwd_kernel<CR>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, Mp, (const char*)g,    \
                     g_cstride, g_coff, (const char*)x, x_cstride, x_coff, wts, rowmask, (char*)out, o_cstride, o_coff)
  UH_GATE8_DISPATCH(c_real, UH_L)
#undef UH_L
  return uh_check_launch("gate8_fwd");
}

static inline int gate8_blocks(long long Mp) {
  long long b = (Mp + 255) / 256;
  if (b > 512) b = 512;
  return (int)b;
}
extern "C" int uh_gate8_parts(long long Mp) { return gate8_blocks(Mp); }

// partial[blk][216] in the layout of wts: dW1 | dW2 | dW3 | db1 | db2 | db3
template <int CR>
__global__ __launch_bounds__(256) void gate8_bwd_kernel(long long Mp, const char* __restrict__ dout, int d_cs, int d_co,
                                                        const char* __restrict__ g, int g_cs, int g_co,
                                                        const char* __restrict__ x, int x_cs, int x_co,
                                                        const float* __restrict__ wts, char* __restrict__ dg,
                                                        int dg_cs, int dg_co, char* __restrict__ dx, int dx_cs,
                                                        int dx_co, float* __restrict__ partial) {
  __shared__ float w[G8_W];
  __shared__ float red[4][G8_W];
  for (int i = threadIdx.x; i < G8_W; i += 256) w[i] = wts[i];
  for (int i = threadIdx.x; i < 4 * G8_W; i += 256) (&red[0][0])[i] = 0.f;     // entries of padding channels stay zero
  __syncthreads();
  float aw[3][CR][CR], ab[3][CR];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int co = 0; co < CR; ++co) {
      ab[q][co] = 0.f;
#pragma unroll
      for (int ci = 0; ci < CR; ++ci) aw[q][co][ci] = 0.f;
    }
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < Mp; m += (long long)gridDim.x * 256) {
    float gv[8], xv[8], d[8], og[8], ox[8];
    unpack8(*(const U4*)(g + (m * g_cs + g_co) * 2), gv);
    unpack8(*(const U4*)(x + (m * x_cs + x_co) * 2), xv);
    unpack8(*(const U4*)(dout + (m * d_cs + d_co) * 2), d);      // zero on pad rows and in padding channels
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) og[ci] = ox[ci] = 0.f;
#pragma unroll
    for (int co = 0; co < CR; ++co) {
      float z1 = w[192 + co], z2 = w[200 + co], z3 = w[208 + co];
#pragma unroll
      for (int ci = 0; ci < CR; ++ci) {
        z1 += w[co * 8 + ci] * gv[ci];
        z2 += w[64 + co * 8 + ci] * gv[ci];
        z3 += w[128 + co * 8 + ci] * (gv[ci] + xv[ci]);
      }
      const float b = sigmoidf_(z1), c = tanhf(z2), a = sigmoidf_(z3);
      const float dz1 = d[co] * c * b * (1.f - b);
      const float dz2 = d[co] * b * (1.f - c * c);
      const float dz3 = d[co] * xv[co] * a * (1.f - a);
      ox[co] += d[co] * a;
#pragma unroll
      for (int ci = 0; ci < CR; ++ci) {
        const float t = w[128 + co * 8 + ci] * dz3;
        og[ci] += w[co * 8 + ci] * dz1 + w[64 + co * 8 + ci] * dz2 + t;
        ox[ci] += t;
        aw[0][co][ci] += dz1 * gv[ci];
        aw[1][co][ci] += dz2 * gv[ci];
        aw[2][co][ci] += dz3 * (gv[ci] + xv[ci]);
      }
      ab[0][co] += dz1;
      ab[1][co] += dz2;
      ab[2][co] += dz3;
    }
    *(U4*)(dg + (m * dg_cs + dg_co) * 2) = pack8(og);
    *(U4*)(dx + (m * dx_cs + dx_co) * 2) = pack8(ox);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int co = 0; co < CR; ++co) {
#pragma unroll
      for (int ci = 0; ci < CR; ++ci) {
        const float v = wave_sum(aw[q][co][ci]);
        if (lane == 0) red[wave][64 * q + co * 8 + ci] = v;
      }
      const float v = wave_sum(ab[q][co]);
      if (lane == 0) red[wave][192 + 8 * q + co] = v;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < G8_W; i += 256)
    partial[(long long)blockIdx.x * G8_W + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

extern "C" int uh_gate8_bwd(long long Mp, int c_real, const void* dout, int d_cstride, int d_coff, const void* g,
                            int g_cstride, int g_coff, const void* x, int x_cstride, int x_coff, const float* wts,
                            void* dg, int dg_cstride, int dg_coff, void* dx, int dx_cstride, int dx_coff, float* partial,
                            uh_stream s) {
  UH_REQUIRE(Mp > 0 && dout && g && x && wts && dg && dx && partial, "gate8_bwd: bad arguments");
  UH_REQUIRE(c_real >= 1 && c_real <= 8, "gate8_bwd: c_real=%d", c_real);
#define UH_L(CR)                                                                                                        \
  hipLaunchKernelGGL(gate8_bwd_kernel<CR>, dim3(gate8_blocks(Mp)), dim3(256), 0, (hipStream_t)s, Mp, (const char*)dout,  \
                     d_cstride, d_coff, (const char*)g, g_cstride, g_coff, (const char*)x, x_cstride, x_coff, wts,      \
                     (char*)dg, dg_cstride, dg_coff, (char*)dx, dx_cstride, dx_coff, partial)
  UH_GATE8_DISPATCH(c_real, UH_L)
#undef UH_L
  return uh_check_launch("gate8_bwd");
}

// ---- bilinear resize, align_corners = False (the size fix after a transposed convolution, lunext.py:226-251) ------------
// torch's rule: src = (dst + 0.5) * in/out - 0.5, clamped at 0; i0 = floor(src), i1 = min(i0 + 1, in - 1), l1 = src - i0.
struct Lerp {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Lerp lerp_of(int dst, float scale, int in_size) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  Lerp r;
  r.i0 = (int)src;
  if (r.i0 > in_size - 1) r.i0 = in_size - 1;
  r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
  r.l1 = src - (float)r.i0;
  r.l0 = 1.f - r.l1;
  return r;
}

__global__ __launch_bounds__(256) void resize_bilinear_fwd_kernel(int B, int Hi, int Wi, int Ho, int Wo, int C,
                                                                  const char* __restrict__ in, int i_cs, int i_co,
                                                                  char* __restrict__ out, int o_cs, int o_co) {
  const int cpr = C / 8, oWp = Wo + 1, oHp = Ho + 1, iWp = Wi + 1, iHp = Hi + 1;
  const long long total = (long long)B * oHp * oWp * cpr;
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % cpr);
    const long long m = i / cpr;
    const int x = (int)(m % oWp);
    const long long r = m / oWp;
    const int y = (int)(r % oHp);
    const long long b = r / oHp;
    U4 o = {0u, 0u, 0u, 0u};
    if (y < Ho && x < Wo) {
      const Lerp ly = lerp_of(y, sy, Hi), lx = lerp_of(x, sx, Wi);
      const long long base = b * iHp;
      float v00[8], v01[8], v10[8], v11[8], f[8];
      unpack8(*(const U4*)(in + ((((base + ly.i0) * iWp + lx.i0) * i_cs) + i_co + cc * 8) * 2), v00);
      unpack8(*(const U4*)(in + ((((base + ly.i0) * iWp + lx.i1) * i_cs) + i_co + cc * 8) * 2), v01);
      unpack8(*(const U4*)(in + ((((base + ly.i1) * iWp + lx.i0) * i_cs) + i_co + cc * 8) * 2), v10);
      unpack8(*(const U4*)(in + ((((base + ly.i1) * iWp + lx.i1) * i_cs) + i_co + cc * 8) * 2), v11);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        f[j] = ly.l0 * (lx.l0 * v00[j] + lx.l1 * v01[j]) + ly.l1 * (lx.l0 * v10[j] + lx.l1 * v11[j]);
      o = pack8(f);
    }
    *(U4*)(out + (m * o_cs + o_co + cc * 8) * 2) = o;
  }
}

// The transpose: every input position gathers, in a fixed order, from the output positions that read it.
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(int B, int Hi, int Wi, int Ho, int Wo, int C,
                                                                  const char* __restrict__ dout, int o_cs, int o_co,
                                                                  char* __restrict__ din, int i_cs, int i_co) {
  const int cpr = C / 8, oWp = Wo + 1, oHp = Ho + 1, iWp = Wi + 1, iHp = Hi + 1;
  const long long total = (long long)B * iHp * iWp * cpr;
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % cpr);
    const long long m = i / cpr;
    const int x = (int)(m % iWp);
    const long long r = m / iWp;
    const int y = (int)(r % iHp);
    const long long b = r / iHp;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (y < Hi && x < Wi) {
      // output rows whose source lies in (y - 1, y + 1): a window around y / scale, widened by two for rounding
      int ylo = (int)floorf(((float)y - 1.f + 0.5f) / sy - 0.5f) - 1, yhi = (int)ceilf(((float)y + 1.f + 0.5f) / sy - 0.5f) + 1;
      int xlo = (int)floorf(((float)x - 1.f + 0.5f) / sx - 0.5f) - 1, xhi = (int)ceilf(((float)x + 1.f + 0.5f) / sx - 0.5f) + 1;
      ylo = ylo < 0 ? 0 : ylo;
      xlo = xlo < 0 ? 0 : xlo;
      yhi = yhi > Ho - 1 ? Ho - 1 : yhi;
      xhi = xhi > Wo - 1 ? Wo - 1 : xhi;
      for (int oy = ylo; oy <= yhi; ++oy) {
        const Lerp ly = lerp_of(oy, sy, Hi);
        const float wy = (ly.i0 == y ? ly.l0 : 0.f) + (ly.i1 == y ? ly.l1 : 0.f);
        if (wy == 0.f) continue;
        for (int ox = xlo; ox <= xhi; ++ox) {
          const Lerp lx = lerp_of(ox, sx, Wi);
          const float wx = (lx.i0 == x ? lx.l0 : 0.f) + (lx.i1 == x ? lx.l1 : 0.f);
          if (wx == 0.f) continue;
          float g[8];
          unpack8(*(const U4*)(dout + ((((b * oHp + oy) * oWp + ox) * o_cs) + o_co + cc * 8) * 2), g);
          const float wgt = wy * wx;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = fmaf(wgt, g[j], acc[j]);
        }
      }
    }
    *(U4*)(din + (m * i_cs + i_co + cc * 8) * 2) = pack8(acc);
  }
}

static int resize_launch(bool fwd, int B, int Hi, int Wi, int Ho, int Wo, int C, const void* src, int s_cs, int s_co,
                         void* dst, int d_cs, int d_co, uh_stream s) {
  EW_CHECK("resize_bilinear", C);
  UH_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && src && dst, "resize_bilinear: bad arguments");
  UH_REQUIRE(s_cs % 8 == 0 && s_co % 8 == 0 && d_cs % 8 == 0 && d_co % 8 == 0, "resize_bilinear: bad views");
  const long long total = (fwd ? uh_mp(B, Ho, Wo) : uh_mp(B, Hi, Wi)) * (C / 8);
  long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (fwd)
    hipLaunchKernelGGL(resize_bilinear_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, B, Hi, Wi, Ho, Wo,
                       C, (const char*)src, s_cs, s_co, (char*)dst, d_cs, d_co);
  else
    hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, B, Hi, Wi, Ho, Wo,
                       C, (const char*)src, s_cs, s_co, (char*)dst, d_cs, d_co);
  return uh_check_launch("resize_bilinear");
}
extern "C" int uh_resize_bilinear_fwd(int B, int Hi, int Wi, int Ho, int Wo, int C, const void* in, int i_cstride,
                                      int i_coff, void* out, int o_cstride, int o_coff, uh_stream s) {
  return resize_launch(true, B, Hi, Wi, Ho, Wo, C, in, i_cstride, i_coff, out, o_cstride, o_coff, s);
}
extern "C" int uh_resize_bilinear_bwd(int B, int Hi, int Wi, int Ho, int Wo, int C, const void* dout, int o_cstride,
                                      int o_coff, void* din, int i_cstride, int i_coff, uh_stream s) {
  return resize_launch(false, B, Hi, Wi, Ho, Wo, C, dout, o_cstride, o_coff, din, i_cstride, i_coff, s);
}

// ---- NCHW fp32 image <-> 8-channel padded-linear bf16 (the top level of LUNeXt reads the image directly) ----------------
__global__ __launch_bounds__(256) void nchw_to_c8_kernel(int B, int Cin, int H, int W, const float* __restrict__ x,
                                                         char* __restrict__ out, int o_cs, int o_co) {
  const int Wp = W + 1, Hp = H + 1;
  const long long Mp = (long long)B * Hp * Wp;
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < Mp; m += (long long)gridDim.x * 256) {
    const int px = (int)(m % Wp);
    const long long r = m / Wp;
    const int py = (int)(r % Hp);
    const long long b = r / Hp;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (px < W && py < H)
      for (int c = 0; c < Cin; ++c) v[c] = x[((b * Cin + c) * H + py) * W + px];
    *(U4*)(out + (m * o_cs + o_co) * 2) = pack8(v);
  }
}

extern "C" int uh_nchw_to_c8(int B, int Cin, int H, int W, const float* x, void* out, int o_cstride, int o_coff,
                             uh_stream s) {
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && Cin >= 1 && Cin <= 8 && x && out, "nchw_to_c8: bad arguments (Cin<=8)");
  const long long Mp = uh_mp(B, H, W);
  long long blocks = (Mp + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(nchw_to_c8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, B, Cin, H, W, x,
                     (char*)out, o_cstride, o_coff);
  return uh_check_launch("nchw_to_c8");
}

// ---- small parameter tables for the 8-channel level, made on the device in one launch each -------------------------
// uh_gate8_pack: the gate's three [C][C] 1x1 weights and three biases (lunext.py:67-73) -> the [3][8][8] | [3][8] table
// uh_gate8_fwd / uh_gate8_bwd take, zero beyond the C real channels.  uh_pad_rows_f32: rows of `cin` floats -> rows of
// `cpad` floats, zero-filled (the head's [oc][cin] weight as [oc][8], a 3-channel bias as 8 channels).
__global__ void gate8_pack_kernel(int C, const float* w1, const float* b1, const float* w2, const float* b2,
                                  const float* w3, const float* b3, float* __restrict__ out) {
  const int i = threadIdx.x;
  if (i >= 216) return;
  const float* w[3] = {w1, w2, w3};
  const float* b[3] = {b1, b2, b3};
  float v = 0.f;
  if (i < 192) {
    const int j = i >> 6, r = (i >> 3) & 7, c = i & 7;
    if (r < C && c < C) v = w[j][r * C + c];
  } else {
    const int j = (i - 192) >> 3, c = (i - 192) & 7;
    if (c < C) v = b[j][c];
  }
  out[i] = v;
}

extern "C" int uh_gate8_pack(int c_real, const float* w1, const float* b1, const float* w2, const float* b2,
                             const float* w3, const float* b3, float* wts216, uh_stream s) {
  UH_REQUIRE(c_real >= 1 && c_real <= 8 && w1 && b1 && w2 && b2 && w3 && b3 && wts216, "gate8_pack: bad arguments");
  hipLaunchKernelGGL(gate8_pack_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, c_real, w1, b1, w2, b2, w3, b3, wts216);
  return uh_check_launch("gate8_pack");
}

__global__ void pad_rows_kernel(int rows, int cin, int cpad, const float* __restrict__ src, float* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cpad) return;
  const int r = i / cpad, c = i - r * cpad;
  dst[i] = c < cin ? src[r * cin + c] : 0.f;
}

extern "C" int uh_pad_rows_f32(int rows, int cin, int cpad, const float* src, float* dst, uh_stream s) {
  UH_REQUIRE(rows > 0 && cin > 0 && cpad >= cin && src && dst && (long long)rows * cpad < (1ll << 30), "pad_rows_f32: bad arguments");
  hipLaunchKernelGGL(pad_rows_kernel, dim3((rows * cpad + 255) / 256), dim3(256), 0, (hipStream_t)s, rows, cin, cpad, src, dst);
  return uh_check_launch("pad_rows_f32");
}
