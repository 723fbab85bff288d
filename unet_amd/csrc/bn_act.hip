// Bandwidth-bound kernels around the convolutions: BatchNorm statistics finalisation,
// normalise+ReLU, their backward, 2x2 max-pool forward/backward, per-channel column sums and
// the 1x1 segmentation head.  All tensors are padded-linear NHWC bf16 (common.h); every thread
// moves 16 bytes (8 channels) per access, lanes of a wave walk the channel axis so a wave
// touches whole rows, and per-channel sums are kept in registers across a grid-stride loop,
// combined through LDS once per workgroup and written as one partial row per workgroup.
#include "ew.h"
#include "reduce_small.h"

__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partial, int P, int C, int c_real,
                                                           double count, const float* gamma, const float* beta,
                                                           float eps, float momentum, float* mean_o, float* invstd_o,
                                                           float* scale_o, float* shift_o, float* rm, float* rv,
                                                           long long* nbt) {
  __shared__ double lds[128 * 2 * 8];
  const int c0 = blockIdx.x * 8;
  double s[2];
  reduce_partials<2>(partial, P, C, c0, s, lds);
  const int t = threadIdx.x;
  if (t < 8 && c0 + t < C) {
    const int c = c0 + t;
    const double mean = s[0] / count;
    double var = s[1] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const bool real = c < c_real;                     // padding channels: gamma = beta = 0, no running statistics
    const double sc = real ? (double)gamma[c] * invstd : 0.0;
    mean_o[c] = (float)mean;
    invstd_o[c] = (float)invstd;
    scale_o[c] = (float)sc;
    shift_o[c] = real ? (float)((double)beta[c] - mean * sc) : 0.f;
    if (rm && real) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      rm[c] = (float)((1.0 - momentum) * rm[c] + momentum * mean);
      rv[c] = (float)((1.0 - momentum) * rv[c] + momentum * unbiased);
    }
  }
  if (nbt && blockIdx.x == 0 && t == 0) *nbt += 1;
}

extern "C" int uh_bn_finalize(const float* partial, int P, int C, int c_real, long long count, const float* gamma,
                              const float* beta, float eps, float momentum, float* mean, float* invstd, float* scale,
                              float* shift, float* running_mean, float* running_var, long long* nbt, uh_stream s) {
  UH_REQUIRE(partial && gamma && beta && mean && invstd && scale && shift && P > 0 && C > 0 && count > 0 &&
                 c_real > 0 && c_real <= C,
             "bn_finalize: bad arguments");
  UH_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats must come in pairs");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 7) / 8), dim3(1024), 0, (hipStream_t)s, partial, P, C, c_real,
                     (double)count, gamma, beta, eps, momentum, mean, invstd, scale, shift, running_mean, running_var, nbt);
  return uh_check_launch("bn_finalize");
}

__global__ void bn_fold_eval_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                                    const float* cb, float eps, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(rv[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] + ((cb ? cb[c] : 0.f) - rm[c]) * sc;
}

extern "C" int uh_bn_fold_eval(int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                               const float* conv_bias, float eps, float* scale, float* shift, uh_stream s) {
  UH_REQUIRE(C > 0 && gamma && beta && rm && rv && scale && shift, "bn_fold_eval: bad arguments");
  hipLaunchKernelGGL(bn_fold_eval_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)s, C, gamma, beta, rm, rv,
                     conv_bias, eps, scale, shift);
  return uh_check_launch("bn_fold_eval");
}

// ---- a = relu(y*scale + shift) * rowmask ------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(long long Mp, int C, const char* __restrict__ y,
                                                            int y_cs, int y_co, const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const uint8_t* __restrict__ rowmask, char* __restrict__ a,
                                                            int a_cs, int a_co) {
  const int cpr = C / 8, rpi = 256 / cpr;
  const int rr = threadIdx.x / cpr, cc = threadIdx.x % cpr;
  if (rr >= rpi) return;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = scale[cc * 8 + j];
    sh[j] = shift[cc * 8 + j];
  }
  // two rows in flight per thread, loads issued before the mask is known (pad rows are readable; they are
  // replaced by zeros afterwards): one memory latency per iteration instead of two
  auto row = [&](long long m, const U4& raw, uint8_t keep) {
    U4 o = {0u, 0u, 0u, 0u};
    if (keep) {
      float f[8];
      unpack8h(raw, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * sc[j] + sh[j], 0.f);
      o = pack8(f);
    }
    *(U4*)(a + (m * a_cs + a_co + cc * 8) * 2) = o;
  };
  const long long step = (long long)gridDim.x * rpi;
  long long m = (long long)blockIdx.x * rpi + rr;
  for (; m + step < Mp; m += 2 * step) {
    const U4 r0 = *(const U4*)(y + (m * y_cs + y_co + cc * 8) * 2);
    const U4 r1 = *(const U4*)(y + ((m + step) * y_cs + y_co + cc * 8) * 2);
    const uint8_t k0 = rowmask[m], k1 = rowmask[m + step];
    row(m, r0, k0);
    row(m + step, r1, k1);
  }
  if (m < Mp) row(m, *(const U4*)(y + (m * y_cs + y_co + cc * 8) * 2), rowmask[m]);
}

extern "C" int uh_bn_relu_apply(long long Mp, int C, const void* y, int y_cstride, int y_coff, const float* scale,
                                const float* shift, const uint8_t* rowmask, void* a, int a_cstride, int a_coff,
                                uh_stream s) {
  EW_CHECK("bn_relu_apply", C);
  UH_REQUIRE(Mp > 0 && y && scale && shift && rowmask && a, "bn_relu_apply: bad arguments");
  UH_REQUIRE(y_cstride % 8 == 0 && y_coff % 8 == 0 && a_cstride % 8 == 0 && a_coff % 8 == 0, "bn_relu_apply: bad views");
  hipLaunchKernelGGL(bn_relu_apply_kernel, dim3(ew_blocks(Mp, C)), dim3(256), 0, (hipStream_t)s, Mp, C, (const char*)y,
                     y_cstride, y_coff, scale, shift, rowmask, (char*)a, a_cstride, a_coff);
  return uh_check_launch("bn_relu_apply");
}

// ---- backward of relu(bn(y)) ------------------------------------------------------------------------
extern "C" int uh_bn_bwd_parts(long long Mp, int C) { return (C > 0 && C % 8 == 0 && C / 8 <= 256) ? ew_stat_blocks(Mp, C) : 0; }

__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(long long Mp, int C, const char* __restrict__ da,
                                                                 int da_cs, int da_co, const char* __restrict__ y,
                                                                 int y_cs, int y_co, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
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
    auto row = [&](const U4& gr, const U4& yr) {
      float g[8], v[8];
      unpack8(gr, g);
      unpack8h(yr, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = (v[j] * sc[j] + sh[j] > 0.f) ? g[j] : 0.f;
        acc[0][j] += d;
        acc[1][j] += d * (v[j] - mu[j]) * is[j];
      }
    };
    // two rows in flight per thread (four were measured: no faster here, slower in the apply kernel)
    const long long step = (long long)gridDim.x * rpi;
    long long m = (long long)blockIdx.x * rpi + rr;
    for (; m + step < Mp; m += 2 * step) {
      const U4 g0 = *(const U4*)(da + (m * da_cs + da_co + cc * 8) * 2);
      const U4 y0 = *(const U4*)(y + (m * y_cs + y_co + cc * 8) * 2);
      const U4 g1 = *(const U4*)(da + ((m + step) * da_cs + da_co + cc * 8) * 2);
      const U4 y1 = *(const U4*)(y + ((m + step) * y_cs + y_co + cc * 8) * 2);
      row(g0, y0);
      row(g1, y1);
    }
    if (m < Mp) row(*(const U4*)(da + (m * da_cs + da_co + cc * 8) * 2), *(const U4*)(y + (m * y_cs + y_co + cc * 8) * 2));
  }
  if (fin.ws.counters) {
    block_combine_store<2, true>(acc, C, cpr, rpi, lds, partial + (long long)blockIdx.x * 2 * C);
    last_arriver_finish<2>(partial, C, fin.ws, fin);
  } else {
    block_combine_store<2>(acc, C, cpr, rpi, lds, partial + (long long)blockIdx.x * 2 * C);
  }
}

extern "C" int uh_bn_relu_bwd_reduce(long long Mp, int C, const void* da, int da_cstride, int da_coff, const void* y,
                                     int y_cstride, int y_coff, const float* mean, const float* invstd,
                                     const float* scale, const float* shift, float* partial, uh_stream s) {
  EW_CHECK("bn_relu_bwd_reduce", C);
  UH_REQUIRE(Mp > 0 && da && y && mean && invstd && scale && shift && partial, "bn_relu_bwd_reduce: bad arguments");
  const int rpi = ew_rows_per_iter(C);
  BnBwdFin fin{};
  hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, dim3(ew_stat_blocks(Mp, C)), dim3(256), (size_t)rpi * 2 * C * 4,
                     (hipStream_t)s, Mp, C, (const char*)da, da_cstride, da_coff, (const char*)y, y_cstride, y_coff,
                     mean, invstd, scale, shift, partial, fin);
  return uh_check_launch("bn_relu_bwd_reduce");
}

// ---- workspace of the launches that finalise their own sums (ew.h: last_arriver_finish) ----------------------------
// ws: uh_fin_ws_bytes(parts, C, nstat) bytes = the group rows (fp64) followed by the partial rows (fp32);
// counters: UH_FIN_COUNTERS unsigned ints, zero before the first such launch on a stream and left zero by each
extern "C" size_t uh_fin_ws_bytes(int parts, int C, int nstat) { return fin_ws_bytes(parts, C, nstat); }
extern "C" int uh_fin_counters() { return FIN_MAX_COUNTERS; }

extern "C" int uh_bn_relu_bwd_reduce_fin(long long Mp, int C, const void* da, int da_cstride, int da_coff,
                                         const void* y, int y_cstride, int y_coff, const float* mean,
                                         const float* invstd, const float* scale, const float* shift, int c_real,
                                         long long count, float* dgamma, float* dbeta, float* coef, void* ws,
                                         size_t ws_bytes, unsigned* counters, uh_stream s) {
  EW_CHECK("bn_relu_bwd_reduce_fin", C);
  UH_REQUIRE(Mp > 0 && da && y && mean && invstd && scale && shift && dgamma && dbeta && coef && count > 0 &&
                 c_real > 0 && c_real <= C,
             "bn_relu_bwd_reduce_fin: bad arguments");
  const int parts = ew_stat_blocks(Mp, C);
  BnBwdFin fin{};
  float* partial = nullptr;
  UH_REQUIRE(fin_split(parts, C, 2, ws, ws_bytes, counters, fin.ws, partial),
             "bn_relu_bwd_reduce_fin: workspace of %zu bytes (8-byte aligned) and counters needed", uh_fin_ws_bytes(parts, C, 2));
  fin.count = (double)count;
  fin.dgamma = dgamma;
  fin.dbeta = dbeta;
  fin.coef = coef;
  fin.c_real = c_real;
  fin.C = C;
  const int rpi = ew_rows_per_iter(C);
  hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, dim3(parts), dim3(256), (size_t)rpi * 2 * C * 4, (hipStream_t)s, Mp, C,
                     (const char*)da, da_cstride, da_coff, (const char*)y, y_cstride, y_coff, mean, invstd, scale, shift,
                     partial, fin);
  return uh_check_launch("bn_relu_bwd_reduce_fin");
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int P, int C,
                                                               int c_real, double count, float* dgamma, float* dbeta,
                                                               float* coef, int accumulate) {
  __shared__ double lds[128 * 2 * 8];
  const int c0 = blockIdx.x * 8;
  double s[2];
  reduce_partials<2>(partial, P, C, c0, s, lds);
  const int t = threadIdx.x;
  if (t < 8 && c0 + t < C) {
    const int c = c0 + t;
    if (c < c_real) {                                 // dgamma/dbeta hold c_real entries
      dbeta[c] = accumulate ? dbeta[c] + (float)s[0] : (float)s[0];
      dgamma[c] = accumulate ? dgamma[c] + (float)s[1] : (float)s[1];
    }
    coef[c] = (float)(s[0] / count);
    coef[C + c] = (float)(s[1] / count);
  }
}

extern "C" int uh_bn_bwd_finalize(const float* partial, int P, int C, int c_real, long long count, float* dgamma,
                                  float* dbeta, float* coef, int accumulate, uh_stream s) {
  UH_REQUIRE(partial && dgamma && dbeta && coef && P > 0 && C > 0 && count > 0 && c_real > 0 && c_real <= C,
             "bn_bwd_finalize: bad arguments");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 7) / 8), dim3(1024), 0, (hipStream_t)s, partial, P, C, c_real,
                     (double)count, dgamma, dbeta, coef, accumulate);
  return uh_check_launch("bn_bwd_finalize");
}

__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(long long Mp, int C, const char* __restrict__ da,
                                                                int da_cs, int da_co, const char* __restrict__ y,
                                                                int y_cs, int y_co, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ coef,
                                                                const uint8_t* __restrict__ rowmask,
                                                                char* __restrict__ dy, int dy_cs, int dy_co) {
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
  auto row = [&](long long m, const U4& gr, const U4& yr, uint8_t keep) {
    U4 o = {0u, 0u, 0u, 0u};
    if (keep) {
      float g[8], v[8];
      unpack8(gr, g);
      unpack8h(yr, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = (v[j] * sc[j] + sh[j] > 0.f) ? g[j] : 0.f;
        const float xh = (v[j] - mu[j]) * is[j];
        g[j] = sc[j] * (d - c0[j] - xh * c1[j]);
      }
      o = pack8(g);
    }
    *(U4*)(dy + (m * dy_cs + dy_co + cc * 8) * 2) = o;
  };
  // two rows in flight per thread, loads issued before the mask is known (pad rows are readable)
  const long long step = (long long)gridDim.x * rpi;
  long long m = (long long)blockIdx.x * rpi + rr;
  for (; m + step < Mp; m += 2 * step) {
    const U4 g0 = *(const U4*)(da + (m * da_cs + da_co + cc * 8) * 2);
    const U4 y0 = *(const U4*)(y + (m * y_cs + y_co + cc * 8) * 2);
    const U4 g1 = *(const U4*)(da + ((m + step) * da_cs + da_co + cc * 8) * 2);
    const U4 y1 = *(const U4*)(y + ((m + step) * y_cs + y_co + cc * 8) * 2);
    const uint8_t k0 = rowmask[m], k1 = rowmask[m + step];
    row(m, g0, y0, k0);
    row(m + step, g1, y1, k1);
  }
  if (m < Mp)
    row(m, *(const U4*)(da + (m * da_cs + da_co + cc * 8) * 2), *(const U4*)(y + (m * y_cs + y_co + cc * 8) * 2),
        rowmask[m]);
}

extern "C" int uh_bn_relu_bwd_apply(long long Mp, int C, const void* da, int da_cstride, int da_coff, const void* y,
                                    int y_cstride, int y_coff, const float* mean, const float* invstd,
                                    const float* scale, const float* shift, const float* coef, const uint8_t* rowmask,
                                    void* dy, int dy_cstride, int dy_coff, uh_stream s) {
  EW_CHECK("bn_relu_bwd_apply", C);
  UH_REQUIRE(Mp > 0 && da && y && mean && invstd && scale && shift && coef && rowmask && dy,
             "bn_relu_bwd_apply: bad arguments");
  hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(ew_blocks(Mp, C)), dim3(256), 0, (hipStream_t)s, Mp, C,
                     (const char*)da, da_cstride, da_coff, (const char*)y, y_cstride, y_coff, mean, invstd, scale, shift,
                     coef, rowmask, (char*)dy, dy_cstride, dy_coff);
  return uh_check_launch("bn_relu_bwd_apply");
}

// ---- per-channel column sums (bias gradients) -------------------------------------------------------
extern "C" int uh_colsum_parts(long long Mp, int C) { return uh_bn_bwd_parts(Mp, C); }

__global__ __launch_bounds__(256) void colsum_kernel(long long Mp, int C, const char* __restrict__ x, int cs, int co,
                                                     float* __restrict__ partial) {
  extern __shared__ float lds[];
  const int cpr = C / 8, rpi = 256 / cpr;
  const int rr = threadIdx.x / cpr, cc = threadIdx.x % cpr;
  float acc[1][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[0][j] = 0.f;
  if (rr < rpi) {
    for (long long m = (long long)blockIdx.x * rpi + rr; m < Mp; m += (long long)gridDim.x * rpi) {
      float v[8];
      unpack8(*(const U4*)(x + (m * cs + co + cc * 8) * 2), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[0][j] += v[j];
    }
  }
  block_combine_store<1>(acc, C, cpr, rpi, lds, partial + (long long)blockIdx.x * C);
}

extern "C" int uh_colsum(long long Mp, int C, const void* x, int cstride, int coff, float* partial, uh_stream s) {
  EW_CHECK("colsum", C);
  UH_REQUIRE(Mp > 0 && x && partial, "colsum: bad arguments");
  const int rpi = ew_rows_per_iter(C);
  hipLaunchKernelGGL(colsum_kernel, dim3(ew_stat_blocks(Mp, C)), dim3(256), (size_t)rpi * C * 4, (hipStream_t)s, Mp, C,
                     (const char*)x, cstride, coff, partial);
  return uh_check_launch("colsum");
}

// out[c] (+)= sum_p sum_{f<fold} partial[p][f*Cout + c],  Cout = C/fold
__global__ __launch_bounds__(1024) void colsum_finalize_kernel(const float* __restrict__ partial, int P, int C, int fold,
                                                               float* out, int accumulate) {
  __shared__ double lds[128 * 8];
  const int Cout = C / fold;
  const int c0 = blockIdx.x * 8;
  const int t = threadIdx.x, c8 = t & 7, slot = t >> 3;
  double s = 0.0;
  if (c0 + c8 < Cout) {
    for (int p = slot; p < P; p += 128)
      for (int f = 0; f < fold; ++f) s += (double)partial[(long long)p * C + f * Cout + c0 + c8];
  }
  lds[slot * 8 + c8] = s;
  __syncthreads();
  for (int stride = 64; stride > 0; stride >>= 1) {
    if (slot < stride) lds[slot * 8 + c8] += lds[(slot + stride) * 8 + c8];
    __syncthreads();
  }
  if (t < 8 && c0 + t < Cout) {
    const float v = (float)lds[t];
    out[c0 + t] = accumulate ? out[c0 + t] + v : v;
  }
}

extern "C" int uh_colsum_finalize(const float* partial, int P, int C, int fold, float* out, int accumulate,
                                  uh_stream s) {
  UH_REQUIRE(partial && out && P > 0 && C > 0 && fold > 0 && C % fold == 0, "colsum_finalize: bad arguments");
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C / fold + 7) / 8), dim3(1024), 0, (hipStream_t)s, partial, P, C,
                     fold, out, accumulate);
  return uh_check_launch("colsum_finalize");
}

// out[i] = sum_p partial[p][cols[i]] for i < n_out: a finalisation that lays its sums out as the caller's gradient
// tensors lie in memory (e.g. an [8][8]-padded 1x1 weight gradient into a [3][3] parameter, several bias gradients of
// one column-sum pass into their separate places) without copy launches behind it.  cols: device array of n_out ints.
__global__ __launch_bounds__(256) void colsum_finalize_gather_kernel(const float* __restrict__ partial, int P, int C,
                                                                     int n_out, const int* __restrict__ cols,
                                                                     float* __restrict__ out) {
  __shared__ double lds[256];
  colsum_gather_body(blockIdx.x, lds, partial, P, C, n_out, cols, out, 0);       // (reduce_small.h)
}

extern "C" int uh_colsum_finalize_gather(const float* partial, int P, int C, int n_out, const int* cols, float* out,
                                         uh_stream s) {
  UH_REQUIRE(partial && cols && out && P > 0 && C > 0 && n_out > 0, "colsum_finalize_gather: bad arguments");
  hipLaunchKernelGGL(colsum_finalize_gather_kernel, dim3((n_out + 7) / 8), dim3(256), 0, (hipStream_t)s, partial, P, C,
                     n_out, cols, out);
  return uh_check_launch("colsum_finalize_gather");
}

// ---- MaxPool2d(2,2) -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(int B, int H, int W, int C, const char* __restrict__ a,
                                                           int a_cs, int a_co, char* __restrict__ out, int o_cs,
                                                           int o_co) {
  const int oH = H / 2, oW = W / 2, oHp = oH + 1, oWp = oW + 1, Wp = W + 1, Hp = H + 1;
  const int cpr = C / 8;
  const long long total = (long long)B * oHp * oWp * cpr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpr);
    long long r = i / cpr;
    const int X = (int)(r % oWp);
    r /= oWp;
    const int Y = (int)(r % oHp);
    const long long b = r / oHp;
    U4 o = {0u, 0u, 0u, 0u};
    if (Y < oH && X < oW) {
      const long long m00 = (b * Hp + 2 * Y) * Wp + 2 * X;
      float v0[8], v1[8], v2[8], v3[8];
      unpack8(*(const U4*)(a + (m00 * a_cs + a_co + cc * 8) * 2), v0);
      unpack8(*(const U4*)(a + ((m00 + 1) * a_cs + a_co + cc * 8) * 2), v1);
      unpack8(*(const U4*)(a + ((m00 + Wp) * a_cs + a_co + cc * 8) * 2), v2);
      unpack8(*(const U4*)(a + ((m00 + Wp + 1) * a_cs + a_co + cc * 8) * 2), v3);
#pragma unroll
      for (int j = 0; j < 8; ++j) v0[j] = fmaxf(fmaxf(v0[j], v1[j]), fmaxf(v2[j], v3[j]));
      o = pack8(v0);
    }
    *(U4*)(out + (((b * oHp + Y) * oWp + X) * o_cs + o_co + cc * 8) * 2) = o;
  }
}

extern "C" int uh_maxpool2_fwd(int B, int H, int W, int C, const void* a, int a_cstride, int a_coff, void* out,
                               int o_cstride, int o_coff, uh_stream s) {
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "maxpool2_fwd: H=%d W=%d must be even", H, W);
  UH_REQUIRE(C > 0 && C % 8 == 0 && a && out, "maxpool2_fwd: bad arguments");
  const long long total = uh_mp(B, H / 2, W / 2) * (C / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, B, H, W, C, (const char*)a,
                     a_cstride, a_coff, (char*)out, o_cstride, o_coff);
  return uh_check_launch("maxpool2_fwd");
}

// a = relu(y*scale + shift) and p = MaxPool2d(2,2)(a) in one pass over y: one work item per 2x2 window and 8-channel
// chunk, windows along the extra row and column included (they hold the pad positions of a, which get zeros).
__global__ __launch_bounds__(256) void bn_relu_apply_pool_kernel(int B, int H, int W, int C, const char* __restrict__ y,
                                                                 int y_cs, int y_co, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, char* __restrict__ a,
                                                                 int a_cs, int a_co, char* __restrict__ p, int p_cs,
                                                                 int p_co) {
  const int oH = H / 2, oW = W / 2, oHp = oH + 1, oWp = oW + 1, Wp = W + 1, Hp = H + 1;
  const int cpr = C / 8;
  const long long total = (long long)B * oHp * oWp * cpr;
  const int cc = threadIdx.x % cpr;             // 256 % cpr == 0 and the stride is a multiple of 256
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = scale[cc * 8 + j];
    sh[j] = shift[cc * 8 + j];
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long r = i / cpr;
    const int X = (int)(r % oWp);
    r /= oWp;
    const int Y = (int)(r % oHp);
    const long long b = r / oHp;
    const long long m00 = (b * Hp + 2 * Y) * Wp + 2 * X;
    U4 o = {0u, 0u, 0u, 0u};
    if (Y < oH && X < oW) {
      const long long mm[4] = {m00, m00 + 1, m00 + Wp, m00 + Wp + 1};
      U4 raw[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) raw[k] = *(const U4*)(y + (mm[k] * y_cs + y_co + cc * 8) * 2);
      float best[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float f[8];
        unpack8h(raw[k], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          f[j] = fmaxf(f[j] * sc[j] + sh[j], 0.f);
          best[j] = k ? fmaxf(best[j], f[j]) : f[j];
        }
        *(U4*)(a + (mm[k] * a_cs + a_co + cc * 8) * 2) = pack8(f);
      }
      o = pack8(best);                          // rounding to bf16 is monotonic: the maximum of the rounded values
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int yy = 2 * Y + (k >> 1), xx = 2 * X + (k & 1);
        if (yy < Hp && xx < Wp) *(U4*)(a + (((b * Hp + yy) * Wp + xx) * a_cs + a_co + cc * 8) * 2) = o;
      }
    }
    *(U4*)(p + (((b * oHp + Y) * oWp + X) * p_cs + p_co + cc * 8) * 2) = o;
  }
}

extern "C" int uh_bn_relu_apply_pool2(int B, int H, int W, int C, const void* y, int y_cstride, int y_coff,
                                      const float* scale, const float* shift, void* a, int a_cstride, int a_coff,
                                      void* p, int p_cstride, int p_coff, uh_stream s) {
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "bn_relu_apply_pool2: H=%d W=%d must be even", H, W);
  UH_REQUIRE(C > 0 && C % 8 == 0 && 256 % (C / 8) == 0, "bn_relu_apply_pool2: C=%d must be 8 times a power of two <= 2048", C);
  UH_REQUIRE(y && scale && shift && a && p, "bn_relu_apply_pool2: null pointer");
  UH_REQUIRE(y_cstride % 8 == 0 && y_coff % 8 == 0 && a_cstride % 8 == 0 && a_coff % 8 == 0 && p_cstride % 8 == 0 &&
                 p_coff % 8 == 0, "bn_relu_apply_pool2: bad views");
  const long long total = uh_mp(B, H / 2, W / 2) * (C / 8);
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bn_relu_apply_pool_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, B, H, W, C,
                     (const char*)y, y_cstride, y_coff, scale, shift, (char*)a, a_cstride, a_coff, (char*)p, p_cstride,
                     p_coff);
  return uh_check_launch("bn_relu_apply_pool2");
}

// STATS: `da` is the gradient at the output of a conv + BatchNorm + ReLU layer (conv{l}_2, whose activation `a` was
// pooled): the layer's BatchNorm-backward sums (what bn_relu_bwd_reduce_kernel computes from da and the layer's fp16
// conv output y in a pass of its own) are taken here, from the values as they are stored (after rounding to bf16).
struct PoolBnStat {
  const char* y;
  int y_cs, y_co;
  const float* mean;
  const float* invstd;
  const float* scale;
  const float* shift;
  float* partial;                   // [gridDim.x][2][C]
};

template <bool STATS>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(int B, int H, int W, int C, const char* __restrict__ a,
                                                           int a_cs, int a_co, const char* __restrict__ dp, int dp_cs,
                                                           int dp_co, const char* __restrict__ add, int add_cs,
                                                           int add_co, char* __restrict__ da, int da_cs, int da_co,
                                                           PoolBnStat st) {
  extern __shared__ float lds[];
  const int oH = H / 2, oW = W / 2, oHp = oH + 1, oWp = oW + 1, Wp = W + 1, Hp = H + 1;
  const int cpr = C / 8;
  const long long total = (long long)B * oHp * oWp * cpr;
  float acc[2][8];
  float sc[8], sh[8], mu[8], is[8];
  if (STATS) {
    const int cc0 = threadIdx.x % cpr;          // 256 % cpr == 0: a thread keeps its channel chunk in every iteration
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[0][j] = acc[1][j] = 0.f;
      sc[j] = st.scale[cc0 * 8 + j];
      sh[j] = st.shift[cc0 * 8 + j];
      mu[j] = st.mean[cc0 * 8 + j];
      is[j] = st.invstd[cc0 * 8 + j];
    }
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpr);
    long long r = i / cpr;
    const int X = (int)(r % oWp);
    r /= oWp;
    const int Y = (int)(r % oHp);
    const long long b = r / oHp;
    const long long m00 = (b * Hp + 2 * Y) * Wp + 2 * X;
    if (Y < oH && X < oW) {
      float v[4][8], g[8];
      const long long mm[4] = {m00, m00 + 1, m00 + Wp, m00 + Wp + 1};
#pragma unroll
      for (int k = 0; k < 4; ++k) unpack8(*(const U4*)(a + (mm[k] * a_cs + a_co + cc * 8) * 2), v[k]);
      unpack8(*(const U4*)(dp + (((b * oHp + Y) * oWp + X) * dp_cs + dp_co + cc * 8) * 2), g);
      int idx[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int best = 0;
        float bv = v[0][j];
        if (v[1][j] > bv) { bv = v[1][j]; best = 1; }
        if (v[2][j] > bv) { bv = v[2][j]; best = 2; }
        if (v[3][j] > bv) { bv = v[3][j]; best = 3; }
        idx[j] = best;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float o[8];
        if (add) {
          unpack8(*(const U4*)(add + (mm[k] * add_cs + add_co + cc * 8) * 2), o);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += (idx[j] == k) ? g[j] : 0.f;
        const U4 packed = pack8(o);
        *(U4*)(da + (mm[k] * da_cs + da_co + cc * 8) * 2) = packed;
        if (STATS) {
          float d8[8], yv[8];
          unpack8(packed, d8);
          unpack8h(*(const U4*)(st.y + (mm[k] * st.y_cs + st.y_co + cc * 8) * 2), yv);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float d = (yv[j] * sc[j] + sh[j] > 0.f) ? d8[j] : 0.f;
            acc[0][j] += d;
            acc[1][j] += d * (yv[j] - mu[j]) * is[j];
          }
        }
      }
    } else {
      // pad position of the pooled grid: zero the pad rows of the input grid it covers
      const U4 z = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int y = 2 * Y + (k >> 1), x = 2 * X + (k & 1);
        if (y < Hp && x < Wp) *(U4*)(da + (((b * Hp + y) * Wp + x) * da_cs + da_co + cc * 8) * 2) = z;
      }
    }
  }
  if (STATS) block_combine_store<2>(acc, C, cpr, 256 / cpr, lds, st.partial + (long long)blockIdx.x * 2 * C);
}

extern "C" int uh_maxpool2_bwd(int B, int H, int W, int C, const void* a, int a_cstride, int a_coff, const void* dp,
                               int dp_cstride, int dp_coff, const void* add, int add_cstride, int add_coff, void* da,
                               int da_cstride, int da_coff, uh_stream s) {
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "maxpool2_bwd: H=%d W=%d must be even", H, W);
  UH_REQUIRE(C > 0 && C % 8 == 0 && a && dp && da, "maxpool2_bwd: bad arguments");
  const long long total = uh_mp(B, H / 2, W / 2) * (C / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL((maxpool2_bwd_kernel<false>), dim3(blocks), dim3(256), 0, (hipStream_t)s, B, H, W, C,
                     (const char*)a, a_cstride, a_coff, (const char*)dp, dp_cstride, dp_coff, (const char*)add,
                     add_cstride, add_coff, (char*)da, da_cstride, da_coff, PoolBnStat{});
  return uh_check_launch("maxpool2_bwd");
}

static int pool_bnstat_blocks(int B, int H, int W, int C) {
  const long long total = uh_mp(B, H / 2, W / 2) * (C / 8);
  long long blocks = (total + 255) / 256 / 8;        // one row of partial sums per workgroup: at least 8 rounds each
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}
extern "C" int uh_maxpool2_bwd_bnstat_parts(int B, int H, int W, int C) {
  return (B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0) ? pool_bnstat_blocks(B, H, W, C) : 0;
}

extern "C" int uh_maxpool2_bwd_bnstat(int B, int H, int W, int C, const void* a, int a_cstride, int a_coff,
                                      const void* dp, int dp_cstride, int dp_coff, const void* add, int add_cstride,
                                      int add_coff, void* da, int da_cstride, int da_coff, const void* y, int y_cstride,
                                      int y_coff, const float* mean, const float* invstd, const float* scale,
                                      const float* shift, float* partial, uh_stream s) {
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "maxpool2_bwd_bnstat: H=%d W=%d must be even", H, W);
  UH_REQUIRE(C > 0 && C % 8 == 0 && 256 % (C / 8) == 0, "maxpool2_bwd_bnstat: C=%d must be 8 times a power of two <= 2048", C);
  UH_REQUIRE(a && dp && da && y && mean && invstd && scale && shift && partial, "maxpool2_bwd_bnstat: null pointer");
  const int blocks = pool_bnstat_blocks(B, H, W, C);
  PoolBnStat st{(const char*)y, y_cstride, y_coff, mean, invstd, scale, shift, partial};
  hipLaunchKernelGGL((maxpool2_bwd_kernel<true>), dim3(blocks), dim3(256), (size_t)2 * 256 * 8 * 4, (hipStream_t)s, B, H, W,
                     C, (const char*)a, a_cstride, a_coff, (const char*)dp, dp_cstride, dp_coff, (const char*)add,
                     add_cstride, add_coff, (char*)da, da_cstride, da_coff, st);
  return uh_check_launch("maxpool2_bwd_bnstat");
}

// ---- 1x1 head: logits[b][oc][y][x] = sum_c a[m][c]*w[oc][c] + bias[oc] ---------------------------------
// C/8 lanes share one pixel (C/8 a power of two <= 64) and combine with xor-shuffles.
// COUNTS: the launch also binarises its logits (sigmoid(x) > 0.5 in fp32, the rule of trainer/k_fold.py:289-290) and
// tallies tp / tn / fp / fn against a mask, per plane = blockIdx.y (one plane per image, or one for the whole batch):
// the evaluation loop then reads neither the logits nor the mask a second time.  The logit arithmetic is the same
// instruction sequence with and without COUNTS, so the tallies equal those of uh_confusion_counts over the logits of
// uh_head_fwd bit for bit.
template <bool COUNTS>
__global__ __launch_bounds__(256) void head_fwd_kernel(int B, int H, int W, int C, int OC, const char* __restrict__ a,
                                                       int a_cs, int a_co, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ logits,
                                                       long long plane_pix, const float* __restrict__ mask,
                                                       unsigned long long* __restrict__ counts) {
  const int cpr = C / 8, ppi = 256 / cpr;
  const int pp = threadIdx.x / cpr, cc = threadIdx.x % cpr;
  const int Wp = W + 1, Hp = H + 1;
  const long long first = (long long)blockIdx.y * plane_pix;
  const long long npix = first + plane_pix;
  unsigned cnt[4] = {0u, 0u, 0u, 0u};
  for (long long base = first + (long long)blockIdx.x * ppi; base < npix; base += (long long)gridDim.x * ppi) {
    const long long pix = base + pp;
    const bool ok = pix < npix;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    long long b = 0;
    int y = 0, x = 0;
    if (ok) {
      x = (int)(pix % W);
      const long long r = pix / W;
      y = (int)(r % H);
      b = r / H;
      const long long m = (b * Hp + y) * Wp + x;
      unpack8(*(const U4*)(a + (m * a_cs + a_co + cc * 8) * 2), v);
    }
    for (int oc = 0; oc < OC; ++oc) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j] * w[oc * C + cc * 8 + j];
      for (int o = cpr >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (ok && cc == 0) {
        const float lg = s + (bias ? bias[oc] : 0.f);
        const long long at = ((b * OC + oc) * H + y) * W + x;
        if (logits) logits[at] = lg;
        if constexpr (COUNTS) {                       // OC == 1 here: `at` is also the mask's index
          const float mk = mask[at];
          const bool p = 1.0f / (1.0f + expf(-lg)) > 0.5f;
          const bool t1 = mk == 1.0f, t0 = mk == 0.0f;
          cnt[0] += (p && t1);
          cnt[1] += (!p && t0);
          cnt[2] += (p && t0);
          cnt[3] += (!p && t1);
        }
      }
    }
  }
  if constexpr (COUNTS) {
    __shared__ unsigned red[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned t = cnt[k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      if (lane == 0) red[wave][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
      const unsigned long long t = (unsigned long long)red[0][threadIdx.x] + red[1][threadIdx.x] +
                                   red[2][threadIdx.x] + red[3][threadIdx.x];
      if (t) atomicAdd(&counts[(long long)blockIdx.y * 4 + threadIdx.x], t);    // integer adds commute
    }
  }
}

static bool head_c_ok(int C) { return C >= 8 && C <= 512 && (C & (C - 1)) == 0; }

extern "C" int uh_head_fwd(int B, int H, int W, int C, int OC, const void* a, int a_cstride, int a_coff,
                           const float* w, const float* bias, float* logits, uh_stream s) {
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && OC > 0 && a && w && logits, "head_fwd: bad arguments");
  UH_REQUIRE(head_c_ok(C), "head_fwd: C=%d must be a power of two in [8,512]", C);
  const long long npix = (long long)B * H * W;
  const int ppi = 256 / (C / 8);
  long long blocks = (npix + ppi - 1) / ppi;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(head_fwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, B, H, W, C, OC,
                     (const char*)a, a_cstride, a_coff, w, bias, logits, npix, nullptr, nullptr);
  return uh_check_launch("head_fwd");
}

extern "C" int uh_head_fwd_counts(int B, int H, int W, int C, const void* a, int a_cstride, int a_coff, const float* w,
                                  const float* bias, float* logits, const float* mask, int per_sample,
                                  long long* counts, uh_stream s) {
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && a && w && mask && counts, "head_fwd_counts: bad arguments");
  UH_REQUIRE(head_c_ok(C), "head_fwd_counts: C=%d must be a power of two in [8,512]", C);
  UH_REQUIRE(!per_sample || B <= 65535, "head_fwd_counts: at most 65535 images per launch with per-sample counts");
  const int planes = per_sample ? B : 1;
  const long long plane_pix = (long long)B * H * W / planes;
  UH_REQUIRE(plane_pix < (1ll << 40), "head_fwd_counts: plane of %lld pixels too large", plane_pix);
  const int ppi = 256 / (C / 8);
  long long blocks = (plane_pix + ppi - 1) / ppi;
  const long long want = (4096 + planes - 1) / planes;      // a thread tallies in 32 bits: its share stays far below 2^32
  if (blocks > want) blocks = want;
  hipLaunchKernelGGL(head_fwd_kernel<true>, dim3((unsigned)blocks, (unsigned)planes), dim3(256), 0, (hipStream_t)s, B, H, W,
                     C, 1, (const char*)a, a_cstride, a_coff, w, bias, logits, plane_pix, mask,
                     reinterpret_cast<unsigned long long*>(counts));
  return uh_check_launch("head_fwd_counts");
}

// backward: da[m][c] = sum_oc dl[oc][m]*w[oc][c] (0 on pads); partial[blk][oc][C+1] = (sum_m dl*a[m][c], sum_m dl)
static inline int head_bwd_blocks(int B, int H, int W, int C) {
  const int rpi = 256 / (C / 8);
  long long b = (uh_mp(B, H, W) + rpi - 1) / rpi;
  if (b > 1024) b = 1024;
  return (int)b;
}
extern "C" int uh_head_bwd_parts(int B, int H, int W, int C) { return head_c_ok(C) ? head_bwd_blocks(B, H, W, C) : 0; }

// STATS: the head's input is the output of a conv + BatchNorm + ReLU layer; that layer's BatchNorm-backward sums are
// taken from da as it is stored (see PoolBnStat), into st.partial[gridDim.x][2][C].
template <bool STATS>
__global__ __launch_bounds__(256) void head_bwd_kernel(int B, int H, int W, int C, int OC, const char* __restrict__ a,
                                                       int a_cs, int a_co, const float* __restrict__ w,
                                                       const float* __restrict__ dl, char* __restrict__ da, int da_cs,
                                                       int da_co, float* __restrict__ partial, PoolBnStat st) {
  extern __shared__ float lds[];
  const int cpr = C / 8, rpi = 256 / cpr;
  const int rr = threadIdx.x / cpr, cc = threadIdx.x % cpr;
  const int Wp = W + 1, Hp = H + 1;
  const long long Mp = (long long)B * Hp * Wp;
  float* out_row = partial + (long long)blockIdx.x * OC * (C + 1);
  float bacc[2][8], sc[8], sh[8], mu[8], is[8];
  if (STATS) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bacc[0][j] = bacc[1][j] = 0.f;
      sc[j] = st.scale[cc * 8 + j];
      sh[j] = st.shift[cc * 8 + j];
      mu[j] = st.mean[cc * 8 + j];
      is[j] = st.invstd[cc * 8 + j];
    }
  }
  for (int oc = 0; oc < OC; ++oc) {
    // one pass per output channel keeps the register footprint fixed; OC is 1 in practice
    float acc[1][8], dbs = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[0][j] = 0.f;
    for (long long m = (long long)blockIdx.x * rpi + rr; m < Mp; m += (long long)gridDim.x * rpi) {
      const int x = (int)(m % Wp);
      const long long r = m / Wp;
      const int y = (int)(r % Hp);
      const long long b = r / Hp;
      float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (x < W && y < H) {
        float v[8];
        unpack8(*(const U4*)(a + (m * a_cs + a_co + cc * 8) * 2), v);
        const float d = dl[((b * OC + oc) * H + y) * W + x];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[0][j] += d * v[j];
        if (cc == 0) dbs += d;
        if (oc == OC - 1) {
          for (int k = 0; k < OC; ++k) {
            const float dk = dl[((b * OC + k) * H + y) * W + x];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += dk * w[k * C + cc * 8 + j];
          }
        }
      }
      if (oc == OC - 1) {
        const U4 packed = pack8(o);
        *(U4*)(da + (m * da_cs + da_co + cc * 8) * 2) = packed;
        if (STATS && x < W && y < H) {
          float d8[8], yv[8];
          unpack8(packed, d8);
          unpack8h(*(const U4*)(st.y + (m * st.y_cs + st.y_co + cc * 8) * 2), yv);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float d = (yv[j] * sc[j] + sh[j] > 0.f) ? d8[j] : 0.f;
            bacc[0][j] += d;
            bacc[1][j] += d * (yv[j] - mu[j]) * is[j];
          }
        }
      }
    }
    __syncthreads();
    block_combine_store<1>(acc, C, cpr, rpi, lds, out_row + oc * (C + 1));
    // bias gradient: threads with cc == 0 hold per-row-slot sums
    __syncthreads();
    if (cc == 0) lds[rr] = dbs;
    __syncthreads();
    if (threadIdx.x == 0) {
      float v = 0.f;
      for (int r = 0; r < rpi; ++r) v += lds[r];
      out_row[oc * (C + 1) + C] = v;
    }
  }
  if (STATS) {
    __syncthreads();
    block_combine_store<2>(bacc, C, cpr, rpi, lds, st.partial + (long long)blockIdx.x * 2 * C);
  }
}

extern "C" int uh_head_bwd(int B, int H, int W, int C, int OC, const void* a, int a_cstride, int a_coff,
                           const float* w, const float* dlogits, void* da, int da_cstride, int da_coff, float* partial,
                           uh_stream s) {
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && OC > 0 && a && w && dlogits && da && partial, "head_bwd: bad arguments");
  UH_REQUIRE(head_c_ok(C), "head_bwd: C=%d must be a power of two in [8,512]", C);
  const int rpi = 256 / (C / 8);
  hipLaunchKernelGGL((head_bwd_kernel<false>), dim3(head_bwd_blocks(B, H, W, C)), dim3(256), (size_t)rpi * C * 4,
                     (hipStream_t)s, B, H, W, C, OC, (const char*)a, a_cstride, a_coff, w, dlogits, (char*)da, da_cstride,
                     da_coff, partial, PoolBnStat{});
  return uh_check_launch("head_bwd");
}

// uh_head_bwd that also leaves the BatchNorm-backward sums of the layer that produced `a` (fp16 conv output y, batch
// statistics mean/invstd, affine scale/shift) in bn_partial[uh_head_bwd_parts()][2][C].
extern "C" int uh_head_bwd_bnstat(int B, int H, int W, int C, int OC, const void* a, int a_cstride, int a_coff,
                                  const float* w, const float* dlogits, void* da, int da_cstride, int da_coff,
                                  float* partial, const void* y, int y_cstride, int y_coff, const float* mean,
                                  const float* invstd, const float* scale, const float* shift, float* bn_partial,
                                  uh_stream s) {
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && OC > 0 && a && w && dlogits && da && partial, "head_bwd_bnstat: bad arguments");
  UH_REQUIRE(y && mean && invstd && scale && shift && bn_partial, "head_bwd_bnstat: null pointer");
  UH_REQUIRE(head_c_ok(C), "head_bwd_bnstat: C=%d must be a power of two in [8,512]", C);
  const int rpi = 256 / (C / 8);
  PoolBnStat st{(const char*)y, y_cstride, y_coff, mean, invstd, scale, shift, bn_partial};
  hipLaunchKernelGGL((head_bwd_kernel<true>), dim3(head_bwd_blocks(B, H, W, C)), dim3(256), (size_t)2 * rpi * C * 4,
                     (hipStream_t)s, B, H, W, C, OC, (const char*)a, a_cstride, a_coff, w, dlogits, (char*)da, da_cstride,
                     da_coff, partial, st);
  return uh_check_launch("head_bwd_bnstat");
}
