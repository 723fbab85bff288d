// Self-regularisation loss of the reference (models/losses.py:6-217), computed from the activations where they
// lie: padded-linear bf16 feature maps in, a handful of floats out.
//
//   differentiable part  0.6 * (1 - mean_b Dice_b(sigmoid(logits), targets))      losses.py:18-41, :72
//                        (the 0.4 * CrossEntropy term is identically zero for one-channel logits: log_softmax over
//                        a single class is 0, and so is its gradient)
//   constants            0.015 * (LSRC + LIFD), computed without gradient          losses.py:63-72
//     LSRC  mean over features of mse(feature[:, subset], avg-pooled last feature)   losses.py:81-155
//     LIFD  mean over features of mse(first half, second half) of the channels ordered by the mean of their
//           spatially L2-normalised maps                                             losses.py:157-217
//
// Every kernel here is HBM-bound: a feature map is read twice (plane sums for the ordering, then the two squared
// differences from an LDS-staged tile), the last feature once more for its pooled pyramid.
#include "ew.h"

// ---- pooled pyramid of the last feature: 2x2 means, padded-linear fp32 out ---------------------------------
// adaptive_avg_pool2d onto an exact divisor of the size is a block mean; chaining 2x2 means gives the same value
// up to fp32 rounding.  Pad rows of the output are written as zeros, so a squared difference against a feature
// map (whose pad rows are zero too) may run over pad rows.
template <bool INF32>
__global__ __launch_bounds__(256) void selfreg_pool2_kernel(int B, int Ho, int Wo, int C, const char* __restrict__ in,
                                                            int cs, int co, float* __restrict__ out) {
  const int c4 = C / 4, Wpo = Wo + 1, Hpo = Ho + 1, Wpi = 2 * Wo + 1, Hpi = 2 * Ho + 1;
  const long long total = (long long)B * Hpo * Wpo * c4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int q = (int)(i % c4);
    const long long mo = i / c4;
    const int x = (int)(mo % Wpo), y = (int)((mo / Wpo) % Hpo), b = (int)(mo / ((long long)Wpo * Hpo));
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x < Wo && y < Ho) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const long long mi = ((long long)b * Hpi + 2 * y + (d >> 1)) * Wpi + 2 * x + (d & 1);
        if (INF32) {
          const float4 v = *(const float4*)(in + (mi * cs + co + q * 4) * 4);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        } else {
          const uint2 u = *(const uint2*)(in + (mi * cs + co + q * 4) * 2);
          acc.x += __uint_as_float(u.x << 16); acc.y += __uint_as_float(u.x & 0xffff0000u);
          acc.z += __uint_as_float(u.y << 16); acc.w += __uint_as_float(u.y & 0xffff0000u);
        }
      }
      acc.x *= 0.25f; acc.y *= 0.25f; acc.z *= 0.25f; acc.w *= 0.25f;
    }
    *(float4*)(out + mo * C + q * 4) = acc;
  }
}

extern "C" int uh_selfreg_pool2(int B, int Ho, int Wo, int C, const void* in, int in_f32, int cstride, int coff,
                                float* out, uh_stream s) {
  UH_REQUIRE(B > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0 && in && out, "selfreg_pool2: B=%d Ho=%d Wo=%d C=%d", B,
             Ho, Wo, C);
  UH_REQUIRE(cstride >= coff + C && coff % 8 == 0 && cstride % 8 == 0, "selfreg_pool2: channel window");
  const long long total = uh_mp(B, Ho, Wo) * (C / 4);
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (in_f32)
    hipLaunchKernelGGL(selfreg_pool2_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, B, Ho, Wo, C,
                       (const char*)in, cstride, coff, out);
  else
    hipLaunchKernelGGL(selfreg_pool2_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, B, Ho, Wo, C,
                       (const char*)in, cstride, coff, out);
  return uh_check_launch("selfreg_pool2");
}

// ---- pass 1: per (image, channel) sum and sum of squares ------------------------------------------------------
// Pad rows are zero, so the sums run over all (H+1)(W+1) rows of an image.  partial[b][part][2][C].
static inline int selfreg_stat_parts(int B, int H, int W, int C) {
  const long long R = (long long)(H + 1) * (W + 1);
  const int rpi = ew_rows_per_iter(C);
  long long p = ((R + rpi - 1) / rpi + 7) / 8;
  long long cap = 2048 / B;
  if (cap < 8) cap = 8;
  if (p > cap) p = cap;
  if (p < 1) p = 1;
  return (int)p;
}
extern "C" int uh_selfreg_stat_parts(int B, int H, int W, int C) {
  return (B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && C <= 1024) ? selfreg_stat_parts(B, H, W, C) : 0;
}

__global__ __launch_bounds__(256) void selfreg_plane_stats_kernel(long long R, int C, const char* __restrict__ f, int cs,
                                                                  int co, float* __restrict__ partial) {
  extern __shared__ float lds[];
  const int cpr = C / 8, rpi = 256 / cpr;
  const int rr = threadIdx.x / cpr, cc = threadIdx.x % cpr;
  const long long base = (long long)blockIdx.y * R;
  float acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = 0.f;
  if (rr < rpi) {
    for (long long r = (long long)blockIdx.x * rpi + rr; r < R; r += (long long)gridDim.x * rpi) {
      float v[8];
      unpack8(*(const U4*)(f + ((base + r) * cs + co + cc * 8) * 2), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc[0][j] += v[j];
        acc[1][j] += v[j] * v[j];
      }
    }
  }
  block_combine_store<2>(acc, C, cpr, rpi, lds, partial + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 2 * C);
}

extern "C" int uh_selfreg_plane_stats(int B, int H, int W, int C, const void* f, int cstride, int coff, float* partial,
                                      uh_stream s) {
  UH_REQUIRE(uh_selfreg_stat_parts(B, H, W, C) > 0 && f && partial, "selfreg_plane_stats: B=%d H=%d W=%d C=%d", B, H, W, C);
  UH_REQUIRE(cstride >= coff + C && coff % 8 == 0 && cstride % 8 == 0, "selfreg_plane_stats: channel window");
  UH_REQUIRE(B <= 65535, "selfreg_plane_stats: B=%d", B);
  const int parts = selfreg_stat_parts(B, H, W, C), rpi = ew_rows_per_iter(C);
  hipLaunchKernelGGL(selfreg_plane_stats_kernel, dim3(parts, B), dim3(256), (size_t)rpi * 2 * C * 4, (hipStream_t)s,
                     (long long)(H + 1) * (W + 1), C, (const char*)f, cstride, coff, partial);
  return uh_check_launch("selfreg_plane_stats");
}

// ---- channel ordering: importance[c] = mean over (b, y, x) of f / (||f[b,c]||_2 + 1e-8), descending -----------
// Step 1, one workgroup per image: add the parts and leave sum / (sqrt(sum of squares) + 1e-8) in the image's
// first partial row (each thread rewrites only the column it alone has read).  Step 2, one workgroup: mean over
// images, then ranks by counting, ties towards the lower channel index (torch.argsort leaves ties unspecified;
// all-zero channels tie at 0 and contribute the same whichever way they are ordered).
__global__ __launch_bounds__(1024) void selfreg_plane_finish_kernel(int parts, int C, float* __restrict__ partial) {
  float* img = partial + (long long)blockIdx.x * parts * 2 * C;
  for (int c = threadIdx.x; c < C; c += 1024) {
    double sum = 0.0, sq = 0.0;
#pragma unroll 8
    for (int p = 0; p < parts; ++p) {
      sum += (double)img[(long long)p * 2 * C + c];
      sq += (double)img[(long long)p * 2 * C + C + c];
    }
    img[c] = (float)sum / (sqrtf((float)sq) + 1e-8f);
  }
}

__global__ __launch_bounds__(1024) void selfreg_rank_kernel(int B, int parts, int C, float inv_count,
                                                            const float* __restrict__ partial, int* __restrict__ order) {
  __shared__ float imp[1024];
  const int c = threadIdx.x;
  float mine = 0.f;
  if (c < C) {
    double tot = 0.0;
    for (int b = 0; b < B; ++b) tot += (double)partial[(long long)b * parts * 2 * C + c];
    mine = (float)tot * inv_count;
    imp[c] = mine;
    order[c] = c;
  }
  __syncthreads();
  if (c < C) {
    int rank = 0;
    for (int k = 0; k < C; ++k) {
      const float v = imp[k];
      rank += (v > mine) || (v == mine && k < c);
    }
    if (rank < C) order[rank] = c;
  }
}

extern "C" int uh_selfreg_rank(int B, int H, int W, int C, float* partial, int* order, uh_stream s) {
  UH_REQUIRE(uh_selfreg_stat_parts(B, H, W, C) > 0 && partial && order, "selfreg_rank: B=%d H=%d W=%d C=%d", B, H, W, C);
  const float inv = (float)(1.0 / ((double)B * H * W));
  const int parts = selfreg_stat_parts(B, H, W, C);
  hipLaunchKernelGGL(selfreg_plane_finish_kernel, dim3(B), dim3(1024), 0, (hipStream_t)s, parts, C, partial);
  hipLaunchKernelGGL(selfreg_rank_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, B, parts, C, inv, partial, order);
  return uh_check_launch("selfreg_rank");
}

// ---- pass 2: both squared differences from one LDS-staged tile of rows ----------------------------------------
//   pair  = sum over rows, j < C/2 of (f[order[j]] - f[order[j + C/2]])^2
//   lsrc  = sum over rows, j < nsel of (f[sel_f[j]] - pooled[sel_p[j]])^2          (pooled == NULL: skipped)
// partial[workgroup][2] = (lsrc, pair).
static inline int selfreg_tile_rows(int C) {
  int r = 8192 / C;
  return r > 128 ? 128 : (r < 1 ? 1 : r);
}
static inline int selfreg_pair_parts(long long Mp, int C) {
  const int rows = selfreg_tile_rows(C);
  long long t = (Mp + rows - 1) / rows;
  return (int)(t > 2048 ? 2048 : t);
}
extern "C" int uh_selfreg_pair_parts(int B, int H, int W, int C) {
  return (B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && C <= 1024) ? selfreg_pair_parts(uh_mp(B, H, W), C) : 0;
}

__device__ __forceinline__ float lds_bf16(const uint16_t* p) { return __uint_as_float((uint32_t)(*p) << 16); }

// LDS: feature tile [rows][C] bf16 | pooled tile [rows][PC] fp32 or bf16 | order[C], sel_f[nsel], sel_p[nsel] (u16)
template <bool PF32>
__global__ __launch_bounds__(256) void selfreg_pair_kernel(long long Mp, int C, int rows, const char* __restrict__ f,
                                                           int cs, int co, const int* __restrict__ order,
                                                           const char* __restrict__ pooled, int pcs, int pco, int PC,
                                                           int nsel, const int* __restrict__ sel_f,
                                                           const int* __restrict__ sel_p, float* __restrict__ partial) {
  extern __shared__ uint16_t tile[];
  uint16_t* ptile = tile + (size_t)rows * C;        // PF32: read as float
  uint16_t* ord = ptile + (pooled ? (size_t)rows * PC * (PF32 ? 2 : 1) : 0);
  uint16_t* sf = ord + C;
  uint16_t* sp = sf + nsel;
  const int t = threadIdx.x, cpr = C / 8, half = C / 2;
  for (int i = t; i < C; i += 256) ord[i] = (uint16_t)order[i];
  for (int i = t; i < nsel; i += 256) {
    sf[i] = (uint16_t)sel_f[i];
    sp[i] = (uint16_t)sel_p[i];
  }
  float pair = 0.f, lsrc = 0.f;
  const long long ntiles = (Mp + rows - 1) / rows;
  for (long long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const long long m0 = tl * rows;
    const int live = (int)((Mp - m0) < rows ? (Mp - m0) : rows);
    __syncthreads();                                // previous tile fully consumed (and the index tables written)
    for (int i = t; i < rows * cpr; i += 256) {
      const int r = i / cpr, cc = i % cpr;
      U4 v = {0u, 0u, 0u, 0u};
      if (r < live) v = *(const U4*)(f + ((m0 + r) * cs + co + cc * 8) * 2);
      *(U4*)(tile + (size_t)r * C + cc * 8) = v;
    }
    if (pooled) {
      if (PF32) {
        const int q4 = PC / 4;
        for (int i = t; i < live * q4; i += 256) {
          const int r = i / q4, q = i % q4;
          *(float4*)(reinterpret_cast<float*>(ptile) + (size_t)r * PC + q * 4) =
              *(const float4*)(pooled + ((m0 + r) * pcs + pco + q * 4) * 4);
        }
      } else {
        const int q8 = PC / 8;
        for (int i = t; i < live * q8; i += 256) {
          const int r = i / q8, q = i % q8;
          *(U4*)(ptile + (size_t)r * PC + q * 8) = *(const U4*)(pooled + ((m0 + r) * pcs + pco + q * 8) * 2);
        }
      }
    }
    __syncthreads();
    {
      // a thread keeps its channel pair while it walks rows whenever the counts allow it
      const bool regular = half < 256 ? (256 % half == 0) : (half % 256 == 0);
      if (regular) {
        const int jstep = half < 256 ? half : 256, qstep = 256 / jstep;
        for (int j = t % jstep; j < half; j += jstep) {
          const int oa = ord[j], ob = ord[j + half];
          for (int q = t / jstep; q < rows; q += qstep) {
            const float d = lds_bf16(tile + (size_t)q * C + oa) - lds_bf16(tile + (size_t)q * C + ob);
            pair += d * d;
          }
        }
      } else {
        for (int i = t; i < rows * half; i += 256) {
          const int q = i / half, j = i % half;
          const float d = lds_bf16(tile + (size_t)q * C + ord[j]) - lds_bf16(tile + (size_t)q * C + ord[j + half]);
          pair += d * d;
        }
      }
    }
    if (pooled) {
      auto term = [&](int q, int fa, int pa) {
        const float pv = PF32 ? reinterpret_cast<const float*>(ptile)[(size_t)q * PC + pa]
                              : lds_bf16(ptile + (size_t)q * PC + pa);
        const float d = lds_bf16(tile + (size_t)q * C + fa) - pv;
        lsrc += d * d;
      };
      const bool regular = nsel < 256 ? (256 % nsel == 0) : (nsel % 256 == 0);
      if (regular) {
        const int jstep = nsel < 256 ? nsel : 256, qstep = 256 / jstep;
        for (int j = t % jstep; j < nsel; j += jstep) {
          const int fa = sf[j], pa = sp[j];
          for (int q = t / jstep; q < live; q += qstep) term(q, fa, pa);
        }
      } else {
        for (int i = t; i < live * nsel; i += 256) term(i / nsel, sf[i % nsel], sp[i % nsel]);
      }
    }
  }
  __shared__ float red[2][4];
  lsrc = wave_sum(lsrc);
  pair = wave_sum(pair);
  if ((t & 63) == 0) {
    red[0][t >> 6] = lsrc;
    red[1][t >> 6] = pair;
  }
  __syncthreads();
  if (t < 2) partial[blockIdx.x * 2 + t] = (red[t][0] + red[t][1]) + (red[t][2] + red[t][3]);
}

extern "C" int uh_selfreg_pair_sums(int B, int H, int W, int C, const void* f, int cstride, int coff, const int* order,
                                    const void* pooled, int pooled_f32, int p_channels, int p_cstride, int p_coff,
                                    int nsel, const int* sel_f, const int* sel_p, float* partial, uh_stream s) {
  UH_REQUIRE(uh_selfreg_pair_parts(B, H, W, C) > 0 && f && order && partial, "selfreg_pair_sums: B=%d H=%d W=%d C=%d", B,
             H, W, C);
  UH_REQUIRE(cstride >= coff + C && coff % 8 == 0 && cstride % 8 == 0, "selfreg_pair_sums: channel window");
  if (pooled) {
    UH_REQUIRE(nsel > 0 && nsel <= C && nsel <= p_channels && sel_f && sel_p, "selfreg_pair_sums: nsel=%d of C=%d", nsel, C);
    UH_REQUIRE(p_channels % 8 == 0 && p_channels <= 1024 && p_cstride >= p_coff + p_channels && p_coff % 8 == 0 &&
                   p_cstride % 8 == 0,
               "selfreg_pair_sums: pooled window %d channels at %d of %d", p_channels, p_coff, p_cstride);
  } else {
    nsel = 0;
    p_channels = 0;
  }
  const long long Mp = uh_mp(B, H, W);
  int rows = selfreg_tile_rows(C);                  // the grid walks tiles in strides, so any tile height serves
  const int p_bytes = pooled_f32 ? 4 : 2;
  if (p_channels > 0 && rows > 32768 / (p_channels * p_bytes)) rows = 32768 / (p_channels * p_bytes);
  const size_t lds = (size_t)rows * C * 2 + (size_t)rows * p_channels * p_bytes + ((size_t)C + 2 * (size_t)nsel) * 2;
  UH_REQUIRE(lds <= 64 * 1024, "selfreg_pair_sums: %zu bytes of LDS (C=%d against %d pooled channels)", lds, C, p_channels);
  const dim3 grid(selfreg_pair_parts(Mp, C));
  if (pooled_f32)
    hipLaunchKernelGGL(selfreg_pair_kernel<true>, grid, dim3(256), lds, (hipStream_t)s, Mp, C, rows, (const char*)f,
                       cstride, coff, order, (const char*)pooled, p_cstride, p_coff, p_channels, nsel, sel_f, sel_p,
                       partial);
  else
    hipLaunchKernelGGL(selfreg_pair_kernel<false>, grid, dim3(256), lds, (hipStream_t)s, Mp, C, rows, (const char*)f,
                       cstride, coff, order, (const char*)pooled, p_cstride, p_coff, p_channels, nsel, sel_f, sel_p,
                       partial);
  return uh_check_launch("selfreg_pair_sums");
}

// aux[0] (LSRC) and aux[1] (LIFD) += scale * sum of the partials of one feature; init != 0 starts from zero
__global__ __launch_bounds__(256) void selfreg_accumulate_kernel(const float* __restrict__ partial, int parts,
                                                                 double scale_lsrc, double scale_pair, int init,
                                                                 float* __restrict__ aux) {
  __shared__ double red[2][256];
  double a = 0.0, b = 0.0;
  for (int p = threadIdx.x; p < parts; p += 256) {
    a += (double)partial[p * 2];
    b += (double)partial[p * 2 + 1];
  }
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = b;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      red[0][threadIdx.x] += red[0][threadIdx.x + st];
      red[1][threadIdx.x] += red[1][threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x < 2) {
    const double add = red[threadIdx.x][0] * (threadIdx.x == 0 ? scale_lsrc : scale_pair);
    aux[threadIdx.x] = (float)((init ? 0.0 : (double)aux[threadIdx.x]) + add);
  }
}

extern "C" int uh_selfreg_accumulate(const float* partial, int parts, double scale_lsrc, double scale_pair, int init,
                                     float* aux, uh_stream s) {
  UH_REQUIRE(partial && aux && parts > 0, "selfreg_accumulate: bad arguments");
  hipLaunchKernelGGL(selfreg_accumulate_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, partial, parts, scale_lsrc,
                     scale_pair, init, aux);
  return uh_check_launch("selfreg_accumulate");
}

// ---- differentiable part: Dice on sigmoid(logits) with squared terms in the denominator ------------------------
static inline int selfreg_loss_parts(int B, long long n) {
  long long p = (n + 256 * 16 - 1) / (256 * 16);
  long long cap = 1024 / B;
  if (cap < 4) cap = 4;
  if (p > cap) p = cap;
  return (int)(p < 1 ? 1 : p);
}
extern "C" size_t uh_selfreg_loss_workspace(int B, long long n) {
  return (B > 0 && n > 0) ? (size_t)B * selfreg_loss_parts(B, n) * 3 * sizeof(float) : 0;
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void selfreg_dice_partial_kernel(long long n, const float* __restrict__ logits,
                                                                   const float* __restrict__ targets,
                                                                   float* __restrict__ partial) {
  const long long base = (long long)blockIdx.y * n;
  float a = 0.f, b = 0.f, c = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float p = sigmoidf(logits[base + i]), t = targets[base + i];
    a += p * t;
    b += p * p;
    c += t * t;
  }
  __shared__ float red[3][4];
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a;
    red[1][threadIdx.x >> 6] = b;
    red[2][threadIdx.x >> 6] = c;
  }
  __syncthreads();
  if (threadIdx.x < 3)
    partial[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 3 + threadIdx.x] =
        (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// loss = w_dice * (1 - mean_b (2 I_b + s) / (U_b + s)) + w_aux * (aux[0] + aux[1]);  s = 1e-6 (losses.py:18)
// coef[b] = (a_b, c_b) with d(1 - mean Dice)/d p_i = a_b * t_i + c_b * p_i for the pixels of image b
__global__ __launch_bounds__(256) void selfreg_loss_finish_kernel(int B, int parts, const float* __restrict__ partial,
                                                                  const float* __restrict__ aux, float w_dice,
                                                                  float w_aux, float* __restrict__ loss,
                                                                  float* __restrict__ coef) {
  __shared__ double red[256];
  double dice = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) {
    double I = 0.0, P2 = 0.0, T2 = 0.0;
    for (int p = 0; p < parts; ++p) {
      const float* row = partial + ((long long)b * parts + p) * 3;
      I += (double)row[0];
      P2 += (double)row[1];
      T2 += (double)row[2];
    }
    const double s = 1e-6, U = P2 + T2 + s, num = 2.0 * I + s;
    dice += num / U;
    coef[b * 2] = (float)(-2.0 / ((double)B * U));
    coef[b * 2 + 1] = (float)(2.0 * num / ((double)B * U * U));
  }
  red[threadIdx.x] = dice;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double v = (double)w_dice * (1.0 - red[0] / (double)B);
    if (aux) v += (double)w_aux * ((double)aux[0] + (double)aux[1]);
    loss[0] = (float)v;
  }
}

extern "C" int uh_selfreg_loss_fwd(int B, long long n, const float* logits, const float* targets, const float* aux,
                                   float w_dice, float w_aux, float* loss, float* coef, void* workspace,
                                   size_t workspace_bytes, uh_stream s) {
  UH_REQUIRE(B > 0 && B <= 65535 && n > 0 && logits && targets && loss && coef && workspace,
             "selfreg_loss_fwd: B=%d n=%lld", B, n);
  UH_REQUIRE(workspace_bytes >= uh_selfreg_loss_workspace(B, n), "selfreg_loss_fwd: workspace of %zu bytes, need %zu",
             workspace_bytes, uh_selfreg_loss_workspace(B, n));
  const int parts = selfreg_loss_parts(B, n);
  hipLaunchKernelGGL(selfreg_dice_partial_kernel, dim3(parts, B), dim3(256), 0, (hipStream_t)s, n, logits, targets,
                     (float*)workspace);
  hipLaunchKernelGGL(selfreg_loss_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, B, parts,
                     (const float*)workspace, aux, w_dice, w_aux, loss, coef);
  return uh_check_launch("selfreg_loss_fwd");
}

// dlogits = gout * w_dice * (a_b t + c_b p) * p (1 - p)
__global__ __launch_bounds__(256) void selfreg_loss_bwd_kernel(long long n, const float* __restrict__ logits,
                                                               const float* __restrict__ targets,
                                                               const float* __restrict__ coef,
                                                               const float* __restrict__ gout, float w_dice,
                                                               float* __restrict__ dlogits) {
  const long long base = (long long)blockIdx.y * n;
  const float g = gout[0] * w_dice, ca = coef[blockIdx.y * 2] * g, cc = coef[blockIdx.y * 2 + 1] * g;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float p = sigmoidf(logits[base + i]), t = targets[base + i];
    dlogits[base + i] = (ca * t + cc * p) * p * (1.0f - p);
  }
}

extern "C" int uh_selfreg_loss_bwd(int B, long long n, const float* logits, const float* targets, const float* coef,
                                   const float* gout, float w_dice, float* dlogits, uh_stream s) {
  UH_REQUIRE(B > 0 && B <= 65535 && n > 0 && logits && targets && coef && gout && dlogits,
             "selfreg_loss_bwd: B=%d n=%lld", B, n);
  long long bx = (n + 256 * 4 - 1) / (256 * 4);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(selfreg_loss_bwd_kernel, dim3((unsigned)bx, B), dim3(256), 0, (hipStream_t)s, n, logits, targets,
                     coef, gout, w_dice, dlogits);
  return uh_check_launch("selfreg_loss_bwd");
}
