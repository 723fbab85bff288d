// First layer: Conv2d(Cin <= 4 -> 64, 3x3, pad 1) straight from the caller's NCHW fp32 image.
// K = 9*Cin <= 36 is too shallow for the matrix cores to pay, and the layer is bound by writing
// its 64-channel output, so it runs on the vector ALUs in fp32: a wave covers 64 consecutive
// padded-linear positions, each of the four waves of a workgroup produces 16 of the 64 output
// channels (weights are wave-uniform, i.e. scalar loads), and the NCHW->NHWC/bf16 layout change
// is fused into the store.  Batch-norm statistics come from the fp32 results.
#include "common.h"

template <int CIN>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(int B, int H, int W, const float* __restrict__ x,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias, int mode,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, char* __restrict__ out,
                                                             int o_cs, int o_co, float* __restrict__ stat_partial) {
  constexpr int KK = CIN * 9;
  const int lane = threadIdx.x & 63;
  const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // 16-channel quarter
  const int Wp = W + 1, Hp = H + 1;
  const long long Mp = (long long)B * Hp * Wp;
  const long long mblk = (long long)blockIdx.x * 512;
  float s1[16], s2[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) s1[j] = s2[j] = 0.f;

  for (int it = 0; it < 8; ++it) {
    const long long m = mblk + it * 64 + lane;
    if (m >= Mp) continue;
    const int px = (int)(m % Wp);
    const long long r = m / Wp;
    const int py = (int)(r % Hp);
    const long long b = r / Hp;
    const bool interior = (px < W && py < H);
    float in[KK];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int yy = py + ky - 1, xx = px + kx - 1;
          const bool ok = interior && yy >= 0 && yy < H && xx >= 0 && xx < W;
          in[ci * 9 + ky * 3 + kx] = ok ? x[((b * CIN + ci) * H + yy) * W + xx] : 0.f;
        }
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = (bias && mode != UH_EPI_AFFINE_RELU) ? bias[q * 16 + j] : 0.f;
#pragma unroll
    for (int k = 0; k < KK; ++k)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] += in[k] * w[(q * 16 + j) * KK + k];
    if (mode == UH_EPI_AFFINE_RELU) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        // folded inference BN: shift already contains the conv bias
        acc[j] = interior ? fmaxf(acc[j] * scale[q * 16 + j] + shift[q * 16 + j], 0.f) : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        acc[j] = interior ? acc[j] : 0.f;
        s1[j] += acc[j];
        s2[j] += acc[j] * acc[j];
      }
    }
    float lo[8], hi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      lo[j] = acc[j];
      hi[j] = acc[8 + j];
    }
    char* dst = out + (m * o_cs + o_co + q * 16) * 2;
    if (mode == UH_EPI_RAW_STATS) {   // BatchNorm input: fp16
      *(U4*)dst = pack8h(lo);
      *(U4*)(dst + 16) = pack8h(hi);
    } else {
      *(U4*)dst = pack8(lo);
      *(U4*)(dst + 16) = pack8(hi);
    }
  }
  if (mode == UH_EPI_RAW_STATS) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float a = wave_sum(s1[j]);
      const float c = wave_sum(s2[j]);
      if (lane == 0) {
        stat_partial[((long long)blockIdx.x * 2 + 0) * 64 + q * 16 + j] = a;
        stat_partial[((long long)blockIdx.x * 2 + 1) * 64 + q * 16 + j] = c;
      }
    }
  }
}

extern "C" int uh_conv_first_parts(int B, int H, int W) { return (int)((uh_mp(B, H, W) + 511) / 512); }

extern "C" int uh_conv_first_fwd(int B, int H, int W, int Cin, int Cout, const float* x, const float* w,
                                 const float* bias, int mode, const float* scale, const float* shift, void* out,
                                 int o_cstride, int o_coff, float* stat_partial, uh_stream s) {
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && x && w && out, "conv_first_fwd: bad arguments");
  UH_REQUIRE(Cin >= 1 && Cin <= 4, "conv_first_fwd: Cin=%d must be 1..4", Cin);
  UH_REQUIRE(Cout == 64, "conv_first_fwd: Cout=%d (only 64 is built)", Cout);
  UH_REQUIRE(mode == UH_EPI_RAW_STATS || mode == UH_EPI_AFFINE_RELU, "conv_first_fwd: bad mode %d", mode);
  UH_REQUIRE(mode != UH_EPI_RAW_STATS || stat_partial, "conv_first_fwd: RAW_STATS needs stat_partial");
  UH_REQUIRE(mode != UH_EPI_AFFINE_RELU || (scale && shift), "conv_first_fwd: AFFINE_RELU needs scale/shift");
  UH_REQUIRE(o_cstride % 8 == 0 && o_coff % 8 == 0 && o_cstride >= o_coff + 64, "conv_first_fwd: bad out view");
  const dim3 grid((unsigned)uh_conv_first_parts(B, H, W));
  hipStream_t st = (hipStream_t)s;
#define LAUNCH(CI)                                                                                                 \
  hipLaunchKernelGGL((conv_first_fwd_kernel<CI>), grid, dim3(256), 0, st, B, H, W, x, w, bias, mode, scale, shift, \
                     (char*)out, o_cstride, o_coff, stat_partial)
  switch (Cin) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    default: LAUNCH(4); break;
  }
#undef LAUNCH
  return uh_check_launch("conv_first_fwd");
}

// ---- weight gradient: dW[co][ci][ky][kx] = sum_pixels dy[m][co] * x[b][ci][y+ky-1][x+kx-1] ---------------
// lane = output channel; a workgroup walks 16 image rows of a 64-pixel-wide strip; per row the
// three input rows it touches are staged in LDS (fp32, zero padded) and read back as broadcasts.
template <int CIN>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(int B, int H, int W, const float* __restrict__ x,
                                                               const char* __restrict__ dy, int dy_cs, int dy_co,
                                                               float* __restrict__ partial) {
  constexpr int KK = CIN * 9;
  __shared__ float xs[CIN * 3 * 66];
  __shared__ float red[4 * 64 * KK];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 16;
  const long long b = blockIdx.z;
  const int Wp = W + 1, Hp = H + 1;
  float acc[KK];
#pragma unroll
  for (int k = 0; k < KK; ++k) acc[k] = 0.f;
  for (int ry = 0; ry < 16; ++ry) {
    const int y = y0 + ry;
    if (y >= H) break;
    __syncthreads();
    for (int i = tid; i < CIN * 3 * 66; i += 256) {
      const int xx = x0 + (i % 66) - 1;
      const int ky = (i / 66) % 3;
      const int ci = i / (66 * 3);
      const int yy = y + ky - 1;
      xs[i] = (xx >= 0 && xx < W && yy >= 0 && yy < H) ? x[((b * CIN + ci) * H + yy) * W + xx] : 0.f;
    }
    __syncthreads();
    for (int p = 0; p < 16; ++p) {
      const int px = wave * 16 + p;
      if (x0 + px >= W) break;
      const long long m = (b * Hp + y) * Wp + x0 + px;
      const float d = bf2f(*(const bf16*)(dy + (m * dy_cs + dy_co + lane) * 2));
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) acc[ci * 9 + ky * 3 + kx] += d * xs[(ci * 3 + ky) * 66 + px + kx];
    }
  }
#pragma unroll
  for (int k = 0; k < KK; ++k) red[(wave * 64 + lane) * KK + k] = acc[k];
  __syncthreads();
  const long long blk = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  for (int i = tid; i < 64 * KK; i += 256)
    partial[blk * 64 * KK + i] = red[i] + red[64 * KK + i] + red[2 * 64 * KK + i] + red[3 * 64 * KK + i];
}

extern "C" int uh_conv_first_wgrad_parts(int B, int H, int W) { return uh_cdiv(W, 64) * uh_cdiv(H, 16) * B; }

extern "C" int uh_conv_first_wgrad(int B, int H, int W, int Cin, int Cout, const float* x, const void* dy,
                                   int dy_cstride, int dy_coff, float* partial, uh_stream s) {
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && x && dy && partial, "conv_first_wgrad: bad arguments");
  UH_REQUIRE(Cin >= 1 && Cin <= 4, "conv_first_wgrad: Cin=%d must be 1..4", Cin);
  UH_REQUIRE(Cout == 64, "conv_first_wgrad: Cout=%d (only 64 is built)", Cout);
  const dim3 grid(uh_cdiv(W, 64), uh_cdiv(H, 16), B);
  hipStream_t st = (hipStream_t)s;
#define LAUNCH(CI)                                                                                              \
  hipLaunchKernelGGL((conv_first_wgrad_kernel<CI>), grid, dim3(256), 0, st, B, H, W, x, (const char*)dy, dy_cstride, \
                     dy_coff, partial)
  switch (Cin) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    default: LAUNCH(4); break;
  }
#undef LAUNCH
  return uh_check_launch("conv_first_wgrad");
}
