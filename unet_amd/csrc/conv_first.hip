// First layer: Conv2d(Cin <= 4 -> 64, 3x3, pad 1) straight from the caller's NCHW fp32 image.
//
// K = 9*Cin <= 36 is one or two MFMA k-blocks, and the layer is bound by writing its 64-channel output, so the
// point is only to keep arithmetic out of the way: on the vector ALUs the 1728 FMAs per pixel took 3.5x the
// time of the store.  The matrix cores take bf16, the layer's contract is fp32 (the input image and these
// weights are not rounded anywhere else in the network), so both operands are split x = hi + lo into two bf16
// values and the product is hi*hi + hi*lo + lo*hi with fp32 accumulation: relative error ~2^-16 per product,
// below the fp16 rounding of the stored result.
//   forward : D[channel][pixel] = W[channel][k] * X^T[k][pixel]; weight rows are permuted so that the four
//             channel tiles a lane holds for its pixel are 16 ADJACENT channels (two 16-byte stores); the
//             im2col fragment of a pixel is gathered straight from NCHW with bounds checks; BatchNorm partial
//             sums come from the fp32 accumulators.
//   wgrad   : D[channel][tap] = sum over pixels dY^T * im2col(X), 32 pixels per MFMA, dY is bf16 already.
#include "common.h"

__device__ __forceinline__ void split_bf16(float v, bf16& hi, bf16& lo) {
  hi = (bf16)v;
  lo = (bf16)(v - (float)hi);
}

// The taps a lane gathers are the same (ci, ky, kx) for every pixel it visits, so everything about them that does not
// depend on the pixel is worked out once per kernel: the element offset from the pixel's own position in channel 0 and
// the row / column displacement for the bounds test (a displacement of 2^20 marks a slot beyond the K taps: it fails
// every test).  Per pixel and tap that leaves two unsigned compares, two selects and the load.
template <int CIN, int SLOTS>
struct FirstTaps {
  int off[SLOTS], dy[SLOTS], dx[SLOTS];
  // slot t holds tap k
  __device__ __forceinline__ void set(int t, int k, int H, int W) {
    const int ci = (k * 57) >> 9, r = k - ci * 9;        // k / 9 for k < 64
    const int ky = (r * 11) >> 5, kx = r - ky * 3;       // r / 3 for r < 9
    const bool real = k < CIN * 9;
    off[t] = real ? (ci * H + ky - 1) * W + kx - 1 : 0;
    dy[t] = real ? ky - 1 : (1 << 20);
    dx[t] = real ? kx - 1 : (1 << 20);
  }
  // xb: &x[b][0][py][px] for a pixel inside the image (any valid address otherwise, with inside = false)
  __device__ __forceinline__ float get(const float* __restrict__ xb, int H, int W, int py, int px, bool inside,
                                       int t) const {
    // bitwise &: as && the second compare is compiled into a branch around it
    const bool ok = inside & ((unsigned)(py + dy[t]) < (unsigned)H) & ((unsigned)(px + dx[t]) < (unsigned)W);
    const float v = xb[ok ? off[t] : 0];      // unconditional load: a select, not a branch per tap
    return ok ? v : 0.f;
  }
};

// MODE = UH_EPI_RAW_STATS: the conv bias rides in the first free K slot (input 1.0 on interior pixels, weight = bias),
// so positions outside the image come out of the matrix cores as exact zeros and the epilogue is the statistics and
// the fp16 store only.  MODE = UH_EPI_AFFINE_RELU: relu(acc * scale + shift), bf16 store.
template <int CIN, int MODE>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(int B, int H, int W, const float* __restrict__ x,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, char* __restrict__ out,
                                                             int o_cs, int o_co, float* __restrict__ stat_partial) {
  constexpr int KK = CIN * 9;
  constexpr int NKB = (KK + 1 + 31) / 32;          // + 1: the bias slot
  constexpr bool STATS = MODE == UH_EPI_RAW_STATS;
  __shared__ float red[4][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int Wp = W + 1, Hp = H + 1;
  const long long Mp = (long long)B * Hp * Wp;
  const long long nblk = (Mp + 511) / 512;         // a workgroup walks blocks of 512 positions: blockIdx.x, + gridDim.x, ...

  // weight fragments (A operand): tile j, row r = l15 is output channel (j/2)*32 + (r/4)*8 + (j%2)*4 + r%4, so that
  // the lane of row group g ends up with channels g*8..g*8+7 (tiles 0, 1) and 32+g*8..32+g*8+7 (tiles 2, 3): each of
  // its two 16-byte stores then lands beside those of the other three groups, 64 contiguous bytes per pixel
  bf16x8 whi[NKB][4], wlo[NKB][4];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = (j >> 1) * 32 + (l15 >> 2) * 8 + (j & 1) * 4 + (l15 & 3);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int k = kb * 32 + g * 8 + t;
        float v = (k < KK) ? w[c * KK + k] : 0.f;
        if (STATS && k == KK && bias) v = bias[c];
        bf16 h, l;
        split_bf16(v, h, l);
        whi[kb][j][t] = h;
        wlo[kb][j][t] = l;
      }
    }
  FirstTaps<CIN, NKB * 8> taps;
#pragma unroll
  for (int t = 0; t < NKB * 8; ++t) taps.set(t, (t / 8) * 32 + g * 8 + t % 8, H, W);
  // the lane's slot that carries the bias (if any): slot index within its NKB * 8
  const int bias_slot = (STATS && KK / 8 % 4 == g) ? (KK / 32) * 8 + KK % 8 : -1;
  // this lane's 16 output channels: value q is channel (q/8)*32 + g*8 + q%8
  float cb[16], csc[16];
  if (!STATS) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      csc[q] = scale[(q >> 3) * 32 + g * 8 + (q & 7)];
      cb[q] = shift[(q >> 3) * 32 + g * 8 + (q & 7)];       // folded inference BN: shift already contains the conv bias
    }
  }
  float s1[16], s2[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) s1[q] = s2[q] = 0.f;

  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
  // the position of the lane's first pixel by division, the seven that follow (16 positions further each) by stepping
  long long m = blk * 512 + wave * 128 + l15;
  int px = (int)(m % Wp);
  long long b = m / Wp;
  int py = (int)(b % Hp);
  b /= Hp;
  for (int tile = 0; tile < 8; ++tile, m += 16, px += 16) {
    while (px >= Wp) {
      px -= Wp;
      if (++py == Hp) { py = 0; ++b; }
    }
    const bool interior = (m < Mp && px < W && py < H);
    const float* xb = interior ? x + ((b * CIN) * H + py) * W + px : x;
    bf16x8 xhi[NKB], xlo[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        float v = taps.get(xb, H, W, py, px, interior, kb * 8 + t);
        if (STATS && kb * 8 + t == bias_slot) v = interior ? 1.f : 0.f;
        bf16 h, l;
        split_bf16(v, h, l);
        xhi[kb][t] = h;
        xlo[kb][t] = l;
      }
    float v[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[kb][j], xhi[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[kb][j], xlo[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[kb][j], xhi[kb], acc, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[j * 4 + e] = acc[e];       // value q = j*4 + e of pixel l15
    }
    if (STATS) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {                           // zero outside the image: every tap was zero
        s1[q] += v[q];
        s2[q] += v[q] * v[q];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float t = fmaxf(v[q] * csc[q] + cb[q], 0.f);
        v[q] = interior ? t : 0.f;
      }
    }
    if (m < Mp) {
      float lo8[8], hi8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { lo8[q] = v[q]; hi8[q] = v[8 + q]; }
      char* dst = out + (m * o_cs + o_co + g * 8) * 2;
      if (STATS) {                      // BatchNorm input: fp16
        *(U4*)dst = pack8h(lo8);
        *(U4*)(dst + 64) = pack8h(hi8);
      } else {
        *(U4*)dst = pack8(lo8);
        *(U4*)(dst + 64) = pack8(hi8);
      }
    }
  }
  }
  if (STATS) {
    // sum over the 16 pixels (lanes l15) of every group g, then over the four waves
#pragma unroll
    for (int q = 0; q < 16; ++q) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        s1[q] += __shfl_xor(s1[q], o, 64);
        s2[q] += __shfl_xor(s2[q], o, 64);
      }
    }
    if (l15 == 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        red[wave][0][(q >> 3) * 32 + g * 8 + (q & 7)] = s1[q];
        red[wave][1][(q >> 3) * 32 + g * 8 + (q & 7)] = s2[q];
      }
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
      stat_partial[((long long)blockIdx.x * 2 + which) * 64 + c] =
          (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
    }
  }
}

// workgroups of the forward launch = rows of its stat_partial: the weight fragments (32 loads and splits per lane) are
// set up once per workgroup, so a workgroup takes several 512-position blocks.  The kernel's ~140 registers allow three
// workgroups per CU: 768 are all resident at once on the 256 CUs (1024 would leave a quarter for a second round).
extern "C" int uh_conv_first_parts(int B, int H, int W) {
  const long long nblk = (uh_mp(B, H, W) + 511) / 512;
  return (int)(nblk < 768 ? nblk : 768);
}

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
  UH_REQUIRE((long long)Cin * H * W < (1ll << 30), "conv_first_fwd: image of %lld elements exceeds the 32-bit tap offsets",
             (long long)Cin * H * W);
  const dim3 grid((unsigned)uh_conv_first_parts(B, H, W));
  hipStream_t st = (hipStream_t)s;
#define LAUNCH(CI)                                                                                                  \
  if (mode == UH_EPI_RAW_STATS)                                                                                     \
    hipLaunchKernelGGL((conv_first_fwd_kernel<CI, UH_EPI_RAW_STATS>), grid, dim3(256), 0, st, B, H, W, x, w, bias,   \
                       scale, shift, (char*)out, o_cstride, o_coff, stat_partial);                                  \
  else                                                                                                              \
    hipLaunchKernelGGL((conv_first_fwd_kernel<CI, UH_EPI_AFFINE_RELU>), grid, dim3(256), 0, st, B, H, W, x, w, bias, \
                       scale, shift, (char*)out, o_cstride, o_coff, stat_partial)
  switch (Cin) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    default: LAUNCH(4); break;
  }
#undef LAUNCH
  return uh_check_launch("conv_first_fwd");
}

// ---- weight gradient: dW[co][k] = sum_pixels dy[m][co] * im2col(x)[m][k] ------------------------------------
// A wave takes 32 padded-linear positions per step: A fragments are dY^T (row = channel, 8 consecutive pixels per
// lane, bf16 as stored), B fragments the im2col taps of the same pixels split into hi + lo.  dY is zero on pad
// rows, so whatever finite value the gather returns there does not matter (it returns 0).
template <int CIN>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(int B, int H, int W, const float* __restrict__ x,
                                                               const char* __restrict__ dy, int dy_cs, int dy_co,
                                                               float* __restrict__ partial) {
  constexpr int KK = CIN * 9;
  constexpr int NT = ((KK + 31) / 32) * 2;        // 16-tap tiles
  __shared__ float red[64][NT * 16 + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int Wp = W + 1, Hp = H + 1;
  const long long Mp = (long long)B * Hp * Wp;
  const long long nblk = (Mp + 31) / 32;
  f32x4 acc[4][NT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) acc[j][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  FirstTaps<CIN, NT> taps;
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) taps.set(tt, tt * 16 + l15, H, W);

  for (long long pb = (long long)blockIdx.x * 4 + wave; pb < nblk; pb += (long long)gridDim.x * 4) {
    const long long m0 = pb * 32 + g * 8;              // this lane's 8 consecutive positions
    // dY^T fragments
    bf16x8 af[4];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const long long m = m0 + t;
      const bool ok = m < Mp;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        af[j][t] = ok ? *(const bf16*)(dy + (m * dy_cs + dy_co + j * 16 + l15) * 2) : (bf16)0.f;
    }
    // positions of the 8 pixels
    int px = (int)(m0 % Wp);
    long long r = m0 / Wp;
    int py = (int)(r % Hp);
    long long b = r / Hp;
    bf16x8 xhi[NT], xlo[NT];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const bool ok = (m0 + t < Mp) && px < W && py < H;
      const float* xb = ok ? x + ((b * CIN) * H + py) * W + px : x;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const float v = taps.get(xb, H, W, py, px, ok, tt);
        bf16 h, l;
        split_bf16(v, h, l);
        xhi[tt][t] = h;
        xlo[tt][t] = l;
      }
      if (++px == Wp) {
        px = 0;
        if (++py == Hp) { py = 0; ++b; }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        acc[j][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j], xlo[tt], acc[j][tt], 0, 0, 0);
        acc[j][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j], xhi[tt], acc[j][tt], 0, 0, 0);
      }
  }
  // D[j][tt]: rows = channel j*16 + g*4 + e, column = tap tt*16 + l15; the four waves add in turn
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float* d = &red[j * 16 + g * 4 + e][tt * 16 + l15];
            *d = (wv == 0) ? acc[j][tt][e] : *d + acc[j][tt][e];
          }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < 64 * KK; i += 256) partial[(long long)blockIdx.x * 64 * KK + i] = red[i / KK][i % KK];
}

extern "C" int uh_conv_first_wgrad_parts(int B, int H, int W) {
  const long long blocks = (uh_mp(B, H, W) + 127) / 128;          // 4 waves x 32 positions per step
  return (int)(blocks < 1024 ? blocks : 1024);
}

extern "C" int uh_conv_first_wgrad(int B, int H, int W, int Cin, int Cout, const float* x, const void* dy,
                                   int dy_cstride, int dy_coff, float* partial, uh_stream s) {
  UH_REQUIRE(B > 0 && H > 0 && W > 0 && x && dy && partial, "conv_first_wgrad: bad arguments");
  UH_REQUIRE(Cin >= 1 && Cin <= 4, "conv_first_wgrad: Cin=%d must be 1..4", Cin);
  UH_REQUIRE(Cout == 64, "conv_first_wgrad: Cout=%d (only 64 is built)", Cout);
  UH_REQUIRE((long long)Cin * H * W < (1ll << 30), "conv_first_wgrad: image of %lld elements exceeds the 32-bit tap offsets",
             (long long)Cin * H * W);
  const dim3 grid((unsigned)uh_conv_first_wgrad_parts(B, H, W));
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
