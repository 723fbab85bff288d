// Implicit-GEMM convolution on gfx950 matrix cores.
//
//   out[m][n] = sum_t sum_k A[m + off(t)][k] * Wp[n][t][k]        (bf16 operands, fp32 accumulate)
//
// A is a padded-linear NHWC activation (common.h): the nine taps of a 3x3 pad-1 convolution
// are nine ROW OFFSETS of one matrix, so the kernel is a plain tiled GEMM whose A-tile source
// address changes per K-step.  Tiles are staged global -> LDS with global_load_lds (16 B per
// lane, no VGPR round trip), double buffered; the LDS image is XOR-swizzled by permuting the
// per-lane SOURCE chunk so ds_read_b128 fragment reads are bank-conflict free; each of the
// four waves owns a 64x64 accumulator block (4x4 tiles of mfma_f32_16x16x32_bf16).
// The epilogue transposes the accumulators through LDS so global stores are 16 B per lane
// along the channel axis, and fuses: bias, BatchNorm batch statistics (per-tile partial sums),
// folded BatchNorm+ReLU (inference), pad-row zeroing, and the pixel-shuffle scatter of a
// stride-2 transposed convolution.
#include "conv_args.h"


template <int BM, int BN>
struct Cfg {
  static constexpr int WN = BN / 64;       // waves along N
  static constexpr int WM = 4 / WN;        // waves along M
  static_assert(BM == WM * 64, "each wave owns 64 rows");
  static constexpr int A_BYTES = BM * 128;
  static constexpr int B_BYTES = BN * 128;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int PITCH = BN * 2 + 16;               // epilogue tile row pitch (bytes)
  static constexpr int TILE_BYTES = BM * PITCH;
  static constexpr int FLAG_OFF = TILE_BYTES;             // float[BM]
  static constexpr int RED_OFF = FLAG_OFF + BM * 4;       // float[WM][2][BN]
  static constexpr int EPI_BYTES = RED_OFF + WM * 2 * BN * 4;
  static constexpr int LDS_BYTES = 2 * STAGE;
  static_assert(EPI_BYTES <= LDS_BYTES, "epilogue reuses the staging LDS");
};

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(GemmArgs p) {
  using C = Cfg<BM, BN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WN, wn = wave % C::WN;
  const long long m0 = (long long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int l15 = lane & 15, g = lane >> 4;

  // ---- staging addresses: one wave-instruction copies 8 rows x 128 B ----------------------
  // LDS row r keeps source chunk c at position c ^ ((r>>1)&7); for the rows a wave stages
  // (groups wave, wave+4, ...) that swizzle is (4*(wave&1) + lane/16) & 7.
  const int lrow = lane >> 3;
  const int src_chunk = (lane & 7) ^ ((4 * (wave & 1) + g) & 7);
  const long long a_row_bytes = (long long)p.a_cstride * 2;
  const long long b_row_bytes = (long long)p.T * p.K * 2;
  const char* a_lane = p.A + (m0 + 8 * wave + lrow) * a_row_bytes + (long long)p.a_coff * 2 + src_chunk * 16;
  const char* b_lane = p.Wt + (long long)(n0 + 8 * wave + lrow) * b_row_bytes + src_chunk * 16;

  auto stage = [&](int buf, int tap, int kc) {
    int off = 0;
    if (p.T == 9) off = (tap / 3 - 1) * p.Wp + (tap % 3 - 1);
    const char* as = a_lane + (long long)off * a_row_bytes + kc * 128;
    const char* bs = b_lane + ((long long)tap * p.K + kc * 64) * 2;
    char* lds = smem + buf * C::STAGE + wave * 1024;
#pragma unroll
    for (int j = 0; j < BM / 32; ++j)
      __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(as + j * 32 * a_row_bytes),
                                       (UH_LDS void*)(lds + j * 4096), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < BN / 32; ++j)
      __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(bs + j * 32 * b_row_bytes),
                                       (UH_LDS void*)(lds + C::A_BYTES + j * 4096), 16, 0, 0);
  };

  // ---- fragment read offsets --------------------------------------------------------------
  // row = w*64 + i*16 + l15 ; chunk = ks*4 + g ; swizzle term (row>>1)&7 == l15>>1
  const int sw = l15 >> 1;
  const int a_rd0 = (wm * 64 + l15) * 128 + (((0 + g) ^ sw) << 4);
  const int a_rd1 = (wm * 64 + l15) * 128 + (((4 + g) ^ sw) << 4);
  const int b_rd0 = (wn * 64 + l15) * 128 + (((0 + g) ^ sw) << 4);
  const int b_rd1 = (wn * 64 + l15) * 128 + (((4 + g) ^ sw) << 4);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int niter = p.T * p.KC;
  int tap = 0, kc = 0;
  stage(0, 0, 0);
  for (int it = 0; it < niter; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (++kc == p.KC) { kc = 0; ++tap; }
    if (it + 1 < niter) stage((it + 1) & 1, tap, kc);
    const char* As = smem + (it & 1) * C::STAGE;
    const char* Bs = As + C::A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(As + (ks ? a_rd1 : a_rd0) + i * 2048);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(Bs + (ks ? b_rd1 : b_rd0) + j * 2048);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue -----------------------------------------------------------------------------
  __syncthreads();  // every wave is done reading the staging buffers
  float* flag = (float*)(smem + C::FLAG_OFF);
  float* red = (float*)(smem + C::RED_OFF);
  if (tid < BM) {
    const long long m = m0 + tid;
    flag[tid] = (m < p.Mp && p.rowmask[m]) ? 1.f : 0.f;
  }
  __syncthreads();

  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = wn * 64 + j * 16 + l15;
    const int n = n0 + col;
    float bias = 0.f, sc = 1.f, sh = 0.f;
    if (p.mode == UH_EPI_CONVT) {
      bias = p.bias ? p.bias[n % p.Cout] : 0.f;
    } else if (p.mode == UH_EPI_AFFINE_RELU) {
      sc = p.scale[n];
      sh = p.shift[n];
    } else {
      bias = p.bias ? p.bias[n] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rbase = wm * 64 + i * 16 + g * 4;
      const f32x4 f = *(const f32x4*)(flag + rbase);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[i][j][e];
        if (p.mode == UH_EPI_AFFINE_RELU) {
          v = fmaxf(v * sc + sh, 0.f);
        } else {
          v += bias;
        }
        v *= f[e];
        if (p.mode == UH_EPI_RAW_STATS) {
          s1[j] += v;
          s2[j] += v * v;
        }
        if (p.mode == UH_EPI_RAW_STATS)
          *(_Float16*)(smem + (rbase + e) * C::PITCH + col * 2) = f2h(v);   // BatchNorm input: fp16
        else
          *(bf16*)(smem + (rbase + e) * C::PITCH + col * 2) = f2bf(v);
      }
    }
  }
  if (p.mode == UH_EPI_RAW_STATS) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s1[j] += __shfl_xor(s1[j], 16, 64);
      s1[j] += __shfl_xor(s1[j], 32, 64);
      s2[j] += __shfl_xor(s2[j], 16, 64);
      s2[j] += __shfl_xor(s2[j], 32, 64);
      if (g == 0) {
        const int col = wn * 64 + j * 16 + l15;
        red[(wm * 2 + 0) * BN + col] = s1[j];
        red[(wm * 2 + 1) * BN + col] = s2[j];
      }
    }
  }
  __syncthreads();

  if (p.mode == UH_EPI_RAW_STATS && tid < 2 * BN) {
    const int which = tid / BN, col = tid % BN;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < C::WM; ++w) v += red[(w * 2 + which) * BN + col];
    p.stat_partial[((long long)blockIdx.x * 2 + which) * p.N + n0 + col] = v;
  }

  constexpr int CPR = BN / 8;  // 16-byte chunks per tile row
  if (p.mode != UH_EPI_CONVT) {
#pragma unroll
    for (int q = tid; q < BM * CPR; q += 256) {
      const int row = q / CPR, cc = q % CPR;
      const long long m = m0 + row;
      if (m < p.Mp) {
        const U4 v = *(const U4*)(smem + row * C::PITCH + cc * 16);
        *(U4*)(p.out + (m * p.o_cstride + p.o_coff + n0 + cc * 8) * 2) = v;
      }
    }
  } else {
    // transposed conv k2 s2: column n = tap*Cout + co of input position (b,y,x) goes to
    // output position (b, 2y + tap/2, 2x + tap%2), channel co; output geometry (B, 2H, 2W).
    const int Hp = p.H + 1, Wp = p.W + 1;
    const int oHp = 2 * p.H + 1, oWp = 2 * p.W + 1;
    for (int q = tid; q < BM * CPR; q += 256) {
      const int row = q / CPR, cc = q % CPR;
      const long long m = m0 + row;
      if (m >= p.Mp) continue;
      const int n = n0 + cc * 8;
      const int t = n / p.Cout, co = n % p.Cout;
      const int x = (int)(m % Wp);
      const long long r = m / Wp;
      const int y = (int)(r % Hp);
      const long long b = r / Hp;
      const int Y = 2 * y + (t >> 1), X = 2 * x + (t & 1);
      if (t < 4 && Y < oHp && X < oWp) {        // t >= 4: zero-padded weight rows (N > 4*Cout)
        const U4 v = *(const U4*)(smem + row * C::PITCH + cc * 16);
        const long long om = (b * oHp + Y) * oWp + X;
        *(U4*)(p.out + (om * p.o_cstride + p.o_coff + co) * 2) = v;
      }
    }
  }
}

static bool use_wide_m(int N) { return N % 128 != 0; }  // N = 64, 192, ...: tiles 64 wide

// M tiles of the T = 9 path (rows of stat_partial in RAW_STATS mode)
extern "C" int uh_conv_gemm_mtiles(int B, int H, int W, int N) {
  (void)N;
  return (int)((uh_mp(B, H, W) + UH_BAND_BM - 1) / UH_BAND_BM);
}

extern "C" size_t uh_conv_gemm_workspace(int B, int H, int W, int K, int N, int T) {
  if (T != 9 || B <= 0 || H <= 0 || W <= 0 || K < 64 || N < 64) return 0;
  return uh_band_workspace_bytes(uh_mp(B, H, W), N, K);
}

extern "C" int uh_conv_gemm(int B, int H, int W, const void* a, int a_cstride, int a_coff, int K,
                            const void* w_packed, int N, int T, int mode, void* out, int o_cstride,
                            int o_coff, const float* bias, const float* scale, const float* shift,
                            float* stat_partial, const uint8_t* rowmask, uh_stream s) {
  return uh_conv_gemm_ws(B, H, W, a, a_cstride, a_coff, K, w_packed, N, T, mode, out, o_cstride, o_coff, bias, scale,
                         shift, stat_partial, rowmask, nullptr, 0, s);
}

extern "C" int uh_conv_gemm_ws(int B, int H, int W, const void* a, int a_cstride, int a_coff, int K,
                               const void* w_packed, int N, int T, int mode, void* out, int o_cstride,
                               int o_coff, const float* bias, const float* scale, const float* shift,
                               float* stat_partial, const uint8_t* rowmask, void* workspace,
                               size_t workspace_bytes, uh_stream s) {
  UH_REQUIRE(B > 0 && H > 0 && W > 0, "conv_gemm: bad geometry B=%d H=%d W=%d", B, H, W);
  UH_REQUIRE(K > 0 && K % 64 == 0, "conv_gemm: K=%d must be a positive multiple of 64", K);
  UH_REQUIRE(N > 0 && N % 64 == 0, "conv_gemm: N=%d must be a positive multiple of 64", N);
  UH_REQUIRE(T == 1 || T == 9, "conv_gemm: T=%d must be 1 or 9", T);
  UH_REQUIRE(mode >= UH_EPI_RAW_STATS && mode <= UH_EPI_CONVT, "conv_gemm: bad mode %d", mode);
  UH_REQUIRE(a && w_packed && out && rowmask, "conv_gemm: null pointer");
  UH_REQUIRE(a_cstride >= a_coff + K && a_cstride % 8 == 0 && a_coff % 8 == 0,
             "conv_gemm: bad A view (cstride=%d coff=%d K=%d)", a_cstride, a_coff, K);
  UH_REQUIRE(o_cstride % 8 == 0 && o_coff % 8 == 0, "conv_gemm: bad out view");
  if (mode == UH_EPI_CONVT) {
    UH_REQUIRE(o_cstride - o_coff >= 8, "conv_gemm: bad transposed-conv view");
    UH_REQUIRE(T == 1, "conv_gemm: transposed conv is a T=1 GEMM");
  } else {
    UH_REQUIRE(o_cstride >= o_coff + N, "conv_gemm: out view too narrow");
  }
  UH_REQUIRE(mode != UH_EPI_RAW_STATS || stat_partial, "conv_gemm: RAW_STATS needs stat_partial");
  UH_REQUIRE(mode != UH_EPI_AFFINE_RELU || (scale && shift), "conv_gemm: AFFINE_RELU needs scale/shift");
  GemmArgs p;
  p.A = (const char*)a;
  p.Wt = (const char*)w_packed;
  p.out = (char*)out;
  p.bias = bias;
  p.scale = scale;
  p.shift = shift;
  p.stat_partial = stat_partial;
  p.rowmask = rowmask;
  p.Mp = uh_mp(B, H, W);
  p.a_cstride = a_cstride;
  p.a_coff = a_coff;
  p.K = K;
  p.N = N;
  p.T = T;
  p.KC = K / 64;
  p.Wp = W + 1;
  p.o_cstride = o_cstride;
  p.o_coff = o_coff;
  p.mode = mode;
  p.H = H;
  p.W = W;
  // transposed conv: Cout = N/4, or the width of the output window when that is narrower (N zero-padded to 64)
  p.Cout = (mode == UH_EPI_CONVT) ? ((N / 4 < o_cstride - o_coff) ? N / 4 : o_cstride - o_coff) : N;
  hipStream_t st = (hipStream_t)s;
  if (T == 9) {
    UH_REQUIRE(((uintptr_t)workspace & 15) == 0, "conv_gemm: workspace must be 16-byte aligned");
    uh_launch_band(p, workspace, workspace_bytes, st);
  } else if (use_wide_m(N)) {
    using C = Cfg<256, 64>;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)conv_gemm_kernel<256, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
      attr = true;
    }
    dim3 grid((unsigned)((p.Mp + 255) / 256), N / 64);
    hipLaunchKernelGGL((conv_gemm_kernel<256, 64>), grid, dim3(256), C::LDS_BYTES, st, p);
  } else {
    using C = Cfg<128, 128>;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)conv_gemm_kernel<128, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
      attr = true;
    }
    dim3 grid((unsigned)((p.Mp + 127) / 128), N / 128);
    hipLaunchKernelGGL((conv_gemm_kernel<128, 128>), grid, dim3(256), C::LDS_BYTES, st, p);
  }
  return uh_check_launch("conv_gemm");
}
