// EXPERIMENT, not part of libunet_hip.so.  A second structure for the 3x3 convolution kernel, built to test
// whether conv_band_kernel (unet_amd/csrc/conv_gemm.hip) is held back by its two barriers per 16 MFMAs and by LDS
// read traffic.  Measured on MI355X against conv_band_kernel in the same process (B = 32 U-Net layer shapes,
// random data): 256x128 / 512x64 tiles with 128x64 wave tiles: within +-3 % of it on every layer (e.g. conv3_2
// 1015 vs 992 TFLOP/s, conv_up2_1 1064 vs 1083); 256x256 tiles with 128x128 wave tiles: 20-30 % slower; the same
// loop on 32x32x16 MFMAs (timing only): 3-6 % slower; without the wait for the staging DMAs (timing only): equal.
// Builds against the product's headers: hipcc -I unet_amd/csrc -c scratch/conv_wide_experiment.hip, then add
// wide_bm / wide_workspace_bytes / launch_wide to the dispatch in conv_gemm.hip.
//
// 3x3 convolution as an implicit GEMM, four waves per workgroup with 128-row wave tiles.
//
// Same operands, band staging, LDS swizzle, persistent unit walk and epilogue fusions as conv_band_kernel
// (conv_gemm.hip); what differs is the shape of the inner loop.  There a wave owns 64x64 outputs, so every
// MFMA needs 512 B of fragment reads and the LDS read path is as busy as the matrix cores; eight waves take turns
// between a load part and an MFMA part behind two barriers per 16 MFMAs.  Here
//   * one wave per SIMD owns 128 x 128 (or 128 x 64) outputs: 256 B (384 B) of fragment reads per MFMA, the
//     accumulators live in the AGPR half of the 512-register file;
//   * the wave pipelines itself: while the 64 MFMAs of one filter tap run it reads the fragments of the next tap
//     into a second register set, and in the last tap of a stage it issues the staging DMAs of the stage after
//     next -- no wave waits for another one except at ONE barrier per stage (192 MFMAs), placed where every wave
//     has finished reading the stage's buffer and has seen its own DMAs of the next stage land;
//   * the LDS ring has two stages of (A band + three weight taps).
#include "conv_gemm.h"

int wide_bm(int N);
size_t wide_workspace_bytes(long long Mp, int N, int K);
bool launch_wide(const GemmArgs& p, void* ws, size_t ws_bytes, hipStream_t st);

template <int WM_, int WN_, int TN_>
struct WideCfg {
  static constexpr int WM = WM_, WN = WN_, TN = TN_;
  static_assert(WM * WN == 4, "four waves");
  static constexpr int BM = WM * 128;
  static constexpr int WCOLS = 16 * TN;           // columns per wave
  static constexpr int BN = WN * WCOLS;
  static constexpr int AG = BM / 16 + 1;          // 1 KiB groups (16 rows x 64 B) of the A band
  static constexpr int BGT = BN / 16;             // groups per weight tap
  static constexpr int SG = AG + 3 * BGT;         // groups per stage
  static constexpr int NQ = (SG + 3) / 4;         // staging slots per wave
  static constexpr int STAGE = SG * 1024;
  static constexpr int FLAG_OFF = 2 * STAGE;                  // 4 waves x 64 dwords of row flags
  static constexpr int RED_OFF = FLAG_OFF + 4 * 256;          // float[WM][2][BN]
  static constexpr int LDS_BYTES = RED_OFF + WM * 2 * BN * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// Epilogue of one unit: lane (l15, g) of wave (wm, wn) holds rows wm*128 + i*16 + g*4 + e and the TN adjacent
// channels wn*16*TN + l15*TN + j of the tile; fw[i] = the four row flags of rows i*16 + g*4 .. +3.
template <class C, int MODE>
__device__ __forceinline__ void wide_epilogue(const GemmArgs& p, const f32x4 (&acc)[8][C::TN], int mt, int nt,
                                              const uint32_t (&fw)[8], int wm, int wn, int l15, int g, float* red) {
  constexpr int TN = C::TN, BN = C::BN, BM = C::BM;
  const int ncol = wn * C::WCOLS + l15 * TN;
  const int n0 = nt * BN;
  float cb[TN], csc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { cb[j] = 0.f; csc[j] = 1.f; }
  if (MODE == UH_EPI_AFFINE_RELU) {
#pragma unroll
    for (int q = 0; q < TN / 4; ++q) {
      const f32x4 a4 = *(const f32x4*)(p.scale + n0 + ncol + 4 * q), b4 = *(const f32x4*)(p.shift + n0 + ncol + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) { csc[4 * q + j] = a4[j]; cb[4 * q + j] = b4[j]; }
    }
  } else if (p.bias) {
#pragma unroll
    for (int q = 0; q < TN / 4; ++q) {
      const f32x4 b4 = *(const f32x4*)(p.bias + n0 + ncol + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) cb[4 * q + j] = b4[j];
    }
  }
  const long long mrow0 = (long long)mt * BM + wm * 128 + g * 4;
  char* orow = p.out + ((mrow0 * p.o_cstride) + p.o_coff + n0 + ncol) * 2;
  const long long o_row_bytes = (long long)p.o_cstride * 2;
  float s1[TN], s2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long m = mrow0 + i * 16 + e;
      const float f = (m < p.Mp) ? (float)((fw[i] >> (8 * e)) & 0xffu) : 0.f;
      float v[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float x = acc[i][j][e];
        if (MODE == UH_EPI_AFFINE_RELU) x = fmaxf(x * csc[j] + cb[j], 0.f);
        else x += cb[j];
        x *= f;
        if (MODE == UH_EPI_RAW_STATS) {
          s1[j] += x;
          s2[j] += x * x;
        }
        v[j] = x;
      }
      uint32_t o[TN / 2];
#pragma unroll
      for (int j = 0; j < TN / 2; ++j)
        o[j] = (MODE == UH_EPI_RAW_STATS) ? pack2h(v[2 * j], v[2 * j + 1]) : pack2(v[2 * j], v[2 * j + 1]);
      char* dst = orow + (i * 16 + e) * o_row_bytes;
      if (TN == 8) {
        U4 u;
        u.x = o[0]; u.y = o[1]; u.z = o[2]; u.w = o[TN / 2 - 1];
        *(U4*)dst = u;
      } else {
        *(uint2*)dst = uint2{o[0], o[1]};
      }
    }
  }
  if (MODE == UH_EPI_RAW_STATS) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      s1[j] += __shfl_xor(s1[j], 16, 64);
      s1[j] += __shfl_xor(s1[j], 32, 64);
      s2[j] += __shfl_xor(s2[j], 16, 64);
      s2[j] += __shfl_xor(s2[j], 32, 64);
    }
    if (g == 0) {
#pragma unroll
      for (int q = 0; q < TN / 4; ++q) {
        *(f32x4*)(red + (wm * 2 + 0) * BN + ncol + 4 * q) = f32x4{s1[4 * q], s1[4 * q + 1], s1[4 * q + 2], s1[4 * q + 3]};
        *(f32x4*)(red + (wm * 2 + 1) * BN + ncol + 4 * q) = f32x4{s2[4 * q], s2[4 * q + 1], s2[4 * q + 2], s2[4 * q + 3]};
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // raw barriers do not wait for LDS writes
  }
}

// red[WM][2][BN] -> stat_partial[mt][2][N]; every wave's red[] writes must be a barrier old
template <class C>
__device__ __forceinline__ void wide_flush_stats(const GemmArgs& p, const float* red, int mt, int nt, int tid) {
  for (int t = tid; t < 2 * C::BN; t += 256) {
    const int which = t / C::BN, col = t % C::BN;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < C::WM; ++w) v += red[(w * 2 + which) * C::BN + col];
    p.stat_partial[((long long)mt * 2 + which) * p.N + nt * C::BN + col] = v;
  }
}

template <class C, int MODE>
__global__ __launch_bounds__(256) void conv_wide_kernel(GemmArgs p, BandSched sc) {
  constexpr int BM = C::BM, BN = C::BN, TN = C::TN, NQ = C::NQ;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int pos = (blockIdx.x & 7) * (sc.G >> 3) + (blockIdx.x >> 3);
  const int niter = sc.niter, ntn = sc.ntn;
  int u_first, u_step, it_first, total;
  if (sc.aligned) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int u_end = min((xcd + 1) * sc.upx, sc.ntm * ntn);
    u_first = xcd * sc.upx + j;
    u_step = sc.G >> 3;
    it_first = 0;
    total = (u_first < u_end) ? ((u_end - u_first + u_step - 1) / u_step) * niter : 0;
  } else {
    const long long i0 = band_start(sc, pos), i1 = band_start(sc, pos + 1);
    u_first = (int)(i0 / niter);
    u_step = 1;
    it_first = (int)(i0 - (long long)u_first * niter);
    total = (int)(i1 - i0);
  }
  if (total <= 0) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int l15 = lane & 15, g = lane >> 4;
  const int KC = p.K / 32;

  // ---- staging: one wave instruction = one group of 16 rows x 64 B; wave w stages groups w, w+4, ... ----------
  const int srow = lane >> 2;
  const int src_chunk = (lane & 3) ^ ((srow >> 1) & 2);
  const long long a_row_bytes = (long long)p.a_cstride * 2;
  const long long b_row_bytes = 9LL * p.K * 2;
  // address of a staged 16-byte piece = wave-uniform (operand base + stage offset + slot offset) + a per-lane
  // 32-bit offset that is the same for every activation slot and for every weight slot
  const unsigned a_lane = (unsigned)(srow * a_row_bytes) + src_chunk * 16;
  const unsigned b_lane = (unsigned)(srow * TN * b_row_bytes) + src_chunk * 16;
  int slot_off[NQ];
  bool slot_is_a[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int gi = min(wave + 4 * q, C::SG - 1);   // past the end: the last group again (same bytes, same place)
    slot_is_a[q] = gi < C::AG;
    const int j = gi - C::AG;
    const int kx = j / C::BGT, nb = j % C::BGT;
    // weight group nb = LDS rows 16*nb .. +15 of the tap = accumulator tile nb % TN of wave column nb / TN;
    // its row c holds channel c*TN + (nb % TN) of that wave column
    const long long oa = ((long long)gi * 16 - 1 - p.Wp) * a_row_bytes + (long long)p.a_coff * 2;
    const long long ob = (long long)((nb / TN) * C::WCOLS + (nb % TN)) * b_row_bytes + (long long)kx * p.K * 2;
    slot_off[q] = __builtin_amdgcn_readfirstlane((int)(slot_is_a[q] ? oa : ob));
  }
  long long offA, offB;
  // slots [q0, q1) of the stage at (offA, offB) into buffer `buf`
  auto issue_slots = [&](int buf, int q0, int q1) {
    const char* ua = p.A + offA;
    const char* ub = p.Wt + offB;
#pragma unroll
    for (int q = q0; q < q1; ++q) {
      const char* base = (slot_is_a[q] ? ua : ub) + slot_off[q];
      const unsigned vo = slot_is_a[q] ? a_lane : b_lane;
      __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(base + vo),
                                       (UH_LDS void*)(smem + buf * C::STAGE + min(wave + 4 * q, C::SG - 1) * 1024), 16, 0, 0);
    }
  };
  // row flags of this wave's 128 rows: 32 dwords of the row mask (lanes 32.. repeat them)
  const long long mask_last_dword = (p.Mp - 1) >> 2;
  auto issue_flags = [&](int mt) {
    long long d = (((long long)mt * BM + wm * 128) >> 2) + (lane & 31);
    if (d > mask_last_dword) d = mask_last_dword;
    __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(p.rowmask + 4 * d),
                                     (UH_LDS void*)(smem + C::FLAG_OFF + wave * 256), 4, 0, 0);
  };

  // ---- fragment read offsets (bytes inside a stage) -----------------------------------------------------
  int a_off[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int r = l15 + kx;
    a_off[kx] = (wm * 128 + r) * 64 + ((g ^ ((r >> 1) & 2)) << 4);
  }
  const int b_off = C::AG * 1024 + (wn * C::WCOLS + l15) * 64 + ((g ^ ((l15 >> 1) & 2)) << 4);
  float* red = (float*)(smem + C::RED_OFF);

  // ---- the stream of stages -------------------------------------------------------------------------------
  int u_n = u_first;
  int mt_n = u_n / ntn, nt_n = u_n % ntn;
  int ky_n = it_first / (KC / 2), kc_n = it_first % (KC / 2);      // it_first counts pairs of stages
  auto stage_offsets = [&]() {
    offA = ((long long)mt_n * BM + (long long)ky_n * p.Wp) * a_row_bytes + kc_n * 64;
    offB = (long long)nt_n * BN * b_row_bytes + ((long long)ky_n * 3 * p.K + kc_n * 32) * 2;
  };
  // next stage of the stream: 64 bytes further along both operands; a new filter row and a new unit are rare
  const int step_mt = u_step / ntn, step_nt = u_step % ntn;
  const long long dA_ky = (long long)p.Wp * a_row_bytes - (long long)KC * 64;
  const long long dB_ky = 3LL * p.K * 2 - (long long)KC * 64;
  auto advance = [&]() {
    ++kc_n;
    offA += 64;
    offB += 64;
    if (kc_n == KC) {
      kc_n = 0;
      ++ky_n;
      offA += dA_ky;
      offB += dB_ky;
      if (ky_n == 3) {
        ky_n = 0;
        nt_n += step_nt;
        mt_n += step_mt;
        if (nt_n >= ntn) { nt_n -= ntn; ++mt_n; }
        stage_offsets();
      }
    }
  };

  bf16x8 fa[2][8], fb[2][TN];
  // read r of the 8 + TN fragment reads of tap kx, in the order the MFMAs of that tap need them: B tiles, then A tiles
  auto read_one = [&](const char* S, int kx, int r, bf16x8 (&a)[8], bf16x8 (&b)[TN]) {
    if (r < TN) b[r] = *(const bf16x8*)(S + b_off + kx * (BN * 64) + r * 1024);
    else a[r - TN] = *(const bf16x8*)(S + a_off[kx] + (r - TN) * 1024);
  };
  auto read_frags = [&](const char* S, int kx, bf16x8 (&a)[8], bf16x8 (&b)[TN]) {
#pragma unroll
    for (int r = 0; r < 8 + TN; ++r) read_one(S, kx, r, a, b);
  };

  // A loop iteration is a PAIR of stages (64 channels of one filter row): six taps that alternate between the
  // two fragment register sets, so every set and both LDS buffers have fixed roles in the loop body.  sc.niter,
  // it_first and total count pairs.
  // prologue: stages 0 and 1 in flight, stage 0 landed and visible, its first fragments in registers
  kc_n *= 2;
  const long long total_st = 2LL * total;       // stages of this workgroup
  long long st_issued = 2;                      // stages whose DMAs have been issued
  stage_offsets();
  issue_slots(0, 0, NQ);
  advance();
  issue_slots(1, 0, NQ);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  const char* S0 = smem;
  const char* S1 = smem + C::STAGE;
  read_frags(S0, 0, fa[0], fb[0]);

  int s = 0;
  int u = u_first, it_a = it_first;
  while (s < total) {
    const int cnt = min(niter - it_a, total - s);
    const bool full = (it_a == 0 && cnt == niter);
    const int mt = u / ntn, nt = u - mt * ntn;
    f32x4 acc[8][TN];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // A tap = 8*TN MFMAs from the current fragment set.  With one wave per SIMD nothing else fills the matrix
    // pipe, so everything else has to be issued in the shadow of an MFMA (16 cycles each): the tap is written as
    // 8*TN slots, "one MFMA + at most a few other instructions", with a scheduling barrier after each slot (left
    // alone the compiler puts the other work in one piece between the MFMAs).  Side work: the 8 + TN fragment reads
    // of the NEXT tap into the other register set (even slots); in the third tap of a stage the staging DMAs of the
    // stage after next (odd slots); in the second tap the bookkeeping that finds that stage.
    // The MFMA is written as asm so that the accumulator is updated IN PLACE: with the builtin the register
    // allocator lets most MFMAs write another AGPR tuple than they read and permutes ~40 tuples back at the loop
    // head.
#ifdef UH_EXPERIMENT_MFMA32
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    f32x16 acc32[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc32[i][e] = 0.f;
#endif
    constexpr int TAP_ADVANCE = 1, TAP_STAGING = 2;   // (0: a plain tap)
    auto tap = [&](auto KIND_, const char* Snext, int kx_next, int buf, bf16x8 (&ca)[8], bf16x8 (&cb_)[TN],
                   bf16x8 (&na)[8], bf16x8 (&nb)[TN]) {
      constexpr int KIND = decltype(KIND_)::value;
#ifdef UH_EXPERIMENT_MFMA32
      // timing experiment only (results are wrong): the same fragment registers fed to 32x32x16 MFMAs
#pragma unroll
      for (int m2 = 0; m2 < 4 * TN; ++m2) {
        asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc32[m2 % 8]) : "v"(ca[m2 % 8]), "v"(cb_[m2 % TN]));
        __builtin_amdgcn_sched_barrier(0);
        { const int m = m2; if (m < 8 + TN) read_one(Snext, kx_next, m, na, nb); }
        if (KIND == TAP_STAGING && m2 < NQ) issue_slots(buf, m2, m2 + 1);
        if (KIND == TAP_ADVANCE && m2 == 4 * TN - 3) { if (st_issued < total_st) advance(); ++st_issued; }
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
#endif
#pragma unroll
      for (int m = 0; m < 8 * TN; ++m) {
        asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m / TN][m % TN]) : "v"(ca[m / TN]), "v"(cb_[m % TN]));
        __builtin_amdgcn_sched_barrier(0);          // the MFMA first, its side work in its shadow
        if (m % 2 == 0 && m / 2 < 8 + TN) read_one(Snext, kx_next, m / 2, na, nb);
        if (KIND == TAP_STAGING && m % 2 == 1 && m / 2 < NQ) issue_slots(buf, m / 2, m / 2 + 1);
        if (KIND == TAP_ADVANCE && m == 8 * TN - 6) {
          // past the end of the stream the last stage is staged again (unused)
          if (st_issued < total_st) advance();
          ++st_issued;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>;
    if (full) issue_flags(mt);                  // lands before the first barrier of the unit
#pragma nounroll
    for (int c = 0; c < cnt; ++c, ++s) {
      __builtin_amdgcn_sched_barrier(0);
      // ---- first stage of the pair (buffer 0): taps from sets 0, 1, 0
      tap(T0{}, S0, 1, 0, fa[0], fb[0], fa[1], fb[1]);
      tap(T1{}, S0, 2, 0, fa[1], fb[1], fa[0], fb[0]);
      // every wave: my reads of buffer 0 are done, my DMAs of the next stage (and the flags) have landed
#ifdef UH_EXPERIMENT_NOWAIT
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
      __builtin_amdgcn_sched_barrier(0);
      tap(T2{}, S1, 0, 0, fa[0], fb[0], fa[1], fb[1]);
      // ---- second stage (buffer 1): taps from sets 1, 0, 1
      tap(T0{}, S1, 1, 0, fa[1], fb[1], fa[0], fb[0]);
      tap(T1{}, S1, 2, 0, fa[0], fb[0], fa[1], fb[1]);
#ifdef UH_EXPERIMENT_NOWAIT
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
      __builtin_amdgcn_sched_barrier(0);
      tap(T2{}, S0, 0, 1, fa[1], fb[1], fa[0], fb[0]);
    }

#ifdef UH_EXPERIMENT_MFMA32
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][e % TN][e] += acc32[i][e] + acc32[i][e + 4] + acc32[i][e + 8] + acc32[i][e + 12];
#endif
    if (full) {
      uint32_t fw[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) fw[i] = *(const uint32_t*)(smem + C::FLAG_OFF + wave * 256 + (i * 4 + g) * 4);
      wide_epilogue<C, MODE>(p, acc, mt, nt, fw, wm, wn, l15, g, red);
      if (MODE == UH_EPI_RAW_STATS) {
        // red[] is next written a whole unit (many barriers) later
        asm volatile("s_barrier" ::: "memory");
        wide_flush_stats<C>(p, red, mt, nt, tid);
      }
    } else {
      // a piece of a unit: park the accumulators for conv_wide_fixup_kernel
      f32x4* slot = (f32x4*)sc.ws + ((long long)(2 * pos + (it_a > 0 ? 0 : 1)) * (8 * TN)) * 256 + tid;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) slot[(i * TN + j) * 256] = acc[i][j];
    }
    u += u_step;
    it_a = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the unused stages staged past the end
}

// One workgroup per cut: boundary k (between positions k-1 and k) that falls inside a unit and is the first
// such boundary of that unit sums the unit's pieces in position order and finishes the unit.
template <class C, int MODE>
__global__ __launch_bounds__(256) void conv_wide_fixup_kernel(GemmArgs p, BandSched sc) {
  constexpr int TN = C::TN;
  __shared__ __attribute__((aligned(16))) float red[C::WM * 2 * C::BN];
  const int k = blockIdx.x + 1;
  const long long sk = band_start(sc, k);
  const int niter = sc.niter;
  if (sk % niter == 0) return;
  const long long u = sk / niter, ub = u * niter, ue = ub + niter;
  if (k >= 2 && band_start(sc, k - 1) > ub) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int l15 = lane & 15, g = lane >> 4;
  f32x4 acc[8][TN];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int pp = k - 1; pp < sc.G; ++pp) {
    const long long ps = band_start(sc, pp), pe = band_start(sc, pp + 1);
    if (ps >= ue) break;
    if (pe <= ps || pe <= ub) continue;
    const int inside = (ps > ub) ? 0 : 1;        // slot 2pp: piece starts inside the unit; 2pp+1: it ends inside
    const f32x4* slot = (const f32x4*)sc.ws + ((long long)(2 * pp + inside) * (8 * TN)) * 256 + tid;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] += slot[(i * TN + j) * 256];
  }
  const int mt = (int)(u / sc.ntn), nt = (int)(u % sc.ntn);
  uint32_t fw[8];
  const long long mask_last_dword = (p.Mp - 1) >> 2;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    long long d = (((long long)mt * C::BM + wm * 128) >> 2) + i * 4 + g;
    if (d > mask_last_dword) d = mask_last_dword;
    fw[i] = *(const uint32_t*)(p.rowmask + 4 * d);
  }
  wide_epilogue<C, MODE>(p, acc, mt, nt, fw, wm, wn, l15, g, red);
  if (MODE == UH_EPI_RAW_STATS) {
    __syncthreads();
    wide_flush_stats<C>(p, red, mt, nt, tid);
  }
}

template <class C>
static size_t wide_ws_bytes(long long Mp, int N, int K) {
  const long long units = ((Mp + C::BM - 1) / C::BM) * (N / C::BN);
  return (size_t)2 * band_workgroups(units, 3 * (K / 32)) * C::BM * C::BN * sizeof(float);
}

template <class C, int MODE>
static void launch_wide_mode(const GemmArgs& p, const BandSched& sc, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_wide_kernel<C, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    attr = true;
  }
  hipLaunchKernelGGL((conv_wide_kernel<C, MODE>), dim3(sc.G), dim3(256), C::LDS_BYTES, st, p, sc);
  if (!sc.aligned)
    hipLaunchKernelGGL((conv_wide_fixup_kernel<C, MODE>), dim3(sc.G - 1), dim3(256), 0, st, p, sc);
}

template <class C>
static void launch_wide_cfg(const GemmArgs& p, void* ws, size_t ws_bytes, hipStream_t st) {
  BandSched sc;
  sc.ntm = (int)((p.Mp + C::BM - 1) / C::BM);
  sc.ntn = p.N / C::BN;
  sc.niter = 3 * (p.K / 64);                      // pairs of stages
  const long long units = (long long)sc.ntm * sc.ntn;
  sc.G = band_workgroups(units, 2 * sc.niter);
  sc.ws = (float*)ws;
  sc.upx = (int)((units + 7) / 8);
  // whole units unless more than a fifth of the CU-time of the last round would idle (and a workspace is there)
  const long long rounds = (units + sc.G - 1) / sc.G;
  const bool wasteful = rounds * sc.G * 100 > units * 125;
  sc.aligned = (ws == nullptr || ws_bytes < wide_ws_bytes<C>(p.Mp, p.N, p.K) || !wasteful) ? 1 : 0;
  if (p.mode == UH_EPI_RAW_STATS) launch_wide_mode<C, UH_EPI_RAW_STATS>(p, sc, st);
  else if (p.mode == UH_EPI_AFFINE_RELU) launch_wide_mode<C, UH_EPI_AFFINE_RELU>(p, sc, st);
  else launch_wide_mode<C, UH_EPI_MASKED>(p, sc, st);
}

using WideX = WideCfg<2, 2, 8>;    // 256 x 256 tiles, N a multiple of 256
using WideA = WideCfg<2, 2, 4>;    // 256 x 128 tiles, N a multiple of 128
using WideB = WideCfg<4, 1, 4>;    // 512 x 64 tiles, other N

#ifndef UH_WIDE_X
#define UH_WIDE_X 0      // 256 x 256 tiles measured 20-30 % slower than 256 x 128 (epilogue spills, bigger cut pieces)
#endif
static bool use_x(int N) { return UH_WIDE_X && N % 256 == 0; }

int wide_bm(int N) { return use_x(N) ? WideX::BM : (N % 128 == 0) ? WideA::BM : WideB::BM; }

size_t wide_workspace_bytes(long long Mp, int N, int K) {
  return use_x(N) ? wide_ws_bytes<WideX>(Mp, N, K)
                  : (N % 128 == 0) ? wide_ws_bytes<WideA>(Mp, N, K) : wide_ws_bytes<WideB>(Mp, N, K);
}

bool launch_wide(const GemmArgs& p, void* ws, size_t ws_bytes, hipStream_t st) {
  if (p.T != 9) return false;
  if (use_x(p.N)) launch_wide_cfg<WideX>(p, ws, ws_bytes, st);
  else if (p.N % 128 == 0) launch_wide_cfg<WideA>(p, ws, ws_bytes, st);
  else launch_wide_cfg<WideB>(p, ws, ws_bytes, st);
  return true;
}
