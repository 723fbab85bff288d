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
#include <cstdlib>
#include "common.h"

#include "conv_gemm.h"

// ===================================================================================================
// 3x3 "band" kernel: the workhorse for T = 9.
//
// Measured on MI355X the tile kernel above is bound by the global->LDS path (staging alone takes 80-97 %
// of its time), and its per-tile prologue and LDS-transposing epilogue are exposed.  (Also measured, on this
// kernel: with all loads removed its barrier/MFMA skeleton is only ~1.3x faster, with the MFMAs removed the
// loads alone take ~70 % of the full time; 128-byte row pieces and 128x64 wave tiles with 512-row tiles each
// left the speed unchanged within 5 %.)  This kernel
//   * stages fewer bytes per FLOP: the three taps of one filter row (kx = 0,1,2) read the SAME activation
//     rows shifted by one, so one staged band of BM+2 rows serves three taps; tiles are 256x128 or 512x64
//     with eight waves of 64x64;
//   * consumes K 32 channels at a time through a three-stage LDS ring (A band + 3 weight taps per stage)
//     behind counted s_waitcnt and raw s_barriers, with the two waves of every SIMD alternating between a
//     load part and an MFMA part (see the main loop);
//   * is persistent: one workgroup per CU walks a list of M tiles and the ring runs on across tile
//     boundaries, so the next tile's first stages are already in LDS while this tile's results are stored;
//   * stores straight from the accumulators: the weight rows are staged in permuted order (LDS row
//     16*j + c holds output channel 4*c + j of the wave's 64), so the four accumulator tiles a lane holds
//     for one activation row are four ADJACENT channels -> one 8-byte store per row, 16 lanes = one
//     128-byte line.  BatchNorm partial sums are reduced across waves through a small LDS array that is
//     flushed to global memory a few barriers later, from inside the next tile.
// LDS rows are 64 B; chunk c of row r sits at position c ^ ((r>>1)&2), which keeps ds_read_b128 fragment
// reads conflict-free for every row shift (checked exhaustively against the gfx950 lane groups).
// Workgroups b, b+8, ... share an XCD; each XCD walks one contiguous range of M tiles so that the band
// rows neighbouring tiles share, and the A rows that all N tiles share, hit in its L2.
template <int BM, int BN>
struct BandCfg {
  static constexpr int WN = BN / 64;
  static constexpr int WM = 8 / WN;
  static_assert(BM == WM * 64, "each of the eight waves owns 64 rows");
  static constexpr int AG = BM / 16 + 1;          // 1 KiB groups (16 rows x 64 B) of the A band: BM+16 >= BM+2 rows
  static constexpr int BGT = BN / 16;             // groups per weight tap
  static constexpr int SG = AG + 3 * BGT;         // groups per stage
  static constexpr int STAGE = SG * 1024;
  static constexpr int NSTAGE = 3;
  static constexpr int FLAG_OFF = NSTAGE * STAGE;             // 8 waves x 64 dwords of row flags
  static constexpr int RED_OFF = FLAG_OFF + 8 * 256;          // float[WM][2][BN]
  static constexpr int COEF_OFF = RED_OFF + WM * 2 * BN * 4;  // 8 waves x float[4][64]: per-channel epilogue constants
  static constexpr int LDS_BYTES = COEF_OFF + 8 * 1024;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// Epilogue of one unit, straight from the accumulators: lane (l15, g) of wave (wm, wn) holds rows
// wm*64 + i*16 + g*4 + e and the 4 adjacent channels wn*64 + l15*4 + j of the tile; fw[i] = the four row
// flags of rows i*16 + g*4 .. +3.  RAW_STATS leaves the wave's channel sums in red[wm][2][BN].
// UH_EPI_MASKED_BNBWD: the consuming layer's fp16 conv output at the lane's rows i_lo*16 .. i_hi*16 (x 4 channels) of
// the tile, for the epilogue's second pass.  Issued at the top of the epilogue: its stores, which are younger, stay in
// flight across the wait for these loads (the memory counter retires in order), and they cover the loads' latency.
// (Issued during the tile's last stage instead, the compiler -- which does not see the main loop's hand-written
// waits -- protects the registers with vmcnt(0) waits of its own all over the loop.)
template <int BM, int BN>
__device__ __forceinline__ void band_load_y(const GemmArgs& p, int mt, int nt, int wm, int wn, int l15, int g, int i_lo,
                                            int i_hi, uint2 (&yraw)[4][4]) {
  const long long mrow0 = (long long)mt * BM + wm * 64 + g * 4;
  const bool ragged = (long long)(mt + 1) * BM > p.Mp;
  const long long y_row_bytes = (long long)p.y_cstride * 2;
  const char* yrow = p.ybn + (long long)(p.y_coff + nt * BN + wn * 64 + l15 * 4) * 2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < i_lo || i >= i_hi) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      long long m = mrow0 + i * 16 + e;
      if (ragged && m >= p.Mp) m = p.Mp - 1;                // masked out in the epilogue; only the address has to be valid
      yraw[i][e] = *(const uint2*)(yrow + m * y_row_bytes);
    }
  }
}

// coef: the wave's float[4][64] in LDS holding bias (or scale) and shift of its 64 columns -- UH_EPI_MASKED_BNBWD: the
// consuming BatchNorm's scale, shift, mean, 1/std -- brought in by DMA during
// the tile (a global load here would be waited for with vmcnt(0), which also drains the staging DMAs of the next
// tile's first stages that are in flight by now); nullptr: read them from global memory.
template <int BM, int BN, int MODE>
__device__ __forceinline__ void band_epilogue(const GemmArgs& p, const f32x4 (&acc)[4][4], int mt, int nt,
                                              const uint32_t (&fw)[4], int wm, int wn, int l15, int g, float* red,
                                              const float* coef, const uint2 (&yraw)[4][4]) {
  // While a workgroup is here none of its waves issues MFMAs, and at K = 64 this is a quarter of a tile's life,
  // so the arithmetic is laid out for the packed fp32 instructions: values are handled as PAIRS OF ROWS (e, e+1),
  // which are adjacent registers of one accumulator, so no register moves are needed to form the operands.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int ncol = wn * 64 + l15 * 4;
  const int n0 = nt * BN;
  f32x2 cb[4], csc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { cb[j] = f32x2{0.f, 0.f}; csc[j] = f32x2{1.f, 1.f}; }
  if (MODE == UH_EPI_AFFINE_RELU) {
    const f32x4 a4 = coef ? *(const f32x4*)(coef + l15 * 4) : *(const f32x4*)(p.scale + n0 + ncol);
    const f32x4 b4 = coef ? *(const f32x4*)(coef + 64 + l15 * 4) : *(const f32x4*)(p.shift + n0 + ncol);
#pragma unroll
    for (int j = 0; j < 4; ++j) { csc[j] = f32x2{a4[j], a4[j]}; cb[j] = f32x2{b4[j], b4[j]}; }
  } else if (p.bias) {
    const f32x4 b4 = coef ? *(const f32x4*)(coef + l15 * 4) : *(const f32x4*)(p.bias + n0 + ncol);
#pragma unroll
    for (int j = 0; j < 4; ++j) cb[j] = f32x2{b4[j], b4[j]};
  }
  const long long mrow0 = (long long)mt * BM + wm * 64 + g * 4;
  char* orow = p.out + ((mrow0 * p.o_cstride) + p.o_coff + n0 + ncol) * 2;
  const long long o_row_bytes = (long long)p.o_cstride * 2;
  const bool ragged = (long long)(mt + 1) * BM > p.Mp;      // only the last M tile has rows past the end
  f32x2 s1[4], s2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { s1[j] = f32x2{0.f, 0.f}; s2[j] = f32x2{0.f, 0.f}; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t fl = fw[i];
    if (ragged) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (mrow0 + i * 16 + e >= p.Mp) fl &= ~(0xffu << (8 * e));
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {                            // rows e = 2h, 2h + 1
      const f32x2 f = f32x2{(float)((fl >> (16 * h)) & 0xffu), (float)((fl >> (16 * h + 8)) & 0xffu)};
      f32x2 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2 x = f32x2{acc[i][j][2 * h], acc[i][j][2 * h + 1]};
        if (MODE == UH_EPI_AFFINE_RELU) {
          x = x * csc[j] + cb[j];
          x = f32x2{fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
        } else {
          x = x + cb[j];
        }
        x = x * f;
        if (MODE == UH_EPI_RAW_STATS) {
          s1[j] = s1[j] + x;
          s2[j] = x * x + s2[j];
        }
        v[j] = x;
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        uint2 o;
        if (MODE == UH_EPI_RAW_STATS) {
          o.x = pack2h(v[0][r], v[1][r]);
          o.y = pack2h(v[2][r], v[3][r]);
        } else {
          o.x = pack2(v[0][r], v[1][r]);
          o.y = pack2(v[2][r], v[3][r]);
        }
        *(uint2*)(orow + (i * 16 + 2 * h + r) * o_row_bytes) = o;
      }
    }
  }
  if (MODE == UH_EPI_MASKED_BNBWD) {
    // Second pass over the accumulators, after all stores of the tile have been issued (they cover the latency of
    // the y loads): d = out * (y*scale + shift > 0); sums of d and of d * xhat per column.
    const f32x4 sc4 = coef ? *(const f32x4*)(coef + l15 * 4) : *(const f32x4*)(p.scale + n0 + ncol);
    const f32x4 sh4 = coef ? *(const f32x4*)(coef + 64 + l15 * 4) : *(const f32x4*)(p.shift + n0 + ncol);
    const f32x4 mu4 = coef ? *(const f32x4*)(coef + 128 + l15 * 4) : *(const f32x4*)(p.bn_mean + n0 + ncol);
    const f32x4 is4 = coef ? *(const f32x4*)(coef + 192 + l15 * 4) : *(const f32x4*)(p.bn_invstd + n0 + ncol);
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t fl = fw[i];
      if (ragged) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (mrow0 + i * 16 + e >= p.Mp) fl &= ~(0xffu << (8 * e));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float f = (float)((fl >> (8 * e)) & 0xffu);
        const h4 yv = *reinterpret_cast<const h4*>(&yraw[i][e]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float yy = (float)yv[j];
          const float d = (yy * sc4[j] + sh4[j] > 0.f) ? acc[i][j][e] * f : 0.f;
          s1[j][e & 1] += d;
          s2[j][e & 1] += d * ((yy - mu4[j]) * is4[j]);
        }
      }
    }
  }
  if (MODE == UH_EPI_RAW_STATS || MODE == UH_EPI_MASKED_BNBWD) {
    float t1[4], t2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t1[j] = s1[j][0] + s1[j][1];
      t2[j] = s2[j][0] + s2[j][1];
      t1[j] += __shfl_xor(t1[j], 16, 64);
      t1[j] += __shfl_xor(t1[j], 32, 64);
      t2[j] += __shfl_xor(t2[j], 16, 64);
      t2[j] += __shfl_xor(t2[j], 32, 64);
    }
    if (g == 0) {
      *(f32x4*)(red + (wm * 2 + 0) * BN + ncol) = f32x4{t1[0], t1[1], t1[2], t1[3]};
      *(f32x4*)(red + (wm * 2 + 1) * BN + ncol) = f32x4{t2[0], t2[1], t2[2], t2[3]};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // raw barriers do not wait for LDS writes
  }
}

// red[WM][2][BN] -> stat_partial[mt][2][N]; every wave's red[] writes must be two barriers old
template <int BM, int BN>
__device__ __forceinline__ void band_flush_stats(const GemmArgs& p, const float* red, int mt, int nt, int tid) {
  if (tid < 2 * BN) {
    const int which = tid / BN, col = tid % BN;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < BandCfg<BM, BN>::WM; ++w) v += red[(w * 2 + which) * BN + col];
    p.stat_partial[((long long)mt * 2 + which) * p.N + nt * BN + col] = v;
  }
}

// KYIN: iteration order inside a unit.  false: filter row, then channel block.  true: channel block, then filter
// row -- the three filter rows of a channel block read the same activation rows one image row apart in consecutive
// stages, so two of the three reads hit in the XCD's L2 instead of going to memory again.
template <int BM, int BN, int MODE, bool KYIN, bool NMAJOR>
__global__ __launch_bounds__(512) void conv_band_kernel(GemmArgs p, BandSched sc) {
  using C = BandCfg<BM, BN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // workgroups b, b+8, ... share an XCD: give each XCD one contiguous run of positions
  const int pos = (blockIdx.x & 7) * (sc.G >> 3) + (blockIdx.x >> 3);
  const int niter = sc.niter, ntn = sc.ntn;
  // this workgroup's units: u_first, u_first + u_step, ...; `total` iterations, starting at iteration it_first
  // of the first unit
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

  // ---- staging: one wave instruction = 16 rows x 64 B; lane -> (row lane/4, position lane%4) ----------
  // A stage is SG groups; wave w stages groups w, w+8, ... ("slots" q = 0..5; only slot 5 can be absent).
  // Slot pointers are fixed up front; a stage adds one of two wave-uniform offsets (activation / weight).
  constexpr int NQ = 6;
  static_assert((C::SG + 7) / 8 == NQ, "six staging slots per wave");
  const int srow = lane >> 2;
  const int src_chunk = (lane & 3) ^ ((srow >> 1) & 2);
  const long long a_row_bytes = (long long)p.a_cstride * 2;
  const long long b_row_bytes = 9LL * p.K * 2;
  const char* gp[NQ];
  bool slot_is_a[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int gi = wave + 8 * q;
    slot_is_a[q] = gi < C::AG;
    const int j = gi - C::AG;
    const int kx = j / C::BGT, nb = j % C::BGT;
    // weight group nb = LDS rows 16*nb .. 16*nb+15 of the tap = accumulator tile (nb%4) of wave column nb/4
    const int chan = (nb >> 2) * 64 + srow * 4 + (nb & 3);
    const char* pa = p.A + (srow + (long long)gi * 16 - 1 - p.Wp) * a_row_bytes + (long long)p.a_coff * 2 + src_chunk * 16;
    const char* pb = p.Wt + (long long)chan * b_row_bytes + (long long)kx * p.K * 2 + src_chunk * 16;
    gp[q] = slot_is_a[q] ? pa : pb;
  }
  const bool has_slot5 = wave + 40 < C::SG;
  auto issue = [&](int q, int buf, long long offA, long long offB) {
    const long long off = slot_is_a[q] ? offA : offB;
    __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(gp[q] + off),
                                     (UH_LDS void*)(smem + buf * C::STAGE + (wave + 8 * q) * 1024), 16, 0, 0);
  };
  // row flags of this wave's 64 rows: 16 dwords of the row mask (lanes 16.. repeat them)
  const long long mask_last_dword = (p.Mp - 1) >> 2;
  auto issue_flags = [&](int mt) {
    long long d = (((long long)mt * BM + wm * 64) >> 2) + l15;
    if (d > mask_last_dword) d = mask_last_dword;
    __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(p.rowmask + 4 * d),
                                     (UH_LDS void*)(smem + C::FLAG_OFF + wave * 256), 4, 0, 0);
  };

  // per-channel epilogue constants of this wave's 64 columns, one float per lane: bias (or scale), then shift
  auto issue_coef = [&](int nt) {
    const bool bn = MODE == UH_EPI_AFFINE_RELU || MODE == UH_EPI_MASKED_BNBWD;
    const float* first = bn ? p.scale : p.bias;
    const long long col = (long long)nt * BN + wn * 64 + lane;
    if (first)
      __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(first + col),
                                       (UH_LDS void*)(smem + C::COEF_OFF + wave * 1024), 4, 0, 0);
    if (bn)
      __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(p.shift + col),
                                       (UH_LDS void*)(smem + C::COEF_OFF + wave * 1024 + 256), 4, 0, 0);
    if (MODE == UH_EPI_MASKED_BNBWD) {
      __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(p.bn_mean + col),
                                       (UH_LDS void*)(smem + C::COEF_OFF + wave * 1024 + 512), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(p.bn_invstd + col),
                                       (UH_LDS void*)(smem + C::COEF_OFF + wave * 1024 + 768), 4, 0, 0);
    }
  };

  // ---- fragment read offsets (bytes inside a stage) -----------------------------------------------------
  int a_off[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int r = l15 + kx;
    a_off[kx] = (wm * 64 + r) * 64 + ((g ^ ((r >> 1) & 2)) << 4);
  }
  const int b_off = C::AG * 1024 + (wn * 64 + l15) * 64 + ((g ^ ((l15 >> 1) & 2)) << 4);
  float* red = (float*)(smem + C::RED_OFF);
  const float* coef = (const float*)(smem + C::COEF_OFF + wave * 1024);

  // ---- main loop ---------------------------------------------------------------------------------------------
  // The (unit, iteration) pairs of this workgroup form one stream of stages s = 0 .. total-1; a stage is one
  // filter row ky and 32 channels.  One iteration consumes one stage in three phases, one per kx.  A phase is
  // a load part (8 fragment reads of this kx + 2 staging slots of a later stage), a barrier, 16 MFMAs, a
  // barrier.  Waves 4-7 (the second wave on each SIMD) run one barrier behind waves 0-3, so on every SIMD one
  // wave is in its MFMA part while the other is in its load part.
  //   stage s is staged by: slots 0,1 in phase 1 of iteration s-2, slots 2,3 in phase 2 of s-2, slots 4,5 in
  //   phase 0 of s-1; it is retired by the vmcnt(2) in phase 2 of s-1 (only slots 0,1 of stage s+1 are
  //   younger) and first read after the second barrier of that phase, which both halves reach after their
  //   wait.  Its buffer was last read in phase 2 of s-3, whose reads each wave has waited for before the
  //   barriers that precede phase 1 of s-2.
  long long offA, offB;
  int u_n = u_first;
  int mt_n, nt_n;
  band_unit<NMAJOR>(sc, u_n, mt_n, nt_n);
  int ky_n = KYIN ? it_first % 3 : it_first / KC, kc_n = KYIN ? it_first / 3 : it_first % KC;
  auto stage_offsets = [&]() {
    offA = ((long long)mt_n * BM + (long long)ky_n * p.Wp) * a_row_bytes + kc_n * 64;
    offB = (long long)nt_n * BN * b_row_bytes + ((long long)ky_n * 3 * p.K + kc_n * 32) * 2;
  };
  auto next_unit = [&]() {
    u_n += u_step;
    band_unit<NMAJOR>(sc, u_n, mt_n, nt_n);
  };
  auto advance = [&]() {
    if (KYIN) {
      if (++ky_n == 3) {
        ky_n = 0;
        if (++kc_n == KC) {
          kc_n = 0;
          next_unit();
        }
      }
    } else {
      if (++kc_n == KC) {
        kc_n = 0;
        if (++ky_n == 3) {
          ky_n = 0;
          next_unit();
        }
      }
    }
    stage_offsets();
  };
  stage_offsets();
#pragma unroll
  for (int q = 0; q < 5; ++q) issue(q, 0, offA, offB);
  if (has_slot5) issue(5, 0, offA, offB);
  if (total > 1) {
    advance();                                  // -> stage 1
#pragma unroll
    for (int q = 0; q < 4; ++q) issue(q, 1, offA, offB);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wave >= 4) __builtin_amdgcn_s_barrier();  // the stagger

  int buf = 0, s = 0;
  int pend_mt = -1, pend_nt = 0;                // unit whose BatchNorm sums still sit in red[]
  int u = u_first, it_a = it_first;
  bf16x8 af[4], bfr[4];
  while (s < total) {
    const int cnt = min(niter - it_a, total - s);
    const bool full = (it_a == 0 && cnt == niter);
    int mt, nt;
    band_unit<NMAJOR>(sc, u, mt, nt);
    f32x4 acc[4][4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[ii][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < cnt; ++c, ++s) {
      const char* S = smem + buf * C::STAGE;
      const int buf1 = (buf == 2) ? 0 : buf + 1;
      const int buf2 = (buf == 0) ? 2 : buf - 1;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(S + b_off + kx * (BN * 64) + j * 1024);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) af[ii] = *(const bf16x8*)(S + a_off[kx] + ii * 1024);
        if (kx == 0) {
          if (c == 0 && full) {
            issue_flags(mt);
            issue_coef(nt);
          }
          if (s + 1 < total) {
            issue(4, buf1, offA, offB);
            if (has_slot5) issue(5, buf1, offA, offB);
          }
        } else if (kx == 1) {
          if ((MODE == UH_EPI_RAW_STATS || MODE == UH_EPI_MASKED_BNBWD) && c == 0 && pend_mt >= 0) {
            band_flush_stats<BM, BN>(p, red, pend_mt, pend_nt, tid);
            pend_mt = -1;
          }
          if (s + 2 < total) {
            advance();
            issue(0, buf2, offA, offB);
            issue(1, buf2, offA, offB);
          }
        } else {
          if (s + 2 < total) {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            issue(2, buf2, offA, offB);
            issue(3, buf2, offA, offB);
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ii], bfr[j], acc[ii][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }
      buf = buf1;
    }

    if (full) {
      uint32_t fw[4];
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) fw[ii] = *(const uint32_t*)(smem + C::FLAG_OFF + wave * 256 + (ii * 4 + g) * 4);
      uint2 yraw[4][4];
      if (MODE == UH_EPI_MASKED_BNBWD) band_load_y<BM, BN>(p, mt, nt, wm, wn, l15, g, 0, 4, yraw);
      band_epilogue<BM, BN, MODE>(p, acc, mt, nt, fw, wm, wn, l15, g, red, coef, yraw);
      if (MODE == UH_EPI_RAW_STATS || MODE == UH_EPI_MASKED_BNBWD) { pend_mt = mt; pend_nt = nt; }
    } else {
      // a piece of a unit: park the accumulators for conv_band_fixup_kernel
      f32x4* slot = (f32x4*)sc.ws + ((long long)(2 * pos + (it_a > 0 ? 0 : 1)) * 16) * 512 + tid;
#pragma unroll
      for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int j = 0; j < 4; ++j) slot[(ii * 4 + j) * 512] = acc[ii][j];
    }
    u += u_step;
    it_a = 0;
  }
  if (wave < 4) __builtin_amdgcn_s_barrier();   // match the second half's extra barrier
  if ((MODE == UH_EPI_RAW_STATS || MODE == UH_EPI_MASKED_BNBWD) && pend_mt >= 0) {
    __syncthreads();
    band_flush_stats<BM, BN>(p, red, pend_mt, pend_nt, tid);
  }
}

// One workgroup per cut: boundary k (between positions k-1 and k) that falls inside a unit and is the first
// such boundary of that unit sums the unit's pieces in position order and finishes the unit.
template <int BM, int BN, int MODE>
__global__ __launch_bounds__(512) void conv_band_fixup_kernel(GemmArgs p, BandSched sc) {
  using C = BandCfg<BM, BN>;
  __shared__ __attribute__((aligned(16))) float red[C::WM * 2 * BN];
  const int k = blockIdx.x + 1;
  const long long sk = band_start(sc, k);
  const int niter = sc.niter;
  if (sk % niter == 0) return;
  const long long u = sk / niter, ub = u * niter, ue = ub + niter;
  if (k >= 2 && band_start(sc, k - 1) > ub) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int l15 = lane & 15, g = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int ii = 0; ii < 4; ++ii)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[ii][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int pp = k - 1; pp < sc.G; ++pp) {
    const long long ps = band_start(sc, pp), pe = band_start(sc, pp + 1);
    if (ps >= ue) break;
    if (pe <= ps || pe <= ub) continue;
    const int inside = (ps > ub) ? 0 : 1;        // slot 2pp: piece starts inside the unit; 2pp+1: it ends inside
    const f32x4* slot = (const f32x4*)sc.ws + ((long long)(2 * pp + inside) * 16) * 512 + tid;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[ii][j] += slot[(ii * 4 + j) * 512];
  }
  int mt, nt;
  if (sc.nmajor) band_unit<true>(sc, (int)u, mt, nt);
  else band_unit<false>(sc, (int)u, mt, nt);
  uint32_t fw[4];
  const long long mask_last_dword = (p.Mp - 1) >> 2;
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) {
    long long d = (((long long)mt * BM + wm * 64) >> 2) + ii * 4 + g;
    if (d > mask_last_dword) d = mask_last_dword;
    fw[ii] = *(const uint32_t*)(p.rowmask + 4 * d);
  }
  uint2 yraw[4][4];
  if (MODE == UH_EPI_MASKED_BNBWD) band_load_y<BM, BN>(p, mt, nt, wm, wn, l15, g, 0, 4, yraw);
  band_epilogue<BM, BN, MODE>(p, acc, mt, nt, fw, wm, wn, l15, g, red, nullptr, yraw);
  if (MODE == UH_EPI_RAW_STATS || MODE == UH_EPI_MASKED_BNBWD) {
    __syncthreads();
    band_flush_stats<BM, BN>(p, red, mt, nt, tid);
  }
}

// ===================================================================================================
// T = 1 "stream" kernel: plain GEMMs out[m][n] = sum_k A[m][k] * Wp[n][k] (the k2s2 transposed convolutions and their
// input gradients, LUNeXt's 1x1 and k2s2 convolutions).  Same machinery as the band kernel -- persistent workgroups of
// eight 64x64 waves, global_load_lds staging with the XOR swizzle on the source chunk, counted s_waitcnt, raw
// s_barriers with waves 4-7 running one barrier behind waves 0-3, stores straight from the accumulators -- but a stage
// is one 32-channel slice of a BM-row A tile plus the matching slice of BN weight rows, consumed in ONE phase, and the
// ring has four stages: stage s+2 is issued while stage s is consumed into the buffer that held stage s-2, whose last
// readers (the lagging half, one barrier late) finished their MFMAs two barriers ago.
//   stage s+1 is retired by the vmcnt(NQ) that follows the issue of stage s+2 in iteration s (only that stage's
//   loads are younger) and first read after the next barrier both halves pass.
// These GEMMs are short (K = 128 ... 2048) and the largest are HBM-bound (transposed convolution at full resolution:
// 408 MB moved for 35 GFLOP), so what matters is that the ring runs on across tile boundaries and that stores of one
// tile overlap the loads of the next.
// NW waves per workgroup: eight 64x64 waves (tiles 256x128 / 512x64), or sixteen for a 256x256 tile, which stages a
// third fewer bytes and DMA instructions per MFMA (32 KB for 256 MFMAs instead of 24 KB for 128).
template <int BM, int BN, int NW = 8>
struct StreamCfg {
  static constexpr int WN = BN / 64;
  static constexpr int WM = NW / WN;
  static_assert(BM == WM * 64, "each wave owns 64 rows");
  static constexpr int AG = BM / 16;              // 1 KiB groups (16 rows x 64 B) of the A slice
  static constexpr int BG = BN / 16;
  static constexpr int SG = AG + BG;
  static constexpr int NQ = (SG + NW - 1) / NW;   // staging slots per wave
  static constexpr int STAGE = SG * 1024;
  // stage s + DIST is issued while stage s is consumed, into the buffer that held stage s - 2 (ring of DIST + 2).
  // DIST = 3 (five stages, 126 KB at 256 x 128) was measured on the four transposed convolutions and their input
  // gradients: no launch faster than with 2; these GEMMs are bound by the bytes staged per MFMA (24 KB for 128
  // MFMAs), not by the latency of a stage.
  static constexpr int DIST = 2;
  static constexpr int NSTAGE = DIST + 2;
  static constexpr int FLAG_OFF = NSTAGE * STAGE;
  static constexpr int RED_OFF = FLAG_OFF + NW * 256;
  static constexpr int COEF_OFF = RED_OFF + WM * 2 * BN * 4;  // NW waves x float[4][64], as in BandCfg
  static constexpr int LDS_BYTES = COEF_OFF + NW * 1024;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// Transposed convolution k2s2 (unet.py:132-164, lunext.py:161-179): column n = tap*Cout + co of input position
// (b, y, x) goes to output position (b, 2y + tap/2, 2x + tap%2), channel co, of the (B, 2H, 2W) output geometry.
template <int BM, int BN>
__device__ __forceinline__ void stream_epilogue_convt(const GemmArgs& p, const f32x4 (&acc)[4][4], int mt, int nt,
                                                      const uint32_t (&fw)[4], int wm, int wn, int l15, int g,
                                                      const float* coef) {
  const int n = nt * BN + wn * 64 + l15 * 4;
  const int otap = n / p.Cout;                     // >= 4: zero-padded weight rows (N > 4*Cout)
  const int co = n - otap * p.Cout;
  float cb[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {                                    // the bias of the wave's 64 columns, brought into LDS during the tile
    const f32x4 b4 = *(const f32x4*)(coef + l15 * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) cb[j] = b4[j];
  }
  const int Hp = p.H + 1, Wp = p.W + 1;
  const int oHp = 2 * p.H + 1, oWp = 2 * p.W + 1;
  // Output row of input position (b, y, x) and this lane's tap: q = b*oHp*oWp + (2y + dy)*oWp + 2x + dx.  The lane's
  // first row is decoded once per tile in 32-bit arithmetic (Mp < 2^31 and the output's row count are checked on the
  // host); from there q is STEPPED with the position (+2 per x, two constants at the end of an image row / image), so a
  // row costs a handful of integer instructions instead of a 64-bit multiply chain: at K = 128 the old form of this
  // epilogue was 1085 instructions per wave and tile against 64 MFMAs.
  const int dy = (otap >> 1) & 1, dx = otap & 1;
  const unsigned mrow0 = (unsigned)mt * BM + wm * 64 + g * 4;
  int x = (int)(mrow0 % (unsigned)Wp);
  const unsigned r = mrow0 / (unsigned)Wp;
  int y = (int)(r % (unsigned)Hp);
  const unsigned b = r / (unsigned)Hp;
  const unsigned row_step = 2u * oWp - 2u * Wp;                      // added on top of the +2 when x wraps
  const unsigned img_step = (unsigned)oHp * oWp - 2u * Hp * oWp;     // added on top of that when y wraps
  unsigned q = (b * oHp + 2u * y + dy) * oWp + 2u * x + dx;
  const long long row_bytes = (long long)p.o_cstride * 2;
  char* const obase = p.out + (long long)(p.o_coff + co) * 2;
  const bool lane_ok = otap < 4;
  const bool ragged = (long long)(mt + 1) * BM > p.Mp;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float f = (float)((fw[i] >> (8 * e)) & 0xffu);
      uint2 o;
      o.x = pack2((acc[i][0][e] + cb[0]) * f, (acc[i][1][e] + cb[1]) * f);
      o.y = pack2((acc[i][2][e] + cb[2]) * f, (acc[i][3][e] + cb[3]) * f);
      // the pad row / column of the input maps to the output's pad row / column for dy = 0 / dx = 0 (zeros are
      // stored there: f = 0) and past it for dy = 1 / dx = 1
      bool ok = lane_ok && !(dy && y == p.H) && !(dx && x == p.W);
      if (ragged) ok = ok && (long long)mrow0 + i * 16 + e < p.Mp;
      if (ok) *(uint2*)(obase + (long long)q * row_bytes) = o;
      q += 2;
      if (++x == Wp) {
        x = 0;
        q += row_step;
        if (++y == Hp) {
          y = 0;
          q += img_step;
        }
      }
    }
    x += 12;                                       // to the lane's next group of four rows, 16 rows on
    q += 24;
    while (x >= Wp) {
      x -= Wp;
      q += row_step;
      if (++y == Hp) {
        y = 0;
        q += img_step;
      }
    }
  }
}

template <int BM, int BN, int MODE, int NW = 8>
__global__ __launch_bounds__(NW * 64) void conv_stream_kernel(GemmArgs p, BandSched sc) {
  using C = StreamCfg<BM, BN, NW>;
  static_assert(NW == 8 || (MODE != UH_EPI_RAW_STATS && MODE != UH_EPI_MASKED_BNBWD),
                "the statistics epilogues reduce over eight waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int niter = sc.niter, ntn = sc.ntn;        // niter = K/32
  // whole units per workgroup; each XCD (workgroups b, b+8, ...) owns one contiguous range of units
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int u_end = min((xcd + 1) * sc.upx, sc.ntm * ntn);
  const int u_first = xcd * sc.upx + jx;
  const int u_step = sc.G >> 3;
  const int total = (u_first < u_end) ? ((u_end - u_first + u_step - 1) / u_step) * niter : 0;
  if (total <= 0) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int l15 = lane & 15, g = lane >> 4;

  // ---- staging: one wave instruction = 16 rows x 64 B; lane -> (row lane/4, position lane%4) ----------
  constexpr int NQ = C::NQ;
  const int srow = lane >> 2;
  const int src_chunk = (lane & 3) ^ ((srow >> 1) & 2);
  const long long a_row_bytes = (long long)p.a_cstride * 2;
  const long long b_row_bytes = (long long)p.K * 2;
  const char* gp[NQ];
  bool slot_is_a[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int gi = wave + NW * q;
    slot_is_a[q] = gi < C::AG;
    const int nb = gi - C::AG;
    const int chan = (nb >> 2) * 64 + srow * 4 + (nb & 3);
    const char* pa = p.A + (srow + (long long)gi * 16) * a_row_bytes + (long long)p.a_coff * 2 + src_chunk * 16;
    const char* pb = p.Wt + (long long)chan * b_row_bytes + src_chunk * 16;
    gp[q] = slot_is_a[q] ? pa : pb;
  }
  const bool has_last = wave + NW * (NQ - 1) < C::SG;
  auto issue_stage = [&](int buf, long long offA, long long offB) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (q == NQ - 1 && !has_last) break;
      const long long off = slot_is_a[q] ? offA : offB;
      __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(gp[q] + off),
                                       (UH_LDS void*)(smem + buf * C::STAGE + (wave + NW * q) * 1024), 16, 0, 0);
    }
  };
  const long long mask_last_dword = (p.Mp - 1) >> 2;
  auto issue_flags = [&](int mt) {
    long long d = (((long long)mt * BM + wm * 64) >> 2) + l15;
    if (d > mask_last_dword) d = mask_last_dword;
    __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(p.rowmask + 4 * d),
                                     (UH_LDS void*)(smem + C::FLAG_OFF + wave * 256), 4, 0, 0);
  };

  // per-channel epilogue constants of this wave's 64 columns, one float per lane (see band_epilogue): bias (or scale),
  // then shift; for a transposed convolution column n carries the bias of output channel n % Cout
  auto issue_coef = [&](int nt) {
    const int n = nt * BN + wn * 64 + lane;
    if (MODE == UH_EPI_CONVT) {
      if (p.bias) {
        const int otap = n / p.Cout;
        const int co = otap < 4 ? n - otap * p.Cout : 0;         // columns of zero-padded weight rows are never stored
        __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(p.bias + co),
                                         (UH_LDS void*)(smem + C::COEF_OFF + wave * 1024), 4, 0, 0);
      }
    } else {
      const bool bn = MODE == UH_EPI_AFFINE_RELU || MODE == UH_EPI_MASKED_BNBWD;
      const float* first = bn ? p.scale : p.bias;
      if (first)
        __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(first + n),
                                         (UH_LDS void*)(smem + C::COEF_OFF + wave * 1024), 4, 0, 0);
      if (bn)
        __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(p.shift + n),
                                         (UH_LDS void*)(smem + C::COEF_OFF + wave * 1024 + 256), 4, 0, 0);
      if (MODE == UH_EPI_MASKED_BNBWD) {
        __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(p.bn_mean + n),
                                         (UH_LDS void*)(smem + C::COEF_OFF + wave * 1024 + 512), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(p.bn_invstd + n),
                                         (UH_LDS void*)(smem + C::COEF_OFF + wave * 1024 + 768), 4, 0, 0);
      }
    }
  };
  const float* coef = (const float*)(smem + C::COEF_OFF + wave * 1024);

  // ---- fragment read offsets (bytes inside a stage) -----------------------------------------------------
  const int a_off = (wm * 64 + l15) * 64 + ((g ^ ((l15 >> 1) & 2)) << 4);
  const int b_off = C::AG * 1024 + (wn * 64 + l15) * 64 + ((g ^ ((l15 >> 1) & 2)) << 4);
  float* red = (float*)(smem + C::RED_OFF);

  long long offA, offB;
  int u_n = u_first, kc_n = 0;
  auto stage_offsets = [&]() {
    int mt_n, nt_n;
    band_unit<false>(sc, u_n, mt_n, nt_n);
    offA = (long long)mt_n * BM * a_row_bytes + kc_n * 64;
    offB = (long long)nt_n * BN * b_row_bytes + kc_n * 64;
  };
  auto advance = [&]() {
    if (++kc_n == niter) {
      kc_n = 0;
      u_n += u_step;
    }
    stage_offsets();
  };
  // all but this wave's `ahead` youngest stages have landed (a stage is NQ or NQ - 1 wave-instructions)
  auto wait_stages = [&](int ahead) {
    if (ahead <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (ahead == 1) {
      if (has_last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ - 1) : "memory");
    } else {
      if (has_last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NQ) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NQ - 2) : "memory");
    }
  };
  constexpr int DIST = C::DIST, NST = C::NSTAGE;
  stage_offsets();
  issue_stage(0, offA, offB);
  int issued = 1;                               // stages issued so far
  for (; issued < DIST && issued < total; ++issued) {
    advance();
    issue_stage(issued, offA, offB);
  }
  wait_stages(issued - 1);                      // stage 0 has landed
  __builtin_amdgcn_s_barrier();
  if (wave >= NW / 2) __builtin_amdgcn_s_barrier();  // the stagger

  int buf = 0, s = 0;
  int pend_mt = -1, pend_nt = 0;
  int u = u_first;
  bf16x8 af[4], bfr[4];
  while (s < total) {
    int mt, nt;
    band_unit<false>(sc, u, mt, nt);
    f32x4 acc[4][4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[ii][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < niter; ++c, ++s) {
      const char* S = smem + buf * C::STAGE;
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(S + b_off + j * 1024);
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) af[ii] = *(const bf16x8*)(S + a_off + ii * 1024);
      if (c == 0) {
        issue_flags(mt);
        issue_coef(nt);
      }
      if constexpr (MODE == UH_EPI_RAW_STATS || MODE == UH_EPI_MASKED_BNBWD) {
        if (c == 1 && pend_mt >= 0) {
          band_flush_stats<BM, BN>(p, red, pend_mt, pend_nt, tid);
          pend_mt = -1;
        }
      }
      if (s + DIST < total) {
        advance();
        int nb = buf + DIST;
        nb = nb >= NST ? nb - NST : nb;
        issue_stage(nb, offA, offB);
        wait_stages(DIST - 1);                  // stage s + 1 has landed; the younger ones stay in flight
      } else {
        wait_stages(total - s - 2);             // the tail: whatever is younger than stage s + 1
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ii], bfr[j], acc[ii][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      buf = buf + 1 >= NST ? 0 : buf + 1;
    }

    uint32_t fw[4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) fw[ii] = *(const uint32_t*)(smem + C::FLAG_OFF + wave * 256 + (ii * 4 + g) * 4);
    if constexpr (MODE == UH_EPI_CONVT) {
      if ((long long)(mt + 1) * BM > p.Mp) {      // rows past the end of the last tile carry no valid flag
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if ((long long)mt * BM + wm * 64 + ii * 16 + g * 4 + e >= p.Mp) fw[ii] &= ~(0xffu << (8 * e));
      }
      stream_epilogue_convt<BM, BN>(p, acc, mt, nt, fw, wm, wn, l15, g, coef);
    } else {
      uint2 yraw[4][4];
      if (MODE == UH_EPI_MASKED_BNBWD) band_load_y<BM, BN>(p, mt, nt, wm, wn, l15, g, 0, 4, yraw);
      band_epilogue<BM, BN, MODE>(p, acc, mt, nt, fw, wm, wn, l15, g, red, coef, yraw);
      if (MODE == UH_EPI_RAW_STATS || MODE == UH_EPI_MASKED_BNBWD) { pend_mt = mt; pend_nt = nt; }
    }
    u += u_step;
  }
  if (wave < NW / 2) __builtin_amdgcn_s_barrier();   // match the second half's extra barrier
  if constexpr (MODE == UH_EPI_RAW_STATS || MODE == UH_EPI_MASKED_BNBWD) {
    if (pend_mt >= 0) {
      __syncthreads();
      band_flush_stats<BM, BN>(p, red, pend_mt, pend_nt, tid);
    }
  }
}

template <int BM, int BN>
static size_t band_workspace_bytes(long long Mp, int N, int K) {
  const long long units = ((Mp + BM - 1) / BM) * (N / BN);
  return (size_t)2 * band_workgroups(units, 3 * (K / 32)) * BM * BN * sizeof(float);
}

// Largest K whose launches walk a unit channel block first (UNET_AMD_BAND_KYIN_MAXK; default: every K; 0: none).
// Measured at B = 32, 256^2: the L2's memory-side reads of a K = 64 launch drop from three times the activation
// bytes to 1.04 times, the launch takes the same time alone (its re-reads were served by the Infinity Cache) and
// the training step gains 0 - 0.1 ms from the bandwidth left to the kernels running beside it.
static int band_kyin_maxk() {
  static const int v = [] {
    const char* e = getenv("UNET_AMD_BAND_KYIN_MAXK");
    return e ? atoi(e) : 1 << 30;
  }();
  return v;
}

// Weight matrices larger than this many KB are walked N tile first (see BandSched; UNET_AMD_NMAJOR_KB, 0: never).
static bool weights_outgrow_l2(long long weight_bytes) {
  static const long long kb = [] {
    const char* e = getenv("UNET_AMD_NMAJOR_KB");
    return e ? atoll(e) : 3072LL;
  }();
  return kb > 0 && weight_bytes > kb * 1024;
}

template <int BM, int BN, int MODE>
static void launch_band_mode(const GemmArgs& p, const BandSched& sc, hipStream_t st) {
  using C = BandCfg<BM, BN>;
  auto go = [&](auto kernel) {
    uh_set_max_lds((const void*)kernel, C::LDS_BYTES);
    hipLaunchKernelGGL(kernel, dim3(sc.G), dim3(512), C::LDS_BYTES, st, p, sc);
  };
  const bool kyin = p.K <= band_kyin_maxk();
  if (sc.nmajor) {                                 // (only cut launches, which only the forward modes have)
    if (kyin) go(conv_band_kernel<BM, BN, MODE, true, true>);
    else go(conv_band_kernel<BM, BN, MODE, false, true>);
  } else {
    if (kyin) go(conv_band_kernel<BM, BN, MODE, true, false>);
    else go(conv_band_kernel<BM, BN, MODE, false, false>);
  }
  if (!sc.aligned)
    hipLaunchKernelGGL((conv_band_fixup_kernel<BM, BN, MODE>), dim3(sc.G - 1), dim3(512), 0, st, p, sc);
}

template <int BM, int BN>
static void launch_band(const GemmArgs& p, void* ws, size_t ws_bytes, hipStream_t st) {
  BandSched sc;
  sc.ntm = (int)((p.Mp + BM - 1) / BM);
  sc.ntn = p.N / BN;
  sc.niter = 3 * (p.K / 32);
  const long long units = (long long)sc.ntm * sc.ntn;
  sc.G = band_workgroups(units, sc.niter);
  sc.ws = (float*)ws;
  sc.upx = (int)((units + 7) / 8);
  // even cuts need the workspace and cost a second launch plus ~2 x 128 KB of traffic per workgroup, and the
  // workgroups of a thinly populated last round run faster than those of a full one: measured, even cuts win
  // from about a third of the CU-time idle (137x4 and 37x8 tiles on 256 CUs), not at a fifth (529x2 tiles)
  const long long rounds = (units + sc.G - 1) / sc.G;
  const bool wasteful = rounds * sc.G * 100 > units * 125;
  sc.aligned = (ws == nullptr || ws_bytes < band_workspace_bytes<BM, BN>(p.Mp, p.N, p.K) || !wasteful) ? 1 : 0;
  // measured on the 32^2 and 16^2 layers (weights 4.7 - 19 MB): N tile first is 15 - 21 % faster with even cuts, where
  // a workgroup's consecutive units then share their weight slice, and no faster with whole units per workgroup
  // (in the training step too: 14.20 ms either way with whole units, 14.51 -> 14.21 ms with even cuts)
  sc.nmajor = (!sc.aligned && sc.ntn > 1 && weights_outgrow_l2(9LL * p.K * p.N * 2)) ? 1 : 0;
  if (p.mode == UH_EPI_RAW_STATS) launch_band_mode<BM, BN, UH_EPI_RAW_STATS>(p, sc, st);
  else if (p.mode == UH_EPI_AFFINE_RELU) launch_band_mode<BM, BN, UH_EPI_AFFINE_RELU>(p, sc, st);
  else if (p.mode == UH_EPI_MASKED_BNBWD) launch_band_mode<BM, BN, UH_EPI_MASKED_BNBWD>(p, sc, st);
  else launch_band_mode<BM, BN, UH_EPI_MASKED>(p, sc, st);
}

template <int BM, int BN, int MODE, int NW = 8>
static void launch_stream_mode(const GemmArgs& p, const BandSched& sc, hipStream_t st) {
  using C = StreamCfg<BM, BN, NW>;
  uh_set_max_lds((const void*)conv_stream_kernel<BM, BN, MODE, NW>, C::LDS_BYTES);
  hipLaunchKernelGGL((conv_stream_kernel<BM, BN, MODE, NW>), dim3(sc.G), dim3(NW * 64), C::LDS_BYTES, st, p, sc);
}

// 256 x 256 tiles on sixteen waves: transposed convolutions and masked GEMMs whose N is a multiple of 256 and that have
// at least two rounds of such tiles (UNET_AMD_STREAM_WIDE=0: never)
static bool stream_wide(const GemmArgs& p) {
  static const bool on = [] { const char* e = getenv("UNET_AMD_STREAM_WIDE"); return !e || atoi(e) != 0; }();
  if (!on || p.N % 256 != 0 || (p.mode != UH_EPI_MASKED && p.mode != UH_EPI_CONVT)) return false;
  return ((p.Mp + 255) / 256) * (p.N / 256) >= 512;
}

static void launch_stream_wide(const GemmArgs& p, hipStream_t st) {
  BandSched sc;
  sc.ntm = (int)((p.Mp + 255) / 256);
  sc.ntn = p.N / 256;
  sc.niter = p.K / 32;
  const long long units = (long long)sc.ntm * sc.ntn;
  sc.G = band_workgroups(units, sc.niter);
  sc.ws = nullptr;
  sc.upx = (int)((units + 7) / 8);
  sc.aligned = 1;
  sc.nmajor = 0;
  if (p.mode == UH_EPI_MASKED) launch_stream_mode<256, 256, UH_EPI_MASKED, 16>(p, sc, st);
  else launch_stream_mode<256, 256, UH_EPI_CONVT, 16>(p, sc, st);
}

template <int BM, int BN>
static void launch_stream(const GemmArgs& p, hipStream_t st) {
  BandSched sc;
  sc.ntm = (int)((p.Mp + BM - 1) / BM);
  sc.ntn = p.N / BN;
  sc.niter = p.K / 32;
  const long long units = (long long)sc.ntm * sc.ntn;
  sc.G = band_workgroups(units, sc.niter);
  sc.ws = nullptr;
  sc.upx = (int)((units + 7) / 8);
  sc.aligned = 1;
  sc.nmajor = 0;                                  // (measured on the 1024 x 2048 transposed convolution: 4 % slower)
  switch (p.mode) {
    case UH_EPI_RAW_STATS: launch_stream_mode<BM, BN, UH_EPI_RAW_STATS>(p, sc, st); break;
    case UH_EPI_AFFINE_RELU: launch_stream_mode<BM, BN, UH_EPI_AFFINE_RELU>(p, sc, st); break;
    case UH_EPI_MASKED: launch_stream_mode<BM, BN, UH_EPI_MASKED>(p, sc, st); break;
    case UH_EPI_MASKED_BNBWD: launch_stream_mode<BM, BN, UH_EPI_MASKED_BNBWD>(p, sc, st); break;
    default: launch_stream_mode<BM, BN, UH_EPI_CONVT>(p, sc, st); break;
  }
}

static bool use_wide_m(int N) { return N % 128 != 0; }  // N = 64, 192, ...: tiles 64 wide

// M tiles of the T = 9 path (rows of stat_partial in RAW_STATS mode)
extern "C" int uh_conv_gemm_mtiles(int B, int H, int W, int N) {
  const long long Mp = uh_mp(B, H, W);
  const int bm = use_wide_m(N) ? 512 : 256;
  return (int)((Mp + bm - 1) / bm);
}

extern "C" size_t uh_conv_gemm_workspace(int B, int H, int W, int K, int N, int T) {
  if (T != 9 || B <= 0 || H <= 0 || W <= 0 || K <= 0 || N <= 0) return 0;
  return use_wide_m(N) ? band_workspace_bytes<512, 64>(uh_mp(B, H, W), N, K)
                       : band_workspace_bytes<256, 128>(uh_mp(B, H, W), N, K);
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
  UH_REQUIRE(uh_mp(B, H, W) < (1ll << 31) - 1024, "conv_gemm: %lld rows exceed the 32-bit row arithmetic of the kernels",
             uh_mp(B, H, W));
  UH_REQUIRE(mode >= UH_EPI_RAW_STATS && mode <= UH_EPI_CONVT, "conv_gemm: bad mode %d", mode);
  UH_REQUIRE(a && w_packed && out && rowmask, "conv_gemm: null pointer");
  UH_REQUIRE(a_cstride >= a_coff + K && a_cstride % 8 == 0 && a_coff % 8 == 0,
             "conv_gemm: bad A view (cstride=%d coff=%d K=%d)", a_cstride, a_coff, K);
  UH_REQUIRE(o_cstride % 8 == 0 && o_coff % 8 == 0, "conv_gemm: bad out view");
  if (mode == UH_EPI_CONVT) {
    UH_REQUIRE(o_cstride - o_coff >= 8, "conv_gemm: bad transposed-conv view");
    UH_REQUIRE(((N / 4 < o_cstride - o_coff) ? N / 4 : o_cstride - o_coff) % 4 == 0,
               "conv_gemm: transposed conv needs an output channel count that is a multiple of 4");
    UH_REQUIRE(T == 1, "conv_gemm: transposed conv is a T=1 GEMM");
    // the pixel-shuffle epilogue steps its OUTPUT row index in 32 bits
    UH_REQUIRE((long long)B * (2ll * H + 1) * (2ll * W + 1) < (1ll << 31),
               "conv_gemm: transposed conv output has %lld rows, beyond the epilogue's 32-bit row arithmetic",
               (long long)B * (2ll * H + 1) * (2ll * W + 1));
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
  p.ybn = nullptr;
  p.y_cstride = p.y_coff = 0;
  p.bn_mean = p.bn_invstd = nullptr;
  hipStream_t st = (hipStream_t)s;
  if (T == 9) {
    UH_REQUIRE(((uintptr_t)workspace & 15) == 0, "conv_gemm: workspace must be 16-byte aligned");
    if (use_wide_m(N))
      launch_band<512, 64>(p, workspace, workspace_bytes, st);
    else
      launch_band<256, 128>(p, workspace, workspace_bytes, st);
  } else if (stream_wide(p)) {
    launch_stream_wide(p, st);
  } else if (use_wide_m(N)) {
    launch_stream<512, 64>(p, st);
  } else {
    launch_stream<256, 128>(p, st);
  }
  return uh_check_launch("conv_gemm");
}

static int gemm_bnbwd(int T, int B, int H, int W, const void* a, int a_cstride, int a_coff, int K,
                      const void* w_packed, int N, void* out, int o_cstride, int o_coff,
                      const uint8_t* rowmask, const void* y, int y_cstride, int y_coff, const float* mean,
                      const float* invstd, const float* scale, const float* shift, float* stat_partial,
                      uh_stream s) {
  UH_REQUIRE(B > 0 && H > 0 && W > 0, "conv_gemm_bnbwd: bad geometry B=%d H=%d W=%d", B, H, W);
  UH_REQUIRE(K > 0 && K % 64 == 0 && N > 0 && N % 64 == 0, "conv_gemm_bnbwd: K=%d N=%d must be multiples of 64", K, N);
  UH_REQUIRE(a && w_packed && out && rowmask && y && mean && invstd && scale && shift && stat_partial,
             "conv_gemm_bnbwd: null pointer");
  UH_REQUIRE(a_cstride >= a_coff + K && a_cstride % 8 == 0 && a_coff % 8 == 0, "conv_gemm_bnbwd: bad A view");
  UH_REQUIRE(o_cstride >= o_coff + N && o_cstride % 8 == 0 && o_coff % 8 == 0, "conv_gemm_bnbwd: bad out view");
  UH_REQUIRE(y_cstride >= y_coff + N && y_cstride % 4 == 0 && y_coff % 4 == 0, "conv_gemm_bnbwd: bad y view");
  GemmArgs p;
  p.A = (const char*)a;
  p.Wt = (const char*)w_packed;
  p.out = (char*)out;
  p.bias = nullptr;
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
  p.mode = UH_EPI_MASKED_BNBWD;
  p.H = H;
  p.W = W;
  p.Cout = N;
  p.ybn = (const char*)y;
  p.y_cstride = y_cstride;
  p.y_coff = y_coff;
  p.bn_mean = mean;
  p.bn_invstd = invstd;
  hipStream_t st = (hipStream_t)s;
  if (T == 1) {
    UH_REQUIRE(uh_mp(B, H, W) < (1ll << 31) - 1024, "conv_gemm_bnbwd: too many rows");
    if (use_wide_m(N))
      launch_stream<512, 64>(p, st);
    else
      launch_stream<256, 128>(p, st);
  } else if (use_wide_m(N)) {
    launch_band<512, 64>(p, nullptr, 0, st);
  } else {
    launch_band<256, 128>(p, nullptr, 0, st);
  }
  return uh_check_launch("conv_gemm_bnbwd");
}

extern "C" int uh_conv_gemm_bnbwd(int B, int H, int W, const void* a, int a_cstride, int a_coff, int K,
                                  const void* w_packed, int N, void* out, int o_cstride, int o_coff,
                                  const uint8_t* rowmask, const void* y, int y_cstride, int y_coff, const float* mean,
                                  const float* invstd, const float* scale, const float* shift, float* stat_partial,
                                  uh_stream s) {
  return gemm_bnbwd(9, B, H, W, a, a_cstride, a_coff, K, w_packed, N, out, o_cstride, o_coff, rowmask, y, y_cstride,
                    y_coff, mean, invstd, scale, shift, stat_partial, s);
}

extern "C" int uh_conv_gemm1_bnbwd(int B, int H, int W, const void* a, int a_cstride, int a_coff, int K,
                                   const void* w_packed, int N, void* out, int o_cstride, int o_coff,
                                   const uint8_t* rowmask, const void* y, int y_cstride, int y_coff, const float* mean,
                                   const float* invstd, const float* scale, const float* shift, float* stat_partial,
                                   uh_stream s) {
  return gemm_bnbwd(1, B, H, W, a, a_cstride, a_coff, K, w_packed, N, out, o_cstride, o_coff, rowmask, y, y_cstride,
                    y_coff, mean, invstd, scale, shift, stat_partial, s);
}
