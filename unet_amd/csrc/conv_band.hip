// 3x3 convolution (pad 1) as an implicit GEMM on gfx950 matrix cores: the kernel behind uh_conv_gemm(T = 9),
// forward and input-gradient.
//
//   out[m][n] = sum_{ky,kx} sum_k A[m + (ky-1)*Wp + (kx-1)][k] * W[n][ky*3+kx][k]
//
// What shapes it (all measured on MI355X):
//   * The global->LDS path, not the MFMA pipe, is the first limit.  Tiles are staged with global_load_lds,
//     16 B per lane, in pieces of 8 rows x 128 B: full cache lines (64 channels of one pixel).  Pieces of
//     16 rows x 64 B moved the same bytes ~1.5x slower.
//   * The three taps of one filter row (kx = 0,1,2) read the SAME activation rows shifted by one, so one
//     staged band of BM+2 rows serves three taps.
//   * One workgroup of eight waves per CU, persistent.  An "iteration" is one filter row ky and 64 channels:
//     the A band (33 KB) plus, per kx "phase", one weight tap (BN rows x 128 B).  A bands and weight taps
//     live in two rings of three, filled two phases ahead behind counted s_waitcnt and raw s_barriers
//     (never vmcnt(0) in the loop).  The two waves of every SIMD alternate between a load part (fragment
//     reads + staging DMAs) and an MFMA part (32 MFMAs), one barrier apart.
//   * The rings run on across tile boundaries; the epilogue stores straight from the accumulators: weight
//     rows are staged in permuted order so that the accumulator tiles a lane holds for one pixel are
//     ADJACENT channels (one 8-byte store per pixel, 16 lanes = one 128-byte line).  BatchNorm partial sums
//     go through a small LDS array that is flushed a few barriers later, from inside the next tile.
//   * Work is cut into (tile, iteration) pieces, evenly over the CUs when whole tiles would leave CUs idle
//     (BandSched below).
// LDS rows are 128 B; 16-byte chunk c of row r sits at position c ^ (r & 7), which keeps ds_read_b128
// fragment reads conflict-free at every row shift (checked exhaustively against the gfx950 lane groups).
#include "conv_args.h"

template <int BN>
struct BandCfg {
  static constexpr int BM = UH_BAND_BM;
  static constexpr int WTN = BN / 2;              // wave tile 64 rows x WTN channels; waves 4 (M) x 2 (N)
  static constexpr int NJ = WTN / 16;             // accumulator tiles along N per wave (4 or 2)
  static constexpr int A_GROUPS = 33;             // 8-row x 128-B pieces of the band: 264 >= BM + 2 rows
  static constexpr int A_THIRD = 11;              // a third of the band is staged per phase
  static constexpr int A_BYTES = A_GROUPS * 1024;
  static constexpr int B_GROUPS = BN / 8;         // pieces of one weight tap
  static constexpr int B_BYTES = BN * 128;
  static constexpr int NB = B_GROUPS / 8;         // weight pieces per wave per tap (2 or 1)
  static constexpr int B_OFF = 3 * A_BYTES;
  static constexpr int FLAG_OFF = B_OFF + 3 * B_BYTES;        // 8 waves x 64 dwords of row flags
  static constexpr int RED_OFF = FLAG_OFF + 8 * 256;          // float[4][2][BN]
  static constexpr int LDS_BYTES = RED_OFF + 4 * 2 * BN * 4;
  static constexpr int SLOT_VEC = 4 * NJ;                     // f32x4 per thread in a parked accumulator set
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// How the work of one launch is cut.  A "unit" is one BM x BN output tile (unit u: M tile u / ntn, N tile
// u % ntn) and takes niter = 3*K/64 iterations; the launch is the sequence of all (unit, iteration) pairs and
// workgroup position p takes the contiguous piece [band_start(p), band_start(p+1)).  With `aligned` the cuts
// fall on unit boundaries (whole tiles per workgroup, uneven when units % G != 0).  Otherwise the cuts are
// even in iterations: a piece that covers only part of a unit leaves its accumulators in `ws`
// (slot 2p: the piece starts inside the unit, slot 2p+1: it ends inside), and conv_band_fixup_kernel sums
// the pieces of every cut unit in position order and runs the epilogue -- a second launch instead of
// workgroups waiting for each other.
struct BandSched {
  int ntm, ntn, niter, G, aligned;
  float* ws;
};

__device__ __forceinline__ long long band_start(const BandSched& sc, int pos) {
  const long long units = (long long)sc.ntm * sc.ntn;
  return sc.aligned ? ((long long)pos * units / sc.G) * sc.niter : ((long long)pos * units * sc.niter) / sc.G;
}

// position of (unit, filter row, 64-channel block) in the iteration stream
struct BandCursor {
  int mt, nt, ky, kc;
  __device__ __forceinline__ void next(int KC, int ntn) {
    if (++kc == KC) {
      kc = 0;
      if (++ky == 3) {
        ky = 0;
        if (++nt == ntn) { nt = 0; ++mt; }
      }
    }
  }
};

__device__ __forceinline__ void band_wait(int vm) {   // vm outstanding DMAs allowed; all LDS reads done
  switch (vm) {
    case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
  }
}

// Epilogue of one unit, straight from the accumulators: lane (l15, g) of wave (wm, wn) holds rows
// wm*64 + i*16 + g*4 + e and the NJ adjacent channels wn*WTN + l15*NJ + j of the tile; fw[i] = the four row
// flags of rows i*16 + g*4 .. +3.  RAW_STATS leaves the wave's channel sums in red[wm][2][BN].
template <int BN, int MODE>
__device__ __forceinline__ void band_epilogue(const GemmArgs& p, const f32x4 (&acc)[4][BandCfg<BN>::NJ], int mt, int nt,
                                              const uint32_t (&fw)[4], int wm, int wn, int l15, int g, float* red) {
  using C = BandCfg<BN>;
  constexpr int NJ = C::NJ;
  const int ncol = wn * C::WTN + l15 * NJ;
  const int n0 = nt * BN;
  float cb[NJ], csc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) { cb[j] = 0.f; csc[j] = 1.f; }
  if (MODE == UH_EPI_AFFINE_RELU) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) { csc[j] = p.scale[n0 + ncol + j]; cb[j] = p.shift[n0 + ncol + j]; }
  } else if (p.bias) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) cb[j] = p.bias[n0 + ncol + j];
  }
  const long long mrow0 = (long long)mt * C::BM + wm * 64 + g * 4;
  char* orow = p.out + ((mrow0 * p.o_cstride) + p.o_coff + n0 + ncol) * 2;
  const long long o_row_bytes = (long long)p.o_cstride * 2;
  float s1[NJ], s2[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long m = mrow0 + i * 16 + e;
      const float f = (m < p.Mp) ? (float)((fw[i] >> (8 * e)) & 0xffu) : 0.f;
      float v[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
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
      uint32_t o[NJ / 2];
#pragma unroll
      for (int j = 0; j < NJ / 2; ++j)
        o[j] = (MODE == UH_EPI_RAW_STATS) ? pack2h(v[2 * j], v[2 * j + 1]) : pack2(v[2 * j], v[2 * j + 1]);
      char* dst = orow + (i * 16 + e) * o_row_bytes;
      if (NJ == 4) *(uint2*)dst = uint2{o[0], o[NJ / 2 - 1]};
      else *(uint32_t*)dst = o[0];
    }
  }
  if (MODE == UH_EPI_RAW_STATS) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      s1[j] += __shfl_xor(s1[j], 16, 64);
      s1[j] += __shfl_xor(s1[j], 32, 64);
      s2[j] += __shfl_xor(s2[j], 16, 64);
      s2[j] += __shfl_xor(s2[j], 32, 64);
    }
    if (g == 0) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        red[(wm * 2 + 0) * BN + ncol + j] = s1[j];
        red[(wm * 2 + 1) * BN + ncol + j] = s2[j];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // raw barriers do not wait for LDS writes
  }
}

// red[4][2][BN] -> stat_partial[mt][2][N]; every wave's red[] writes must be two barriers old
template <int BN>
__device__ __forceinline__ void band_flush_stats(const GemmArgs& p, const float* red, int mt, int nt, int tid) {
  if (tid < 2 * BN) {
    const int which = tid / BN, col = tid % BN;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += red[(w * 2 + which) * BN + col];
    p.stat_partial[((long long)mt * 2 + which) * p.N + nt * BN + col] = v;
  }
}

template <int BN, int MODE>
__global__ __launch_bounds__(512) void conv_band_kernel(GemmArgs p, BandSched sc) {
  using C = BandCfg<BN>;
  constexpr int NJ = C::NJ;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // workgroups b, b+8, ... share an XCD: give each XCD one contiguous run of positions
  const int pos = (blockIdx.x & 7) * (sc.G >> 3) + (blockIdx.x >> 3);
  const long long i0 = band_start(sc, pos), i1 = band_start(sc, pos + 1);
  if (i0 >= i1) return;
  const int total = (int)(i1 - i0);
  const int niter = sc.niter, ntn = sc.ntn;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  const int KC = p.K / 64;

  // ---- staging: one wave instruction = 8 rows x 128 B; lane -> (row lane/8, position lane%8) ---------------
  const int lrow = lane >> 3;
  const int src_chunk = (lane & 7) ^ lrow;        // pieces start at multiples of 8 rows: (row & 7) == lrow
  const long long a_row_bytes = (long long)p.a_cstride * 2;
  const long long b_row_bytes = 9LL * p.K * 2;
  const long long a_third_bytes = (long long)C::A_THIRD * 8 * a_row_bytes;
  // the band starts one row above and one pixel left of the tile: row -1 - Wp
  const char* gpA[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
    gpA[q] = p.A + ((long long)(wave + 8 * q) * 8 + lrow - 1 - p.Wp) * a_row_bytes + (long long)p.a_coff * 2 + src_chunk * 16;
  const bool hasA1 = wave + 8 < C::A_THIRD;       // 11 pieces per third: waves 0-2 stage two
  const int nA = hasA1 ? 2 : 1;
  const char* gpB[C::NB];
#pragma unroll
  for (int q = 0; q < C::NB; ++q) {
    // LDS row rl of a tap = accumulator tile j, column c of wave column w: holds output channel w*WTN + c*NJ + j
    const int rl = (wave + 8 * q) * 8 + lrow;
    const int w = rl / C::WTN, rr = rl % C::WTN;
    const int chan = w * C::WTN + (rr & 15) * NJ + (rr >> 4);
    gpB[q] = p.Wt + (long long)chan * b_row_bytes + src_chunk * 16;
  }
  auto offA = [&](const BandCursor& c) {
    return ((long long)c.mt * C::BM + (long long)c.ky * p.Wp) * a_row_bytes + c.kc * 128;
  };
  auto issueA = [&](const BandCursor& c, int third, int aslot) {
    const long long off = offA(c) + third * a_third_bytes;
    char* dst = smem + aslot * C::A_BYTES + (third * C::A_THIRD + wave) * 1024;
    __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(gpA[0] + off), (UH_LDS void*)dst, 16, 0, 0);
    if (hasA1)
      __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(gpA[1] + off), (UH_LDS void*)(dst + 8 * 1024), 16, 0, 0);
  };
  auto issueB = [&](const BandCursor& c, int kx, int bslot) {
    const long long off = (long long)c.nt * BN * b_row_bytes + ((long long)(c.ky * 3 + kx) * p.K + c.kc * 64) * 2;
#pragma unroll
    for (int q = 0; q < C::NB; ++q)
      __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(gpB[q] + off),
                                       (UH_LDS void*)(smem + C::B_OFF + bslot * C::B_BYTES + (wave + 8 * q) * 1024), 16, 0, 0);
  };
  // row flags of this wave's 64 rows: 16 dwords of the row mask (lanes 16.. repeat them)
  const long long mask_last_dword = (p.Mp - 1) >> 2;
  auto issue_flags = [&](int mt) {
    long long d = (((long long)mt * C::BM + wm * 64) >> 2) + l15;
    if (d > mask_last_dword) d = mask_last_dword;
    __builtin_amdgcn_global_load_lds((const UH_GLOBAL void*)(p.rowmask + 4 * d),
                                     (UH_LDS void*)(smem + C::FLAG_OFF + wave * 256), 4, 0, 0);
  };

  // ---- fragment read offsets ---------------------------------------------------------------------------------
  int a_off[3][2], b_off[2];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = wm * 64 + l15 + kx;
      a_off[kx][h] = r * 128 + (((g + 4 * h) ^ (r & 7)) << 4);          // + i*2048: 16 rows further, same r & 7
    }
#pragma unroll
  for (int h = 0; h < 2; ++h) b_off[h] = C::B_OFF + (wn * C::WTN + l15) * 128 + (((g + 4 * h) ^ (l15 & 7)) << 4);
  float* red = (float*)(smem + C::RED_OFF);

  // ---- main loop ---------------------------------------------------------------------------------------------
  // Phase (s, kx) of iteration s: load part = fragment reads of tap kx from A(s) and weight slot kx, then the
  // DMAs of the weight tap two phases ahead and of one third of a later A band, then one wait; barrier; 32
  // MFMAs; barrier.  Waves 4-7 (the second wave on each SIMD) run one barrier behind waves 0-3.
  //   A(s+1) is staged in phases (s-1,2), (s,0), (s,1); weight tap of phase q in phase q-2.
  //   The wait at the end of a load part leaves only the DMAs issued in that same load part outstanding, so
  //   everything a later phase reads has been retired by every wave before the barriers in between; it also
  //   drains the wave's own LDS reads, so a slot is free for DMA as soon as the readers' next barrier is passed.
  BandCursor c0;
  {
    const int u0 = (int)(i0 / niter);
    const int it0 = (int)(i0 - (long long)u0 * niter);
    c0.mt = u0 / ntn;
    c0.nt = u0 - c0.mt * ntn;
    c0.ky = it0 / KC;
    c0.kc = it0 - c0.ky * KC;
  }
  BandCursor c1 = c0;
  c1.next(KC, ntn);
  BandCursor c2 = c1;
  c2.next(KC, ntn);
  issueA(c0, 0, 0);
  issueA(c0, 1, 0);
  issueA(c0, 2, 0);
  issueB(c0, 0, 0);
  issueB(c0, 1, 1);
  if (total > 1) {
    issueA(c1, 0, 1);
    band_wait(C::NB + nA);
  } else {
    band_wait(C::NB);
  }
  __builtin_amdgcn_s_barrier();
  if (wave >= 4) __builtin_amdgcn_s_barrier();  // the stagger

  int a_cur = 0, s = 0;
  int pend_mt = -1, pend_nt = 0;                // unit whose BatchNorm sums still sit in red[]
  long long i = i0;
  bf16x8 af[2][4], bfr[2][NJ];
  while (i < i1) {
    const int u = (int)(i / niter);
    const int it_a = (int)(i - (long long)u * niter);
    const int cnt = min(niter - it_a, (int)(i1 - i));
    const bool full = (it_a == 0 && cnt == niter);
    const int mt = u / ntn, nt = u - mt * ntn;
    f32x4 acc[4][NJ];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[ii][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < cnt; ++c, ++s) {
      const int a1 = (a_cur == 2) ? 0 : a_cur + 1;
      const int a2 = (a_cur == 0) ? 2 : a_cur - 1;
      const char* SA = smem + a_cur * C::A_BYTES;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const char* SB = smem + kx * C::B_BYTES;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) bfr[h][j] = *(const bf16x8*)(SB + b_off[h] + j * 2048);
#pragma unroll
          for (int ii = 0; ii < 4; ++ii) af[h][ii] = *(const bf16x8*)(SA + a_off[kx][h] + ii * 2048);
        }
        int issued = 0;
        if (kx == 0) {
          if (c == 0 && full) issue_flags(mt);
          issueB(c0, 2, 2);                              // two phases ahead: tap 2 of this iteration
          issued = C::NB;
          if (s + 1 < total) { issueA(c1, 1, a1); issued += nA; }
        } else if (kx == 1) {
          if (MODE == UH_EPI_RAW_STATS && c == 0 && pend_mt >= 0) {
            band_flush_stats<BN>(p, red, pend_mt, pend_nt, tid);
            pend_mt = -1;
          }
          if (s + 1 < total) {
            issueB(c1, 0, 0);
            issueA(c1, 2, a1);
            issued = C::NB + nA;
          }
        } else {
          if (s + 1 < total) { issueB(c1, 1, 1); issued = C::NB; }
          if (s + 2 < total) { issueA(c2, 0, a2); issued += nA; }
        }
        band_wait(issued);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[h][ii], bfr[h][j], acc[ii][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }
      a_cur = a1;
      c0 = c1;
      c1 = c2;
      c2.next(KC, ntn);
    }
    i += cnt;

    if (full) {
      uint32_t fw[4];
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) fw[ii] = *(const uint32_t*)(smem + C::FLAG_OFF + wave * 256 + (ii * 4 + g) * 4);
      band_epilogue<BN, MODE>(p, acc, mt, nt, fw, wm, wn, l15, g, red);
      if (MODE == UH_EPI_RAW_STATS) { pend_mt = mt; pend_nt = nt; }
    } else {
      // a piece of a unit: park the accumulators for conv_band_fixup_kernel
      f32x4* slot = (f32x4*)sc.ws + ((long long)(2 * pos + (it_a > 0 ? 0 : 1)) * C::SLOT_VEC) * 512 + tid;
#pragma unroll
      for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int j = 0; j < NJ; ++j) slot[(ii * NJ + j) * 512] = acc[ii][j];
    }
  }
  if (wave < 4) __builtin_amdgcn_s_barrier();   // match the second half's extra barrier
  if (MODE == UH_EPI_RAW_STATS && pend_mt >= 0) {
    __syncthreads();
    band_flush_stats<BN>(p, red, pend_mt, pend_nt, tid);
  }
}

// One workgroup per cut: boundary k (between positions k-1 and k) that falls inside a unit and is the first
// such boundary of that unit sums the unit's pieces in position order and finishes the unit.
template <int BN, int MODE>
__global__ __launch_bounds__(512) void conv_band_fixup_kernel(GemmArgs p, BandSched sc) {
  using C = BandCfg<BN>;
  constexpr int NJ = C::NJ;
  __shared__ __attribute__((aligned(16))) float red[4 * 2 * BN];
  const int k = blockIdx.x + 1;
  const long long sk = band_start(sc, k);
  const int niter = sc.niter;
  if (sk % niter == 0) return;
  const long long u = sk / niter, ub = u * niter, ue = ub + niter;
  if (k >= 2 && band_start(sc, k - 1) > ub) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  f32x4 acc[4][NJ];
#pragma unroll
  for (int ii = 0; ii < 4; ++ii)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[ii][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int pp = k - 1; pp < sc.G; ++pp) {
    const long long ps = band_start(sc, pp), pe = band_start(sc, pp + 1);
    if (ps >= ue) break;
    if (pe <= ps || pe <= ub) continue;
    const int inside = (ps > ub) ? 0 : 1;        // slot 2pp: piece starts inside the unit; 2pp+1: it ends inside
    const f32x4* slot = (const f32x4*)sc.ws + ((long long)(2 * pp + inside) * C::SLOT_VEC) * 512 + tid;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[ii][j] += slot[(ii * NJ + j) * 512];
  }
  const int mt = (int)(u / sc.ntn), nt = (int)(u % sc.ntn);
  uint32_t fw[4];
  const long long mask_last_dword = (p.Mp - 1) >> 2;
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) {
    long long d = (((long long)mt * C::BM + wm * 64) >> 2) + ii * 4 + g;
    if (d > mask_last_dword) d = mask_last_dword;
    fw[ii] = *(const uint32_t*)(p.rowmask + 4 * d);
  }
  band_epilogue<BN, MODE>(p, acc, mt, nt, fw, wm, wn, l15, g, red);
  if (MODE == UH_EPI_RAW_STATS) {
    __syncthreads();
    band_flush_stats<BN>(p, red, mt, nt, tid);
  }
}

static int band_workgroups(long long units, int niter) {
  const long long total = units * niter;
  long long g = (total / 2) & ~7LL;               // at least ~2 iterations (six phases) per workgroup
  if (g > 256) g = 256;                           // one workgroup per CU
  if (g < 8) g = 8;
  return (int)g;
}

static int band_bn(int N) { return (N % 128 == 0) ? 128 : 64; }

size_t uh_band_workspace_bytes(long long Mp, int N, int K) {
  const int bn = band_bn(N);
  const long long units = ((Mp + UH_BAND_BM - 1) / UH_BAND_BM) * (N / bn);
  return (size_t)2 * band_workgroups(units, 3 * (K / 64)) * UH_BAND_BM * bn * sizeof(float);
}

template <int BN, int MODE>
static void launch_band_mode(const GemmArgs& p, const BandSched& sc, hipStream_t st) {
  using C = BandCfg<BN>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_band_kernel<BN, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    attr = true;
  }
  hipLaunchKernelGGL((conv_band_kernel<BN, MODE>), dim3(sc.G), dim3(512), C::LDS_BYTES, st, p, sc);
  if (!sc.aligned)
    hipLaunchKernelGGL((conv_band_fixup_kernel<BN, MODE>), dim3(sc.G - 1), dim3(512), 0, st, p, sc);
}

template <int BN>
static void launch_band_bn(const GemmArgs& p, void* ws, size_t ws_bytes, hipStream_t st) {
  BandSched sc;
  sc.ntm = (int)((p.Mp + UH_BAND_BM - 1) / UH_BAND_BM);
  sc.ntn = p.N / BN;
  sc.niter = 3 * (p.K / 64);
  const long long units = (long long)sc.ntm * sc.ntn;
  sc.G = band_workgroups(units, sc.niter);
  sc.ws = (float*)ws;
  // even cuts need the workspace and cost a second launch plus ~2 x 128 KB of traffic per workgroup: use them
  // when whole tiles would leave more than ~15 % of the CU-time idle in the last round
  const long long rounds = (units + sc.G - 1) / sc.G;
  const bool wasteful = rounds * sc.G * 100 > units * 115;
  sc.aligned = (ws == nullptr || ws_bytes < uh_band_workspace_bytes(p.Mp, p.N, p.K) || !wasteful) ? 1 : 0;
  if (p.mode == UH_EPI_RAW_STATS) launch_band_mode<BN, UH_EPI_RAW_STATS>(p, sc, st);
  else if (p.mode == UH_EPI_AFFINE_RELU) launch_band_mode<BN, UH_EPI_AFFINE_RELU>(p, sc, st);
  else launch_band_mode<BN, UH_EPI_MASKED>(p, sc, st);
}

void uh_launch_band(const GemmArgs& p, void* ws, size_t ws_bytes, hipStream_t st) {
  if (band_bn(p.N) == 128) launch_band_bn<128>(p, ws, ws_bytes, st);
  else launch_band_bn<64>(p, ws, ws_bytes, st);
}
