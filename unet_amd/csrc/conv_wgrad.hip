// Weight gradient of the implicit-GEMM convolution on gfx950 matrix cores.
//
//   dW[n][t][k] = sum_m dY[m][n] * A[m + off(t)][k]
//
// The reduction runs over the rows (pixels) of two padded-linear NHWC tensors, so BOTH MFMA
// operands must be read "down the columns" of their LDS tiles.  The tiles are staged as they
// sit in memory ([pixel][channel], 128 B per pixel row, global_load_lds) and the fragments are
// read with ds_read_b64_tr_b16, the gfx950 transposing LDS read: each 16-lane group fetches a
// 4-pixel x 16-channel block and every lane receives the 4 pixels of ONE channel.
// One workgroup owns a 64(n) x 64(k) x T block of dW and a contiguous range of pixels; per
// 128-pixel stage it keeps dY[128][64] and, for T = 9, three row bands of A (dy = -1,0,+1,
// 136 pixel rows each) so that all nine taps are served from LDS.  Each wave owns one 16-wide
// k slice: 4 (n) x T accumulator tiles.  Partial results of the pixel splits are written to a
// workspace slab and summed by uh_wgrad_reduce (deterministic, no atomics).
#include "common.h"
#include "reduce_small.h"
#include <cstdlib>
// The LDS-DMA statements below write M0 and name it as clobbered, so that the compiler's own M0 set-up (it merges
// identical M0 initialisations when nothing between them modifies the register) never assumes a value across them.
// M0 is a reserved register, for which clang warns on every clobber list; the declaration is intended.
#pragma clang diagnostic ignored "-Winline-asm"

struct WgradArgs {
  const char* dY;
  const char* A;
  float* partial;
  long long Mp;
  long long rows_per_split;  // multiple of 128
  int dy_cstride, dy_coff, a_cstride, a_coff;
  int N, K, Wp;
  int S;        // pixel splits
  int remap;    // 1: 1-D grid, every split's tiles on one XCD (see the kernel)
  // conv_wgrad_col_kernel only: rows_per_split counts stages, a stage index i is (strip i / total_rows, row i % total_rows)
  int total_rows;   // B * (H + 1)
  int nstrip;       // W / 64
};

template <int T>
struct WCfg {
  static constexpr int BANDS = (T == 9) ? 3 : 1;
  static constexpr int BAND_GROUPS = (T == 9) ? 17 : 16;  // 8-row groups per band (T=9: 130 rows needed)
  static constexpr int DY_BYTES = 128 * 128;
  static constexpr int BAND_BYTES = BAND_GROUPS * 1024;
  static constexpr int LDS_BYTES = DY_BYTES + BANDS * BAND_BYTES;
  static constexpr int NGROUPS = 16 + BANDS * BAND_GROUPS;
};

// MERGED (T = 9, narrow images): the three A bands of a stage are the same rows one image row apart, so where
// 130 + 2 Wp rows fit the three band slots ONE image of that many rows is staged and filter row ky reads it ky * Wp rows
// further down: at W = 32 a stage is 41 KB instead of 67 (cycle stamps had the 17 DMA instructions per wave of a stage
// take as long to issue as its 144 MFMAs).
template <int T, bool MERGED>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs p) {
  using C = WCfg<T>;
  static_assert(!MERGED || T == 9, "one tap has one band");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // All (n, k) tiles of one pixel split read the same rows of dY and A.  Workgroups are dealt to the 8 XCDs
  // round-robin by their linear id, so with a 1-D grid decoded as id -> (xcd = id % 8, q = id / 8),
  // split = xcd + 8 * (q / tiles), tile = q % tiles, the tiles of a split run on ONE XCD, next to each other
  // in time, and those rows are fetched into one L2 once instead of into all eight.
  int n0, k0, split;
  if (p.remap) {
    const int tn = p.N / 64, tiles = tn * (p.K / 64);
    const int q = blockIdx.x >> 3;
    split = (blockIdx.x & 7) + 8 * (q / tiles);
    if (split >= p.S) return;
    const int t = q % tiles;
    n0 = (t % tn) * 64;
    k0 = (t / tn) * 64;
  } else {
    n0 = blockIdx.x * 64;
    k0 = blockIdx.y * 64;
    split = blockIdx.z;
  }
  const long long ms = (long long)split * p.rows_per_split;
  long long me = ms + p.rows_per_split;
  if (me > ((p.Mp + 127) / 128) * 128) me = ((p.Mp + 127) / 128) * 128;

  // staging: LDS row r keeps source chunk c at position c ^ (((r>>1)&3)<<1); an 8-row group
  // always starts at a multiple of 8, so the swizzle depends on the lane only.
  // A DMA's source is a wave-uniform base (scalar registers, scalar arithmetic) plus this lane's fixed 32-bit offset
  // inside a group of eight rows.
  const int lrow = lane >> 3;
  const int src_chunk = (lane & 7) ^ (((lane >> 4) & 3) << 1);
  const long long dy_row_bytes = (long long)p.dy_cstride * 2, a_row_bytes = (long long)p.a_cstride * 2;
  const uint32_t dy_voff = (uint32_t)(lrow * (int)dy_row_bytes + src_chunk * 16);
  const uint32_t a_voff = (uint32_t)(lrow * (int)a_row_bytes + src_chunk * 16);
  const char* dy_base = p.dY + (long long)(p.dy_coff + n0) * 2;
  const char* a_base = p.A + (long long)(p.a_coff + k0) * 2;

  // transposed-read lane geometry: group g = lane/16 reads pixels 4g+q (first read) and 16+4g+q
  // (second read); q = (lane/4)&3 is the block row this lane addresses, pc = lane&3 the 4-channel column.
  const int g = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3;
  const int px = 4 * g + q;                       // pixel of the first read within a 32-pixel k-step

  f32x4 acc[T][4];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (long long mb = ms; mb < me; mb += 128) {
    __syncthreads();  // previous stage fully consumed
    // ---- stage dY[128 rows] and the A bands -------------------------------------------------
    const int ngroups = MERGED ? 16 + (130 + 2 * p.Wp + 7) / 8 : C::NGROUPS;
    for (int gi = wave; gi < ngroups; gi += 4) {
      const char* base;
      uint32_t voff;
      if (gi < 16) {
        base = dy_base + (mb + gi * 8) * dy_row_bytes;
        voff = dy_voff;
      } else if (MERGED) {
        base = a_base + (mb - p.Wp - 1 + (gi - 16) * 8) * a_row_bytes;
        voff = a_voff;
      } else {
        const int band = (gi - 16) / C::BAND_GROUPS, bg = (gi - 16) % C::BAND_GROUPS;
        const long long r0 = (T == 9) ? (mb + (long long)(band - 1) * p.Wp - 1) : mb;
        base = a_base + (r0 + bg * 8) * a_row_bytes;
        voff = a_voff;
      }
      const uint32_t lds_addr = (uint32_t)(uintptr_t)(UH_LDS char*)(smem + gi * 1024);
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory", "m0");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      // A-operand fragments: dY^T, 4 n-subtiles of 16 channels
      bf16x8 af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bf16x4 lo, hi;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = ks * 32 + px + 16 * h;
          const int chunk = (2 * i + (pc >> 1)) ^ (((r >> 1) & 3) << 1);
          const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (UH_LDS bf16x4*)(smem + r * 128 + chunk * 16 + (pc & 1) * 8));
          if (h == 0) lo = v; else hi = v;
        }
        af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int band = (T == 9) ? t / 3 : 0;
        const int kx = (T == 9) ? t % 3 : 0;
        bf16x4 lo, hi;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          // row inside the band (MERGED: inside the one image, filter row `band` sits band * Wp rows further down)
          const int r = ks * 32 + px + 16 * h + kx + (MERGED ? band * p.Wp : 0);
          const int chunk = (2 * wave + (pc >> 1)) ^ (((r >> 1) & 3) << 1);
          const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (UH_LDS bf16x4*)(smem + C::DY_BYTES + (MERGED ? 0 : band * C::BAND_BYTES) + r * 128 + chunk * 16 +
                               (pc & 1) * 8));
          if (h == 0) lo = v; else hi = v;
        }
        const bf16x8 bfr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[t][i], 0, 0, 0);
      }
    }
  }

  // ---- store the partial block: partial[split][n][t][k] -----------------------------------
  const int l15 = lane & 15;
  float* base = p.partial + (long long)split * p.N * T * p.K;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + i * 16 + g * 4 + e;
        const int k = k0 + wave * 16 + l15;
        base[((long long)n * T + t) * p.K + k] = acc[t][i][e];
      }
}

// ---- T = 1, N and K multiples of 128: a 128 x 128 block per workgroup ---------------------------------------------------
// With one tap a 64 x 64 block stages 32 KB per 128 pixels for 16 MFMAs per wave: cycle stamps put the issue of its eight
// DMA instructions per wave at 1500-1700 cycles a stage against 650 for the arithmetic, and every staged byte is fetched
// from L2 once per block that needs it.  A 128 x 128 block stages 64 KB for four times the MFMAs: half the DMA
// instructions and half the L2 reads per MFMA.  LDS holds four images of 128 pixels x 64 channels (dY low / high half,
// A low / high half), each laid out and read like the 64-channel images of conv_wgrad_kernel; wave w owns k columns
// 32 w .. 32 w + 31 of the block and all 128 n rows (acc: 8 x 2 tiles).
struct WWideCfg {
  static constexpr int IMG_BYTES = 128 * 128;      // one image: 128 pixels x 64 channels
  static constexpr int LDS_BYTES = 4 * IMG_BYTES;
  static constexpr int NGROUPS = 64;               // 16 eight-row groups per image
};

__global__ __launch_bounds__(256, 2) void conv_wgrad1_wide_kernel(WgradArgs p) {
  using C = WWideCfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int n0, k0, split;                 // as in conv_wgrad_kernel, with 128 x 128 blocks
  if (p.remap) {
    const int tn = p.N / 128, tiles = tn * (p.K / 128);
    const int q = blockIdx.x >> 3;
    split = (blockIdx.x & 7) + 8 * (q / tiles);
    if (split >= p.S) return;
    const int t = q % tiles;
    n0 = (t % tn) * 128;
    k0 = (t / tn) * 128;
  } else {
    n0 = blockIdx.x * 128;
    k0 = blockIdx.y * 128;
    split = blockIdx.z;
  }
  const long long ms = (long long)split * p.rows_per_split;
  long long me = ms + p.rows_per_split;
  if (me > ((p.Mp + 127) / 128) * 128) me = ((p.Mp + 127) / 128) * 128;

  const int lrow = lane >> 3;
  const int src_chunk = (lane & 7) ^ (((lane >> 4) & 3) << 1);
  const long long dy_row_bytes = (long long)p.dy_cstride * 2, a_row_bytes = (long long)p.a_cstride * 2;
  const uint32_t dy_voff = (uint32_t)(lrow * (int)dy_row_bytes + src_chunk * 16);
  const uint32_t a_voff = (uint32_t)(lrow * (int)a_row_bytes + src_chunk * 16);
  const char* dy_base = p.dY + (long long)(p.dy_coff + n0) * 2;
  const char* a_base = p.A + (long long)(p.a_coff + k0) * 2;
  const int g = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3;
  const int px = 4 * g + q;

  f32x4 acc[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) acc[i][kk] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (long long mb = ms; mb < me; mb += 128) {
    __syncthreads();  // previous stage fully consumed
    for (int gi = wave; gi < C::NGROUPS; gi += 4) {
      const int img = gi >> 4, gr = gi & 15;       // image 0, 1: dY channels 0-63, 64-127; 2, 3: A likewise
      const char* base = (img < 2 ? dy_base + (mb + gr * 8) * dy_row_bytes : a_base + (mb + gr * 8) * a_row_bytes) +
                         (img & 1) * 128;
      const uint32_t voff = img < 2 ? dy_voff : a_voff;
      const uint32_t lds_addr = (uint32_t)(uintptr_t)(UH_LDS char*)(smem + gi * 1024);
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory", "m0");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[8];                                // dY^T: 8 n-subtiles of 16 channels
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        bf16x4 lo, hi;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = ks * 32 + px + 16 * h;
          const int chunk = (2 * (i & 3) + (pc >> 1)) ^ (((r >> 1) & 3) << 1);
          const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (UH_LDS bf16x4*)(smem + (i >> 2) * C::IMG_BYTES + r * 128 + chunk * 16 + (pc & 1) * 8));
          if (h == 0) lo = v; else hi = v;
        }
        af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int k16 = wave * 2 + kk;             // 16-column subtile of the block's 128 k columns
        bf16x4 lo, hi;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = ks * 32 + px + 16 * h;
          const int chunk = (2 * (k16 & 3) + (pc >> 1)) ^ (((r >> 1) & 3) << 1);
          const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (UH_LDS bf16x4*)(smem + (2 + (k16 >> 2)) * C::IMG_BYTES + r * 128 + chunk * 16 + (pc & 1) * 8));
          if (h == 0) lo = v; else hi = v;
        }
        const bf16x8 bfr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          acc[i][kk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[i][kk], 0, 0, 0);
      }
    }
  }

  // partial[split][n][k]
  const int l15 = lane & 15;
  float* base = p.partial + (long long)split * p.N * p.K;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + i * 16 + g * 4 + e;
        const int k = k0 + (wave * 2 + kk) * 16 + l15;
        base[(long long)n * p.K + k] = acc[i][kk][e];
      }
}

// ---- T = 9, W a multiple of 64: column walk with a rotating band ring and one stage of prefetch ------------------------
// A split walks DOWN a 64-pixel-wide column strip, one image row per stage (the pad column, whose dY is zero, is
// never visited).  The A bands of filter rows 0, 1, 2 of a stage are then the bands of rows 1, 2 of the previous stage
// and ONE new band: four LDS slots rotate, three in use and one being filled, and dY has two buffers, so the loads of
// stage i+1 (8 KB of dY + 9 KB of A instead of 16 + 51 per 128 pixels) are in flight while stage i is consumed.  One
// barrier per stage: behind it every wave has finished reading stage i and all of stage i+1 has landed.
// A stage index i is (strip i / total_rows, row i % total_rows); where a split passes from one strip to the next the
// three bands are fetched afresh, without prefetch.
struct WColCfg {
  static constexpr int DY_BYTES = 64 * 128;        // one dY buffer: 64 pixels x 64 channels
  static constexpr int BAND_GROUPS = 9;            // 72 >= 66 rows
  static constexpr int BAND_BYTES = BAND_GROUPS * 1024;
  static constexpr int BAND_OFF = 2 * DY_BYTES;
  static constexpr int LDS_BYTES = BAND_OFF + 4 * BAND_BYTES;
};

__global__ __launch_bounds__(256, 2) void conv_wgrad_col_kernel(WgradArgs p) {
  using C = WColCfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int n0, k0, split;                 // as in conv_wgrad_kernel
  if (p.remap) {
    const int tn = p.N / 64, tiles = tn * (p.K / 64);
    const int q = blockIdx.x >> 3;
    split = (blockIdx.x & 7) + 8 * (q / tiles);
    if (split >= p.S) return;
    const int t = q % tiles;
    n0 = (t % tn) * 64;
    k0 = (t / tn) * 64;
  } else {
    n0 = blockIdx.x * 64;
    k0 = blockIdx.y * 64;
    split = blockIdx.z;
  }
  const long long total = (long long)p.nstrip * p.total_rows;
  const long long ms = (long long)split * p.rows_per_split;
  long long me = ms + p.rows_per_split;
  if (me > total) me = total;

  // A DMA's source is a wave-uniform base (scalar registers, scalar arithmetic) plus this lane's fixed 32-bit offset
  // inside a group of eight rows: two VGPRs for addressing in all, and no vector instructions per DMA.
  const int lrow = lane >> 3;
  const int src_chunk = (lane & 7) ^ (((lane >> 4) & 3) << 1);
  const int dy_rb = p.dy_cstride * 2, a_rb = p.a_cstride * 2;
  const uint32_t dy_voff = (uint32_t)(lrow * dy_rb + src_chunk * 16);
  const uint32_t a_voff = (uint32_t)(lrow * a_rb + src_chunk * 16);
  const char* dy_base = p.dY + (long long)(p.dy_coff + n0) * 2;
  const char* a_base = p.A + (long long)(p.a_coff + k0) * 2;
  const int g = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3;
  const int px = 4 * g + q;

  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Written as assembly on purpose: behind the builtin the compiler puts a full s_waitcnt vmcnt(0) in front of
  // every transposing LDS read that follows (it cannot tell which slot the DMA fills), which would undo the
  // prefetch; the waits for these loads are the explicit ones at the end of each stage.
  auto dma = [&](const char* base, uint32_t voff, int dst) {
    const uint32_t lds_addr = (uint32_t)(uintptr_t)(UH_LDS char*)(smem + dst);
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory", "m0");
  };
  // dY of the stage at pixel mb into buffer db; bands first .. 2 of it into their slots (rot + d) & 3
  auto issue = [&](long long mb, int db, int rot, int first) {
    const int ngroups = 8 + (3 - first) * C::BAND_GROUPS;
    for (int gi = wave; gi < ngroups; gi += 4) {
      if (gi < 8) {
        dma(dy_base + (mb + gi * 8) * dy_rb, dy_voff, db * C::DY_BYTES + gi * 1024);
      } else {
        const int bi = (gi - 8) / C::BAND_GROUPS, bg = (gi - 8) % C::BAND_GROUPS;
        const int d = first + bi;
        dma(a_base + (mb + (long long)(d - 1) * p.Wp - 1 + bg * 8) * a_rb, a_voff,
            C::BAND_OFF + ((rot + d) & 3) * C::BAND_BYTES + bg * 1024);
      }
    }
  };
  // The common case, the next image row of the same strip (dY + the one new band, into slot (rot + 2) & 3 counted from
  // the rotation the ring will then have), with nothing but scalar additions per stage: a wave's two dY groups
  // (wave, wave + 4) and two or three band groups (wave, wave + 4, and 8 for wave 0) sit at fixed offsets from the
  // stage's first pixel, and the bases of that pixel advance by one image row.  (Measured with cycle stamps: the
  // general form above, with its 64-bit products and the division that finds a stage's strip, took as long per
  // stage as the stage's 72 MFMAs.)
  const int dy_g0 = wave * 8 * dy_rb, dy_g1 = (wave + 4) * 8 * dy_rb;
  const int a_g0 = wave * 8 * a_rb, a_g1 = (wave + 4) * 8 * a_rb, a_g2 = 8 * 8 * a_rb;
  const long long dy_step = (long long)p.Wp * dy_rb, a_step = (long long)p.Wp * a_rb;
  const char* a_new = a_base + (long long)(p.Wp - 1) * a_rb;       // band 2 of a stage starts Wp - 1 pixels behind it
  auto issue_next_row = [&](const char* dyp, const char* ap, int db_n, int rot_n) {
    const int dyb = db_n * C::DY_BYTES + wave * 1024;
    const int bnd = C::BAND_OFF + ((rot_n + 2) & 3) * C::BAND_BYTES + wave * 1024;
    dma(dyp + dy_g0, dy_voff, dyb);
    dma(dyp + dy_g1, dy_voff, dyb + 4096);
    dma(ap + a_g0, a_voff, bnd);
    dma(ap + a_g1, a_voff, bnd + 4096);
    if (wave == 0) dma(ap + a_g2, a_voff, bnd + 8192);
  };

  if (ms < me) {
    int rot = 0, db = 0;
    int strip = (int)(ms / p.total_rows);
    int row = (int)(ms - (long long)strip * p.total_rows);
    long long mb = (long long)row * p.Wp + strip * 64;
    const char* dyp = dy_base + mb * dy_rb;                         // this stage's first pixel (wave-uniform)
    const char* ap = a_new + mb * a_rb;
    issue(mb, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (long long it = ms; it < me; ++it) {
      const bool more = it + 1 < me;
      const bool follows = more && row + 1 < p.total_rows;   // the next stage is the next image row of this strip
      if (follows) {
        dyp += dy_step;
        ap += a_step;
        issue_next_row(dyp, ap, db ^ 1, (rot + 1) & 3);
      }
      const char* dyb = smem + db * C::DY_BYTES;
      int boff[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) boff[d] = C::BAND_OFF + ((rot + d) & 3) * C::BAND_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bf16x4 lo, hi;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int r = ks * 32 + px + 16 * h;
            const int chunk = (2 * i + (pc >> 1)) ^ (((r >> 1) & 3) << 1);
            const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (UH_LDS bf16x4*)(dyb + r * 128 + chunk * 16 + (pc & 1) * 8));
            if (h == 0) lo = v; else hi = v;
          }
          af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int band = t / 3, kx = t % 3;
          bf16x4 lo, hi;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int r = ks * 32 + px + 16 * h + kx;
            const int chunk = (2 * wave + (pc >> 1)) ^ (((r >> 1) & 3) << 1);
            const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (UH_LDS bf16x4*)(smem + boff[band] + r * 128 + chunk * 16 + (pc & 1) * 8));
            if (h == 0) lo = v; else hi = v;
          }
          const bf16x8 bfr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[t][i], 0, 0, 0);
        }
      }
      if (follows) {
        rot = (rot + 1) & 3;
        ++row;
      } else if (more) {                                     // a new strip: nothing of this stage is reused
        __syncthreads();
        rot = 0;
        row = 0;
        ++strip;
        mb = (long long)strip * 64;
        dyp = dy_base + mb * dy_rb;
        ap = a_new + mb * a_rb;
        issue(mb, db ^ 1, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      db ^= 1;
    }
  }

  const int l15 = lane & 15;
  float* base = p.partial + (long long)split * p.N * 9 * p.K;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + i * 16 + g * 4 + e;
        const int k = k0 + wave * 16 + l15;
        base[((long long)n * 9 + t) * p.K + k] = acc[t][i][e];
      }
}

// T = 1 launches whose N and K are multiples of 128 take 128 x 128 blocks if they are large (UNET_AMD_WGRAD1_WIDE=0:
// never).  ~512 workgroups write 33 MB of partial blocks instead of 8, which the reduction then reads: measured, the
// U-Net's four transposed convolutions (35-39 GFLOP each) go from 395 to 261 us and its step from 13.66 to 13.55 ms,
// while LUNeXt's 1x1 and 2x2 layers (<= 10 GFLOP) lose more in the reduction than they gain (step 5.61 -> 5.69 ms).
static bool wgrad1_wide(long long Mp, int N, int K, int T) {
  static const bool on = [] { const char* e = getenv("UNET_AMD_WGRAD1_WIDE"); return !e || atoi(e) != 0; }();
  return on && T == 1 && N % 128 == 0 && K % 128 == 0 && 2.0 * (double)Mp * N * K >= 2e10;
}

static int wgrad_splits(long long Mp, int N, int K, int T) {
  const long long stages = (Mp + 127) / 128;
  const int edge = wgrad1_wide(Mp, N, K, T) ? 128 : 64;
  const long long tiles = (long long)(N / edge) * (K / edge);
  static const int target = [] { const char* e = getenv("UNET_AMD_WGRAD_WGS"); return e ? atoi(e) : 512; }();
  long long want = (target + tiles - 1) / tiles;     // aim at ~512 workgroups (two per CU): every split costs a slab
  if (want > stages) want = stages;
  if (want < 1) want = 1;
  if (want > 512) want = 512;
  return (int)want;
}

extern "C" int uh_conv_wgrad_splits(int B, int H, int W, int N, int K, int T) {
  return wgrad_splits(uh_mp(B, H, W), N, K, T);
}

extern "C" size_t uh_conv_wgrad_workspace(int B, int H, int W, int N, int K, int T) {
  return (size_t)wgrad_splits(uh_mp(B, H, W), N, K, T) * N * T * K * sizeof(float);
}

extern "C" int uh_conv_wgrad(int B, int H, int W, const void* dy, int dy_cstride, int dy_coff, int N,
                             const void* a, int a_cstride, int a_coff, int K, int T, float* partial,
                             uh_stream s) {
  UH_REQUIRE(B > 0 && H > 0 && W > 0, "conv_wgrad: bad geometry");
  UH_REQUIRE(N > 0 && N % 64 == 0 && K > 0 && K % 64 == 0, "conv_wgrad: N=%d K=%d must be multiples of 64", N, K);
  UH_REQUIRE(T == 1 || T == 9, "conv_wgrad: T=%d must be 1 or 9", T);
  UH_REQUIRE(dy && a && partial, "conv_wgrad: null pointer");
  UH_REQUIRE(dy_cstride >= dy_coff + N && a_cstride >= a_coff + K && dy_cstride % 8 == 0 && a_cstride % 8 == 0 &&
                 dy_coff % 8 == 0 && a_coff % 8 == 0, "conv_wgrad: bad views");
  WgradArgs p;
  p.dY = (const char*)dy;
  p.A = (const char*)a;
  p.partial = partial;
  p.Mp = uh_mp(B, H, W);
  const int S = wgrad_splits(p.Mp, N, K, T);
  const long long stages = (p.Mp + 127) / 128;
  p.rows_per_split = ((stages + S - 1) / S) * 128;
  // UNET_AMD_WGRAD_COLWALK=0: the linear walk of conv_wgrad_kernel everywhere (default: conv_wgrad_col_kernel wherever
  // W is a multiple of 64)
  static const bool colwalk_on = [] { const char* e = getenv("UNET_AMD_WGRAD_COLWALK"); return !e || atoi(e) != 0; }();
  const bool col64 = T == 9 && W % 64 == 0 && colwalk_on;
  p.total_rows = B * (H + 1);
  p.nstrip = W / 64;
  if (col64) {
    const long long cstages = (long long)p.nstrip * p.total_rows;     // >= Mp / 128 >= S: no split is empty
    p.rows_per_split = (cstages + S - 1) / S;
  }
  p.dy_cstride = dy_cstride;
  p.dy_coff = dy_coff;
  p.a_cstride = a_cstride;
  p.a_coff = a_coff;
  p.N = N;
  p.K = K;
  p.Wp = W + 1;
  p.S = S;
  p.remap = (S >= 8) ? 1 : 0;
  const int edge = wgrad1_wide(p.Mp, N, K, T) ? 128 : 64;
  dim3 grid(N / edge, K / edge, S);
  if (p.remap) grid = dim3((unsigned)(8 * ((S + 7) / 8) * (N / edge) * (K / edge)), 1, 1);
  hipStream_t st = (hipStream_t)s;
  if (edge == 128) {
    uh_set_max_lds((const void*)conv_wgrad1_wide_kernel, WWideCfg::LDS_BYTES);
    hipLaunchKernelGGL(conv_wgrad1_wide_kernel, grid, dim3(256), WWideCfg::LDS_BYTES, st, p);
  } else if (col64) {
    hipLaunchKernelGGL(conv_wgrad_col_kernel, grid, dim3(256), WColCfg::LDS_BYTES, st, p);
  } else if (T == 9) {
    static const bool merge_on = [] { const char* e = getenv("UNET_AMD_WGRAD_MERGE"); return !e || atoi(e) != 0; }();
    if (merge_on && 130 + 2 * p.Wp <= 3 * WCfg<9>::BAND_GROUPS * 8) {
      uh_set_max_lds((const void*)conv_wgrad_kernel<9, true>, WCfg<9>::LDS_BYTES);
      hipLaunchKernelGGL((conv_wgrad_kernel<9, true>), grid, dim3(256), WCfg<9>::LDS_BYTES, st, p);
    } else {
      uh_set_max_lds((const void*)conv_wgrad_kernel<9, false>, WCfg<9>::LDS_BYTES);
      hipLaunchKernelGGL((conv_wgrad_kernel<9, false>), grid, dim3(256), WCfg<9>::LDS_BYTES, st, p);
    }
  } else {
    hipLaunchKernelGGL((conv_wgrad_kernel<1, false>), grid, dim3(256), WCfg<1>::LDS_BYTES, st, p);
  }
  return uh_check_launch("conv_wgrad");
}

// grad[n*sn + t*st + k*sk] (+)= sum_s partial[s*split_stride + n*on + t*ot + k]
// A block owns 256/SL consecutive source elements; SL threads share each element and take every SL-th split
// (many splits of a small tensor would otherwise be one long serial loop per thread).
template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int S, long long split_stride,
                                                           int N, int T, int K, long long on, long long ot, long long sn,
                                                           long long st, long long sk, float* __restrict__ grad,
                                                           int accumulate) {
  __shared__ float red[256];
  wgrad_reduce_body<SL>(blockIdx.x, red, partial, S, split_stride, N, T, K, on, ot, sn, st, sk, grad, accumulate);
}

// (body in reduce_small.h)
__global__ __launch_bounds__(256) void wgrad_reduce_convt_kernel(const float* __restrict__ partial, int S,
                                                                 long long split_stride, int Cout, int K,
                                                                 float* __restrict__ grad, int accumulate) {
  __shared__ float red[UH_WGRAD_REDUCE_CONVT_FLOATS];
  wgrad_reduce_convt_body(blockIdx.x, blockIdx.y, red, partial, S, split_stride, Cout, K, grad, accumulate);
}

// The common case, partial [n][9][k] -> OIHW grad[n][k][9]: a direct write would scatter 4-byte stores 36 bytes
// apart.  One block sums a (n, 64 k, 9 taps) piece -- four threads share each k and take every fourth split,
// reading 256-byte runs -- and transposes it through LDS, so the 576 results leave as one contiguous run.
__global__ __launch_bounds__(256) void wgrad_reduce_oihw9_kernel(const float* __restrict__ partial, int S,
                                                                  long long split_stride, int K,
                                                                  float* __restrict__ grad, int accumulate) {
  __shared__ float red[4][9 * 64 + 8];
  const int n = blockIdx.y, k0 = blockIdx.x * 64;
  const int kk = threadIdx.x & 63, sl = threadIdx.x >> 6;
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  const float* base = partial + (long long)n * 9 * K + k0 + kk;
  for (int sp = sl; sp < S; sp += 4) {
    const float* q = base + sp * split_stride;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] += q[(long long)t * K];
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) red[sl][kk * 9 + t] = acc[t];
  __syncthreads();
  float* dst = grad + ((long long)n * K + k0) * 9;
  for (int j = threadIdx.x; j < 9 * 64; j += 256) {
    const float v = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
    dst[j] = accumulate ? dst[j] + v : v;
  }
}

// The same for MANY splits of a small tensor (the full-resolution layers: 512 splits of 64 x 9 x 64): a block per
// (n, tap, 64 k) and sixteen slices of the splits, so that the launch has thousands of waves in flight instead of
// N * K / 64 workgroups walking 128 splits each.  The 64 results of a block go out 36 bytes apart (a 147 KB tensor).
__global__ __launch_bounds__(1024) void wgrad_reduce_oihw9_wide_kernel(const float* __restrict__ partial, int S,
                                                                        long long split_stride, int K,
                                                                        float* __restrict__ grad, int accumulate) {
  __shared__ float red[16][64];
  const int n = blockIdx.z, t = blockIdx.y, k0 = blockIdx.x * 64;
  const int kk = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const float* q = partial + ((long long)n * 9 + t) * K + k0 + kk;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int sp = sl;
  for (; sp + 48 < S; sp += 64) {
    a0 += q[sp * split_stride];
    a1 += q[(sp + 16) * split_stride];
    a2 += q[(sp + 32) * split_stride];
    a3 += q[(sp + 48) * split_stride];
  }
  for (; sp < S; sp += 16) a0 += q[sp * split_stride];
  red[sl][kk] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0) {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) v += red[j][kk];
    float* dst = grad + ((long long)n * K + k0 + kk) * 9 + t;
    *dst = accumulate ? *dst + v : v;
  }
}

extern "C" int uh_wgrad_reduce_strided(const float* partial, int S, long long split_stride, int N, int T, int K,
                                       long long on, long long ot, long long sn, long long st, long long sk,
                                       float* grad, int accumulate, uh_stream s) {
  UH_REQUIRE(partial && grad && S > 0 && N > 0 && T > 0 && K > 0 && split_stride > 0, "wgrad_reduce: bad arguments");
  if (T == 9 && K % 64 == 0 && on == 9LL * K && ot == K && sn == 9LL * K && st == 1 && sk == 9 && N <= 65535 &&
      S >= 64) {
    hipLaunchKernelGGL(wgrad_reduce_oihw9_wide_kernel, dim3(K / 64, 9, N), dim3(1024), 0, (hipStream_t)s, partial, S,
                       split_stride, K, grad, accumulate);
    return uh_check_launch("wgrad_reduce");
  }
  if (T == 9 && K % 64 == 0 && on == 9LL * K && ot == K && sn == 9LL * K && st == 1 && sk == 9 && N <= 65535) {
    hipLaunchKernelGGL(wgrad_reduce_oihw9_kernel, dim3(K / 64, N), dim3(256), 0, (hipStream_t)s, partial, S,
                       split_stride, K, grad, accumulate);
    return uh_check_launch("wgrad_reduce");
  }
  if (T == 4 && K % 64 == 0 && N % 4 == 0 && on == K && ot == (long long)N * K && sn == 4 && st == 1 &&
      sk == 4LL * N && N / 4 <= 65535) {
    hipLaunchKernelGGL(wgrad_reduce_convt_kernel, dim3(K / 64, N / 4), dim3(256), 0, (hipStream_t)s, partial, S,
                       split_stride, N, K, grad, accumulate);
    return uh_check_launch("wgrad_reduce");
  }
  const long long total = (long long)N * T * K;
  if (S >= 64 && total <= 65536) {
    hipLaunchKernelGGL((wgrad_reduce_kernel<32>), dim3((unsigned)((total + 7) / 8)), dim3(256), 0, (hipStream_t)s, partial,
                       S, split_stride, N, T, K, on, ot, sn, st, sk, grad, accumulate);
  } else {
    hipLaunchKernelGGL((wgrad_reduce_kernel<4>), dim3((unsigned)((total + 63) / 64)), dim3(256), 0, (hipStream_t)s, partial,
                       S, split_stride, N, T, K, on, ot, sn, st, sk, grad, accumulate);
  }
  return uh_check_launch("wgrad_reduce");
}

extern "C" int uh_wgrad_reduce(const float* partial, int S, int N, int T, int K, long long on, long long ot,
                               long long sn, long long st, long long sk, float* grad, int accumulate, uh_stream s) {
  return uh_wgrad_reduce_strided(partial, S, (long long)N * T * K, N, T, K, on, ot, sn, st, sk, grad, accumulate, s);
}
