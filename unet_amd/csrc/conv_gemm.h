// Argument blocks of the convolution GEMM kernels (conv_gemm.hip).
#pragma once
#include "common.h"

struct GemmArgs {
  const char* A;
  const char* Wt;
  char* out;
  const float* bias;
  const float* scale;
  const float* shift;
  float* stat_partial;
  const uint8_t* rowmask;
  long long Mp;
  int a_cstride, a_coff, K, N, T, KC;  // KC = K/64
  int Wp;                              // row pitch of the A geometry (W+1)
  int o_cstride, o_coff, mode;
  int H, W;                            // A geometry (for the transposed-conv scatter)
  int Cout;                            // transposed conv: N = 4*Cout
  // UH_EPI_MASKED_BNBWD: the consuming layer's fp16 conv output and batch statistics (scale/shift above)
  const char* ybn;
  int y_cstride, y_coff;
  const float* bn_mean;
  const float* bn_invstd;
};

// How the work of one launch is cut.  A "unit" is one BM x BN output tile (numbered as band_unit() below says) and
// takes niter = 3*K/32 iterations; the launch is the sequence of all (unit, iteration) pairs and
// workgroup position p takes a list of them.  With `aligned` a workgroup takes whole units: the 8 XCDs split
// the units into 8 contiguous ranges, and inside a range workgroup j takes units j, j + G/8, j + 2G/8, ... so that
// the workgroups that run at the same time on one XCD work on NEIGHBOURING tiles (the bands of neighbouring M
// tiles overlap and all N tiles of an M tile read the same rows: their re-reads then hit in that XCD's L2).
// Whole units are uneven when units % G != 0; otherwise (`aligned` = 0) position p takes the contiguous piece
// [band_start(p), band_start(p+1)) of the (unit, iteration) sequence, cut evenly in iterations: a piece that
// covers only part of a unit leaves its accumulators in `ws` (slot 2p: the piece starts inside the unit,
// slot 2p+1: it ends inside), and conv_band_fixup_kernel sums the pieces of every cut unit in position order
// and runs the epilogue -- a second launch instead of workgroups waiting for each other.
// Unit order.  nmajor = 0: unit u is M tile u / ntn, N tile u % ntn -- the N tiles of an M tile follow each other, an
// activation tile is fetched once and the whole weight matrix passes through every XCD's L2 again and again: right
// while the matrix stays resident there.  nmajor = 1: N tile u / ntm, M tile u % ntm -- an XCD's contiguous range of
// units then has one or two N tiles, whose weight slice stays in its L2, and the (small, deep-layer) activations are
// re-read per N tile instead: for the layers whose weights are several times the 4 MB of an L2.
struct BandSched {
  int ntm, ntn, niter, G, aligned;
  int upx;        // aligned: units per XCD range
  int nmajor;
  float* ws;
};

// (The order is a template parameter of the kernels' main loops: chosen at run time there, the extra scalar code in
// the loop cost the 3x3 kernel 15 - 25 %.)
template <bool NMAJOR>
__device__ __forceinline__ void band_unit(const BandSched& sc, int u, int& mt, int& nt) {
  if (NMAJOR) {
    nt = u / sc.ntm;
    mt = u - nt * sc.ntm;
  } else {
    mt = u / sc.ntn;
    nt = u - mt * sc.ntn;
  }
}

__device__ __forceinline__ long long band_start(const BandSched& sc, int pos) {
  const long long units = (long long)sc.ntm * sc.ntn;
  return ((long long)pos * units * sc.niter) / sc.G;
}

static inline int band_workgroups(long long units, int niter) {
  const long long total = units * niter;
  long long g = (total / 4) & ~7LL;               // at least ~4 iterations per workgroup
  if (g > 256) g = 256;                           // one workgroup per CU
  if (g < 8) g = 8;
  return (int)g;
}
