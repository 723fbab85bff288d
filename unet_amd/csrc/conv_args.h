// Arguments shared by the implicit-GEMM convolution kernels (conv_gemm.hip, conv_band.hip).
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
};

// conv_band.hip: the T = 9 kernel.  Rows per M tile (= rows per stat_partial entry), scratch bytes for even
// cuts, and the launch (ws may be null).
constexpr int UH_BAND_BM = 256;
size_t uh_band_workspace_bytes(long long Mp, int N, int K);
void uh_launch_band(const GemmArgs& p, void* ws, size_t ws_bytes, hipStream_t st);
