// Shared device/host helpers for the gfx950 kernels of libunet_hip.so.
//
// Activation layout ("padded-linear NHWC"):
//   A logical tensor [B][H][W][C] is stored as rows of C bf16 channels, one row per
//   position of a grid that has ONE extra zero column and ONE extra zero row per image:
//       Wp = W + 1, Hp = H + 1, row index m = (b*Hp + y)*Wp + x,  Mp = B*Hp*Wp rows.
//   A position is "interior" iff y < H and x < W; every other row is all zeros.
//   The buffer also carries guard rows before row 0 and after row Mp-1 (uh_guard_rows(W) rows on each side).
//   They start as zeros.  Kernels read them (3x3 neighbours of the first and last image rows, the overhang of
//   the last tile) and the GEMM epilogues also WRITE the rows of a ragged last M tile that lie past Mp -- up to
//   511 rows, always with zeros (the row flags of those rows are cleared before the multiply) -- which is why the
//   guard is at least 512 rows long.
//   With this layout the 3x3 neighbour (dy,dx) of row m is row m + dy*Wp + dx for EVERY
//   interior m, and whatever lies outside the image reads as zero -- so a pad-1
//   convolution is a GEMM whose A operand is the same matrix at nine row offsets,
//   with no per-lane bounds checks.
//   A tensor may live inside a wider buffer (skip-concat): `cstride` is the number of
//   channels per row of the buffer and `coff` the first channel of the view.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/unet_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define UH_LDS __attribute__((address_space(3)))
#define UH_GLOBAL __attribute__((address_space(1)))

// ---- error plumbing (host) -------------------------------------------------
void uh_set_error(const char* fmt, ...);
int uh_check_launch(const char* what);
// Raises a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) once per (kernel, device); the
// attribute belongs to the device that is current when it is set.  Thread-safe.
void uh_set_max_lds(const void* kernel, int bytes);

#define UH_REQUIRE(cond, ...)                         \
  do {                                                \
    if (!(cond)) {                                    \
      uh_set_error(__VA_ARGS__);                      \
      return UH_ERR_SHAPE;                            \
    }                                                 \
  } while (0)

static inline int uh_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long uh_mp(int B, int H, int W) { return (long long)B * (H + 1) * (W + 1); }

// ---- device helpers --------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// 16-byte vector of 8 bf16 as a plain struct for global/LDS moves
struct __attribute__((aligned(16))) U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ void unpack8(const U4& u, float (&f)[8]) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v = {(bf16)lo, (bf16)hi};
  return *reinterpret_cast<uint32_t*>(&v);
}

// fp16 variants: the raw convolution output that feeds train-mode BatchNorm is stored as IEEE
// half (11 significant bits).  BatchNorm subtracts the batch mean, which amplifies the storage
// rounding of its input by |mean|/std; bf16 (8 bits) there costs ~10x the end-to-end error of
// every other bf16 rounding in the network.  Values are clamped to the finite half range.
__device__ __forceinline__ void unpack8h(const U4& u, float (&f)[8]) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const h2 v = *reinterpret_cast<const h2*>(&w[i]);
    f[2 * i] = (float)v[0];
    f[2 * i + 1] = (float)v[1];
  }
}
__device__ __forceinline__ _Float16 f2h(float v) { return (_Float16)fminf(fmaxf(v, -65504.f), 65504.f); }
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  h2 v = {f2h(lo), f2h(hi)};
  return *reinterpret_cast<uint32_t*>(&v);
}
__device__ __forceinline__ U4 pack8h(const float (&f)[8]) {
  U4 u;
  u.x = pack2h(f[0], f[1]);
  u.y = pack2h(f[2], f[3]);
  u.z = pack2h(f[4], f[5]);
  u.w = pack2h(f[6], f[7]);
  return u;
}

__device__ __forceinline__ U4 pack8(const float (&f)[8]) {
  U4 u;
  u.x = pack2(f[0], f[1]);
  u.y = pack2(f[2], f[3]);
  u.z = pack2(f[4], f[5]);
  u.w = pack2(f[6], f[7]);
  return u;
}
