// clDice soft-skeleton loss (losses/cl_dice.py of the reference) on gfx950: fp32, bandwidth bound.
//
// The reference iterates   img <- erode(img); delta = relu(img - dilate(erode(img))); skel += relu(delta - skel*delta)
// ten times.  With E_0 = clamp(x,0,1) and E_{k+1} = erode(E_k) the same values are
//     delta_k = relu(E_k - dilate(E_{k+1})),  skel_0 = delta_0,  skel_k = skel_{k-1} + relu(delta_k - skel_{k-1}*delta_k)
// for k = 0..10, i.e. ONE chain of eleven erosions instead of twenty-one (the reference erodes every level twice,
// once inside soft_open and once at the top of the next iteration; both give the same tensor and their gradients add).
// erode  = 5-point-cross minimum, out-of-image neighbours ignored (cl_dice.py:15-17),
// dilate = 3x3 maximum, out-of-image neighbours ignored            (cl_dice.py:27).
//
// Forward: one launch per level k reads E_k (centre), the 3x3 neighbourhood of E_{k+1} and skel_{k-1}, and writes
// skel_k and E_{k+2}.  Backward: one launch per level, 32x32 tiles with halos staged in LDS, three in-tile stages
// (pointwise masks -> dilation scatter turned into a gather -> erosion routing), reproducing autograd's tie rules:
// max-pool sends the gradient to the FIRST extremum in row-major window order, torch.min splits ties 1/2:1/2,
// relu'(0) = 0, clamp passes the gradient on its closed interval.
// All sums are two-stage (per-workgroup partials, then one finalising workgroup): deterministic, no atomics.
#include "common.h"
#include <cstdlib>
#include <type_traits>
#pragma clang fp contract(off)   // keep delta - skel*delta as two roundings, like the reference's separate ops

#define SKEL_LEVELS 11            // E_1..E_11, skel_0..skel_10  (num_iter = 10, cl_dice.py:103)

struct Img {
  int H, W;
  __device__ __forceinline__ bool in(int y, int x) const { return y >= 0 && y < H && x >= 0 && x < W; }
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

// 5-point-cross minimum at (y,x) of plane p; CLAMP: the plane holds raw input and E_0 = clamp01(input)
template <bool CLAMP>
__device__ __forceinline__ float erode_at(const float* __restrict__ p, Img g, int y, int x) {
  float v = p[y * g.W + x];
  if (y > 0) v = fminf(v, p[(y - 1) * g.W + x]);
  if (y + 1 < g.H) v = fminf(v, p[(y + 1) * g.W + x]);
  if (x > 0) v = fminf(v, p[y * g.W + x - 1]);
  if (x + 1 < g.W) v = fminf(v, p[y * g.W + x + 1]);
  return CLAMP ? clamp01(v) : v;   // clamp is monotone: min(clamp(.)) == clamp(min(.))
}

__device__ __forceinline__ float dilate_at(const float* __restrict__ p, Img g, int y, int x) {
  float v = -INFINITY;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx)
      if (g.in(y + dy, x + dx)) v = fmaxf(v, p[(y + dy) * g.W + x + dx]);
  return v;
}

// ---- stand-alone morphology (also the first erosion of every chain) -----------------------------------
template <bool CLAMP, bool DILATE>
__global__ __launch_bounds__(256) void morph_kernel(int planes, int H, int W, const float* __restrict__ in,
                                                    float* __restrict__ out) {
  const Img g{H, W};
  const long long n = (long long)planes * H * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W);
    const long long r = i / W;
    const int y = (int)(r % H);
    const float* p = in + (r / H) * H * W;
    out[i] = DILATE ? dilate_at(p, g, y, x) : erode_at<CLAMP>(p, g, y, x);
  }
}

// ---- forward level k ------------------------------------------------------------------------------------
// skel_k = (k==0) ? delta : skel_prev + relu(delta - skel_prev*delta), delta = relu(E_k - dilate(E_{k+1}));
// E_{k+2} = erode(E_{k+1}) when ek2 != null.  CLAMP0: ek is the raw input (level 0).
template <bool CLAMP0>
__global__ __launch_bounds__(256) void skel_level_kernel(int planes, int H, int W, const float* __restrict__ ek,
                                                         const float* __restrict__ ek1,
                                                         const float* __restrict__ skel_prev,
                                                         float* __restrict__ skel_out, float* __restrict__ ek2) {
  const Img g{H, W};
  const long long n = (long long)planes * H * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W);
    const long long r = i / W;
    const int y = (int)(r % H);
    const float* p1 = ek1 + (r / H) * H * W;
    float e = ek[i];
    if (CLAMP0) e = clamp01(e);
    const float d = dilate_at(p1, g, y, x);
    const float delta = fmaxf(e - d, 0.f);
    float s = delta;
    if (skel_prev) {
      const float sp = skel_prev[i];
      const float prod = sp * delta;
      s = sp + fmaxf(delta - prod, 0.f);
    }
    skel_out[i] = s;
    if (ek2) ek2[i] = erode_at<false>(p1, g, y, x);
  }
}

// Same, four pixels of one row per thread (W % 4 == 0): the three rows of E_{k+1} are read as one aligned
// 16-byte load plus the two neighbours each instead of nine scalar loads per pixel, and all index arithmetic
// is 32-bit.  min/max are exact and order-independent, so the results equal the scalar kernel's bit for bit.
template <bool CLAMP0>
__global__ __launch_bounds__(256) void skel_level_v4_kernel(int planes, int H, int W, const float* __restrict__ ek,
                                                            const float* __restrict__ ek1,
                                                            const float* __restrict__ skel_prev,
                                                            float* __restrict__ skel_out, float* __restrict__ ek2) {
  const unsigned w4 = (unsigned)W >> 2;
  const unsigned total = (unsigned)planes * H * w4;
  for (unsigned it = blockIdx.x * 256u + threadIdx.x; it < total; it += gridDim.x * 256u) {
    const unsigned xq = it % w4;
    const unsigned row = it / w4;                       // plane * H + y
    const int y = (int)(row % (unsigned)H);
    const int x0 = (int)(xq << 2);
    const size_t base = (size_t)row * W + x0;           // index of the first of the four pixels
    float dmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    float emin[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      if (y + dy < 0 || y + dy >= H) continue;
      const float* r = ek1 + base + (long long)dy * W;
      const f32x4 c = *(const f32x4*)r;
      const float l = (x0 > 0) ? r[-1] : -INFINITY;
      const float rr = (x0 + 4 < W) ? r[4] : -INFINITY;
      const float v[6] = {l, c[0], c[1], c[2], c[3], rr};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dmax[j] = fmaxf(dmax[j], fmaxf(fmaxf(v[j], v[j + 1]), v[j + 2]));
        emin[j] = fminf(emin[j], v[j + 1]);
      }
      if (dy == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j > 0 || x0 > 0) emin[j] = fminf(emin[j], v[j]);
          if (j < 3 || x0 + 4 < W) emin[j] = fminf(emin[j], v[j + 2]);
        }
      }
    }
    f32x4 e = *(const f32x4*)(ek + base);
    f32x4 sp = {0.f, 0.f, 0.f, 0.f};
    if (skel_prev) sp = *(const f32x4*)(skel_prev + base);
    f32x4 so;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float ev = e[j];
      if (CLAMP0) ev = clamp01(ev);
      const float delta = fmaxf(ev - dmax[j], 0.f);
      float sv = delta;
      if (skel_prev) {
        const float prod = sp[j] * delta;
        sv = sp[j] + fmaxf(delta - prod, 0.f);
      }
      so[j] = sv;
    }
    *(f32x4*)(skel_out + base) = so;
    if (ek2) *(f32x4*)(ek2 + base) = f32x4{emin[0], emin[1], emin[2], emin[3]};
  }
}

// ---- sums ---------------------------------------------------------------------------------------------------
// partial[blk][7] = (sum t*p, sum t, sum p, sum skel_p*t, sum skel_p, sum skel_t*p, sum skel_t), t,p clamped
__global__ __launch_bounds__(256) void cldice_sums_kernel(long long n, const float* __restrict__ yt,
                                                          const float* __restrict__ yp,
                                                          const float* __restrict__ skp,
                                                          const float* __restrict__ skt, float* __restrict__ partial) {
  float s[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float t = clamp01(yt[i]), p = clamp01(yp[i]);
    const float a = skp[i], b = skt[i];
    s[0] += t * p;
    s[1] += t;
    s[2] += p;
    s[3] += a * t;
    s[4] += a;
    s[5] += b * p;
    s[6] += b;
  }
  __shared__ float red[4][7];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const float v = wave_sum(s[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 7)
    partial[blockIdx.x * 7 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// coef: [0] loss, [1..3] direct gradient dL/dp = c1*t + c2 + c3*skel_t, [4..5] g_skel_pred = c4*t + c5,
//       [6] dice, [7] cl_dice, [8] tprec, [9] tsens, [10..16] the seven sums
__global__ __launch_bounds__(256) void cldice_finalize_kernel(const float* __restrict__ partial, int P, float alpha,
                                                              float smooth, float* __restrict__ coef,
                                                              float* __restrict__ loss_out) {
  __shared__ double red[256][7];
  double s[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int p = threadIdx.x; p < P; p += 256)
#pragma unroll
    for (int k = 0; k < 7; ++k) s[k] += (double)partial[p * 7 + k];
#pragma unroll
  for (int k = 0; k < 7; ++k) red[threadIdx.x][k] = s[k];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st)
#pragma unroll
      for (int k = 0; k < 7; ++k) red[threadIdx.x][k] += red[threadIdx.x + st][k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double I = red[0][0], St = red[0][1], Sp = red[0][2], A = red[0][3], Ssp = red[0][4], Bv = red[0][5],
                 Sst = red[0][6];
    const double den = St + Sp + 1.0, num = 2.0 * I + 1.0;     // soft_dice hard-codes smooth = 1 (cl_dice.py:83)
    const double dice = 1.0 - num / den;
    const double tprec = (A + smooth) / (Ssp + smooth), tsens = (Bv + smooth) / (Sst + smooth);
    const double cl = 1.0 - 2.0 * tprec * tsens / (tprec + tsens);
    const double dcl_dprec = -2.0 * tsens * tsens / ((tprec + tsens) * (tprec + tsens));
    const double dcl_dsens = -2.0 * tprec * tprec / ((tprec + tsens) * (tprec + tsens));
    const float loss = (float)((1.0 - alpha) * dice + alpha * cl);
    coef[0] = loss;
    *loss_out = loss;
    coef[1] = (float)(-(1.0 - alpha) * 2.0 / den);
    coef[2] = (float)((1.0 - alpha) * num / (den * den));
    coef[3] = (float)(alpha * dcl_dsens / (Sst + smooth));
    coef[4] = (float)(alpha * dcl_dprec / (Ssp + smooth));
    coef[5] = (float)(-alpha * dcl_dprec * (A + smooth) / ((Ssp + smooth) * (Ssp + smooth)));
    coef[6] = (float)dice;
    coef[7] = (float)cl;
    coef[8] = (float)tprec;
    coef[9] = (float)tsens;
    for (int k = 0; k < 7; ++k) coef[10 + k] = (float)red[0][k];
  }
}

// ---- backward level k ------------------------------------------------------------------------------------------
#define BT 32                       // core tile edge
struct BwdArgs {
  const float* ek;                  // E_k (raw input when level0)
  const float* ek1;                 // E_{k+1}
  const float* skel_prev;           // skel_{k-1} (null at level 0)
  const float* gs_in;               // dL/dskel_k (null at the top level: computed from coef and y_true)
  const float* ge1_in;              // dL/dE_{k+1} without the dilation term of this level (null at the top level)
  float* gs_out;                    // dL/dskel_{k-1} (null at level 0)
  float* ge_out;                    // dL/dE_k without the dilation term of level k-1; at level 0: dL/dy_pred
  const float* yt;                  // y_true (top level and level 0)
  const float* skt;                 // skel_true (level 0)
  const float* coef;
  const float* gout;                // upstream gradient (scalar on the device)
  int H, W, level0, top;
};

__global__ __launch_bounds__(256) void skel_bwd_level_kernel(BwdArgs a) {
  constexpr int N3 = BT + 6, N2 = BT + 4, N1 = BT + 2;
  __shared__ float sE1[N3 * N3];    // E_{k+1}, halo 3; -inf outside the image (never a maximum)
  __shared__ float sE[N2 * N2];     // E_k, halo 2; +inf outside the image (never a minimum)
  __shared__ float sGD[N2 * N2];    // dL/dD_k, halo 2
  __shared__ signed char sAM[N2 * N2];  // arg-max slot (0..8) of the 3x3 window of E_{k+1} at each position, halo 2
  __shared__ float sGV[N1 * N1];    // halo 1: dL/dE_{k+1}(q) times the share of the vertical minimum of window q
  __shared__ float sGH[N1 * N1];    //         ... of the horizontal minimum
  __shared__ signed char sRT[N1 * N1];  // where those minima sit: (varg + 1) | (harg + 1) << 2; 15 outside the image
  __shared__ float sDir[BT * BT];   // direct term dL/ddelta_k * mask at the core
  const Img g{a.H, a.W};
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * BT, y0 = blockIdx.y * BT;
  const long long plane = (long long)blockIdx.z * a.H * a.W;
  const float* ek = a.ek + plane;
  const float* ek1 = a.ek1 + plane;
  const float gout = *a.gout;

  // Everything this workgroup reads from global memory is requested up front: the per-position operands of stages
  // A and B go into registers now, so no stage waits for a load in the middle of its loop (with four workgroups
  // per CU those exposed latencies, not instructions or bandwidth, set the kernel's time).
  constexpr int KA = (N2 * N2 + 255) / 256, KB = (N1 * N1 + 255) / 256;
  float pre_gs[KA], pre_sp[KA], pre_g1[KB];
#pragma unroll
  for (int k = 0; k < KA; ++k) {
    const int i = tid + 256 * k;
    const int y = y0 - 2 + i / N2, x = x0 - 2 + i % N2;
    pre_gs[k] = 0.f;
    pre_sp[k] = 0.f;
    if (i < N2 * N2 && g.in(y, x)) {
      const long long gi = plane + (long long)y * a.W + x;
      pre_gs[k] = a.top ? a.yt[gi] : a.gs_in[gi];
      if (a.skel_prev) pre_sp[k] = a.skel_prev[gi];
    }
  }
#pragma unroll
  for (int k = 0; k < KB; ++k) {
    const int i = tid + 256 * k;
    const int y = y0 - 1 + i / N1, x = x0 - 1 + i % N1;
    pre_g1[k] = (a.ge1_in && i < N1 * N1 && g.in(y, x)) ? a.ge1_in[plane + (long long)y * a.W + x] : 0.f;
  }
  const float c4 = a.top ? a.coef[4] : 0.f, c5 = a.top ? a.coef[5] : 0.f;
  // the two LDS images likewise: all loads first (from element 0 of the plane where the position is outside the image),
  // then the stores; as a plain loop every iteration would wait for its own load before the next one is issued
  constexpr int K3 = (N3 * N3 + 255) / 256;
  float e1v[K3], ev[KA];
#pragma unroll
  for (int k = 0; k < K3; ++k) {
    const int i = tid + 256 * k;
    const int y = y0 - 3 + i / N3, x = x0 - 3 + i % N3;
    const bool ok = i < N3 * N3 && g.in(y, x);
    const float v = ek1[ok ? y * a.W + x : 0];
    e1v[k] = ok ? v : -INFINITY;
  }
#pragma unroll
  for (int k = 0; k < KA; ++k) {
    const int i = tid + 256 * k;
    const int y = y0 - 2 + i / N2, x = x0 - 2 + i % N2;
    const bool ok = i < N2 * N2 && g.in(y, x);
    const float v = ek[ok ? y * a.W + x : 0];
    ev[k] = ok ? (a.level0 ? clamp01(v) : v) : INFINITY;
  }
#pragma unroll
  for (int k = 0; k < K3; ++k)
    if (tid + 256 * k < N3 * N3) sE1[tid + 256 * k] = e1v[k];
#pragma unroll
  for (int k = 0; k < KA; ++k)
    if (tid + 256 * k < N2 * N2) sE[tid + 256 * k] = ev[k];
  __syncthreads();

  // stage A: pointwise on core+2
#pragma unroll
  for (int k = 0; k < KA; ++k) {
    const int i = tid + 256 * k;
    if (i >= N2 * N2) break;
    const int ly = i / N2, lx = i % N2;
    const int y = y0 - 2 + ly, x = x0 - 2 + lx;
    float gd = 0.f;
    int am = -1;
    if (g.in(y, x)) {
      // 3x3 window of E_{k+1} in row-major order, first maximum wins
      float win[9];                     // all nine reads first: one LDS round trip instead of nine
#pragma unroll
      for (int t = 0; t < 9; ++t) win[t] = sE1[(ly + t / 3) * N3 + lx + t % 3];
      float best = -INFINITY;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const bool up = win[t] > best;
        best = up ? win[t] : best;
        am = up ? t : am;
      }
      const float e = sE[i];
      const float diff = e - best;
      const float delta = fmaxf(diff, 0.f);
      const long long gi = plane + (long long)y * a.W + x;
      const float gs = a.top ? gout * (c4 * clamp01(pre_gs[k]) + c5) : pre_gs[k];
      float gdelta = gs;
      const bool core = (ly >= 2 && ly < 2 + BT && lx >= 2 && lx < 2 + BT);
      if (a.skel_prev) {
        const float sp = pre_sp[k];
        const float prod = sp * delta;
        const bool on = (delta - prod) > 0.f;
        gdelta = on ? gs * (1.f - sp) : 0.f;
        if (core) a.gs_out[gi] = on ? gs * (1.f - delta) : gs;
      }
      const float dmask = diff > 0.f ? gdelta : 0.f;
      gd = -dmask;
      if (core) sDir[(ly - 2) * BT + lx - 2] = dmask;
    }
    sGD[i] = gd;
    sAM[i] = (signed char)am;
  }
  __syncthreads();

  // stage B on core+1: complete dL/dE_{k+1}(q) = incoming part + the dilation windows whose arg-max is q, and how
  // window q of the erosion E_{k+1} = erode(E_k) routes it: min(vertical min, horizontal min), torch.min splits a tie
  // 1/2 : 1/2, inside each 3-window the FIRST minimum (top/left, centre, bottom/right) takes the gradient
#pragma unroll
  for (int k = 0; k < KB; ++k) {
    const int i = tid + 256 * k;
    if (i >= N1 * N1) break;
    const int ly = i / N1, lx = i % N1;
    const int y = y0 - 1 + ly, x = x0 - 1 + lx;
    float gv = 0.f, gh = 0.f;
    int code = 15;
    if (g.in(y, x)) {
      float v = pre_g1[k];
      // window centred at q + (dy,dx), t = (dy+1)*3 + (dx+1): q sits at slot 8 - t of that window.  All reads
      // first, then selects (adding 0 where the arg-max is elsewhere leaves the sum unchanged)
      float gdw[9];
      int amw[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int qi = (ly + t / 3) * N2 + lx + t % 3;
        gdw[t] = sGD[qi];
        amw[t] = sAM[qi];
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) v += (amw[t] == 8 - t) ? gdw[t] : 0.f;
      const int ci = (ly + 1) * N2 + lx + 1;               // q in sE coordinates
      const float c = sE[ci];
      float vmin = sE[ci - N2];
      int varg = -1;
      if (c < vmin) { vmin = c; varg = 0; }
      { const float b = sE[ci + N2]; if (b < vmin) { vmin = b; varg = 1; } }
      float hmin = sE[ci - 1];
      int harg = -1;
      if (c < hmin) { hmin = c; harg = 0; }
      { const float b = sE[ci + 1]; if (b < hmin) { hmin = b; harg = 1; } }
      const float wv = vmin < hmin ? 1.f : (vmin == hmin ? 0.5f : 0.f);
      gv = v * wv;
      gh = v * (1.f - wv);
      code = (varg + 1) | ((harg + 1) << 2);
    }
    sGV[i] = gv;
    sGH[i] = gh;
    sRT[i] = (signed char)code;
  }
  __syncthreads();

  // stage C at the core: p collects from the five erosion windows that contain it (centred at p, above, below, left,
  // right of p, in that order) the shares whose minimum sits on p
  for (int i = tid; i < BT * BT; i += 256) {
    const int cy = i / BT, cx = i % BT;
    const int y = y0 + cy, x = x0 + cx;
    if (!g.in(y, x)) continue;
    float acc = sDir[i];
    const int q = (cy + 1) * N1 + cx + 1;
    int c = sRT[q];
    if ((c & 3) == 1) acc += sGV[q];
    if ((c >> 2) == 1) acc += sGH[q];
    c = sRT[q - N1];                                        // window above: p is its bottom row
    if (c != 15 && (c & 3) == 2) acc += sGV[q - N1];
    c = sRT[q + N1];                                        // window below: p is its top row
    if (c != 15 && (c & 3) == 0) acc += sGV[q + N1];
    c = sRT[q - 1];                                         // window to the left: p is its right column
    if (c != 15 && (c >> 2) == 2) acc += sGH[q - 1];
    c = sRT[q + 1];                                         // window to the right: p is its left column
    if (c != 15 && (c >> 2) == 0) acc += sGH[q + 1];
    const long long gi = plane + (long long)y * a.W + x;
    if (a.level0) {
      const float raw = ek[y * a.W + x];
      const float t = clamp01(a.yt[gi]);
      const float direct = gout * (a.coef[1] * t + a.coef[2] + a.coef[3] * a.skt[gi]);
      acc = (raw >= 0.f && raw <= 1.f) ? acc + direct : 0.f;
    }
    a.ge_out[gi] = acc;
  }
}

// ---- backward level k, strip walk --------------------------------------------------------------------------------
// The same arithmetic as skel_bwd_level_kernel (term for term, in the same order: bit-identical results) without LDS,
// barriers or index divisions.  A wave owns a strip of 64 columns (58 core columns + 3 halo columns on either side,
// lane = column) and walks down R core rows; the three stages run as a software pipeline three rows deep in
// registers: with E_{k+1} row r just arrived, stage A (masks, arg-max of the dilation window) is evaluated for row
// r-1, stage B (gather of the dilation gradient, routing of the erosion window) for row r-2, stage C (collect at the
// pixel) for row r-3.  Horizontal neighbours come from the adjacent lanes by DPP wave shifts; every global access is
// one 256-byte row piece per wave, requested one iteration ahead of its use.
#define SK_CORE 58
// wave_shr:1 / wave_shl:1 with bound_ctrl: lanes 0 / 63 receive 0 (their results are halo, never stored)
__device__ __forceinline__ int from_left(int v) { return __builtin_amdgcn_mov_dpp(v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int from_right(int v) { return __builtin_amdgcn_mov_dpp(v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ float from_left(float v) {    // the value held by lane - 1 (column x - 1)
  return __builtin_bit_cast(float, from_left(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float from_right(float v) {   // the value held by lane + 1 (column x + 1)
  return __builtin_bit_cast(float, from_right(__builtin_bit_cast(int, v)));
}

template <bool TOP, bool L0>
__global__ __launch_bounds__(256) void skel_bwd_strip_kernel(BwdArgs a, int planes, int R, int nstrips, int nchunks) {
  const int lane = threadIdx.x & 63;
  // the wave's strip: the same for all 64 lanes, and said so (readfirstlane), or the row counter and every row test
  // below would be vector values with a branch each
  // Workgroups are handed to the eight XCDs in turn, and each XCD has its own L2: workgroup b works on the b/8-th block
  // of four strips of XCD b%8's contiguous share, so that strips whose halo rows overlap meet in one L2.
  // (The grid is a multiple of eight workgroups.)
  const unsigned per_xcd = gridDim.x >> 3;
  const unsigned lblk = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  unsigned item = lblk * 4u + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (item >= (unsigned)planes * nstrips * nchunks) return;
  const int chunk = (int)(item % (unsigned)nchunks);
  item /= (unsigned)nchunks;
  const int strip = (int)(item % (unsigned)nstrips);
  const long long plane = (long long)(item / (unsigned)nstrips) * a.H * a.W;
  const int H = a.H, W = a.W;
  const int Y0 = chunk * R, Y1 = (Y0 + R < H) ? Y0 + R : H;
  const int x = strip * SK_CORE - 3 + lane;
  const bool inx = x >= 0 && x < W;
  const bool core = inx && lane >= 3 && lane < 3 + SK_CORE;
  const long long col = plane + (inx ? x : 0);
  const float* __restrict__ p_e1 = a.ek1 + col;
  const float* __restrict__ p_e = a.ek + col;
  const float* __restrict__ p_sp = L0 ? nullptr : a.skel_prev + col;
  const float* __restrict__ p_gs = TOP ? a.yt + col : a.gs_in + col;     // top level: dL/dskel comes from y_true
  const float* __restrict__ p_g1 = TOP ? nullptr : a.ge1_in + col;
  const float* __restrict__ p_yt = a.yt + col;
  const float* __restrict__ p_skt = L0 ? a.skt + col : nullptr;
  const float gout = *a.gout;
  const float c4 = TOP ? a.coef[4] : 0.f, c5 = TOP ? a.coef[5] : 0.f;
  const float k1 = L0 ? a.coef[1] : 0.f, k2 = L0 ? a.coef[2] : 0.f, k3 = L0 ? a.coef[3] : 0.f;
  // row y of a plane, `dflt` outside the image; the load itself is unconditional (from a row inside the image)
  auto ld = [&](const float* __restrict__ p, int y, float dflt) -> float {
    const bool ok = (y >= 0) & (y < H);
    const float v = p[(long long)(ok ? y : 0) * W];
    return (ok & inx) ? v : dflt;
  };
  // the same in two halves, so that a value requested in one iteration is first touched in the next
  auto ld_raw = [&](const float* __restrict__ p, int y) -> float {
    const bool ok = (y >= 0) & (y < H);
    return p[(ok ? y : 0) * W];                        // a plane has fewer than 2^31 pixels (checked on the host)
  };
  auto sel = [&](float raw, int y, float dflt) -> float { return ((y >= 0) & (y < H) & inx) ? raw : dflt; };
  // first maximum of (left, centre, right) of a row of E_{k+1}
  auto rowmax = [&](float c, float& m, int& arg) {
    const float l = from_left(c), r = from_right(c);
    m = l;
    arg = 0;
    if (c > m) { m = c; arg = 1; }
    if (r > m) { m = r; arg = 2; }
  };
  // ---- pipeline state.  Three-row windows are rings indexed by the iteration's phase (r - (Y0 - 1)) mod 3: in phase PH
  // slot PH holds the oldest row, PH+1 the middle one, PH+2 (mod 3) receives the new one; the loop body is written
  // once as a lambda over the phase and instantiated three times, so the windows rotate by renaming, without moves.
  float mx[3];                       // row maxima of E_{k+1}: rows r-2, r-1, r
  int ax[3];
  float gdl[3], gdc[3], gdr[3];      // stage A rows r-3, r-2, r-1: dL/dD and the arg-max slot, with both neighbours
  int aml[3], amc[3], amr[3];
  float ec[3];                       // E_k rows r-3, r-2, r-1 (clamped at level 0; +inf outside the image)
  float rawc[3];                     // level 0: the same rows as given (the clamp's pass-through test)
  float gvc[3], ghc[3];              // stage B rows r-4, r-3, r-2
  int cdc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    gdl[i] = gdc[i] = gdr[i] = 0.f;
    aml[i] = amc[i] = amr[i] = -1;
    ec[i] = INFINITY;
    rawc[i] = 0.f;
    gvc[i] = ghc[i] = 0.f;
    cdc[i] = 15;
  }
  rowmax(ld(p_e1, Y0 - 3, -INFINITY), mx[1], ax[1]);      // phase 0: slots 1, 2 hold rows r-2, r-1; slot 0 takes row r
  rowmax(ld(p_e1, Y0 - 2, -INFINITY), mx[2], ax[2]);
  // operands of the first iteration (r = Y0 - 1): E_{k+1} row r, and row r-1 of E_k, dL/dskel, skel_{k-1}
  float q_e1 = ld_raw(p_e1, Y0 - 1);
  float q_e = ld_raw(p_e, Y0 - 2);
  float q_gs = ld_raw(p_gs, Y0 - 2);
  float q_sp = L0 ? 0.f : ld_raw(p_sp, Y0 - 2);
  float q_g1 = 0.f, q_yt = 0.f, q_skt = 0.f;

  auto body = [&](auto phase, int r) {
    constexpr int PH = decltype(phase)::value;
    constexpr int O = PH % 3, M = (PH + 1) % 3, N = (PH + 2) % 3;     // oldest, middle, newest slot of the A / E rings
    // requests for the next iteration
    const float n_e1 = ld_raw(p_e1, r + 1);
    const float n_e = ld_raw(p_e, r);
    const float n_gs = ld_raw(p_gs, r);
    const float n_sp = L0 ? 0.f : ld_raw(p_sp, r);
    const float n_g1 = TOP ? 0.f : ld_raw(p_g1, r - 1);
    const float n_yt = L0 ? ld_raw(p_yt, r - 2) : 0.f;
    const float n_skt = L0 ? ld_raw(p_skt, r - 2) : 0.f;
    // this iteration's operands: what the previous one requested
    const float c_e1 = sel(q_e1, r, -INFINITY);
    const float c_e = sel(q_e, r - 1, INFINITY);
    const float c_gs = sel(q_gs, r - 1, 0.f);
    const float c_sp = sel(q_sp, r - 1, 0.f);
    const float c_g1 = (TOP || r == Y0 - 1) ? 0.f : sel(q_g1, r - 2, 0.f);
    const float c_yt = sel(q_yt, r - 3, 0.f);
    const float c_skt = sel(q_skt, r - 3, 0.f);

    // ---- stage A, row r-1: the row maxima ring holds rows r-2, r-1 in slots M, N and takes row r in slot O
    rowmax(c_e1, mx[O], ax[O]);
    {
      const int y = r - 1;
      const bool in = inx & (y >= 0) & (y < H);
      float best = mx[M];
      int am = ax[M];
      if (mx[N] > best) { best = mx[N]; am = 3 + ax[N]; }
      if (mx[O] > best) { best = mx[O]; am = 6 + ax[O]; }
      const float e = L0 ? clamp01(c_e) : c_e;
      const float diff = e - best;
      const float delta = fmaxf(diff, 0.f);
      const float gs = TOP ? gout * (c4 * clamp01(c_gs) + c5) : c_gs;
      float gdelta = gs;
      if (!L0) {
        const float sp = c_sp;
        const float prod = sp * delta;
        const bool on = (delta - prod) > 0.f;
        gdelta = on ? gs * (1.f - sp) : 0.f;
        if (core & in & (y >= Y0) & (y < Y1)) a.gs_out[plane + (long long)y * W + x] = on ? gs * (1.f - delta) : gs;
      }
      const float dmask = diff > 0.f ? gdelta : 0.f;
      const float gd = in ? -dmask : 0.f;
      const int amv = in ? am : -1;
      // the new row replaces the oldest one: afterwards slots M, N, O hold rows r-3, r-2, r-1
      gdc[O] = gd; gdl[O] = from_left(gd); gdr[O] = from_right(gd);
      amc[O] = amv; aml[O] = from_left(amv); amr[O] = from_right(amv);
      ec[O] = in ? e : INFINITY;
      if (L0) rawc[O] = c_e;
    }
    // ---- stage B, row r-2 (first needed: row Y0 - 1); stage-A rows r-3, r-2, r-1 are in slots M, N, O
    if (r > Y0) {
      const int y = r - 2;
      const bool in = inx & (y >= 0) & (y < H);
      float v = c_g1;
      v += (aml[M] == 8) ? gdl[M] : 0.f;
      v += (amc[M] == 7) ? gdc[M] : 0.f;
      v += (amr[M] == 6) ? gdr[M] : 0.f;
      v += (aml[N] == 5) ? gdl[N] : 0.f;
      v += (amc[N] == 4) ? gdc[N] : 0.f;
      v += (amr[N] == 3) ? gdr[N] : 0.f;
      v += (aml[O] == 2) ? gdl[O] : 0.f;
      v += (amc[O] == 1) ? gdc[O] : 0.f;
      v += (amr[O] == 0) ? gdr[O] : 0.f;
      const float c = ec[N];
      const float el = from_left(c), er = from_right(c);
      float vmin = ec[M];
      int varg = -1;
      if (c < vmin) { vmin = c; varg = 0; }
      if (ec[O] < vmin) { vmin = ec[O]; varg = 1; }
      float hmin = el;
      int harg = -1;
      if (c < hmin) { hmin = c; harg = 0; }
      if (er < hmin) { hmin = er; harg = 1; }
      const float wv = vmin < hmin ? 1.f : (vmin == hmin ? 0.5f : 0.f);
      // stage-B ring: the new row (r-2) replaces the oldest in slot O; rows r-4, r-3 stay in slots M, N
      gvc[O] = in ? v * wv : 0.f;
      ghc[O] = in ? v * (1.f - wv) : 0.f;
      cdc[O] = in ? ((varg + 1) | ((harg + 1) << 2)) : 15;
    }
    // ---- stage C, row r-3: stage-B rows r-4, r-3, r-2 are in slots M, N, O; the direct term dmask of row r-3 is
    // minus the stage-A value in slot M
    if (r >= Y0 + 3) {
      const int y = r - 3;
      const float ghl = from_left(ghc[N]), ghr = from_right(ghc[N]);
      const int cdl = from_left(cdc[N]), cdr = from_right(cdc[N]);
      float acc = -gdc[M];
      int c = cdc[N];
      if ((c & 3) == 1) acc += gvc[N];
      if ((c >> 2) == 1) acc += ghc[N];
      c = cdc[M];                                            // window above: p is its bottom row
      if (c != 15 && (c & 3) == 2) acc += gvc[M];
      c = cdc[O];                                            // window below: p is its top row
      if (c != 15 && (c & 3) == 0) acc += gvc[O];
      if (cdl != 15 && (cdl >> 2) == 2) acc += ghl;          // window to the left: p is its right column
      if (cdr != 15 && (cdr >> 2) == 0) acc += ghr;          // window to the right: p is its left column
      if (L0) {
        const float raw = rawc[M];
        const float t = clamp01(c_yt);
        const float direct = gout * (k1 * t + k2 + k3 * c_skt);
        acc = (raw >= 0.f && raw <= 1.f) ? acc + direct : 0.f;
      }
      if (core) a.ge_out[plane + (long long)y * W + x] = acc;        // Y0 <= y < Y1 by the loop bounds
    }
    q_e1 = n_e1; q_e = n_e; q_gs = n_gs; q_sp = n_sp; q_g1 = n_g1; q_yt = n_yt; q_skt = n_skt;
  };
  const int last = Y1 + 2;
  for (int r = Y0 - 1; r <= last; r += 3) {
    body(std::integral_constant<int, 0>{}, r);
    if (r + 1 <= last) body(std::integral_constant<int, 1>{}, r + 1);
    if (r + 2 <= last) body(std::integral_constant<int, 2>{}, r + 2);
  }
}

// ---- forward, strip walk over several levels at once -----------------------------------------------------------------
// NLEV consecutive levels k0 .. k0+NLEV-1 of the chain in one pass, as a register pipeline down a strip of 64 columns
// (NLEV+1 halo columns on either side, lane = column): with row t of E_{k0} just arrived, E_{k0+j} is evaluated for row
// t-j (j = 1..NLEV), delta and skel of level k0+j for row t-j-2.  The skeleton value of a pixel travels from level to
// level, one level per iteration.  Every operation is the one skel_level_kernel performs (min / max are exact, the
// skeleton update has the same two roundings), so the planes are bit-identical to those of the level-by-level kernels;
// what changes is the traffic: E_{k0} and skel_{k0-1} are read once, and only what a later pass needs is written
// (KEEP: every E and skel plane, for the backward pass; otherwise the last skeleton and, if asked for, the last E).
struct FwdArgs {
  const float* in;       // E_{k0}; the raw input when FIRST (clamped on the fly)
  const float* sk_in;    // skel_{k0-1}; unused when FIRST
  float* e_out;          // KEEP: plane set of E_{k0+1}, those of the following levels n floats apart;
                         // otherwise the plane set of E_{k0+NLEV}, or null
  float* sk_out;         // KEEP: plane set of skel_{k0}, the following n floats apart; otherwise that of skel_{k0+NLEV-1}
  long long n;           // floats per plane set
  int H, W, planes;
};

template <int NLEV, bool FIRST, bool KEEP>
__global__ __launch_bounds__(256) void skel_fwd_strip_kernel(FwdArgs a, int R, int nstrips, int nchunks) {
  constexpr int HALO = NLEV + 1, CORE = 64 - 2 * HALO;
  const int lane = threadIdx.x & 63;
  const unsigned per_xcd = gridDim.x >> 3;                 // XCD-contiguous strips, as in skel_bwd_strip_kernel
  const unsigned lblk = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  unsigned item = lblk * 4u + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (item >= (unsigned)a.planes * nstrips * nchunks) return;
  const int chunk = (int)(item % (unsigned)nchunks);
  item /= (unsigned)nchunks;
  const int strip = (int)(item % (unsigned)nstrips);
  const long long plane = (long long)(item / (unsigned)nstrips) * a.H * a.W;
  const int H = a.H, W = a.W;
  const int Y0 = chunk * R, Y1 = (Y0 + R < H) ? Y0 + R : H;
  const int x = strip * CORE - HALO + lane;
  const bool inx = x >= 0 && x < W;
  const bool core = inx && lane >= HALO && lane < 64 - HALO;
  const long long col = plane + (inx ? x : 0);
  const float* __restrict__ p_in = a.in + col;
  const float* __restrict__ p_sk = FIRST ? nullptr : a.sk_in + col;
  auto ld_raw = [&](const float* __restrict__ p, int y) -> float {
    const bool ok = (y >= 0) & (y < H);
    return p[(ok ? y : 0) * W];
  };
  auto rowin = [&](int y) -> bool { return (y >= 0) & (y < H); };
  // rings of three rows, slot = row mod 3 in terms of the iteration's phase (see skel_bwd_strip_kernel)
  float ec[NLEV + 1][3];        // E_{k0+j}, +inf outside the image (an erosion ignores it)
  float hm[NLEV + 1][3];        // max over (x-1, x, x+1) of E_{k0+j}, -inf outside the image (a dilation ignores it)
  float skc[NLEV];              // skel_{k0+j} of the row level j handled in the previous iteration
#pragma unroll
  for (int j = 0; j <= NLEV; ++j)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      ec[j][q] = INFINITY;
      hm[j][q] = -INFINITY;
    }
#pragma unroll
  for (int j = 0; j < NLEV; ++j) skc[j] = 0.f;
  const int t0 = Y0 - HALO, t1 = Y1 - 1 + HALO;
  float q_in = ld_raw(p_in, t0);
  float q_sk = FIRST ? 0.f : ld_raw(p_sk, t0 - 2);

  auto body = [&](auto phase, int t) {
    constexpr int PH = decltype(phase)::value;
    const float n_in = ld_raw(p_in, t + 1);
    const float n_sk = FIRST ? 0.f : ld_raw(p_sk, t - 1);
    {
      const float v = FIRST ? clamp01(q_in) : q_in;
      ec[0][PH % 3] = (rowin(t) & inx) ? v : INFINITY;
    }
    float below = (rowin(t - 2) & inx) ? q_sk : 0.f;      // skel of the level below for the row level 0 handles now
#pragma unroll
    for (int j = 0; j < NLEV; ++j) {
      // E_{j}: rows t-j (slot N), t-j-1 (M), t-j-2 (O)
      // (constants once the loop is unrolled)
      const int N = ((PH - j) % 3 + 3) % 3, M = (N + 2) % 3, O = (N + 1) % 3;
      const int N1 = ((PH - j - 1) % 3 + 3) % 3;          // where E_{j+1} keeps its new row t-j-1
      const float mid = ec[j][M];
      float e1 = fminf(fminf(ec[j][O], ec[j][N]), fminf(mid, fminf(from_left(mid), from_right(mid))));
      const int r1 = t - j - 1;
      const bool in1 = rowin(r1) & inx;
      e1 = in1 ? e1 : INFINITY;
      ec[j + 1][N1] = e1;
      const float d = in1 ? e1 : -INFINITY;
      hm[j + 1][N1] = fmaxf(d, fmaxf(from_left(d), from_right(d)));
      const bool st1 = core & (r1 >= Y0) & (r1 < Y1);
      if (KEEP) {
        if (st1) a.e_out[(long long)j * a.n + plane + (long long)r1 * W + x] = e1;
      } else if (j == NLEV - 1 && a.e_out) {
        if (st1) a.e_out[plane + (long long)r1 * W + x] = e1;
      }
      // level j's delta and skeleton for row y = t-j-2
      const int y = t - j - 2;
      const float D = fmaxf(fmaxf(hm[j + 1][0], hm[j + 1][1]), hm[j + 1][2]);
      const float delta = fmaxf(ec[j][O] - D, 0.f);
      float sv = delta;
      if (!(FIRST && j == 0)) {
        const float prod = below * delta;
        sv = below + fmaxf(delta - prod, 0.f);
      }
      below = skc[j];            // what level j made in the previous iteration: the row level j+1 handles now
      skc[j] = sv;
      const bool st = core & (y >= Y0) & (y < Y1);
      if (KEEP) {
        if (st) a.sk_out[(long long)j * a.n + plane + (long long)y * W + x] = sv;
      } else if (j == NLEV - 1) {
        if (st) a.sk_out[plane + (long long)y * W + x] = sv;
      }
    }
    q_in = n_in;
    q_sk = n_sk;
  };
  for (int t = t0; t <= t1; t += 3) {
    body(std::integral_constant<int, 0>{}, t);
    if (t + 1 <= t1) body(std::integral_constant<int, 1>{}, t + 1);
    if (t + 2 <= t1) body(std::integral_constant<int, 2>{}, t + 2);
  }
}

// ---- host entry points ----------------------------------------------------------------------------------------
static int grid_for(long long n) {
  long long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  return (int)b;
}

extern "C" int uh_soft_erode(int planes, int H, int W, const float* in, float* out, int clamp, uh_stream s) {
  UH_REQUIRE(planes > 0 && H > 0 && W > 0 && in && out, "soft_erode: bad arguments");
  const long long n = (long long)planes * H * W;
  if (clamp)
    hipLaunchKernelGGL((morph_kernel<true, false>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, planes, H, W, in, out);
  else
    hipLaunchKernelGGL((morph_kernel<false, false>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, planes, H, W, in, out);
  return uh_check_launch("soft_erode");
}

extern "C" int uh_soft_dilate(int planes, int H, int W, const float* in, float* out, uh_stream s) {
  UH_REQUIRE(planes > 0 && H > 0 && W > 0 && in && out, "soft_dilate: bad arguments");
  const long long n = (long long)planes * H * W;
  hipLaunchKernelGGL((morph_kernel<false, true>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, planes, H, W, in, out);
  return uh_check_launch("soft_dilate");
}

// E planes: e[0] = input (clamped on the fly), e[k] for k >= 1 at E + (k-1)*n; skel planes at SK + k*n.
// keep_all = 0: three rotating E buffers and one skeleton plane (E needs 3n floats, SK n floats).
static int skel_forward(int planes, int H, int W, const float* x, float* E, float* SK, int keep_all, hipStream_t st) {
  const long long n = (long long)planes * H * W;
  // UNET_AMD_SKEL_FWD=0: one launch per level (skel_level kernels below) instead of the whole chain in one strip walk;
  // both give the same bits.  UNET_AMD_SKEL_FWD_R: core rows per strip-walk wave.
  const char* env = getenv("UNET_AMD_SKEL_FWD");
  if ((!env || atoi(env) != 0) && (long long)H * W < (1ll << 31)) {
    constexpr int CORE = 64 - 2 * (SKEL_LEVELS + 1);
    const int nstrips = uh_cdiv(W, CORE);
    int R = 32;
    if (const char* e = getenv("UNET_AMD_SKEL_FWD_R")) R = atoi(e) > 0 ? atoi(e) : R;
    const int nchunks = uh_cdiv(H, R);
    const long long items = (long long)planes * nstrips * nchunks;
    if (items < (1ll << 32)) {
      const dim3 sgrid((unsigned)(((items + 3) / 4 + 7) / 8 * 8));
      FwdArgs a;
      a.in = x;
      a.sk_in = nullptr;
      a.e_out = keep_all ? E : nullptr;
      a.sk_out = SK;
      a.n = n;
      a.H = H;
      a.W = W;
      a.planes = planes;
      if (keep_all)
        hipLaunchKernelGGL((skel_fwd_strip_kernel<SKEL_LEVELS, true, true>), sgrid, dim3(256), 0, st, a, R, nstrips, nchunks);
      else
        hipLaunchKernelGGL((skel_fwd_strip_kernel<SKEL_LEVELS, true, false>), sgrid, dim3(256), 0, st, a, R, nstrips, nchunks);
      return uh_check_launch("soft_skel");
    }
  }
  const int grid = grid_for(n);
  // four pixels per thread when rows are multiples of 4 pixels, every plane set is 16-byte aligned and the
  // item count fits 32 bits
  const bool vec4 = (W % 4 == 0) && n / 4 < 0x7fffffffLL && (((uintptr_t)x | (uintptr_t)E | (uintptr_t)SK) & 15) == 0;
  const int grid4 = (int)((n / 4 + 255) / 256);
  auto eptr = [&](int k) -> float* { return keep_all ? E + (long long)(k - 1) * n : E + (long long)((k - 1) % 3) * n; };
  auto sptr = [&](int k) -> float* { return keep_all ? SK + (long long)k * n : SK; };
  hipLaunchKernelGGL((morph_kernel<true, false>), dim3(grid), dim3(256), 0, st, planes, H, W, x, eptr(1));
  for (int k = 0; k < SKEL_LEVELS; ++k) {
    const float* ek = (k == 0) ? x : eptr(k);
    float* ek2 = (k + 1 < SKEL_LEVELS) ? eptr(k + 2) : nullptr;
    const float* sp = (k == 0) ? nullptr : sptr(k - 1);
    if (vec4) {
      if (k == 0)
        hipLaunchKernelGGL((skel_level_v4_kernel<true>), dim3(grid4), dim3(256), 0, st, planes, H, W, ek, eptr(k + 1), sp, sptr(k), ek2);
      else
        hipLaunchKernelGGL((skel_level_v4_kernel<false>), dim3(grid4), dim3(256), 0, st, planes, H, W, ek, eptr(k + 1), sp, sptr(k), ek2);
    } else if (k == 0) {
      hipLaunchKernelGGL((skel_level_kernel<true>), dim3(grid), dim3(256), 0, st, planes, H, W, ek, eptr(k + 1), sp, sptr(k), ek2);
    } else {
      hipLaunchKernelGGL((skel_level_kernel<false>), dim3(grid), dim3(256), 0, st, planes, H, W, ek, eptr(k + 1), sp, sptr(k), ek2);
    }
  }
  return uh_check_launch("soft_skel");
}

extern "C" size_t uh_soft_skel_workspace(int planes, int H, int W) {
  return (size_t)planes * H * W * 3 * sizeof(float);
}

extern "C" int uh_soft_skel(int planes, int H, int W, const float* x, float* skel, float* workspace, uh_stream s) {
  UH_REQUIRE(planes > 0 && H > 0 && W > 0 && x && skel && workspace, "soft_skel: bad arguments");
  return skel_forward(planes, H, W, x, workspace, skel, 0, (hipStream_t)s);
}

// workspace layout (floats), n = planes*H*W:
//   E_pred[11n] | SK_pred[11n] | skel_true[n] | scratch[4n] | partial[7*SUM_BLOCKS] | coef[32]
#define SUM_BLOCKS 1024
static size_t cldice_ws_floats(long long n) { return (size_t)(27 * n + 7 * SUM_BLOCKS + 32); }

extern "C" size_t uh_cldice_workspace(int planes, int H, int W) {
  return cldice_ws_floats((long long)planes * H * W) * sizeof(float);
}

// with_true: the soft skeleton of y_true is made here; otherwise uh_cldice_skel_true has left it in the workspace
static int cldice_forward_impl(const char* what, int planes, int H, int W, const float* y_true, const float* y_pred,
                               float alpha, float smooth, float* workspace, float* loss_out, bool with_true, uh_stream s) {
  UH_REQUIRE(planes > 0 && H > 0 && W > 0 && y_true && y_pred && workspace && loss_out, "%s: bad arguments", what);
  hipStream_t st = (hipStream_t)s;
  const long long n = (long long)planes * H * W;
  float* E = workspace;
  float* SK = E + 11 * n;
  float* SKT = SK + 11 * n;
  float* scratch = SKT + n;
  float* partial = scratch + 4 * n;
  float* coef = partial + 7 * SUM_BLOCKS;
  int rc = skel_forward(planes, H, W, y_pred, E, SK, 1, st);
  if (rc) return rc;
  if (with_true) {
    rc = skel_forward(planes, H, W, y_true, scratch, SKT, 0, st);
    if (rc) return rc;
  }
  int blocks = grid_for(n);
  if (blocks > SUM_BLOCKS) blocks = SUM_BLOCKS;
  hipLaunchKernelGGL(cldice_sums_kernel, dim3(blocks), dim3(256), 0, st, n, y_true, y_pred, SK + 10 * n, SKT, partial);
  hipLaunchKernelGGL(cldice_finalize_kernel, dim3(1), dim3(256), 0, st, partial, blocks, alpha, smooth, coef, loss_out);
  return uh_check_launch(what);
}

extern "C" int uh_cldice_forward(int planes, int H, int W, const float* y_true, const float* y_pred, float alpha,
                                 float smooth, float* workspace, float* loss_out, uh_stream s) {
  return cldice_forward_impl("cldice_forward", planes, H, W, y_true, y_pred, alpha, smooth, workspace, loss_out, true, s);
}

// The half of uh_cldice_forward that needs no prediction: the soft skeleton of y_true, left in the workspace (it touches
// only the part of the workspace that the prediction's skeleton does not, so it may run on another stream beside
// whatever produces y_pred); uh_cldice_forward_pred, ordered after it, then does the rest.  Together: the same bits as
// uh_cldice_forward.
extern "C" int uh_cldice_skel_true(int planes, int H, int W, const float* y_true, float* workspace, uh_stream s) {
  UH_REQUIRE(planes > 0 && H > 0 && W > 0 && y_true && workspace, "cldice_skel_true: bad arguments");
  const long long n = (long long)planes * H * W;
  float* SKT = workspace + 22 * n;
  return skel_forward(planes, H, W, y_true, SKT + n, SKT, 0, (hipStream_t)s);
}

extern "C" int uh_cldice_forward_pred(int planes, int H, int W, const float* y_true, const float* y_pred, float alpha,
                                      float smooth, float* workspace, float* loss_out, uh_stream s) {
  return cldice_forward_impl("cldice_forward_pred", planes, H, W, y_true, y_pred, alpha, smooth, workspace, loss_out,
                             false, s);
}

extern "C" int uh_cldice_backward(int planes, int H, int W, const float* y_true, const float* y_pred,
                                  float* workspace, const float* grad_out, float* grad_pred, uh_stream s) {
  UH_REQUIRE(planes > 0 && H > 0 && W > 0 && y_true && y_pred && workspace && grad_out && grad_pred,
             "cldice_backward: bad arguments");
  hipStream_t st = (hipStream_t)s;
  const long long n = (long long)planes * H * W;
  float* E = workspace;
  float* SK = E + 11 * n;
  float* SKT = SK + 11 * n;
  float* scratch = SKT + n;
  float* coef = scratch + 4 * n + 7 * SUM_BLOCKS;
  float* gs[2] = {scratch, scratch + n};
  float* ge[2] = {scratch + 2 * n, scratch + 3 * n};
  const dim3 grid(uh_cdiv(W, BT), uh_cdiv(H, BT), planes);
  // UNET_AMD_SKEL_TILES=1: the LDS-tile kernel (one workgroup per 32x32 tile) instead of the strip walk; both give the
  // same bits.  UNET_AMD_SKEL_R: core rows per strip-walk wave.
  const char* env = getenv("UNET_AMD_SKEL_TILES");
  const bool tiles = env && atoi(env) != 0;
  const int nstrips = uh_cdiv(W, SK_CORE);
  int R = 0;
  if (const char* e = getenv("UNET_AMD_SKEL_R")) R = atoi(e);
  if (R <= 0) {
    // A wave walks R + 4 rows, so longer strips waste less; but a wave's rows follow one another through ~250 dependent
    // instructions each, and it takes about five waves per SIMD (5120 on the chip) to keep the ALUs issuing.  Measured
    // per loss call at 32 x 256 x 256: R = 8 (5120 waves) 385-392 us, 16 (2560) 421-426, 32 (1280) 487, 64 (640) 635;
    // at 4 x 1024 x 1024: R = 8 750 us, 16 727, 32 878.
    R = ((long long)planes * nstrips * uh_cdiv(H, 16) >= 5120) ? 16 : 8;
  }
  const int nchunks = uh_cdiv(H, R);
  const long long items = (long long)planes * nstrips * nchunks;
  UH_REQUIRE(items < (1ll << 32) && (long long)H * W < (1ll << 31), "cldice_backward: %lld strips of a %d x %d plane",
             items, H, W);
  const dim3 sgrid((unsigned)(((items + 3) / 4 + 7) / 8 * 8));     // four strips per workgroup, eight workgroups a round
  for (int k = SKEL_LEVELS - 1; k >= 0; --k) {
    BwdArgs a;
    a.ek = (k == 0) ? y_pred : E + (long long)(k - 1) * n;
    a.ek1 = E + (long long)k * n;
    a.skel_prev = (k == 0) ? nullptr : SK + (long long)(k - 1) * n;
    a.top = (k == SKEL_LEVELS - 1);
    a.level0 = (k == 0);
    a.gs_in = a.top ? nullptr : gs[k & 1];
    a.ge1_in = a.top ? nullptr : ge[k & 1];
    a.gs_out = (k == 0) ? nullptr : gs[(k - 1) & 1];
    a.ge_out = (k == 0) ? grad_pred : ge[(k - 1) & 1];
    a.yt = y_true;
    a.skt = SKT;
    a.coef = coef;
    a.gout = grad_out;
    a.H = H;
    a.W = W;
    if (tiles)
      hipLaunchKernelGGL(skel_bwd_level_kernel, grid, dim3(256), 0, st, a);
    else if (a.top)
      hipLaunchKernelGGL((skel_bwd_strip_kernel<true, false>), sgrid, dim3(256), 0, st, a, planes, R, nstrips, nchunks);
    else if (a.level0)
      hipLaunchKernelGGL((skel_bwd_strip_kernel<false, true>), sgrid, dim3(256), 0, st, a, planes, R, nstrips, nchunks);
    else
      hipLaunchKernelGGL((skel_bwd_strip_kernel<false, false>), sgrid, dim3(256), 0, st, a, planes, R, nstrips, nchunks);
  }
  return uh_check_launch("cldice_backward");
}

// plain soft_dice (cl_dice.py:73-94) from the same sums, for reporting
extern "C" int uh_cldice_read_terms(const float* workspace, int planes, int H, int W, float* out16, uh_stream s) {
  UH_REQUIRE(workspace && out16, "cldice_read_terms: bad arguments");
  const long long n = (long long)planes * H * W;
  const float* coef = workspace + 27 * n + 7 * SUM_BLOCKS;
  hipError_t e = hipMemcpyAsync(out16, coef, 17 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)s);
  if (e != hipSuccess) {
    uh_set_error("cldice_read_terms: %s", hipGetErrorString(e));
    return UH_ERR_HIP;
  }
  return UH_OK;
}
