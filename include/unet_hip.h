/* libunet_hip.so -- C ABI of the gfx950 (MI355X) kernels for the SaviorDT/UNet hot path.
 *
 * The reference (a pure-PyTorch repository) has no FFI of its own: the operations below are
 * the ATen calls its nn.Module / loss code dispatches, cited per entry point as
 * reference file:line.  Every function
 *   - takes plain device pointers and sizes (no framework types),
 *   - allocates nothing, never synchronises the device, and keeps no state between calls except two mutex-guarded,
 *     per-device records (which kernels have had their dynamic-LDS limit raised; the address of the scratch array of
 *     uh_bce_logits_noalloc), so one process may drive several GPUs from several threads,
 *   - enqueues its kernels on `stream` (a hipStream_t passed as void*),
 *   - returns UH_OK or a negative uh_status; uh_last_error() gives the thread-local message.
 *
 * Activation layout ("padded-linear NHWC", bf16): a logical [B][H][W][C] tensor is stored as
 * Mp = B*(H+1)*(W+1) rows of C channels, row m = (b*(H+1)+y)*(W+1)+x; rows with y==H or x==W
 * are zero, and uh_guard_rows(W) guard rows precede row 0 and follow row Mp-1 (zero on entry; kernels read them and
 * may rewrite them with zeros).  Pointers passed
 * here address row 0.  `cstride`/`coff` select a channel window of a wider buffer (skip concat).
 */
#ifndef UNET_HIP_H
#define UNET_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum { UH_OK = 0, UH_ERR_SHAPE = -1, UH_ERR_HIP = -2, UH_ERR_UNSUPPORTED = -3 } uh_status;
typedef void* uh_stream;

const char* uh_last_error(void);
int uh_version(void);
/* zero guard rows required on each side of a padded-linear buffer of image width W */
int uh_guard_rows(int W);
/* rowmask[m] = 1 for interior rows, 0 for pad rows; m in [0, Mp).  Kernels read the mask four rows at a time:
 * the buffer must be 4-byte aligned and readable up to the next multiple of 4 bytes (contents there are ignored). */
int uh_rowmask_build(int B, int H, int W, uint8_t* rowmask, uh_stream s);

/* ---- weight packing -------------------------------------------------------------------
 * Generic fp32 -> bf16 repack: dst[n*on + t*ot + k] = src[n*sn + tmap(t)*st + k*sk],
 * tmap(t) = reverse ? T-1-t : t.  Turns OIHW conv weights (unet.py:19-76), [Cin][Cout][2][2]
 * transposed-conv weights (unet.py:51-72) and 2x2/1x1 conv weights (lunext.py:67-185) into
 * the [N][T][K] GEMM operand the MFMA kernels read, for forward and for dgrad (reverse=1
 * gives the 180-degree rotated filter). */
int uh_pack_weights(const float* src, void* dst_bf16, int N, int T, int K,
                    long long sn, long long st, long long sk, long long on, long long ot,
                    int reverse, uh_stream s);

/* The same repack for `count` tensors in one launch; the descriptors live in DEVICE memory (all fields 8 bytes).
 * reverse: bit 0 = rotated taps as above; bit 1 = the destination holds fp32 instead of bf16 (zero-padded fp32 parameter
 * tables of the 8-channel kernels, e.g. the [oc][8] head weight, made in the launch that repacks the GEMM operands). */
typedef struct {
  const float* src;
  void* dst_bf16;
  long long N, T, K, reverse, sn, st, sk, on, ot;
} uh_pack_desc;
int uh_pack_weights_multi(const uh_pack_desc* descs_device, int count, int blocks_per_desc, uh_stream s);

/* ---- implicit-GEMM convolution (MFMA bf16, fp32 accumulate) ----------------------------
 * out[m][n] = sum_{t<T} sum_{k<K} a[m + off(t)][k] * w[n][t][k]   (T = 9: off = (ty-1)*(W+1)+(tx-1); T = 1: off = 0)
 * Replaces nn.Conv2d 3x3 pad 1 forward (unet.py:99-169), its input gradient (autograd of the same,
 * with rotated weights), nn.ConvTranspose2d k2 s2 (unet.py:132,144,154,164; lunext.py:223-249) as a
 * T=1 GEMM with a pixel-shuffle store, and 1x1 / 2x2-stride-2 convs after space-to-depth (lunext.py).
 * K and N must be multiples of 64.  Epilogue modes: */
enum {
  UH_EPI_RAW_STATS = 0,   /* out = fp16(acc + bias) (IEEE half, same 2-byte rows), zero on pad rows; per-M-tile channel sums of v and v*v
                             over interior rows -> stat_partial[mtile][2][N]   (train-mode BN input) */
  UH_EPI_AFFINE_RELU = 1, /* out = relu(acc*scale[n] + shift[n]) on interior rows, 0 on pads (eval-mode BN folded) */
  UH_EPI_MASKED = 2,      /* out = acc (+bias if given) on interior rows, 0 on pads (dgrad, plain convs) */
  UH_EPI_CONVT = 3        /* n = tap*Cout+co: out[(b,2y+tap/2,2x+tap%2)][co] = acc + bias[co]; Cout = min(N/4, o_cstride-o_coff)
                             (weight rows beyond 4*Cout are zero padding and are skipped);
                             out has geometry (B,2H,2W); pad rows of out are written as zero */
  ,
  UH_EPI_MASKED_BNBWD = 4 /* uh_conv_gemm_bnbwd only: out as UH_EPI_MASKED, plus the BatchNorm-backward sums of the
                             layer that consumes out (see there) */
};
int uh_conv_gemm(int B, int H, int W,
                 const void* a, int a_cstride, int a_coff, int K,
                 const void* w_packed, int N, int T, int mode,
                 void* out, int o_cstride, int o_coff,
                 const float* bias, const float* scale, const float* shift,
                 float* stat_partial, const uint8_t* rowmask, uh_stream s);
/* Same, with scratch memory that lets the T = 9 kernel cut its work evenly over the CUs even when the number of
 * output tiles is not a multiple of the CU count (pieces of tiles are summed in a second launch).  Results do
 * not depend on the cut.  workspace: 16-byte aligned, uh_conv_gemm_workspace() bytes; NULL or too small = whole
 * tiles per workgroup, as uh_conv_gemm. */
size_t uh_conv_gemm_workspace(int B, int H, int W, int K, int N, int T);
int uh_conv_gemm_ws(int B, int H, int W,
                    const void* a, int a_cstride, int a_coff, int K,
                    const void* w_packed, int N, int T, int mode,
                    void* out, int o_cstride, int o_coff,
                    const float* bias, const float* scale, const float* shift,
                    float* stat_partial, const uint8_t* rowmask,
                    void* workspace, size_t workspace_bytes, uh_stream s);
/* Input gradient of a 3x3 convolution (T = 9, as uh_conv_gemm with UH_EPI_MASKED and no bias) whose result `out` is
 * the gradient with respect to relu(bn(y)) of the PRECEDING conv+BatchNorm+ReLU layer (unet.py:99-169: conv x_2 reads
 * conv x_1's activation).  While a tile of out is still in the accumulators, that layer's BatchNorm-backward sums are
 * taken from it: with d = out * (y*scale + shift > 0),
 *   stat_partial[mtile][0][n] = sum_rows d,   stat_partial[mtile][1][n] = sum_rows d * (y - mean) * invstd
 * (uh_conv_gemm_mtiles(B,H,W,N) rows: what uh_bn_relu_bwd_reduce would compute in a pass of its own, ready for
 * uh_bn_bwd_finalize).  y: the layer's fp16 conv output (N channels in a window of y_cstride). */
int uh_conv_gemm_bnbwd(int B, int H, int W, const void* a, int a_cstride, int a_coff, int K, const void* w_packed,
                       int N, void* out, int o_cstride, int o_coff, const uint8_t* rowmask, const void* y,
                       int y_cstride, int y_coff, const float* mean, const float* invstd, const float* scale,
                       const float* shift, float* stat_partial, uh_stream s);
/* the same for a T = 1 GEMM: the input gradient of a k2 s2 transposed convolution (unet.py:132-164 + autograd), whose
 * output is the gradient at the output of conv_up{k-1}_2 / conv5_2 */
int uh_conv_gemm1_bnbwd(int B, int H, int W, const void* a, int a_cstride, int a_coff, int K, const void* w_packed,
                       int N, void* out, int o_cstride, int o_coff, const uint8_t* rowmask, const void* y,
                       int y_cstride, int y_coff, const float* mean, const float* invstd, const float* scale,
                       const float* shift, float* stat_partial, uh_stream s);

/* number of M tiles (= rows of stat_partial) uh_conv_gemm uses for this problem */
int uh_conv_gemm_mtiles(int B, int H, int W, int N);

/* ---- weight gradient (MFMA bf16, fp32 accumulate, split over pixels) --------------------
 * partial[s][n][t][k] = sum_{m in split s} dy[m][n] * a[m + off(t)][k]; then
 * grad[n*sn + t*st + k*sk] = sum_s partial[s][n*on + t*ot + k].
 * Replaces the weight-gradient half of convolution_backward for every conv above. */
int uh_conv_wgrad_splits(int B, int H, int W, int N, int K, int T);
size_t uh_conv_wgrad_workspace(int B, int H, int W, int N, int K, int T);
int uh_conv_wgrad(int B, int H, int W,
                  const void* dy, int dy_cstride, int dy_coff, int N,
                  const void* a, int a_cstride, int a_coff, int K, int T,
                  float* partial, uh_stream s);
int uh_wgrad_reduce(const float* partial, int S, int N, int T, int K,
                    long long on, long long ot, long long sn, long long st, long long sk,
                    float* grad, int accumulate, uh_stream s);
/* same, with an explicit distance between split slabs (partials whose rows are zero-padded beyond N*T*K) */
int uh_wgrad_reduce_strided(const float* partial, int S, long long split_stride, int N, int T, int K,
                            long long on, long long ot, long long sn, long long st, long long sk,
                            float* grad, int accumulate, uh_stream s);

/* ---- first layer: Conv2d(Cin<=4 -> Cout, 3x3, pad 1) on NCHW fp32 input (unet.py:99) ------ */
int uh_conv_first_fwd(int B, int H, int W, int Cin, int Cout,
                      const float* x_nchw, const float* w_oihw, const float* bias,
                      int mode, const float* scale, const float* shift,
                      void* out, int o_cstride, int o_coff, float* stat_partial, uh_stream s);
int uh_conv_first_parts(int B, int H, int W);
int uh_conv_first_wgrad(int B, int H, int W, int Cin, int Cout,
                        const float* x_nchw, const void* dy, int dy_cstride, int dy_coff,
                        float* partial, uh_stream s);     /* partial[parts][Cout][Cin*9] */
int uh_conv_first_wgrad_parts(int B, int H, int W);

/* ---- BatchNorm2d (train) + ReLU (unet.py:99-169, defaults eps 1e-5 momentum 0.1) --------- */
/* stats: partial[P][2][C] -> mean, invstd, scale = gamma*invstd, shift = beta - mean*scale;
 * running_mean/var (unbiased var) and num_batches_tracked updated in place when non-null.
 * c_real <= C: gamma, beta and the running statistics hold c_real entries; channels from c_real on are layout
 * padding (scale = shift = 0). */
int uh_bn_finalize(const float* partial, int P, int C, int c_real, long long count,
                   const float* gamma, const float* beta, float eps, float momentum,
                   float* mean, float* invstd, float* scale, float* shift,
                   float* running_mean, float* running_var, long long* num_batches_tracked, uh_stream s);
/* eval mode: scale = gamma/sqrt(rv+eps); shift = beta + (conv_bias - rm)*scale */
int uh_bn_fold_eval(int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                    const float* conv_bias, float eps, float* scale, float* shift, uh_stream s);
/* a = relu(y*scale + shift) on interior rows, 0 on pads; y is the fp16 RAW_STATS output, a is bf16 */
int uh_bn_relu_apply(long long Mp, int C, const void* y, int y_cstride, int y_coff,
                     const float* scale, const float* shift, const uint8_t* rowmask,
                     void* a, int a_cstride, int a_coff, uh_stream s);
/* backward of relu(bn(y)):  dyh = da * (y*scale+shift > 0); partial[P][2][C] = (sum dyh, sum dyh*xhat) */
int uh_bn_bwd_parts(long long Mp, int C);
int uh_bn_relu_bwd_reduce(long long Mp, int C, const void* da, int da_cstride, int da_coff,
                          const void* y, int y_cstride, int y_coff,
                          const float* mean, const float* invstd, const float* scale, const float* shift,
                          float* partial, uh_stream s);
/* dgamma = sum dyh*xhat, dbeta = sum dyh (c_real entries each), coef[0][c] = dbeta/count, coef[1][c] = dgamma/count
 * (C entries each) */
int uh_bn_bwd_finalize(const float* partial, int P, int C, int c_real, long long count,
                       float* dgamma, float* dbeta, float* coef, int accumulate, uh_stream s);

/* ---- the same sums, finalised inside the launch that takes them ------------------------------------------------------
 * The last workgroups to finish add up the per-workgroup rows (fixed order, fp64: run-independent results) and
 * evaluate the per-channel result, so no finalisation launch follows.  ws: uh_fin_ws_bytes(parts, C, 2) bytes, 8-byte
 * aligned, `parts` the workgroup count of the launch (uh_bn_bwd_parts / uh_dw_parts); counters: uh_fin_counters()
 * unsigned ints, zero before the first such launch on a stream; every launch leaves them zero.  Launches that share
 * counters must be ordered on one stream. */
size_t uh_fin_ws_bytes(int parts, int C, int nstat);
int uh_fin_counters(void);
/* uh_bn_relu_bwd_reduce + uh_bn_bwd_finalize (autograd of BatchNorm2d + ReLU, unet.py:19-76) */
int uh_bn_relu_bwd_reduce_fin(long long Mp, int C, const void* da, int da_cstride, int da_coff, const void* y,
                              int y_cstride, int y_coff, const float* mean, const float* invstd, const float* scale,
                              const float* shift, int c_real, long long count, float* dgamma, float* dbeta, float* coef,
                              void* ws, size_t ws_bytes, unsigned* counters, uh_stream s);
/* uh_bn_res_relu_bwd_reduce + uh_bn_bwd_finalize (autograd of DepthwiseConvBlock's BatchNorm + residual + ReLU,
 * lunext.py:19-30) */
int uh_bn_res_relu_bwd_reduce_fin(long long Mp, int C, const void* da, int da_cstride, int da_coff, const void* y,
                                  int y_cstride, int y_coff, const void* x, int x_cstride, int x_coff,
                                  const float* mean, const float* invstd, const float* scale, const float* shift,
                                  int c_real, long long count, float* dgamma, float* dbeta, float* coef, void* ws,
                                  size_t ws_bytes, unsigned* counters, uh_stream s);
/* uh_dwconv3x3_fwd in train mode + uh_bn_finalize (lunext.py:14-21: depthwise conv, then BatchNorm's batch statistics,
 * scale / shift and running-statistics update) */
int uh_dwconv3x3_fwd_bn(int B, int H, int W, int C, const void* x, int x_cstride, int x_coff, const float* w, int c_real,
                        const float* bias, const uint8_t* rowmask, void* out, int o_cstride, int o_coff,
                        const float* gamma, const float* beta, float eps, float momentum, float* mean, float* invstd,
                        float* scale, float* shift, float* running_mean, float* running_var, long long* nbt, void* ws,
                        size_t ws_bytes, unsigned* counters, uh_stream s);
/* dy = scale * (dyh - coef0 - xhat*coef1) on interior rows, 0 on pads */
int uh_bn_relu_bwd_apply(long long Mp, int C, const void* da, int da_cstride, int da_coff,
                         const void* y, int y_cstride, int y_coff,
                         const float* mean, const float* invstd, const float* scale, const float* shift,
                         const float* coef, const uint8_t* rowmask,
                         void* dy, int dy_cstride, int dy_coff, uh_stream s);

/* ---- MaxPool2d(2,2) (unet.py:103-124) ---------------------------------------------------- */
int uh_maxpool2_fwd(int B, int H, int W, int C, const void* a, int a_cstride, int a_coff,
                    void* out, int o_cstride, int o_coff, uh_stream s);
/* uh_bn_relu_apply followed by uh_maxpool2_fwd in one pass over y (the encoder's conv{l}_2 -> pool{l}, unet.py:100-124):
 * a = relu(y*scale + shift) (pads zero) and p = MaxPool2d(2,2)(a).  C = 8 * a power of two. */
int uh_bn_relu_apply_pool2(int B, int H, int W, int C, const void* y, int y_cstride, int y_coff,
                           const float* scale, const float* shift,
                           void* a, int a_cstride, int a_coff, void* p, int p_cstride, int p_coff, uh_stream s);
/* da[m] = (dp routed to the FIRST maximum of the window, row-major order) + (add ? add[m] : 0) */
int uh_maxpool2_bwd(int B, int H, int W, int C, const void* a, int a_cstride, int a_coff,
                    const void* dp, int dp_cstride, int dp_coff,
                    const void* add, int add_cstride, int add_coff,
                    void* da, int da_cstride, int da_coff, uh_stream s);

/* uh_maxpool2_bwd where da is the gradient at the output of a conv + BatchNorm + ReLU layer (the encoder's conv{l}_2,
 * whose activation was pooled, unet.py:100-124): also leaves that layer's BatchNorm-backward sums, as
 * uh_bn_relu_bwd_reduce would compute them from the stored da and the layer's fp16 conv output y, in
 * partial[uh_maxpool2_bwd_bnstat_parts()][2][C].  C = 8 * a power of two. */
int uh_maxpool2_bwd_bnstat_parts(int B, int H, int W, int C);
int uh_maxpool2_bwd_bnstat(int B, int H, int W, int C, const void* a, int a_cstride, int a_coff,
                           const void* dp, int dp_cstride, int dp_coff,
                           const void* add, int add_cstride, int add_coff,
                           void* da, int da_cstride, int da_coff,
                           const void* y, int y_cstride, int y_coff,
                           const float* mean, const float* invstd, const float* scale, const float* shift,
                           float* partial, uh_stream s);

/* ---- space-to-depth 2x2: out[(b,y,x)][t*C + c] = in[(b,2y+t/2,2x+t%2)][c]; in geometry (B,2H,2W) ----
 * Turns the gradient of a k2 s2 transposed conv, and the input of a k2 s2 conv (lunext.py:139-154),
 * into a T=1 GEMM operand. */
int uh_space_to_depth2(int B, int H, int W, int C, const void* in, int i_cstride, int i_coff,
                       void* out, int o_cstride, int o_coff, uh_stream s);
/* The same pair with a large geometry of (B, bigH, bigW), bigH in {2H, 2H+1}, bigW in {2W, 2W+1}: a k2 s2 convolution
 * ignores the last row / column of an odd-sized input (lunext.py:139-154 on sizes not divisible by 16), and the
 * inverse scatter writes zeros there. */
int uh_space_to_depth2_of(int B, int H, int W, int C, int bigH, int bigW, const void* in, int i_cstride, int i_coff,
                          void* out, int o_cstride, int o_coff, uh_stream s);
int uh_depth_to_space2_to(int B, int H, int W, int C, int bigH, int bigW, const void* in, int i_cstride, int i_coff,
                          void* out, int o_cstride, int o_coff, uh_stream s);
/* uh_depth_to_space2_to plus a tensor of the large geometry (pads zero): the gradient that comes back through a k2 s2
 * convolution joined with the gradient of the skip connection taken before it (lunext.py:201-216 + autograd) */
int uh_depth_to_space2_add(int B, int H, int W, int C, int bigH, int bigW, const void* in, int i_cstride, int i_coff,
                           const void* add, int a_cstride, int a_coff, void* out, int o_cstride, int o_coff, uh_stream s);
/* inverse scatter: out[(b,2y+t/2,2x+t%2)][c] = in[(b,y,x)][t*C + c], pads of out zero */
int uh_depth_to_space2(int B, int H, int W, int C, const void* in, int i_cstride, int i_coff,
                       void* out, int o_cstride, int o_coff, uh_stream s);

/* ---- per-channel sum over rows: partial-free two-pass (bias gradients) ---------------------- */
int uh_colsum_parts(long long Mp, int C);
int uh_colsum(long long Mp, int C, const void* x, int cstride, int coff, float* partial, uh_stream s);
int uh_colsum_finalize(const float* partial, int P, int C, int fold, float* out, int accumulate, uh_stream s);
/* out[i] = sum_p partial[p][cols[i]], i < n_out (cols: n_out ints on the device, each in [0, C)): the sums of one pass
 * written straight into the layout of the parameter gradients they are (autograd's accumulation of lunext.py's padded
 * 3-channel level and of several biases summed in one pass), no copies behind it */
int uh_colsum_finalize_gather(const float* partial, int P, int C, int n_out, const int* cols, float* out, uh_stream s);
/* Up to UH_REDUCE_MAX of these small reductions in ONE launch (a LUNeXt backward pass has some thirty of them, nothing
 * but the optimizer waits for their results): descs is a HOST array, copied into the launch's arguments.
 *   UH_REDUCE_COLSUM: out[i] (+)= sum_p partial[p][cols ? cols[i] : i], i < n_out  (uh_colsum_finalize with fold 1,
 *                     uh_colsum_finalize_gather); rows = P
 *   UH_REDUCE_WGRAD:  uh_wgrad_reduce_strided's arguments; rows = S.  3x3 weight gradients (T = 9 in OIHW order) are not
 *                     taken: their reductions have kernels of their own. */
#define UH_REDUCE_MAX 24
enum { UH_REDUCE_COLSUM = 0, UH_REDUCE_WGRAD = 1 };
typedef struct {
  int kind, rows, accumulate;
  int C, n_out;                              /* column sums */
  int N, T, K;                               /* weight gradient */
  const float* partial;
  float* out;
  const int* cols;                           /* device array or NULL */
  long long split_stride, on, ot, sn, st, sk;
} uh_reduce_desc;
int uh_reduce_multi(const uh_reduce_desc* descs, int count, uh_stream s);

/* ---- 1x1 segmentation head Conv2d(C -> OC) (unet.py:173; lunext.py:257), fp32 NCHW logits ------ */
int uh_head_fwd(int B, int H, int W, int C, int OC, const void* a, int a_cstride, int a_coff,
                const float* w, const float* bias, float* logits_nchw, uh_stream s);
/* The one-channel head in evaluation (trainer/k_fold.py:281-290, test_helper/utils.py:124-131): the launch that makes
 * the logits also binarises them (sigmoid(x) > 0.5, fp32) and adds tp, tn, fp, fn against `mask` ([B,1,H,W] fp32) to
 * counts[plane][4] (int64; one plane per image if per_sample, else one for the batch).  logits_nchw may be NULL when
 * only the counts are wanted.  Counts equal uh_confusion_counts over uh_head_fwd's logits exactly. */
int uh_head_fwd_counts(int B, int H, int W, int C, const void* a, int a_cstride, int a_coff, const float* w,
                       const float* bias, float* logits_nchw, const float* mask, int per_sample, long long* counts,
                       uh_stream s);
int uh_head_bwd_parts(int B, int H, int W, int C);
int uh_head_bwd(int B, int H, int W, int C, int OC, const void* a, int a_cstride, int a_coff,
                const float* w, const float* dlogits_nchw,
                void* da, int da_cstride, int da_coff, float* partial /*[P][OC][C+1]*/, uh_stream s);
/* uh_head_bwd that also leaves the BatchNorm-backward sums of the layer in front of the head (conv_up4_2,
 * unet.py:168-173) in bn_partial[uh_head_bwd_parts()][2][C], as uh_bn_relu_bwd_reduce would compute them */
int uh_head_bwd_bnstat(int B, int H, int W, int C, int OC, const void* a, int a_cstride, int a_coff,
                       const float* w, const float* dlogits_nchw,
                       void* da, int da_cstride, int da_coff, float* partial,
                       const void* y, int y_cstride, int y_coff,
                       const float* mean, const float* invstd, const float* scale, const float* shift,
                       float* bn_partial, uh_stream s);

/* ---- LUNeXt blocks (models/lunext.py), padded-linear NHWC bf16 with C a multiple of 8 ----------------
 * The 3-channel full-resolution level is carried zero-padded to 8 channels. */
/* rows of `partial` of the depthwise weight-gradient kernel, of uh_dwconv3x3_fwd_bn and of the eval-mode forward */
int uh_dw_parts(int B, int H, int W, int C);
/* rows of `partial` that uh_dwconv3x3_fwd writes in RAW_STATS mode (the training forward runs as a per-wave strip walk,
 * one row per strip and column block; UNET_AMD_DWFWD_TILES=1: the one-thread-per-column kernel, uh_dw_parts rows) */
int uh_dwconv3x3_fwd_parts(int B, int H, int W, int C);
/* depthwise Conv2d(C,C,3,pad 1,groups=C) (lunext.py:14-15,24).  w and bias are the module's own fp32 tensors
 * ([c_real][1][3][3] and [c_real], 16-byte aligned); channels c_real..C-1 of the activations are zero padding.
 * RAW_STATS: out = fp16(v) and partial[uh_dwconv3x3_fwd_parts][2][C] = (sum v, sum v^2); AFFINE_RELU:
 * out = relu(v*scale + shift + x) (eval-mode block; scale, shift of C floats). */
int uh_dwconv3x3_fwd(int B, int H, int W, int C, const void* x, int x_cstride, int x_coff,
                     const float* w, int c_real, const float* bias, int mode, const float* scale, const float* shift,
                     const uint8_t* rowmask, void* out, int o_cstride, int o_coff, float* partial, uh_stream s);
/* The training forward of the SECOND DepthwiseConvBlock of a pair (lunext.py:187-259 chains them two by two): its input
 * a = relu(scale_q * yq + shift_q + xq), the first block's BatchNorm + residual + ReLU (lunext.py:27-29, what
 * uh_bn_res_relu_apply computes), is evaluated here, written to `a` and convolved; out and partial as uh_dwconv3x3_fwd
 * in RAW_STATS mode.  Same bits as the two launches it replaces, without reading `a` back.
 * uh_dwconv3x3_fwd_takes_apply(C) is 1 when the kernel in use can do this for C channels. */
int uh_dwconv3x3_fwd_takes_apply(int C);
int uh_dwconv3x3_fwd_applied(int B, int H, int W, int C, const void* yq, int yq_cstride, int yq_coff,
                             const void* xq, int xq_cstride, int xq_coff, const float* scale_q, const float* shift_q,
                             void* a, int a_cstride, int a_coff, const float* w, int c_real, const float* bias,
                             void* out, int o_cstride, int o_coff, float* partial, uh_stream s);
/* a = relu(y*scale + shift + x)   (lunext.py:25-28) and its backward: d = da*(.. > 0);
 * reduce -> partial[P][2][C] (P = uh_bn_bwd_parts); apply -> dy = scale*(d - c0 - xhat*c1), dr = d */
int uh_bn_res_relu_apply(long long Mp, int C, const void* y, int y_cstride, int y_coff,
                         const void* x, int x_cstride, int x_coff, const float* scale, const float* shift,
                         const uint8_t* rowmask, void* a, int a_cstride, int a_coff, uh_stream s);
/* uh_bn_res_relu_apply for a block whose output also feeds the 2x2 stride-2 convolution of the next level
 * (lunext.py:139-154, 203-214): interior values are stored a second time in the space-to-depth layout that convolution's
 * GEMM reads, depth[(b, y/2, x/2)][d_coff + t*C + c], t = 2 (y & 1) + (x & 1) -- uh_space_to_depth2_of's result without
 * its pass.  Pad rows of `depth` are not written: they must be zero and stay zero (a buffer only this kernel and
 * uh_space_to_depth2_of write).  (B, H, W): the block's own geometry. */
int uh_bn_res_relu_apply_s2d(int B, int H, int W, int C, const void* y, int y_cstride, int y_coff,
                             const void* x, int x_cstride, int x_coff, const float* scale, const float* shift,
                             void* a, int a_cstride, int a_coff, void* depth, int d_cstride, int d_coff, uh_stream s);
int uh_bn_res_relu_bwd_reduce(long long Mp, int C, const void* da, int da_cstride, int da_coff,
                              const void* y, int y_cstride, int y_coff, const void* x, int x_cstride, int x_coff,
                              const float* mean, const float* invstd, const float* scale, const float* shift,
                              float* partial, uh_stream s);
int uh_bn_res_relu_bwd_apply(long long Mp, int C, const void* da, int da_cstride, int da_coff,
                             const void* y, int y_cstride, int y_coff, const void* x, int x_cstride, int x_coff,
                             const float* mean, const float* invstd, const float* scale, const float* shift,
                             const float* coef, const uint8_t* rowmask,
                             void* dy, int dy_cstride, int dy_coff, void* dr, int dr_cstride, int dr_coff, uh_stream s);
/* dx = dwconv(dy, flipped w) + dr;  partial[blk][9][C] = sum_m dy[m][c]*x[m+off(t)][c] (blk < uh_dw_parts) */
int uh_dwconv3x3_dgrad(int B, int H, int W, int C, const void* dy, int dy_cstride, int dy_coff,
                       const float* w, int c_real, const void* dr, int dr_cstride, int dr_coff,
                       const uint8_t* rowmask, void* dx, int dx_cstride, int dx_coff, uh_stream s);
int uh_dwconv3x3_wgrad(int B, int H, int W, int C, const void* dy, int dy_cstride, int dy_coff,
                       const void* x, int x_cstride, int x_coff, float* partial, uh_stream s);
/* AttentionGate pointwise parts (lunext.py:100-117); z = [W1 g+b1 | W2 g+b2 | W3 (g+x)+b3], 3C channels */
int uh_gate_fwd(long long Mp, int C, const void* z, int z_cstride, int z_coff, const void* x, int x_cstride,
                int x_coff, void* out, int o_cstride, int o_coff, uh_stream s);
int uh_gate_bwd(long long Mp, int C, const void* dout, int d_cstride, int d_coff, const void* z, int z_cstride,
                int z_coff, const void* x, int x_cstride, int x_coff, const uint8_t* rowmask,
                void* dz, int dz_cstride, int dz_coff, void* dxd, int dx_cstride, int dx_coff, uh_stream s);
/* uh_gate_bwd that also takes the column sums of dz as it stores it (the gradients of the biases of the gate's three
 * 1x1 convolutions, lunext.py:88-95): partial[uh_gate_bwd_parts(Mp, C)][3 * C] for
 * uh_colsum_finalize(partial, parts, 3 * C, ...), in place of a uh_colsum pass over the 3C-channel tensor. */
int uh_gate_bwd_parts(long long Mp, int C);
int uh_gate_bwd_colsum(long long Mp, int C, const void* dout, int d_cstride, int d_coff, const void* z, int z_cstride,
                       int z_coff, const void* x, int x_cstride, int x_coff, const uint8_t* rowmask,
                       void* dz, int dz_cstride, int dz_coff, void* dxd, int dx_cstride, int dx_coff,
                       float* partial, uh_stream s);
/* out = a + b (+ c when c != null) */
int uh_add(long long Mp, int C, const void* a, int a_cstride, int a_coff, const void* b, int b_cstride, int b_coff,
           const void* c, int c_cstride, int c_coff, void* out, int o_cstride, int o_coff, uh_stream s);
/* weight gradient of the depthwise convolution: the partial[parts][9][C] rows of uh_dwconv3x3_wgrad summed into the
 * module's own layout out[c_real][1][3][3] (lunext.py:14), padding channels dropped */
int uh_dw_wgrad_finalize(const float* partial, int parts, int C, int c_real, float* out, uh_stream s);
/* the same for several layers in one launch (a training step has 18 of them, each a few microseconds of work behind a
 * launch and a stream dependency of its own): descs is a HOST array, copied into the launch's arguments */
#define UH_DWFIN_MAX 32
typedef struct {
  const float* partial;   /* [parts][9][C] */
  float* out;             /* [c_real][1][3][3] */
  int parts, C, c_real;
} uh_dwfin_desc;
int uh_dw_wgrad_finalize_multi(const uh_dwfin_desc* descs, int count, uh_stream s);
/* The backward of one DepthwiseConvBlock (lunext.py:6-30) behind its BatchNorm sums, in one pass: what
 * uh_bn_res_relu_bwd_apply + uh_dwconv3x3_dgrad + uh_dwconv3x3_wgrad compute, without storing dy and dr
 * (dy is used in fp32, not rounded to bf16).  coef is uh_bn_bwd_finalize's output; partial[parts][9][C] goes to
 * uh_dw_wgrad_finalize.  C a power of two in [8, 512]. */
int uh_dwblock_bwd_parts(int B, int H, int W, int C);
int uh_dwblock_bwd(int B, int H, int W, int C, const void* da, int da_cstride, int da_coff,
                   const void* y, int y_cstride, int y_coff, const void* x, int x_cstride, int x_coff,
                   const float* w, int c_real,
                   const float* mean, const float* invstd, const float* scale, const float* shift, const float* coef,
                   void* dx, int dx_cstride, int dx_coff, float* partial, uh_stream s);
/* uh_dwblock_bwd for the second block of a pair (lunext.py:187-259 chains DepthwiseConvBlocks two by two: the second
 * block's input x is the first block's output relu(bn(yq) + its input), read by nothing else, and dx is the first
 * block's upstream gradient).  Also writes the FIRST block's BatchNorm-backward partial sums -- what
 * uh_bn_res_relu_bwd_reduce would make of dx, yq and the first block's input -- to
 * prev_partial[uh_dwblock_bwd_parts][2][C] for uh_bn_bwd_finalize, from the dx values as stored, the sign of x and one
 * more tensor read, so that block needs no reduce pass of its own.  mean_q / invstd_q: the first block's batch
 * statistics.  uh_dwblock_bwd_takes_prev() is 1 when the kernel in use can do this (0 under
 * UNET_AMD_DWBLOCK_LDS=1, and uh_dwblock_bwd_prev then returns an error). */
int uh_dwblock_bwd_takes_prev(void);
/* uh_dwblock_bwd for the first block behind a 2x2 stride-2 convolution (lunext.py:139-154): dx is that convolution's
 * output gradient, whose column sums are its bias gradient.  They are taken from dx as it is stored:
 * colsum[uh_dwblock_bwd_parts][C] for uh_colsum_finalize(colsum, parts, C, ...), in place of a uh_colsum pass.
 * Available when uh_dwblock_bwd_takes_prev() is 1. */
int uh_dwblock_bwd_colsum(int B, int H, int W, int C, const void* da, int da_cstride, int da_coff,
                          const void* y, int y_cstride, int y_coff, const void* x, int x_cstride, int x_coff,
                          const float* w, int c_real,
                          const float* mean, const float* invstd, const float* scale, const float* shift, const float* coef,
                          void* dx, int dx_cstride, int dx_coff, float* partial, float* colsum, uh_stream s);
int uh_dwblock_bwd_prev(int B, int H, int W, int C, const void* da, int da_cstride, int da_coff,
                        const void* y, int y_cstride, int y_coff, const void* x, int x_cstride, int x_coff,
                        const float* w, int c_real,
                        const float* mean, const float* invstd, const float* scale, const float* shift, const float* coef,
                        void* dx, int dx_cstride, int dx_coff, float* partial,
                        const void* yq, int yq_cstride, int yq_coff, const float* mean_q, const float* invstd_q,
                        float* prev_partial, uh_stream s);

/* F.interpolate(x, size=(Ho, Wo), mode='bilinear', align_corners=False) and its backward (lunext.py:226-251: the
 * size fix after a transposed convolution when the skip connection is one row / column larger), padded-linear NHWC
 * bf16 on both sides, fp32 arithmetic with torch's source-index rule. */
int uh_resize_bilinear_fwd(int B, int Hi, int Wi, int Ho, int Wo, int C, const void* in, int i_cstride, int i_coff,
                           void* out, int o_cstride, int o_coff, uh_stream s);
int uh_resize_bilinear_bwd(int B, int Hi, int Wi, int Ho, int Wo, int C, const void* dout, int o_cstride, int o_coff,
                           void* din, int i_cstride, int i_coff, uh_stream s);
/* fully fused gate for the 8-channel top level; wts = W1[8][8]|W2[8][8]|W3[8][8]|b1[8]|b2[8]|b3[8] fp32
 * ([co][ci], zero beyond the c_real real channels; only those are computed),
 * partial[blk][216] (blk < uh_gate8_parts) holds the weight gradients in the same layout */
int uh_gate8_fwd(long long Mp, int c_real, const void* g, int g_cstride, int g_coff, const void* x, int x_cstride,
                 int x_coff, const float* wts, const uint8_t* rowmask, void* out, int o_cstride, int o_coff, uh_stream s);
int uh_gate8_parts(long long Mp);
int uh_gate8_bwd(long long Mp, int c_real, const void* dout, int d_cstride, int d_coff, const void* g, int g_cstride, int g_coff,
                 const void* x, int x_cstride, int x_coff, const float* wts,
                 void* dg, int dg_cstride, int dg_coff, void* dx, int dx_cstride, int dx_coff,
                 float* partial, uh_stream s);
/* the table uh_gate8_fwd / uh_gate8_bwd take, from the gate's three [C][C] weights and three biases where they lie
 * (lunext.py:67-73), in one launch; and rows of `cin` floats zero-padded to rows of `cpad` (the head's [oc][cin] weight
 * as [oc][8], a 3-channel bias as 8 channels) */
int uh_gate8_pack(int c_real, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                  const float* b3, float* wts216, uh_stream s);
int uh_pad_rows_f32(int rows, int cin, int cpad, const float* src, float* dst, uh_stream s);
/* NCHW fp32 image (Cin <= 8) -> 8-channel padded-linear bf16 */
int uh_nchw_to_c8(int B, int Cin, int H, int W, const float* x, void* out, int o_cstride, int o_coff, uh_stream s);

/* ---- optimizer step: torch.optim.Adam(params, lr) as the reference's trainer builds it (training_utils.py:107),
 * over ONE flat fp32 buffer (p, g, m, v: n floats each, n a multiple of 4, 16-byte aligned); step counts from 1:
 *   g' = g + wd*p; m += (1-b1)(g'-m); v = b2 v + (1-b2) g'^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps) */
int uh_adam_step(long long n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int step, uh_stream s);
/* The same step under torch.amp.GradScaler (scaler.step(optimizer); training_utils.py:153-155): grad_scale and found_inf
 * are DEVICE scalars (either may be NULL).  The gradient is read as g * (1 / *grad_scale), so no unscale pass over the
 * gradients is needed, and when *found_inf != 0 the launch changes nothing (the skipped step of GradScaler).  A
 * skipped step is not counted, which the host cannot know without reading found_inf back: step_dev, when not NULL, is
 * a device scalar holding the count of this step (the caller adds 1 - *found_inf to it before the launch) and
 * replaces `step` in the bias corrections. */
int uh_adam_step_amp(long long n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int step, const float* grad_scale, const float* found_inf,
                     const float* step_dev, uh_stream s);

/* ---- probs = sigmoid(outputs).clamp(lo, hi), the glue in front of the clDice loss (trainer/training_utils.py:146,162,
 * 201), and its backward: dlogits = dprobs * s (1 - s) where lo <= s <= hi (s = the unclamped sigmoid), else 0.
 * fp32, n elements, 16-byte aligned buffers. */
int uh_sigmoid_clamp(long long n, const float* logits, float lo, float hi, float* probs, uh_stream s);
int uh_sigmoid_clamp_bwd(long long n, const float* logits, float lo, float hi, const float* dprobs, float* dlogits,
                         uh_stream s);

/* ---- nn.BCEWithLogitsLoss() as the trainer builds it (trainer/training_utils.py:105: mean reduction, no weights),
 * forward and backward in one pass over fp32 logits and targets of n elements:
 *   loss[0] = mean( max(x,0) - x t + log(1 + exp(-|x|)) );  dlogits = grad_scale * (sigmoid(x) - t) / n
 * (dlogits may be NULL: forward only).  workspace: uh_bce_logits_workspace(n) bytes. */
size_t uh_bce_logits_workspace(long long n);
int uh_bce_logits(long long n, const float* logits, const float* targets, float grad_scale, float* loss,
                  float* dlogits, void* workspace, size_t workspace_bytes, uh_stream s);
/* The same computation, same results, for callers that cannot allocate (code under HIP graph capture, serving loops
 * that must stay off the allocator).  It has no counterpart in the reference.  The per-workgroup partial sums live in a
 * 4 KiB array inside the library's code object (one copy per GPU, held until the process ends) instead of a caller
 * workspace; the call allocates and frees nothing.  n <= 2^32 elements, larger n is refused with UH_ERR_SHAPE.  The
 * device address of the array is looked up on the first call on each device: make that call outside graph capture,
 * with the device current.  All calls on one GPU share the array: calls that overlap on one GPU race. */
int uh_bce_logits_noalloc(long long n, const float* logits, const float* targets, float grad_scale, float* loss,
                          float* dlogits, uh_stream s);

/* ---- self-regularisation loss (models/losses.py:6-217; trainer/training_utils.py:100-101,142-143) ------------
 * loss = 0.4*CE + 0.6*Dice + 0.015*(LSRC + LIFD) (losses.py:72).  For one-channel logits the CE term and its
 * gradient are identically zero; LSRC and LIFD enter as constants (losses.py:63-69).
 *
 * Constants, from padded-linear bf16 feature maps (C a multiple of 8, at most 1024):
 *   uh_selfreg_pool2        2x2 mean of a map (bf16 window, or fp32 rows of C floats when in_f32) -> fp32 rows of
 *                           C floats, padded-linear for (B, Ho, Wo), pad rows zero: the levels of
 *                           adaptive_avg_pool2d(last feature, size of feature i) (losses.py:117)
 *   uh_selfreg_plane_stats  partial[b][part][2][C]: sum and sum of squares per image and channel (losses.py:185-189)
 *   uh_selfreg_rank         order[C]: channels by descending mean of f / (||f||_2 + 1e-8) (losses.py:185-192),
 *                           ties towards the lower index; uses up `partial` (its contents are overwritten)
 *   uh_selfreg_pair_sums    partial[part][2] = ( sum (f[sel_f[j]] - pooled[sel_p[j]])^2 over rows and j < nsel
 *                           (losses.py:128-147; pooled: a window of p_channels channels, bf16 or fp32; NULL: 0),  sum (f[order[j]] - f[order[j + C/2]])^2 over rows
 *                           and j < C/2 (losses.py:195-206) )
 *   uh_selfreg_accumulate   aux[0] += scale_lsrc * sum partial[.][0]; aux[1] += scale_pair * sum partial[.][1]
 *                           (init != 0: start from zero); the scales carry 1/elements and 1/(features - 1)
 *                           (losses.py:147,155,206,217)
 * Differentiable part, fp32 logits and targets [B][n]:
 *   uh_selfreg_loss_fwd     loss[0] = w_dice * (1 - mean_b (2 sum(p t) + 1e-6) / (sum(p^2) + sum(t^2) + 1e-6)),
 *                           p = sigmoid(logits) (losses.py:18-41), + w_aux * (aux[0] + aux[1]) when aux != NULL;
 *                           coef[B][2] is kept for the backward pass
 *   uh_selfreg_loss_bwd     dlogits = gout[0] * d loss / d logits */
int uh_selfreg_pool2(int B, int Ho, int Wo, int C, const void* in, int in_f32, int cstride, int coff, float* out,
                     uh_stream s);
int uh_selfreg_stat_parts(int B, int H, int W, int C);
int uh_selfreg_plane_stats(int B, int H, int W, int C, const void* f, int cstride, int coff, float* partial,
                           uh_stream s);
int uh_selfreg_rank(int B, int H, int W, int C, float* partial, int* order, uh_stream s);
int uh_selfreg_pair_parts(int B, int H, int W, int C);
int uh_selfreg_pair_sums(int B, int H, int W, int C, const void* f, int cstride, int coff, const int* order,
                         const void* pooled, int pooled_f32, int p_channels, int p_cstride, int p_coff, int nsel,
                         const int* sel_f, const int* sel_p, float* partial, uh_stream s);
int uh_selfreg_accumulate(const float* partial, int parts, double scale_lsrc, double scale_pair, int init, float* aux,
                          uh_stream s);
size_t uh_selfreg_loss_workspace(int B, long long n);
int uh_selfreg_loss_fwd(int B, long long n, const float* logits, const float* targets, const float* aux, float w_dice,
                        float w_aux, float* loss, float* coef, void* workspace, size_t workspace_bytes, uh_stream s);
int uh_selfreg_loss_bwd(int B, long long n, const float* logits, const float* targets, const float* coef,
                        const float* gout, float w_dice, float* dlogits, uh_stream s);

/* ---- input pipeline: FloatTensor(images).permute(0, 3, 1, 2) + TensorDataset + shuffling DataLoader
 * (trainer/training_utils.py:52-84) as one gather per batch from a data set that stays on the device in the loaders'
 * own layout.  data: [n_items][H][W][C], bytes (every value of the loader's fp32 array is lut[byte]; lut[k] =
 * fp32(k)/fp32(255) for arrays made by data_helpers/utils.py:58-62) or fp32; idx: b int64 item numbers on the
 * device (an index outside [0, n_items) leaves its slice of out untouched); out: [b][C][H][W] fp32, exactly the
 * values FloatTensor(images).permute(0, 3, 1, 2)[idx] holds. */
int uh_gather_batch_u8(int b, int n_items, int H, int W, int C, const uint8_t* data, const long long* idx,
                       const float* lut, float* out, uh_stream s);
int uh_gather_batch_f32(int b, int n_items, int H, int W, int C, const float* data, const long long* idx,
                        float* out, uh_stream s);

/* ---- evaluation: binarised predictions and confusion counts, fp32 logits and masks [planes][n] ---------
 * pred = sigmoid(logit) > 0.5 evaluated in fp32 (trainer/k_fold.py:289-290; test_helper/utils.py:130-131), so
 * logit 0 and NaN are background; counted against mask == 1 / mask == 0 exactly (k_fold.py:371-380: other
 * mask values fall into no class).  counts[plane][4] = tp, tn, fp, fn are ADDED to (the caller zeroes them
 * once and may keep accumulating over batches: one plane = k_fold.py::evaluate_model's whole-set counts,
 * one plane per image = test_helper/utils.py::save_results' per-sample counts). */
int uh_confusion_counts(int planes, long long n, const float* logits, const float* mask, long long* counts,
                        uh_stream s);

/* ---- clDice soft-skeleton loss, fp32 planes [planes][H][W] (losses/cl_dice.py) ---------------------
 * soft_erode (cl_dice.py:12-17): 5-point-cross minimum, out-of-image neighbours ignored; clamp != 0 first
 * clamps the input to [0,1] (cl_dice.py:116-117).  soft_dilate (cl_dice.py:24-27): 3x3 maximum. */
int uh_soft_erode(int planes, int H, int W, const float* in, float* out, int clamp, uh_stream s);
int uh_soft_dilate(int planes, int H, int W, const float* in, float* out, uh_stream s);
/* soft_skel with num_iter = 10 (cl_dice.py:35-46,103) of clamp(x,0,1); workspace = 3 planesets */
size_t uh_soft_skel_workspace(int planes, int H, int W);
int uh_soft_skel(int planes, int H, int W, const float* x, float* skel, float* workspace, uh_stream s);
/* soft_dice_cldice.forward(y_true, y_pred) (cl_dice.py:106-125): loss_out[0] = (1-alpha)*dice + alpha*clDice.
 * The workspace (uh_cldice_workspace bytes) keeps what backward needs: the eleven erosion levels and
 * skeleton states of y_pred, skel(y_true), and the scalar coefficients. */
size_t uh_cldice_workspace(int planes, int H, int W);
int uh_cldice_forward(int planes, int H, int W, const float* y_true, const float* y_pred,
                      float alpha, float smooth, float* workspace, float* loss_out, uh_stream s);
/* uh_cldice_forward in two halves, for a training loop that has the labels before the prediction
 * (trainer/training_utils.py:125-148: the batch is there when the model's forward pass starts): uh_cldice_skel_true
 * leaves skel(y_true) (cl_dice.py:119) in the workspace and touches no other part of it that the second half or the
 * backward pass read, so it may run on a second stream beside the network; uh_cldice_forward_pred, ordered after it
 * on the device, does the rest.  Together they give uh_cldice_forward's bits. */
int uh_cldice_skel_true(int planes, int H, int W, const float* y_true, float* workspace, uh_stream s);
int uh_cldice_forward_pred(int planes, int H, int W, const float* y_true, const float* y_pred,
                           float alpha, float smooth, float* workspace, float* loss_out, uh_stream s);
/* grad_pred = grad_out[0] * dLoss/dy_pred, with autograd's tie rules (first extremum of a pooling window,
 * 1/2:1/2 split in min(), relu'(0)=0, clamp passes the gradient on [0,1]).  Uses the forward's workspace. */
int uh_cldice_backward(int planes, int H, int W, const float* y_true, const float* y_pred,
                       float* workspace, const float* grad_out, float* grad_pred, uh_stream s);
/* out17 = loss, 5 gradient coefficients, dice, clDice, tprec, tsens, then the seven global sums
 * (sum t*p, sum t, sum p, sum skel_p*t, sum skel_p, sum skel_t*p, sum skel_t) */
int uh_cldice_read_terms(const float* workspace, int planes, int H, int W, float* out17, uh_stream s);

#ifdef __cplusplus
}
#endif
#endif
