// Many small reductions in one launch.  A LUNeXt backward pass ends some thirty tables of partial rows (weight gradients
// of its 1x1 / 2x2 / transposed convolutions, bias gradients as column sums), each a few microseconds of work that
// nothing but the optimizer waits for; launched one by one they cost the step their launch boundaries and stream
// dependencies.  uh_reduce_multi takes up to UH_REDUCE_MAX descriptors (a HOST array: they travel as kernel arguments)
// and gives every table the blocks its single launch would have had.
#include "reduce_small.h"

struct ReduceTable {
  uh_reduce_desc d[UH_REDUCE_MAX];
  int first[UH_REDUCE_MAX + 1];     // first block of each table
  int route[UH_REDUCE_MAX];         // UH_RED_* for weight gradients
  int n;
};

__global__ __launch_bounds__(256) void reduce_multi_kernel(ReduceTable t) {
  __shared__ double smem[(UH_WGRAD_REDUCE_CONVT_FLOATS * 4 + 7) / 8];
  int i = 0;
  while (i + 1 < t.n && (int)blockIdx.x >= t.first[i + 1]) ++i;            // (uniform over the block)
  const uh_reduce_desc& d = t.d[i];
  const int blk = (int)blockIdx.x - t.first[i];
  if (d.kind == UH_REDUCE_COLSUM) {
    colsum_gather_body(blk, smem, d.partial, d.rows, d.C, d.n_out, d.cols, d.out, d.accumulate);
  } else if (t.route[i] == UH_RED_CONVT) {
    const int nbx = d.K / 64;
    wgrad_reduce_convt_body(blk % nbx, blk / nbx, (float*)smem, d.partial, d.rows, d.split_stride, d.N, d.K, d.out,
                            d.accumulate);
  } else if (t.route[i] == UH_RED_GENERIC32) {
    wgrad_reduce_body<32>(blk, (float*)smem, d.partial, d.rows, d.split_stride, d.N, d.T, d.K, d.on, d.ot, d.sn, d.st,
                          d.sk, d.out, d.accumulate);
  } else {
    wgrad_reduce_body<4>(blk, (float*)smem, d.partial, d.rows, d.split_stride, d.N, d.T, d.K, d.on, d.ot, d.sn, d.st,
                         d.sk, d.out, d.accumulate);
  }
}

extern "C" int uh_reduce_multi(const uh_reduce_desc* descs, int count, uh_stream s) {
  UH_REQUIRE(descs && count > 0 && count <= UH_REDUCE_MAX, "reduce_multi: 1..%d tables, got %d", UH_REDUCE_MAX, count);
  ReduceTable t;
  long long blocks = 0;
  for (int i = 0; i < count; ++i) {
    const uh_reduce_desc& d = descs[i];
    UH_REQUIRE(d.partial && d.out && d.rows > 0, "reduce_multi: bad table %d", i);
    long long nb = 0;
    t.route[i] = -1;
    if (d.kind == UH_REDUCE_COLSUM) {
      UH_REQUIRE(d.C > 0 && d.n_out > 0 && (d.cols || d.n_out <= d.C), "reduce_multi: table %d: C=%d, n_out=%d", i, d.C, d.n_out);
      nb = (d.n_out + 7) / 8;
    } else {
      UH_REQUIRE(d.kind == UH_REDUCE_WGRAD && d.N > 0 && d.T > 0 && d.K > 0 && d.split_stride > 0,
                 "reduce_multi: table %d: kind %d, N=%d T=%d K=%d", i, d.kind, d.N, d.T, d.K);
      t.route[i] = wgrad_reduce_route(d.rows, d.N, d.T, d.K, d.on, d.ot, d.sn, d.st, d.sk, &nb);
      UH_REQUIRE(t.route[i] != UH_RED_OIHW9, "reduce_multi: table %d is a 3x3 weight gradient (uh_wgrad_reduce_strided has kernels of its own for those)", i);
    }
    t.d[i] = d;
    t.first[i] = (int)blocks;
    blocks += nb;
    UH_REQUIRE(blocks < (1ll << 30), "reduce_multi: too many blocks");
  }
  t.first[count] = (int)blocks;
  t.n = count;
  hipLaunchKernelGGL(reduce_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, t);
  return uh_check_launch("reduce_multi");
}
