"""Drop-in `LUNeXt` module (models/lunext.py:124-259 of the reference): same constructor, parameters,
state_dict keys (176) and forward contract, executed by the HIP kernels of libunet_hip.so.

Layout notes.  Activations are padded-linear NHWC bf16.  The full-resolution level works on
`in_channels` (3) channels; it is carried zero-padded to 8 channels so that one pixel is one 16-byte
access, with parameters zero-padded to match (a padded BatchNorm channel has gamma = 0 and therefore
stays exactly zero).  The 2x2 stride-2 convolutions are space-to-depth + MFMA GEMM, the transposed
convolutions MFMA GEMM + pixel-shuffle store, the gate's 1x1 convolutions MFMA GEMMs at C >= 64 and one
fused vector kernel at the 8-channel level; GEMM operands narrower than 64 are zero-padded to 64.
`att*.output_conv` is declared but never called by the reference (lunext.py:74,120): its parameters
exist in the state_dict and receive no gradient, as there.
"""
import contextlib
import os

import torch
import torch.nn as nn

from . import _lib
from ._lib import call, EPI_RAW_STATS, EPI_AFFINE_RELU, EPI_MASKED, EPI_CONVT
from .engine import BN_EPS, BN_MOMENTUM, _stream, side_stream_for, ranged, fused_finalize
from .layout import pool_for, Act
from .packing import lunext_packer_for

L = _lib.lib

# UNET_AMD_LUNEXT_FUSED_DW=0: the backward of a DepthwiseConvBlock behind its BatchNorm sums runs as three passes
# (uh_bn_res_relu_bwd_apply, uh_dwconv3x3_dgrad, uh_dwconv3x3_wgrad) instead of one (uh_dwblock_bwd).
FUSED_DW_BWD = os.environ.get("UNET_AMD_LUNEXT_FUSED_DW", "1") == "1"
# UNET_AMD_LUNEXT_PAIR_SUMS=0: the first block of a pair makes its BatchNorm-backward sums in a pass of its own
# (uh_bn_res_relu_bwd_reduce) instead of getting them from the second block's backward (uh_dwblock_bwd_prev).
PAIR_SUMS = os.environ.get("UNET_AMD_LUNEXT_PAIR_SUMS", "1") == "1"
# UNET_AMD_LUNEXT_PAIR_APPLY=0: the first block of a pair writes its output in a launch of its own
# (uh_bn_res_relu_apply) instead of leaving it to the second block's convolution (uh_dwconv3x3_fwd_applied).
PAIR_APPLY = os.environ.get("UNET_AMD_LUNEXT_PAIR_APPLY", "1") == "1"
# UNET_AMD_LUNEXT_GATE_CONCAT=0: the attention gate adds g + x in a pass of its own and runs two GEMMs, instead of one
# GEMM over [g | x] held side by side in one buffer (the skip connection and the transposed convolution write there).
GATE_CONCAT = os.environ.get("UNET_AMD_LUNEXT_GATE_CONCAT", "1") == "1"
# UNET_AMD_LUNEXT_S2D_FUSED=0: the space-to-depth operand of a 2x2 stride-2 convolution is made by a pass of its own
# (uh_space_to_depth2_of) instead of by the launch that writes the block output it copies (uh_bn_res_relu_apply_s2d).
S2D_FUSED = os.environ.get("UNET_AMD_LUNEXT_S2D_FUSED", "1") == "1"


def _padc(c):
    return max(8, (c + 7) // 8 * 8)


def _padv(t, n, fill=0.0):
    if t.numel() == n:
        return t.detach()
    out = torch.full((n,), fill, dtype=torch.float32, device=t.device)
    out[:t.numel()] = t.detach()
    return out


class _Handle:
    """context manager that just hands out a stream handle"""
    __slots__ = ("h",)

    def __init__(self, h):
        self.h = h

    def __enter__(self):
        return self.h

    def __exit__(self, *exc):
        return False


class _DWRec:
    __slots__ = ("name", "x", "y", "a", "C", "Cp", "lvl", "wt", "mean", "invstd", "scale", "shift")


class LUNeXtRun:
    def __init__(self, module, x, training):
        self.m, self.training, self.device = module, training, x.device
        self.pool = pool_for(x.device)
        self.lease = self.pool.lease()
        B, Cin, H, W = x.shape
        if H < 16 or W < 16:
            raise ValueError(f"LUNeXt halves the image four times: H and W must be at least 16, got {H}x{W}")
        if Cin != module.in_channels or Cin > 8:
            raise ValueError(f"expected {module.in_channels} (<= 8) input channels, got {Cin}")
        self.x = x
        self.B = B
        # a k2 s2 convolution floors odd sizes (lunext.py:139-154); the decoder then resizes what the transposed
        # convolution returns to the skip connection's size (lunext.py:226-251)
        self.geo = [(B, H >> l, W >> l) for l in range(5)]
        self.masks = [self.pool.rowmask(*g) for g in self.geo]
        self.chans = [Cin, 64, 128, 256, 512]
        self.cp = [_padc(c) for c in self.chans]
        self.recs = {}
        self.saved = {}
        self._side_used = None
        self._bn_sums = {}
        self._pending = {}
        self._dx_colsum = {}
        self._dw_wgrads = []
        self._reduces = []
        self._reduces2 = []
        self._x4s = {}
        self._depth_made = set()
        self.packer = lunext_packer_for(module, x.device)       # bf16 GEMM operands, one repack launch per direction
        self.wf = None
        # set by a caller that keeps all parameter gradients in one flat buffer (dp.FlatGrads, laid out by
        # flat_param_order below): every gradient is then written where it lives, by the kernel that makes it
        self.gflat = None

    def _gnew(self, name, shape):
        """the gradient tensor of parameter `name`: the caller's view of its flat buffer, or a new tensor"""
        if self.gflat is not None:
            return self.gflat.views[name]
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _table(self, key, build):
        """a small int32 index table on the device, built once per module (uh_colsum_finalize_gather's `cols`)"""
        tabs = self.m.__dict__.setdefault("_gather_tables", {})
        t = tabs.get(key)
        if t is None or t.device != self.device:
            t = tabs[key] = torch.tensor(build(), dtype=torch.int32, device=self.device)
        return t

    # ------------------------------------------------------------------ depthwise block
    @ranged(lambda name, *a, **k: "fwd/" + name)
    def _dw(self, name, x, lvl, defer_apply=False, out=None, depth=None):
        """One DepthwiseConvBlock forward.  defer_apply (training, first block of a pair): the block's output buffer
        is handed on unwritten and noted in self._pending with the record of what is to be applied; the next _dw call
        then makes it inside its own convolution launch (uh_dwconv3x3_fwd_applied) instead of reading it back.
        out: where the block's output goes (a view of a buffer shared with another tensor), else a buffer of its own.
        depth: the space-to-depth operand of the 2x2 stride-2 convolution that reads this block's output: filled by the
        same launch that writes the output (training), see _down."""
        blk = getattr(self.m, name)
        B, H, W = self.geo[lvl]
        C, Cp = blk.depthwise.out_channels, x.C
        s, Ls = self._sh, self.lease
        wt, bias = blk.depthwise.weight.detach(), blk.depthwise.bias.detach()      # read in place by the kernels
        if wt.data_ptr() % 16 or not wt.is_contiguous():
            wt = wt.clone(memory_format=torch.contiguous_format)
        mask = self.masks[lvl]
        a = out if out is not None else Ls.act(B, H, W, Cp)
        if not self.training:
            gamma, beta = _padv(blk.bn.weight, Cp), _padv(blk.bn.bias, Cp)
            rm, rv = _padv(blk.bn.running_mean, Cp), _padv(blk.bn.running_var, Cp, 1.0)
            f = Ls.f32(2 * Cp)
            call("uh_bn_fold_eval", Cp, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), None, BN_EPS,
                 f[:Cp].data_ptr(), f[Cp:].data_ptr(), s)
            call("uh_dwconv3x3_fwd", B, H, W, Cp, x.ptr, x.cstride, x.coff, wt.data_ptr(), C, bias.data_ptr(),
                 EPI_AFFINE_RELU, f[:Cp].data_ptr(), f[Cp:].data_ptr(), mask.data_ptr(), a.ptr, a.cstride, a.coff,
                 None, s)
            return a
        rec = _DWRec()
        rec.name, rec.x, rec.a, rec.C, rec.Cp, rec.lvl, rec.wt = name, x, a, C, Cp, lvl, wt
        rec.y = Ls.act(B, H, W, Cp)
        parts = L.uh_dw_parts(B, H, W, Cp) if fused_finalize() else L.uh_dwconv3x3_fwd_parts(B, H, W, Cp)
        f = Ls.f32(4 * Cp)
        rec.mean, rec.invstd, rec.scale, rec.shift = f[:Cp], f[Cp:2 * Cp], f[2 * Cp:3 * Cp], f[3 * Cp:]
        # the C real channels of the parameters and running statistics are read and updated where they are
        if fused_finalize():
            # the convolution's last workgroups also turn the batch statistics into scale / shift: no second launch
            nbytes = L.uh_fin_ws_bytes(parts, Cp, 2)
            ws = Ls.f32((nbytes + 3) // 4)
            call("uh_dwconv3x3_fwd_bn", B, H, W, Cp, x.ptr, x.cstride, x.coff, wt.data_ptr(), C, bias.data_ptr(),
                 mask.data_ptr(), rec.y.ptr, rec.y.cstride, rec.y.coff, blk.bn.weight.data_ptr(),
                 blk.bn.bias.data_ptr(), BN_EPS, BN_MOMENTUM, rec.mean.data_ptr(), rec.invstd.data_ptr(),
                 rec.scale.data_ptr(), rec.shift.data_ptr(), blk.bn.running_mean.data_ptr(),
                 blk.bn.running_var.data_ptr(), blk.bn.num_batches_tracked.data_ptr(), ws.data_ptr(), nbytes,
                 self.pool.counters(s).data_ptr(), s)
        else:
            stat = Ls.f32(parts * 2 * Cp)
            q = self._pending.pop(id(x), None)
            if q is not None:                        # x = relu(bn(q.y) + q.x) is made by this launch
                call("uh_dwconv3x3_fwd_applied", B, H, W, Cp, q.y.ptr, q.y.cstride, q.y.coff, q.x.ptr, q.x.cstride,
                     q.x.coff, q.scale.data_ptr(), q.shift.data_ptr(), x.ptr, x.cstride, x.coff, wt.data_ptr(), C,
                     bias.data_ptr(), rec.y.ptr, rec.y.cstride, rec.y.coff, stat.data_ptr(), s)
            else:
                call("uh_dwconv3x3_fwd", B, H, W, Cp, x.ptr, x.cstride, x.coff, wt.data_ptr(), C, bias.data_ptr(),
                     EPI_RAW_STATS, None, None, mask.data_ptr(), rec.y.ptr, rec.y.cstride, rec.y.coff, stat.data_ptr(), s)
            call("uh_bn_finalize", stat.data_ptr(), parts, Cp, C, B * H * W, blk.bn.weight.data_ptr(),
                 blk.bn.bias.data_ptr(), BN_EPS, BN_MOMENTUM, rec.mean.data_ptr(), rec.invstd.data_ptr(),
                 rec.scale.data_ptr(), rec.shift.data_ptr(), blk.bn.running_mean.data_ptr(),
                 blk.bn.running_var.data_ptr(), blk.bn.num_batches_tracked.data_ptr(), s)
        self.recs[name] = rec
        if defer_apply and PAIR_APPLY and not fused_finalize() and L.uh_dwconv3x3_fwd_takes_apply(Cp):
            self._pending[id(a)] = rec
            return a
        if depth is not None and S2D_FUSED:
            call("uh_bn_res_relu_apply_s2d", B, H, W, Cp, rec.y.ptr, rec.y.cstride, rec.y.coff, x.ptr, x.cstride, x.coff,
                 rec.scale.data_ptr(), rec.shift.data_ptr(), a.ptr, a.cstride, a.coff, depth.ptr, depth.cstride, 0, s)
            self._depth_made.add(id(depth))
            return a
        call("uh_bn_res_relu_apply", x.Mp, Cp, rec.y.ptr, rec.y.cstride, rec.y.coff, x.ptr, x.cstride, x.coff,
             rec.scale.data_ptr(), rec.shift.data_ptr(), mask.data_ptr(), a.ptr, a.cstride, a.coff, s)
        return a

    @ranged(lambda name, *a: "bwd/" + name)
    def _dw_backward(self, name, da, grads, prev=None, colsum=False):
        """`prev`: name of the depthwise block whose output is this block's input and is read by nothing else (the
        first block of a pair).  Its BatchNorm-backward sums are then taken while this block's input gradient is
        stored (uh_dwblock_bwd_prev), and its own _dw_backward call finds them in self._bn_sums.
        `colsum`: the returned gradient goes to _down_backward, which wants its column sums."""
        rec = self.recs[name]
        B, H, W = self.geo[rec.lvl]
        C, Cp, x, y = rec.C, rec.Cp, rec.x, rec.y
        s, Ls = self._sh, self.lease
        mask = self.masks[rec.lvl]
        st = (rec.mean.data_ptr(), rec.invstd.data_ptr(), rec.scale.data_ptr(), rec.shift.data_ptr())
        coef = Ls.f32(2 * Cp)
        if self.gflat is not None:
            dgam, dbet = self.gflat.views[name + ".bn.weight"], self.gflat.views[name + ".bn.bias"]
        else:
            dgam, dbet = torch.empty(2, C, dtype=torch.float32, device=self.device).unbind(0)
        ready = self._bn_sums.pop(name, None)
        if ready is not None:                        # the block after this one has left the sums behind
            partial, parts = ready
            call("uh_bn_bwd_finalize", partial.data_ptr(), parts, Cp, C, B * H * W, dgam.data_ptr(), dbet.data_ptr(),
                 coef.data_ptr(), 0, s)
        elif fused_finalize():                      # sums and their finalisation in one launch
            parts = L.uh_bn_bwd_parts(x.Mp, Cp)
            nbytes = L.uh_fin_ws_bytes(parts, Cp, 2)
            ws = Ls.f32((nbytes + 3) // 4)
            call("uh_bn_res_relu_bwd_reduce_fin", x.Mp, Cp, da.ptr, da.cstride, da.coff, y.ptr, y.cstride, y.coff,
                 x.ptr, x.cstride, x.coff, *st, C, B * H * W, dgam.data_ptr(), dbet.data_ptr(), coef.data_ptr(),
                 ws.data_ptr(), nbytes, self.pool.counters(s).data_ptr(), s)
        else:
            parts = L.uh_bn_bwd_parts(x.Mp, Cp)
            partial = Ls.f32(parts * 2 * Cp)
            call("uh_bn_res_relu_bwd_reduce", x.Mp, Cp, da.ptr, da.cstride, da.coff, y.ptr, y.cstride, y.coff, x.ptr,
                 x.cstride, x.coff, *st, partial.data_ptr(), s)
            call("uh_bn_bwd_finalize", partial.data_ptr(), parts, Cp, C, B * H * W, dgam.data_ptr(), dbet.data_ptr(),
                 coef.data_ptr(), 0, s)                                      # the kernel writes the gradient tensors
        grads[name + ".bn.weight"], grads[name + ".bn.bias"] = dgam, dbet
        if FUSED_DW_BWD:
            wparts = L.uh_dwblock_bwd_parts(B, H, W, Cp)
            wp = Ls.f32(wparts * 9 * Cp)
            dx = Ls.act(B, H, W, Cp)
            dw = self._gnew(name + ".depthwise.weight", (C, 1, 3, 3))
            q = self.recs.get(prev) if PAIR_SUMS and prev is not None and L.uh_dwblock_bwd_takes_prev() else None
            if q is not None and q.a is x and q.Cp == Cp:
                qpart = Ls.f32(wparts * 2 * Cp)
                call("uh_dwblock_bwd_prev", B, H, W, Cp, da.ptr, da.cstride, da.coff, y.ptr, y.cstride, y.coff, x.ptr,
                     x.cstride, x.coff, rec.wt.data_ptr(), C, *st, coef.data_ptr(), dx.ptr, dx.cstride,
                     dx.coff, wp.data_ptr(), q.y.ptr, q.y.cstride, q.y.coff, q.mean.data_ptr(), q.invstd.data_ptr(),
                     qpart.data_ptr(), s)
                self._bn_sums[prev] = (qpart, wparts)
            elif colsum and PAIR_SUMS and L.uh_dwblock_bwd_takes_prev():
                # dx is the output gradient of the convolution in front of the block: its column sums (that
                # convolution's bias gradient) are taken here, _down_backward finds them in self._dx_colsum
                cpart = Ls.f32(wparts * Cp)
                call("uh_dwblock_bwd_colsum", B, H, W, Cp, da.ptr, da.cstride, da.coff, y.ptr, y.cstride, y.coff, x.ptr,
                     x.cstride, x.coff, rec.wt.data_ptr(), C, *st, coef.data_ptr(), dx.ptr, dx.cstride,
                     dx.coff, wp.data_ptr(), cpart.data_ptr(), s)
                self._dx_colsum[id(dx)] = (cpart, wparts)
            else:
                call("uh_dwblock_bwd", B, H, W, Cp, da.ptr, da.cstride, da.coff, y.ptr, y.cstride, y.coff, x.ptr,
                     x.cstride, x.coff, rec.wt.data_ptr(), C, *st, coef.data_ptr(), dx.ptr, dx.cstride,
                     dx.coff, wp.data_ptr(), s)
            # the partial rows are added up later, many layers per launch (_flush_small)
            self._dw_wgrads.append(_lib.DwFinDesc(wp.data_ptr(), dw.data_ptr(), wparts, Cp, C))
            grads[name + ".depthwise.weight"] = dw
            grads[name + ".depthwise.bias"] = self._gzero(name + ".depthwise.bias", C)
            return dx
        dy, dr = Ls.act(B, H, W, Cp), Ls.act(B, H, W, Cp)
        call("uh_bn_res_relu_bwd_apply", x.Mp, Cp, da.ptr, da.cstride, da.coff, y.ptr, y.cstride, y.coff, x.ptr,
             x.cstride, x.coff, *st, coef.data_ptr(), mask.data_ptr(), dy.ptr, dy.cstride, dy.coff, dr.ptr, dr.cstride,
             dr.coff, s)
        wparts = L.uh_dw_parts(B, H, W, Cp)
        wp = Ls.f32(wparts * 9 * Cp)
        dw = self._gnew(name + ".depthwise.weight", (C, 1, 3, 3))
        with self._side() as s2:                     # beside the input gradient below: both only read dy and x
            call("uh_dwconv3x3_wgrad", B, H, W, Cp, dy.ptr, dy.cstride, dy.coff, x.ptr, x.cstride, x.coff, wp.data_ptr(), s2)
            call("uh_dw_wgrad_finalize", wp.data_ptr(), wparts, Cp, C, dw.data_ptr(), s2)
        grads[name + ".depthwise.weight"] = dw
        # conv bias in front of train-mode BatchNorm: exactly zero gradient
        grads[name + ".depthwise.bias"] = self._gzero(name + ".depthwise.bias", C)
        dx = Ls.act(B, H, W, Cp)
        call("uh_dwconv3x3_dgrad", B, H, W, Cp, dy.ptr, dy.cstride, dy.coff, rec.wt.data_ptr(), C, dr.ptr, dr.cstride,
             dr.coff, mask.data_ptr(), dx.ptr, dx.cstride, dx.coff, s)
        return dx

    def _later(self, desc, second=False):
        """a small reduction (a _lib.ReduceDesc) whose result only the optimizer waits for: run by _flush_small.
        second: in a launch after the others' (it adds to what one of them writes)"""
        (self._reduces2 if second else self._reduces).append(desc)

    def _flush_small(self):
        """The small reductions that are waiting -- depthwise weight-gradient tables, weight gradients of the 1x1 / 2x2 /
        transposed convolutions, column sums -- in one visit to the second stream and two launches (one by one, each
        behind a launch boundary and most behind a stream dependency of their own, the 18 + 30 of a step cost it
        0.13 + 0.05 ms)."""
        dws, self._dw_wgrads = self._dw_wgrads, []
        reds, self._reduces = self._reduces, []
        reds2, self._reduces2 = self._reduces2, []
        if not dws and not reds and not reds2:
            return
        with self._side() as s2:
            for lo in range(0, len(dws), _lib.DWFIN_MAX):
                part = dws[lo:lo + _lib.DWFIN_MAX]
                call("uh_dw_wgrad_finalize_multi", (_lib.DwFinDesc * len(part))(*part), len(part), s2)
            for group in (reds, reds2):
                for lo in range(0, len(group), _lib.REDUCE_MAX):
                    part = group[lo:lo + _lib.REDUCE_MAX]
                    call("uh_reduce_multi", (_lib.ReduceDesc * len(part))(*part), len(part), s2)

    # ------------------------------------------------------------------ 2x2 stride-2 conv (lunext.py:139-154)
    def _x4(self, l, cpi):
        """the space-to-depth operand of downconv{l} (a private, zero-initialised buffer: its pad rows and the channels
        beyond 4 * cpi are never written); one per pass"""
        if l not in self._x4s:
            Bs, Hs, Ws = self.geo[l]
            self._x4s[l] = self.lease.act(Bs, Hs, Ws, max(64, 4 * cpi), tag=f"down{l}.x4")
        return self._x4s[l]

    @ranged(lambda l, *a: f"fwd/downconv{l + 1}")
    def _down(self, l, x):
        """level l-1 -> level l"""
        conv = getattr(self.m, f"downconv{l}")
        cin, cpi, cout = self.chans[l - 1], x.C, self.chans[l]
        Bs, Hs, Ws = self.geo[l]
        K = max(64, 4 * cpi)
        s, Ls = self._sh, self.lease
        x4 = self._x4(l, cpi)
        if id(x4) in self._depth_made:                   # the launch that wrote x has filled it
            self._depth_made.discard(id(x4))
        else:
            call("uh_space_to_depth2_of", Bs, Hs, Ws, cpi, x.H, x.W, x.ptr, x.cstride, x.coff, x4.ptr, x4.cstride, 0, s)
        wp = self.wf[f"down{l}"]
        out = Ls.act(Bs, Hs, Ws, cout)
        call("uh_conv_gemm", Bs, Hs, Ws, x4.ptr, x4.cstride, 0, K, wp.data_ptr(), cout, 1, EPI_MASKED, out.ptr,
             out.cstride, out.coff, conv.bias.data_ptr(), None, None, None, self.masks[l].data_ptr(), s)
        self.saved[f"down{l}"] = (x4, K, cpi)
        return out

    @ranged(lambda l, *a: f"bwd/downconv{l + 1}")
    def _down_backward(self, l, dy, grads, skip_grad):
        conv = getattr(self.m, f"downconv{l}")
        x4, K, cpi = self.saved[f"down{l}"]
        cin, cout = self.chans[l - 1], self.chans[l]
        Bs, Hs, Ws = self.geo[l]
        s, Ls = self._sh, self.lease
        S = L.uh_conv_wgrad_splits(Bs, Hs, Ws, cout, K, 1)
        part = Ls.f32(S * cout * K)
        dw = self._gnew(f"downconv{l}.weight", conv.weight.shape)
        with self._side() as s2:
            ready = self._dx_colsum.pop(id(dy), None)
            if ready is not None:                     # the launch that wrote dy has left its column sums behind
                db = self._gnew(f"downconv{l}.bias", (cout,))
                self._later(_lib.ReduceDesc.colsum(ready[0].data_ptr(), ready[1], cout, db.data_ptr()))
                grads[f"downconv{l}.bias"] = db
            else:
                grads[f"downconv{l}.bias"] = self._colsum(dy, cout, s2, out=self._gnew(f"downconv{l}.bias", (cout,)))
            call("uh_conv_wgrad", Bs, Hs, Ws, dy.ptr, dy.cstride, dy.coff, cout, x4.ptr, x4.cstride, 0, K, 1,
                 part.data_ptr(), s2)
            self._later(_lib.ReduceDesc.wgrad(part.data_ptr(), S, cout * K, cout, 4, cin, K, cpi, cin * 4, 1, 4, dw.data_ptr()))
        grads[f"downconv{l}.weight"] = dw
        # dx4[m][t*cpi + ci] = sum_co dy[m][co] * W[co][ci][t]
        wd = self.wb[f"down{l}"]
        dx4 = Ls.act(Bs, Hs, Ws, K)
        call("uh_conv_gemm", Bs, Hs, Ws, dy.ptr, dy.cstride, dy.coff, cout, wd.data_ptr(), K, 1, EPI_MASKED, dx4.ptr,
             dx4.cstride, 0, None, None, None, None, self.masks[l].data_ptr(), s)
        B, H, W = self.geo[l - 1]
        dx = Ls.act(B, H, W, cpi)
        # joined with the gradient of the skip connection taken in front of the convolution
        call("uh_depth_to_space2_add", Bs, Hs, Ws, cpi, H, W, dx4.ptr, dx4.cstride, 0, skip_grad.ptr, skip_grad.cstride,
             skip_grad.coff, dx.ptr, dx.cstride, dx.coff, s)
        return dx

    # ------------------------------------------------------------------ transposed conv (lunext.py:161-179)
    @ranged(lambda k, *a: f"fwd/upconv{k}")
    def _up(self, k, x, out=None):
        """level 5-k -> level 4-k; `out`: where the result goes (else a buffer of its own)"""
        l = 4 - k
        up = getattr(self.m, f"upconv{k}")
        cin, cout, cpo = self.chans[l + 1], self.chans[l], self.cp[l]
        N = max(64, 4 * cpo)
        Bs, Hs, Ws = self.geo[l + 1]
        s, Ls = self._sh, self.lease
        wp = self.wf[f"up{k}"]
        B, H, W = self.geo[l]
        dest, exact = out, (2 * Hs, 2 * Ws) == (H, W)
        out = dest if dest is not None and exact else Ls.act(B, 2 * Hs, 2 * Ws, cpo)
        bias = up.bias.detach()
        if cpo != cout:                                    # 3 real channels carried as 8: padded by the weight repack
            bias = self.wf[f"up{k}.bias.f32"]
        call("uh_conv_gemm", Bs, Hs, Ws, x.ptr, x.cstride, x.coff, cin, wp.data_ptr(), N, 1, EPI_CONVT, out.ptr,
             out.cstride, out.coff, bias.data_ptr(), None, None, None, self.masks[l + 1].data_ptr(), s)
        self.saved[f"up{k}"] = (x, N)
        if not exact:                                      # lunext.py:226-227 (and :234, :242, :250)
            fit = dest if dest is not None else Ls.act(B, H, W, cpo)
            call("uh_resize_bilinear_fwd", B, 2 * Hs, 2 * Ws, H, W, cpo, out.ptr, out.cstride, out.coff, fit.ptr,
                 fit.cstride, fit.coff, s)
            return fit
        return out

    @ranged(lambda k, *a: f"bwd/upconv{k}")
    def _up_backward(self, k, dup, grads):
        l = 4 - k
        up = getattr(self.m, f"upconv{k}")
        x, N = self.saved[f"up{k}"]
        cin, cout, cpo = self.chans[l + 1], self.chans[l], self.cp[l]
        Bs, Hs, Ws = self.geo[l + 1]
        s, Ls = self._sh, self.lease
        B, H, W = self.geo[l]
        if (2 * Hs, 2 * Ws) != (H, W):                     # back through the size fix
            raw = Ls.act(B, 2 * Hs, 2 * Ws, cpo)
            call("uh_resize_bilinear_bwd", B, 2 * Hs, 2 * Ws, H, W, cpo, dup.ptr, dup.cstride, dup.coff, raw.ptr,
                 raw.cstride, raw.coff, s)
            dup = raw
        d4 = Ls.act(Bs, Hs, Ws, N, tag=f"up{k}.d4")
        call("uh_space_to_depth2", Bs, Hs, Ws, cpo, dup.ptr, dup.cstride, dup.coff, d4.ptr, d4.cstride, 0, s)
        S = L.uh_conv_wgrad_splits(Bs, Hs, Ws, N, cin, 1)
        part = Ls.f32(S * N * cin)
        dw = self._gnew(f"upconv{k}.weight", up.weight.shape)  # [cin][cout][2][2]
        # one visit to the second stream for the bias and the weight gradient (every visit is an event on the main
        # stream, which costs it a few microseconds)
        with self._side() as s2:
            if self.gflat is None:
                grads[f"upconv{k}.bias"] = self._colsum(dup, cpo, s2)[:cout]   # a view of a tensor made for this gradient
            elif cpo == cout:
                grads[f"upconv{k}.bias"] = self._colsum(dup, cpo, s2, out=self.gflat.views[f"upconv{k}.bias"])
            else:                                                  # padded channels: only the first cout sums are wanted
                grads[f"upconv{k}.bias"] = self._colsum(dup, cpo, s2, out=self.gflat.views[f"upconv{k}.bias"],
                                                        cols=self._table(("first", cout), lambda: list(range(cout))))
            call("uh_conv_wgrad", Bs, Hs, Ws, d4.ptr, d4.cstride, 0, N, x.ptr, x.cstride, x.coff, cin, 1, part.data_ptr(), s2)
            self._later(_lib.ReduceDesc.wgrad(part.data_ptr(), S, N * cin, cout, 4, cin, cin, cpo * cin, 4, 1, cout * 4,
                                              dw.data_ptr()))
        grads[f"upconv{k}.weight"] = dw
        wd = self.wb[f"up{k}"]
        dx = Ls.act(Bs, Hs, Ws, cin)
        call("uh_conv_gemm", Bs, Hs, Ws, d4.ptr, d4.cstride, 0, N, wd.data_ptr(), cin, 1, EPI_MASKED, dx.ptr,
             dx.cstride, dx.coff, None, None, None, None, self.masks[l + 1].data_ptr(), s)
        return dx

    def _bind_streams(self):
        """Looks the current stream up ONCE per pass (torch.cuda.current_stream() costs ~9 us, and a pass asked for it
        ~60 times: the LUNeXt step is bound by what the host spends per launch, not by the GPU)."""
        self._main = torch.cuda.current_stream(self.device)
        self._sh = self._main.cuda_stream
        self._two_streams = os.environ.get("UNET_AMD_WGRAD_STREAM", "1") != "0"
        if self._two_streams:
            self._side_st = side_stream_for(self.device)
            self._fork_event = torch.cuda.Event()

    def _side(self):
        """`with self._side() as s2:` -- kernels launched on s2 run on a second stream, ordered after everything issued
        so far on the main one (UNET_AMD_WGRAD_STREAM=0: s2 is the main stream).  Only launches into buffers that
        already exist belong there.  torch's current stream is NOT switched (every launch takes its stream explicitly):
        no torch operation may sit inside the block.  backward() joins the stream before it returns."""
        if not self._two_streams:
            return _Handle(self._sh)
        self._side_used = self._side_st
        self._fork_event.record(self._main)
        self._side_st.wait_event(self._fork_event)
        return _Handle(self._side_st.cuda_stream)

    def _zeros(self, n):
        """n zeros out of one per-run buffer (gradients that are identically zero; nothing writes to them)."""
        if getattr(self, "_zero_pool", None) is None or self._zero_used + n > self._zero_pool.numel():
            self._zero_pool = torch.zeros(max(4096, n), dtype=torch.float32, device=self.device)
            self._zero_used = 0
        out = self._zero_pool[self._zero_used:self._zero_used + n]
        self._zero_used += n
        return out

    def _gzero(self, name, n):
        """the gradient of a parameter whose gradient is identically zero: in a flat gradient buffer its place is never
        written (the buffer starts as zeros), otherwise n zeros out of a per-run buffer"""
        if self.gflat is not None:
            return self.gflat.views[name]
        return self._zeros(n)

    def _colsum(self, a, C, s=None, out=None, cols=None):
        """column sums of `a` (a bias gradient) into `out` if given; with `cols` (int32 device table) out[i] is the sum of
        column cols[i].  The pass over `a` runs on stream s (default: the main stream), its partial rows are added up
        later with the other small reductions (_flush_small)."""
        s, Ls = (self._sh if s is None else s), self.lease
        parts = L.uh_colsum_parts(a.Mp, C)
        cp = Ls.f32(parts * C)
        call("uh_colsum", a.Mp, C, a.ptr, a.cstride, a.coff, cp.data_ptr(), s)
        if out is None:
            out = torch.empty(C, dtype=torch.float32, device=self.device)
        if cols is not None:
            self._later(_lib.ReduceDesc.colsum(cp.data_ptr(), parts, C, out.data_ptr(), cols.numel(), cols.data_ptr()))
        else:
            self._later(_lib.ReduceDesc.colsum(cp.data_ptr(), parts, C, out.data_ptr()))
        return out

    # ------------------------------------------------------------------ attention gate (lunext.py:76-122)
    @ranged(lambda k, *a: f"fwd/att{k}")
    def _gate(self, k, g, x):
        att = getattr(self.m, f"att{k}")
        l = 4 - k
        B, H, W = self.geo[l]
        C, Cp = self.chans[l], g.C
        s, Ls = self._sh, self.lease
        out = Ls.act(B, H, W, Cp)
        if Cp == 8:
            wts = self.wf[f"att{k}.gate8.f32"]               # the table uh_gate8_pack makes, written by the weight repack
            call("uh_gate8_fwd", g.Mp, C, g.ptr, g.cstride, g.coff, x.ptr, x.cstride, x.coff, wts.data_ptr(),
                 self.masks[l].data_ptr(), out.ptr, out.cstride, out.coff, s)
            self.saved[f"att{k}"] = (g, x, wts)
            return out
        z = Ls.act(B, H, W, 3 * C)
        mask = self.masks[l]
        if g.buf is x.buf and g.cstride == 2 * C and g.coff == 0 and x.coff == C:
            # [g | x] side by side: one GEMM with the rows [W1 0; W2 0; W3 W3] gives all three pre-activations,
            # W3 (g + x) included, without g + x (the backward pass makes that sum on its second stream)
            b1, b2, b3 = (c.bias.detach() for c in (att.sigmoid_conv, att.tanh_conv, att.combine_conv))
            if b1.data_ptr() + 4 * C == b2.data_ptr() and b2.data_ptr() + 4 * C == b3.data_ptr():
                b123 = b1                                 # side by side in a flat parameter buffer: read where they are
            else:
                b123 = torch.cat([b1, b2, b3])
            call("uh_conv_gemm", B, H, W, g.ptr, g.cstride, 0, 2 * C, self.wf[f"att{k}.wcat"].data_ptr(), 3 * C, 1,
                 EPI_MASKED, z.ptr, z.cstride, 0, b123.data_ptr(), None, None, None, mask.data_ptr(), s)
            call("uh_gate_fwd", g.Mp, C, z.ptr, z.cstride, 0, x.ptr, x.cstride, x.coff, out.ptr, out.cstride, out.coff, s)
            self.saved[f"att{k}"] = (g, x, None, z)
            return out
        ssum = Ls.act(B, H, W, C)
        call("uh_add", g.Mp, C, g.ptr, g.cstride, g.coff, x.ptr, x.cstride, x.coff, None, 0, 0, ssum.ptr, ssum.cstride,
             ssum.coff, s)
        w12, w3 = self.wf[f"att{k}.w12"], self.wf[f"att{k}.w3"]
        b1, b2 = att.sigmoid_conv.bias.detach(), att.tanh_conv.bias.detach()
        if b1.data_ptr() + 4 * C == b2.data_ptr():        # side by side (parameters living in a flat buffer laid out
            b12 = b1                                      # by flat_param_order): read as one 2C vector where they are
        else:
            b12 = torch.cat([b1, b2])
        call("uh_conv_gemm", B, H, W, g.ptr, g.cstride, g.coff, C, w12.data_ptr(), 2 * C, 1, EPI_MASKED, z.ptr,
             z.cstride, 0, b12.data_ptr(), None, None, None, mask.data_ptr(), s)
        call("uh_conv_gemm", B, H, W, ssum.ptr, ssum.cstride, ssum.coff, C, w3.data_ptr(), C, 1, EPI_MASKED, z.ptr,
             z.cstride, 2 * C, att.combine_conv.bias.data_ptr(), None, None, None, mask.data_ptr(), s)
        call("uh_gate_fwd", g.Mp, C, z.ptr, z.cstride, 0, x.ptr, x.cstride, x.coff, out.ptr, out.cstride, out.coff, s)
        self.saved[f"att{k}"] = (g, x, ssum, z)
        return out

    @ranged(lambda k, *a: f"bwd/att{k}")
    def _gate_backward(self, k, dout, grads):
        """returns (dg, dx)"""
        att = getattr(self.m, f"att{k}")
        l = 4 - k
        B, H, W = self.geo[l]
        C = self.chans[l]
        s, Ls = self._sh, self.lease
        pre = f"att{k}."
        if self.cp[l] == 8 and C < 64:
            g, x, wts = self.saved[f"att{k}"]
            dg, dx = Ls.act(B, H, W, 8), Ls.act(B, H, W, 8)
            parts = L.uh_gate8_parts(g.Mp)
            part = Ls.f32(parts * 216)
            call("uh_gate8_bwd", g.Mp, C, dout.ptr, dout.cstride, dout.coff, g.ptr, g.cstride, g.coff, x.ptr, x.cstride,
                 x.coff, wts.data_ptr(), dg.ptr, dg.cstride, dg.coff, dx.ptr, dx.cstride, dx.coff, part.data_ptr(), s)
            names = ("sigmoid_conv", "tanh_conv", "combine_conv")
            if self.gflat is not None:
                # the three [C][C] weights, then the three biases, lie side by side in the flat buffer
                # (flat_param_order): one finalisation picks their 3 C (C + 1) sums out of the padded [3][8][8] + [3][8]
                cols = self._table(("gate8", C), lambda: [64 * j + 8 * r + c for j in range(3) for r in range(C)
                                                          for c in range(C)] + [192 + 8 * j + c for j in range(3)
                                                                                for c in range(C)])
                out = self.gflat.span(pre + "sigmoid_conv.weight", cols.numel())
                self._later(_lib.ReduceDesc.colsum(part.data_ptr(), parts, 216, out.data_ptr(), cols.numel(), cols.data_ptr()))
                for nm in names:
                    grads[pre + nm + ".weight"] = self.gflat.views[pre + nm + ".weight"]
                    grads[pre + nm + ".bias"] = self.gflat.views[pre + nm + ".bias"]
                return dg, dx
            gw = torch.empty(216, dtype=torch.float32, device=self.device)   # made for these gradients: views below
            self._later(_lib.ReduceDesc.colsum(part.data_ptr(), parts, 216, gw.data_ptr()))
            for j, nm in enumerate(names):
                grads[pre + nm + ".weight"] = gw[64 * j:64 * j + 64].view(8, 8)[:C, :C].reshape(C, C, 1, 1)
                grads[pre + nm + ".bias"] = gw[192 + 8 * j:192 + 8 * j + C]
            return dg, dx
        g, x, ssum, z = self.saved[f"att{k}"]
        mask = self.masks[l]
        dz, dxd = Ls.act(B, H, W, 3 * C), Ls.act(B, H, W, C)
        # the three bias gradients are the column sums of dz: taken by the launch that writes it
        bparts = L.uh_gate_bwd_parts(g.Mp, C)
        bpart = Ls.f32(bparts * 3 * C)
        call("uh_gate_bwd_colsum", g.Mp, C, dout.ptr, dout.cstride, dout.coff, z.ptr, z.cstride, 0, x.ptr, x.cstride,
             x.coff, mask.data_ptr(), dz.ptr, dz.cstride, 0, dxd.ptr, dxd.cstride, dxd.coff, bpart.data_ptr(), s)
        # bias and weight gradients only read dz, g and g + x: on the second stream, beside the input gradients below
        # (weight gradients: [dW1; dW2] = dz[:, :2C]^T g,  dW3 = dz[:, 2C:]^T (g + x))
        concat = ssum is None                         # the forward pass ran on [g | x] and never made g + x
        names = ("sigmoid_conv", "tanh_conv", "combine_conv")
        dws = [self._gnew(pre + nm + ".weight", (C, C, 1, 1)) for nm in names]
        wd3 = self.wb[f"att{k}.wd3"]
        ds = Ls.act(B, H, W, C)
        dx = Ls.act(B, H, W, C)
        if concat:
            # one weight-gradient GEMM dz^T [g | x] (3C x 2C): its blocks (1, g), (2, g) are dW1, dW2 and
            # (3, g) + (3, x) is dW3 = dz_3^T (g + x) -- the same bytes as the two GEMMs over g and g + x, one launch
            # less and no g + x
            S = L.uh_conv_wgrad_splits(B, H, W, 3 * C, 2 * C, 1)
            part = Ls.f32(S * 6 * C * C)
        else:
            S = L.uh_conv_wgrad_splits(B, H, W, 2 * C, C, 1)
            part = Ls.f32(S * 2 * C * C)
            S3 = L.uh_conv_wgrad_splits(B, H, W, C, C, 1)
            part3 = Ls.f32(S3 * C * C)
        with self._side() as s2:
            # in a flat gradient buffer the three biases lie side by side (flat_param_order)
            db = (torch.empty(3 * C, dtype=torch.float32, device=self.device) if self.gflat is None else
                  self.gflat.span(pre + "sigmoid_conv.bias", 3 * C))
            self._later(_lib.ReduceDesc.colsum(bpart.data_ptr(), bparts, 3 * C, db.data_ptr()))
            if concat:
                call("uh_conv_wgrad", B, H, W, dz.ptr, dz.cstride, 0, 3 * C, g.ptr, g.cstride, 0, 2 * C, 1,
                     part.data_ptr(), s2)
                for j in range(3):                            # the blocks over g: rows j C .. of 2C columns each
                    self._later(_lib.ReduceDesc.wgrad(part.data_ptr() + j * C * 2 * C * 4, S, 6 * C * C, C, 1, C, 2 * C,
                                                      0, C, 0, 1, dws[j].data_ptr()))
                # ... and the block (3, x) on top of dW3, in a launch after the one that wrote it
                self._later(_lib.ReduceDesc.wgrad(part.data_ptr() + (2 * C * 2 * C + C) * 4, S, 6 * C * C, C, 1, C, 2 * C,
                                                  0, C, 0, 1, dws[2].data_ptr(), accumulate=1), second=True)
            else:
                call("uh_conv_wgrad", B, H, W, dz.ptr, dz.cstride, 0, 2 * C, g.ptr, g.cstride, g.coff, C, 1,
                     part.data_ptr(), s2)
                for j in range(2):
                    self._later(_lib.ReduceDesc.wgrad(part.data_ptr() + j * C * C * 4, S, 2 * C * C, C, 1, C, C, 0, C, 0, 1,
                                                      dws[j].data_ptr()))
                call("uh_conv_wgrad", B, H, W, dz.ptr, dz.cstride, 2 * C, C, ssum.ptr, ssum.cstride, ssum.coff, C, 1,
                     part3.data_ptr(), s2)
                self._later(_lib.ReduceDesc.wgrad(part3.data_ptr(), S3, C * C, C, 1, C, C, 0, C, 0, 1, dws[2].data_ptr()))
            # dx = dout * a + dz_3 W_3 is the gradient of the skip connection: nothing needs it before the encoder's
            # backward reaches this level, so its two launches are on the second stream too, in the same visit
            # (backward() makes the main stream wait for `ready` there)
            call("uh_conv_gemm", B, H, W, dz.ptr, dz.cstride, 2 * C, C, wd3.data_ptr(), C, 1, EPI_MASKED, ds.ptr,
                 ds.cstride, ds.coff, None, None, None, None, mask.data_ptr(), s2)
            call("uh_add", g.Mp, C, dxd.ptr, dxd.cstride, dxd.coff, ds.ptr, ds.cstride, ds.coff, None, 0, 0, dx.ptr,
                 dx.cstride, dx.coff, s2)
        if self._two_streams:
            ready = torch.cuda.Event()
            ready.record(self._side_st)
            self._skip_ready[l] = ready
        for j, nm in enumerate(("sigmoid_conv", "tanh_conv", "combine_conv")):
            grads[pre + nm + ".bias"] = db[j * C:(j + 1) * C]
            grads[pre + nm + ".weight"] = dws[j]
        # input gradients: dg = sum_j dz_j W_j (one GEMM over K = 3C),  ds = dz_3 W_3,  dx = dout*a + ds
        wd = self.wb[f"att{k}.wd"]
        dg = Ls.act(B, H, W, C)
        call("uh_conv_gemm", B, H, W, dz.ptr, dz.cstride, 0, 3 * C, wd.data_ptr(), C, 1, EPI_MASKED, dg.ptr,
             dg.cstride, dg.coff, None, None, None, None, mask.data_ptr(), s)
        return dg, dx

    # ------------------------------------------------------------------ whole network
    def forward(self):
        self._bind_streams()
        m, Ls, s = self.m, self.lease, self._sh
        if self.training:
            self.wf, self.wb = self.packer.get_both(s)          # a training step repacks both directions at once
        else:
            self.wf = self.packer.get("fwd", s)
        B, H, W = self.geo[0]
        cin = self.chans[0]
        cur = Ls.act(B, H, W, 8)
        call("uh_nchw_to_c8", B, cin, H, W, self.x.data_ptr(), cur.ptr, cur.cstride, cur.coff, s)
        skips, pair = [], {}
        for l in range(1, 5):                                   # lunext.py:199-216
            lv, Cl = l - 1, self.chans[l - 1]
            if GATE_CONCAT and Cl >= 64:
                # the skip connection x of this level and the transposed convolution's output g that meets it in the
                # attention gate share one buffer, [g | x] per position, which the gate's GEMM reads in one go
                Bl, Hl, Wl = self.geo[lv]
                gv = Ls.act(Bl, Hl, Wl, Cl, cstride=2 * Cl)
                pair[lv] = (gv, Act(gv.buf, Bl, Hl, Wl, Cl, 2 * Cl, Cl, gv.guard))
            cur = self._dw(f"conv{l}_2", self._dw(f"conv{l}_1", cur, lv, defer_apply=True), lv,
                           out=pair[lv][1] if lv in pair else None,
                           depth=self._x4(l, self.cp[lv]) if self.training else None)
            skips.append(cur)
            cur = self._down(l, cur)
        cur = self._dw("conv5_2", self._dw("conv5_1", cur, 4, defer_apply=True), 4)
        for k in range(1, 5):                                   # lunext.py:223-254
            l = 4 - k
            up = self._up(k, cur, out=pair[l][0] if l in pair else None)
            cur = self._gate(k, up, skips[l])
            cur = self._dw(f"conv_up{k}_2", self._dw(f"conv_up{k}_1", cur, l, defer_apply=True), l)
        oc = m.out_channels
        fc = m.final_conv
        w = self.wf["head.w.f32"].view(oc, 8)                    # [oc][cin] zero-padded to 8 by the weight repack
        logits = torch.empty(B, oc, H, W, dtype=torch.float32, device=self.device)
        call("uh_head_fwd", B, H, W, 8, oc, cur.ptr, cur.cstride, cur.coff, w.data_ptr(), fc.bias.data_ptr(),
             logits.data_ptr(), s)
        self.head_in, self.head_w = cur, w
        if self._pending:
            raise RuntimeError("a depthwise block's output was left to a launch that never ran: " +
                               ", ".join(r.name for r in self._pending.values()))
        return logits

    def backward(self, dlogits, gflat=None):
        """{parameter name: fp32 gradient}.  gflat: a dp.FlatGrads laid out by flat_param_order(module): the gradients
        are written into its views (those of att*.output_conv, which the forward pass never uses, stay untouched)."""
        if not self.training:
            raise RuntimeError("backward through an eval-mode forward is not built; call model.train() first")
        self.gflat = gflat
        self._bind_streams()
        m, Ls, s = self.m, self.lease, self._sh
        self.wb = self.packer.get("bwd", s)
        grads = {}
        B, H, W = self.geo[0]
        cin, oc = self.chans[0], m.out_channels
        dl = dlogits.contiguous().float()
        a = self.head_in
        da = Ls.act(B, H, W, 8)
        parts = L.uh_head_bwd_parts(B, H, W, 8)
        hp = Ls.f32(parts * oc * 9)
        call("uh_head_bwd", B, H, W, 8, oc, a.ptr, a.cstride, a.coff, self.head_w.data_ptr(), dl.data_ptr(), da.ptr,
             da.cstride, da.coff, hp.data_ptr(), s)
        if self.gflat is not None:
            cols = self._table(("head", oc, cin), lambda: [o * 9 + c for o in range(oc) for c in range(cin)] +
                               [o * 9 + 8 for o in range(oc)])
            out = self.gflat.span("final_conv.weight", cols.numel())       # weight, then bias: side by side
            self._later(_lib.ReduceDesc.colsum(hp.data_ptr(), parts, oc * 9, out.data_ptr(), cols.numel(), cols.data_ptr()))
            grads["final_conv.weight"] = self.gflat.views["final_conv.weight"]
            grads["final_conv.bias"] = self.gflat.views["final_conv.bias"]
        else:
            hw = torch.empty(oc, 9, dtype=torch.float32, device=self.device)    # made for these gradients: views below
            self._later(_lib.ReduceDesc.colsum(hp.data_ptr(), parts, oc * 9, hw.data_ptr()))
            grads["final_conv.weight"] = hw[:, :cin].reshape(oc, cin, 1, 1)
            grads["final_conv.bias"] = hw[:, 8]
        skip_grads = [None] * 4
        self._skip_ready = {}
        for k in range(4, 0, -1):
            l = 4 - k
            da = self._dw_backward(f"conv_up{k}_1", self._dw_backward(f"conv_up{k}_2", da, grads, prev=f"conv_up{k}_1"), grads)
            dg, dx = self._gate_backward(k, da, grads)
            skip_grads[l] = dx
            da = self._up_backward(k, dg, grads)
        da = self._dw_backward("conv5_1", self._dw_backward("conv5_2", da, grads, prev="conv5_1"), grads, colsum=True)
        self._flush_small()                         # the decoder's and the bridge's, beside the encoder's backward
        for l in range(4, 0, -1):
            if (l - 1) in self._skip_ready:
                self._main.wait_event(self._skip_ready[l - 1])
            tot = self._down_backward(l, da, grads, skip_grads[l - 1])
            self._flush_small()                     # beside this level's blocks; only their own tables are left for the end
            da = self._dw_backward(f"conv{l}_1", self._dw_backward(f"conv{l}_2", tot, grads, prev=f"conv{l}_1"), grads,
                                   colsum=l > 1)
        self._flush_small()
        if self._side_used is not None:
            self._main.wait_stream(self._side_used)
        return grads

    def release(self):
        self.recs, self.saved = {}, {}
        self.lease.release()


def flat_param_order(module):
    """Parameter names in the order a flat gradient buffer must hold them for LUNeXtRun.backward(gflat=...): declaration
    order, except that each attention gate's three used 1x1 weights come first, then its three biases, so that the
    tensors one kernel pass produces together lie side by side."""
    order = []
    for n, _ in module.named_parameters():
        if n.startswith("att") and ".output_conv." not in n:
            pre = n.split(".")[0]
            if n == pre + ".sigmoid_conv.weight":
                order += [f"{pre}.{c}.weight" for c in ("sigmoid_conv", "tanh_conv", "combine_conv")]
                order += [f"{pre}.{c}.bias" for c in ("sigmoid_conv", "tanh_conv", "combine_conv")]
            continue
        order.append(n)
    assert sorted(order) == sorted(n for n, _ in module.named_parameters())
    return order


class _LUNeXtFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, module, x, *params):
        run = LUNeXtRun(module, x, training=True)
        logits = run.forward()
        ctx.run = run
        ctx.names = module._param_names()
        return logits

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dlogits):
        run = ctx.run
        if run is None:
            raise RuntimeError("the buffers of this forward pass were already released (second backward?)")
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("gradient with respect to the input image is not built")
        grads = run.backward(dlogits)
        run.release()
        ctx.run = None
        out = [None, None]
        for i, n in enumerate(ctx.names):
            out.append(grads.get(n) if ctx.needs_input_grad[2 + i] else None)   # att*.output_conv.*: None
        return tuple(out)


class _DWParams(nn.Module):
    """Parameter container with the keys of DepthwiseConvBlock (lunext.py:14-16)."""

    def __init__(self, c):
        super().__init__()
        self.depthwise = nn.Conv2d(c, c, kernel_size=3, padding=1, groups=c)
        self.bn = nn.BatchNorm2d(c)


class _GateParams(nn.Module):
    """Parameter container with the keys of AttentionGate (lunext.py:67-74)."""

    def __init__(self, c):
        super().__init__()
        self.sigmoid_conv = nn.Conv2d(c, c, kernel_size=1, bias=True)
        self.tanh_conv = nn.Conv2d(c, c, kernel_size=1, bias=True)
        self.combine_conv = nn.Conv2d(c, c, kernel_size=1, bias=True)
        self.output_conv = nn.Conv2d(c, c, kernel_size=1, bias=True)


class LUNeXt(nn.Module):
    def __init__(self, in_channels=3, out_channels=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        ch = [in_channels, 64, 128, 256, 512]
        for l in range(1, 5):                                   # lunext.py:137-154
            setattr(self, f"conv{l}_1", _DWParams(ch[l - 1]))
            setattr(self, f"conv{l}_2", _DWParams(ch[l - 1]))
            setattr(self, f"downconv{l}", nn.Conv2d(ch[l - 1], ch[l], kernel_size=2, stride=2))
        self.conv5_1 = _DWParams(512)
        self.conv5_2 = _DWParams(512)
        for k in range(1, 5):                                   # lunext.py:161-182
            ci, co = ch[5 - k], ch[4 - k]
            setattr(self, f"upconv{k}", nn.ConvTranspose2d(ci, co, kernel_size=2, stride=2))
            setattr(self, f"att{k}", _GateParams(co))
            setattr(self, f"conv_up{k}_1", _DWParams(co))
            setattr(self, f"conv_up{k}_2", _DWParams(co))
        self.final_conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)

    def _param_names(self):
        names = self.__dict__.get("_names_cache")
        if names is None:
            names = self.__dict__["_names_cache"] = [n for n, _ in self.named_parameters()]
        return names

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("unet_amd.LUNeXt runs on the HIP kernels only: move the model and the input to a "
                               "ROCm device (there is no CPU path)")
        if x.dim() != 4:
            raise ValueError(f"expected a [B,C,H,W] tensor, got shape {tuple(x.shape)}")
        params = list(self.parameters())
        need_grad = torch.is_grad_enabled() and self.training and any(p.requires_grad for p in params)
        if need_grad:
            return _LUNeXtFn.apply(self, x.float().contiguous(), *params)
        with torch.no_grad():
            run = LUNeXtRun(self, x.detach().float().contiguous(), training=self.training)
            logits = run.forward()
            run.release()
        return logits
