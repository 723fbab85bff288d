"""Host-side schedule of the U-Net forward and backward passes over the HIP kernels.

The layer order, channel widths, skip wiring and what is saved follow models/unet.py:86-176 of
the reference; everything the device executes is an entry point of libunet_hip.so.  One
`UNetRun` object holds the buffers of one forward pass and can run its backward pass.
"""
import contextlib
import os

import torch

from . import _lib
from ._lib import call, EPI_RAW_STATS, EPI_AFFINE_RELU, EPI_MASKED, EPI_CONVT
from .layout import pool_for
from .packing import packer_for

BN_EPS = 1e-5        # nn.BatchNorm2d defaults used by the reference (unet.py:20)
BN_MOMENTUM = 0.1

WIDTHS = (64, 128, 256, 512, 1024)


# UNET_AMD_FUSE_BN_BWD=1: BatchNorm-backward sums are taken in the epilogue of the input gradient that produces the
# tensor they are sums of (uh_conv_gemm_bnbwd) instead of in a pass of their own (uh_bn_relu_bwd_reduce).  Off by
# default: measured on the training step (batch 32, 256^2) the nine passes it saves and the longer epilogue cancel
# (14.03 against 14.06 ms; the 3x3 kernel drops from 904 to 853 TFLOP/s in the step) -- the epilogue waits for the
# layer's fp16 conv output, and loads that return in order cannot stay in flight across the main loop's staging waits.
def fuse_bn_bwd(lvl=0):
    if os.environ.get("UNET_AMD_FUSE_BN_BWD", "0") == "1":
        return True
    return str(lvl) in os.environ.get("UNET_AMD_FUSE_BN_BWD_LEVELS", "")



# UNET_AMD_DGRAD_CUT_LEVELS: resolution levels (0 = full size) whose 3x3 input gradients are cut evenly over the CUs
# (workspace + a second kernel for the cut tiles) instead of taking whole tiles per workgroup.
def dgrad_cut_levels():
    return os.environ.get("UNET_AMD_DGRAD_CUT_LEVELS", "")


# UNET_AMD_FUSE_POOL_STATS=0: the BatchNorm-backward sums of conv{l}_2 and conv_up4_2 are taken in a pass of their own
# (uh_bn_relu_bwd_reduce).  By default the bandwidth-bound kernels that write those layers' output gradients
# (uh_maxpool2_bwd_bnstat, uh_head_bwd_bnstat) take them on the way: one read of the gradient and one launch less each.
def fuse_pool_stats():
    return os.environ.get("UNET_AMD_FUSE_POOL_STATS", "1") == "1"


# UNET_AMD_FUSE_POOL_FWD=0: train-mode MaxPool2d is a pass of its own (uh_maxpool2_fwd) after uh_bn_relu_apply instead of
# being written by the pass that applies BatchNorm + ReLU to conv{l}_2 (uh_bn_relu_apply_pool2).
def fuse_pool_fwd():
    return os.environ.get("UNET_AMD_FUSE_POOL_FWD", "1") == "1"


# UNET_AMD_FUSE_T1_STATS=1: the input gradient of a transposed convolution also takes the BatchNorm-backward sums of
# the layer whose output gradient it writes (uh_conv_gemm1_bnbwd) instead of leaving them to uh_bn_relu_bwd_reduce.
# Off by default, as UNET_AMD_FUSE_BN_BWD: measured on the training step the four passes it saves and the longer
# epilogues cancel (13.72 against 13.74 ms).
def fuse_t1_stats():
    return os.environ.get("UNET_AMD_FUSE_T1_STATS", "0") == "1"



# UNET_AMD_FUSED_FINALIZE=1: BatchNorm sums are added up by the last workgroups of the launch that takes them
# (uh_bn_relu_bwd_reduce_fin; LUNeXt: uh_bn_res_relu_bwd_reduce_fin, uh_dwconv3x3_fwd_bn) instead of by a launch of their
# own (uh_bn_bwd_finalize / uh_bn_finalize).  Off by default: 18 (U-Net) and 36 (LUNeXt) launches fewer per step, but the
# two dependent reduction stages at the end of the producing kernel take what the small launch and its gap took.
# Measured in alternation in one call (batch 32, 256^2): U-Net 13.27 / 13.25 ms with, 13.24 / 13.26 without; LUNeXt +
# clDice 5.30 / 5.29 with, 5.23 / 5.23 without.
def fused_finalize():
    return os.environ.get("UNET_AMD_FUSED_FINALIZE", "0") == "1"


_SIDE_STREAMS = {}


# UNET_AMD_ROCTX=1: every layer's launches are bracketed by a roctx range ("fwd/conv3_1", "bwd/conv3_1", ...), which
# rocprofv3 --marker-trace / the torch profiler show beside the kernel trace.  Off by default (two host calls per layer).
ROCTX = os.environ.get("UNET_AMD_ROCTX", "0") == "1"       # read once, at import
_NO_RANGE = contextlib.nullcontext()


def layer_range(name):
    return torch.cuda.nvtx.range(name) if ROCTX else _NO_RANGE


def ranged(label):
    """Method decorator: label(*args) names a roctx range around the call; the method is returned as it is when
    ranges are off."""
    def deco(fn):
        if not ROCTX:
            return fn

        def wrapped(self, *args, **kwargs):
            with torch.cuda.nvtx.range(label(*args)):
                return fn(self, *args, **kwargs)
        return wrapped
    return deco


def side_stream_for(device, which="wgrad"):
    """A further stream of a device (one per process, device and `which`): "wgrad" is the engine's second stream,
    "opt" the data-parallel trainer's third (collective waits + optimizer).

    It should own a hardware queue.  A process has few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and HIP binds a
    stream to one of them, in turn, when the stream is first used; a second stream first used after a process group
    had made its streams was observed on the main stream's queue, which serialises the two (training step 15.2 instead of
    14.4 ms).  So the stream is used once right here, at creation, and the data-parallel trainer creates it before its
    first collective.  UNET_AMD_SIDE_PRIORITY picks its priority (default 0, normal).  High priority (-1) also gets a
    queue of its own, but then the queue whose head is a wait for an event of the main stream outranks the main
    stream's queue: when both were placed on one hardware pipe (seen when two processes shared a GPU) the main queue
    was not served any more and the step never finished."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _SIDE_STREAMS.get((idx, which))
    if st is None:
        prio = int(os.environ.get("UNET_AMD_SIDE_PRIORITY", "0"))
        st = _SIDE_STREAMS[(idx, which)] = torch.cuda.Stream(device=torch.device("cuda", idx), priority=prio)
        with torch.cuda.stream(st):
            torch.zeros(1, device=torch.device("cuda", idx))      # first use: binds the hardware queue now
        st.synchronize()
    return st


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return t.data_ptr() if t is not None else None


class _ConvRec:
    """Saved state of one conv3x3+BN+ReLU layer."""
    __slots__ = ("name", "a_in", "y", "a_out", "cin", "cout", "mean", "invstd", "scale", "shift")


class UNetRun:
    """One forward pass (and optionally its backward) of the U-Net on one device."""

    def __init__(self, module, x, training):
        self.m = module
        self.training = training
        self.device = x.device
        self.pool = pool_for(x.device)
        self.lease = self.pool.lease()
        B, Cin, H, W = x.shape
        if H % 16 or W % 16:
            raise ValueError(f"UNet needs H and W divisible by 16 (four 2x2 poolings), got {H}x{W}")
        if Cin != module.in_channels:
            raise ValueError(f"expected {module.in_channels} input channels, got {Cin}")
        if Cin > 4:
            raise NotImplementedError("the first-layer kernel is built for in_channels <= 4")
        self.B, self.H, self.W, self.Cin = B, H, W, Cin
        self.x = x
        self.recs = {}
        self.geo = [(B, H >> l, W >> l) for l in range(5)]
        self.masks = [self.pool.rowmask(*g) for g in self.geo]
        self.grad_alloc = self.on_ready = None
        # set by a caller whose grad_alloc hands out the same memory every step and never writes to it: the gradients
        # that are identically zero (conv biases in front of train-mode BatchNorm) then need no fill after the first step
        self.zero_grads_kept = False
        self._fused_stats = {}     # layer name -> (partial, parts) left by the input gradient that feeds it
        self.timer = None          # optional KernelTimer (bench): times selected kernel launches with events
        self._ws, self._ws_bytes = None, 0
        self.packer = packer_for(module, x.device)
        # Packed bf16 operands.  The first layer reads the fp32 weights themselves, so in training both repacks go to
        # the second stream (which first waits for the optimizer step that changed the weights): the forward operands
        # are awaited just before the second layer, the operands of the input gradients at the start of backward.
        side = self._side_stream() if training else None
        self._wf_ready = None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            self.wf = self.packer.get("fwd", side.cuda_stream)
            self._wf_ready = torch.cuda.Event()
            self._wf_ready.record(side)
            self.packer.get("bwd", side.cuda_stream)
        else:
            self.wf = self.packer.get("fwd", _stream())   # by layer name

    # ------------------------------------------------------------------ helpers
    def _gemm(self, *args, whole_tiles=False):
        """uh_conv_gemm; with a KernelTimer attached, bracketed by events on the launching stream.
        whole_tiles: no workspace, so the launch is not cut evenly over the CUs (no second kernel for the cut tiles)."""
        B, H, W, K, N, T = args[0], args[1], args[2], args[6], args[8], args[9]
        # one scratch buffer per run serves every 3x3 launch (they are ordered on the stream)
        need = 0 if whole_tiles else _lib.lib.uh_conv_gemm_workspace(B, H, W, K, N, T)
        if need > self._ws_bytes:
            self._ws = self.lease.f32((need + 3) // 4)
            self._ws_bytes = need
        args = args[:-1] + (self._ws.data_ptr() if need else None, self._ws_bytes if need else 0, args[-1])
        if self.timer is None:
            return call("uh_conv_gemm_ws", *args)
        # T = 9 runs conv_band_kernel, T = 1 (transposed convs as GEMMs) conv_gemm_kernel
        with self.timer.launch("conv_band" if T == 9 else "conv_gemm_t1", 2.0 * B * H * W * K * N * T):
            call("uh_conv_gemm_ws", *args)

    def _side_stream(self):
        """second stream for the weight gradients of the GEMM layers; UNET_AMD_WGRAD_STREAM=0 keeps everything on
        one stream (measured on the training step at batch 32, 256^2: 15.04 ms with one stream, 14.54 ms with two)"""
        if os.environ.get("UNET_AMD_WGRAD_STREAM", "1") == "0":
            return None
        st = side_stream_for(self.device)
        self._side_used = st
        return st

    def join_side_stream(self):
        """the current stream waits for everything issued on the second stream"""
        st = getattr(self, "_side_used", None)
        if st is not None:
            torch.cuda.current_stream().wait_stream(st)

    def _wgrad(self, *args):
        if self.timer is None:
            return call("uh_conv_wgrad", *args)
        B, H, W, N, K, T = args[0], args[1], args[2], args[6], args[10], args[11]
        with self.timer.launch("conv_wgrad", 2.0 * B * H * W * K * N * T):
            call("uh_conv_wgrad", *args)

    def _bn_params(self, C):
        f = self.lease.f32(4 * C)
        return f[0:C], f[C:2 * C], f[2 * C:3 * C], f[3 * C:4 * C]     # mean, invstd, scale, shift

    def _finish_bn(self, rec, lvl, stat, parts, conv, bn, pooled=None):
        """statistics -> scale/shift (+ running stats), then a_out = relu(y*scale+shift); pooled = MaxPool2d(2,2) of
        a_out in the same pass when given."""
        B, H, W = self.geo[lvl]
        C = rec.cout
        rec.mean, rec.invstd, rec.scale, rec.shift = self._bn_params(C)
        call("uh_bn_finalize", stat.data_ptr(), parts, C, C, B * H * W, bn.weight.data_ptr(), bn.bias.data_ptr(),
             BN_EPS, BN_MOMENTUM, rec.mean.data_ptr(), rec.invstd.data_ptr(), rec.scale.data_ptr(),
             rec.shift.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
             bn.num_batches_tracked.data_ptr(), _stream())
        y, a = rec.y, rec.a_out
        if pooled is not None:
            call("uh_bn_relu_apply_pool2", B, H, W, C, y.ptr, y.cstride, y.coff, rec.scale.data_ptr(),
                 rec.shift.data_ptr(), a.ptr, a.cstride, a.coff, pooled.ptr, pooled.cstride, pooled.coff, _stream())
            return
        call("uh_bn_relu_apply", y.Mp, C, y.ptr, y.cstride, y.coff, rec.scale.data_ptr(), rec.shift.data_ptr(),
             self.masks[lvl].data_ptr(), a.ptr, a.cstride, a.coff, _stream())

    def _cbr(self, name, a_in, lvl, a_out, pooled=None):
        """conv3x3 + BN + ReLU (unet.py:99 etc.).  a_in is None for the first layer (reads self.x).  pooled: where the
        2x2 max pooling of the output goes (train mode: written by the pass that writes a_out); returns True if done."""
        with layer_range("fwd/" + name):
            return self._cbr_launch(name, a_in, lvl, a_out, pooled)

    def _cbr_launch(self, name, a_in, lvl, a_out, pooled=None):
        conv = getattr(self.m, name)
        bn = getattr(self.m, name.replace("conv", "bn", 1))
        B, H, W = self.geo[lvl]
        cout = conv.out_channels
        cin = conv.in_channels
        rec = _ConvRec()
        rec.name, rec.a_in, rec.a_out, rec.cin, rec.cout = name, a_in, a_out, cin, cout
        s = _stream()
        if self.training:
            rec.y = self.lease.act(B, H, W, cout)
            if a_in is None:
                parts = _lib.lib.uh_conv_first_parts(B, H, W)
                stat = self.lease.f32(parts * 2 * cout)
                call("uh_conv_first_fwd", B, H, W, cin, cout, self.x.data_ptr(), conv.weight.data_ptr(),
                     conv.bias.data_ptr(), EPI_RAW_STATS, None, None, rec.y.ptr, rec.y.cstride, rec.y.coff,
                     stat.data_ptr(), s)
            else:
                if self._wf_ready is not None:
                    torch.cuda.current_stream().wait_event(self._wf_ready)
                    self._wf_ready = None
                wp = self.wf[name]
                parts = _lib.lib.uh_conv_gemm_mtiles(B, H, W, cout)
                stat = self.lease.f32(parts * 2 * cout)
                self._gemm(B, H, W, a_in.ptr, a_in.cstride, a_in.coff, cin, wp.data_ptr(), cout, 9,
                     EPI_RAW_STATS, rec.y.ptr, rec.y.cstride, rec.y.coff, conv.bias.data_ptr(), None, None,
                     stat.data_ptr(), self.masks[lvl].data_ptr(), s)
            self._finish_bn(rec, lvl, stat, parts, conv, bn, pooled)
            self.recs[name] = rec
            return pooled is not None
        else:
            f = self.lease.f32(2 * cout)
            scale, shift = f[:cout], f[cout:]
            call("uh_bn_fold_eval", cout, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                 bn.running_var.data_ptr(), conv.bias.data_ptr(), BN_EPS, scale.data_ptr(), shift.data_ptr(), s)
            if a_in is None:
                call("uh_conv_first_fwd", B, H, W, cin, cout, self.x.data_ptr(), conv.weight.data_ptr(),
                     conv.bias.data_ptr(), EPI_AFFINE_RELU, scale.data_ptr(), shift.data_ptr(), a_out.ptr,
                     a_out.cstride, a_out.coff, None, s)
            else:
                wp = self.wf[name]
                self._gemm(B, H, W, a_in.ptr, a_in.cstride, a_in.coff, cin, wp.data_ptr(), cout, 9,
                     EPI_AFFINE_RELU, a_out.ptr, a_out.cstride, a_out.coff, None, scale.data_ptr(),
                     shift.data_ptr(), None, self.masks[lvl].data_ptr(), s)
        return False

    # ------------------------------------------------------------------ forward
    def forward(self, count_into=None):
        """count_into = (masks [B,1,H,W] fp32, counts int64 [4] or [B,4], per_sample, want_logits): the head launch
        also binarises the logits and adds the confusion counts (evaluation, k_fold.py:289-290); returns None instead
        of the logits when want_logits is False."""
        m, L = self.m, self.lease
        s = _stream()
        acts = {}
        # concat buffers, one per decoder stage: [skip | upsampled] (unet.py:137,147,157,167)
        cat = [L.act(*self.geo[l], WIDTHS[l], cstride=2 * WIDTHS[l]) for l in range(4)]
        self.cat = cat
        cur = None
        for l in range(5):                                   # encoder + bridge (unet.py:99-128)
            B, H, W = self.geo[l]
            w = WIDTHS[l]
            a1 = L.act(B, H, W, w)
            self._cbr(f"conv{l + 1}_1", cur, l, a1)
            a2 = cat[l].window(0, w) if l < 4 else L.act(B, H, W, w)
            p = L.act(B, H // 2, W // 2, w) if l < 4 else None
            pooled = self._cbr(f"conv{l + 1}_2", a1, l, a2, p if fuse_pool_fwd() else None)
            acts[f"conv{l + 1}_1"], acts[f"conv{l + 1}_2"] = a1, a2
            if l < 4:
                if not pooled:
                    call("uh_maxpool2_fwd", B, H, W, w, a2.ptr, a2.cstride, a2.coff, p.ptr, p.cstride, p.coff, s)
                acts[f"pool{l + 1}"] = p
                cur = p
            else:
                cur = a2
        for k in range(1, 5):                                # decoder (unet.py:132-169)
            l = 4 - k                                        # level of the output
            w = WIDTHS[l]
            up = getattr(m, f"upconv{k}")
            Bs, Hs, Ws = self.geo[l + 1]
            wp = self.wf[f"upconv{k}"]                        # GEMM rows n = (a*2+b)*Cout + co
            dst = cat[l].window(w, w)
            with layer_range(f"fwd/upconv{k}"):
                self._gemm(Bs, Hs, Ws, cur.ptr, cur.cstride, cur.coff, 2 * w, wp.data_ptr(), 4 * w, 1,
                           EPI_CONVT, dst.ptr, dst.cstride, dst.coff, up.bias.data_ptr(), None, None, None,
                           self.masks[l + 1].data_ptr(), s)
            acts[f"upconv{k}_in"] = cur
            B, H, W = self.geo[l]
            a1 = L.act(B, H, W, w)
            self._cbr(f"conv_up{k}_1", cat[l].window(0, 2 * w), l, a1)
            a2 = L.act(B, H, W, w)
            self._cbr(f"conv_up{k}_2", a1, l, a2)
            acts[f"conv_up{k}_1"], acts[f"conv_up{k}_2"] = a1, a2
            cur = a2
        oc = m.out_channels
        fc = m.final_conv
        self.acts = acts
        self.head_in = cur
        if count_into is not None:
            masks, counts, per_sample, want_logits = count_into
            if oc != 1:
                raise NotImplementedError("confusion counts inside the head are built for one output channel")
            logits = (torch.empty(self.B, 1, self.H, self.W, dtype=torch.float32, device=self.device)
                      if want_logits else None)
            call("uh_head_fwd_counts", self.B, self.H, self.W, 64, cur.ptr, cur.cstride, cur.coff,
                 fc.weight.data_ptr(), fc.bias.data_ptr(), _ptr(logits), masks.data_ptr(), int(per_sample),
                 counts.data_ptr(), s)
            return logits
        logits = torch.empty(self.B, oc, self.H, self.W, dtype=torch.float32, device=self.device)
        call("uh_head_fwd", self.B, self.H, self.W, 64, oc, cur.ptr, cur.cstride, cur.coff, fc.weight.data_ptr(),
             fc.bias.data_ptr(), logits.data_ptr(), s)
        return logits

    # ------------------------------------------------------------------ backward
    def _galloc(self, name, shape, zero=False):
        """Gradient tensor for parameter `name`: from the caller's allocator (flat all-reduce buckets
        in data-parallel training) or a fresh tensor."""
        if self.grad_alloc is not None:
            t = self.grad_alloc(name, shape)
            if zero and not self.zero_grads_kept:
                t.zero_()
            return t
        f = torch.zeros if zero else torch.empty
        return f(shape, dtype=torch.float32, device=self.device)

    def _ready(self, *names):
        if self.on_ready is not None:
            self.on_ready(names)

    def _cbr_backward(self, name, lvl, da, grads, want_dgrad=True, stats_for=None):
        """Backward of conv3x3+BN+ReLU.  `da` = gradient w.r.t. the layer's activated output.
        Returns the gradient w.r.t. the layer's input activation (an Act with a_in's channel count)."""
        with layer_range("bwd/" + name):
            return self._cbr_backward_launch(name, lvl, da, grads, want_dgrad, stats_for)

    def _cbr_backward_launch(self, name, lvl, da, grads, want_dgrad, stats_for):
        rec = self.recs[name]
        conv = getattr(self.m, name)
        L, s = self.lease, _stream()
        B, H, W = self.geo[lvl]
        C, cin = rec.cout, rec.cin
        y = rec.y
        Mp = y.Mp
        bnname = name.replace("conv", "bn", 1)
        fused = self._fused_stats.pop(name, None)
        dgamma = self._galloc(bnname + ".weight", (C,))
        dbeta = self._galloc(bnname + ".bias", (C,))
        coef = L.f32(2 * C)
        if fused is None and fused_finalize():
            # the sums and, in the launch's last workgroups, their finalisation: no second launch
            parts = _lib.lib.uh_bn_bwd_parts(Mp, C)
            nbytes = _lib.lib.uh_fin_ws_bytes(parts, C, 2)
            ws = L.f32((nbytes + 3) // 4)
            call("uh_bn_relu_bwd_reduce_fin", Mp, C, da.ptr, da.cstride, da.coff, y.ptr, y.cstride, y.coff,
                 rec.mean.data_ptr(), rec.invstd.data_ptr(), rec.scale.data_ptr(), rec.shift.data_ptr(), C, B * H * W,
                 dgamma.data_ptr(), dbeta.data_ptr(), coef.data_ptr(), ws.data_ptr(), nbytes,
                 self.pool.counters(s).data_ptr(), s)
        else:
            if fused is not None:                  # the kernel that produced da left the sums behind
                partial, parts = fused
            else:
                parts = _lib.lib.uh_bn_bwd_parts(Mp, C)
                partial = L.f32(parts * 2 * C)
                call("uh_bn_relu_bwd_reduce", Mp, C, da.ptr, da.cstride, da.coff, y.ptr, y.cstride, y.coff,
                     rec.mean.data_ptr(), rec.invstd.data_ptr(), rec.scale.data_ptr(), rec.shift.data_ptr(),
                     partial.data_ptr(), s)
            call("uh_bn_bwd_finalize", partial.data_ptr(), parts, C, C, B * H * W, dgamma.data_ptr(),
                 dbeta.data_ptr(), coef.data_ptr(), 0, s)
        grads[bnname + ".weight"], grads[bnname + ".bias"] = dgamma, dbeta
        dy = L.act(B, H, W, C)
        call("uh_bn_relu_bwd_apply", Mp, C, da.ptr, da.cstride, da.coff, y.ptr, y.cstride, y.coff,
             rec.mean.data_ptr(), rec.invstd.data_ptr(), rec.scale.data_ptr(), rec.shift.data_ptr(),
             coef.data_ptr(), self.masks[lvl].data_ptr(), dy.ptr, dy.cstride, dy.coff, s)
        # the bias of a conv that feeds train-mode BatchNorm has an exactly zero gradient:
        # BN subtracts the batch mean, so sum_m dy[m][c] == 0 for every channel.
        grads[name + ".bias"] = self._galloc(name + ".bias", (C,), zero=True)
        dw = self._galloc(name + ".weight", tuple(conv.weight.shape))
        done = (bnname + ".weight", bnname + ".bias", name + ".weight", name + ".bias")
        if rec.a_in is None:
            parts = _lib.lib.uh_conv_first_wgrad_parts(B, H, W)
            wpart = L.f32(parts * C * cin * 9)
            call("uh_conv_first_wgrad", B, H, W, cin, C, self.x.data_ptr(), dy.ptr, dy.cstride, dy.coff,
                 wpart.data_ptr(), s)
            call("uh_wgrad_reduce", wpart.data_ptr(), parts, C, 1, cin * 9, cin * 9, 0, cin * 9, 0, 1,
                 dw.data_ptr(), 0, s)
            grads[name + ".weight"] = dw
            self._ready(*done)
            return None
        a_in = rec.a_in
        S = _lib.lib.uh_conv_wgrad_splits(B, H, W, C, cin, 9)
        wpart = L.f32(S * C * 9 * cin)
        # the weight gradient and the input gradient both read dy and nothing else of each other: with a second stream
        # the weight gradient's workgroups fill CUs that the other kernels leave idle.  The input gradient, which the
        # rest of backward waits for, is enqueued first so that its workgroups get their CUs first.
        side = self._side_stream()
        first = side is not None and want_dgrad
        dx = None

        def dgrad():
            # dx[m][ci] = sum_t sum_co dy[m + off(t)][co] * W[co][ci][8-t]
            wd = self.wb[name]
            out = L.act(B, H, W, cin)
            # Beside a weight gradient on the second stream, whole tiles per workgroup beat even cuts: the kernel that
            # finishes the cut tiles waits for CUs the weight gradient holds (85 instead of 16 us), and workgroups that
            # start late undo the even split anyway (step 14.31 -> 14.12 ms).  Alone on the chip even cuts win, but
            # the choice does not follow the stream setting: both settings then give bit-identical results.
            prev = self.recs.get(stats_for) if stats_for else None
            if prev is not None and fuse_bn_bwd(lvl) and prev.cout == cin:
                # `out` is the gradient at the output of layer `stats_for` (conv + BatchNorm + ReLU): its
                # BatchNorm-backward sums are taken while the tiles are still in the accumulators
                parts = _lib.lib.uh_conv_gemm_mtiles(B, H, W, cin)
                partial = L.f32(parts * 2 * cin)
                args = (B, H, W, dy.ptr, dy.cstride, dy.coff, C, wd.data_ptr(), cin, out.ptr, out.cstride, out.coff,
                        self.masks[lvl].data_ptr(), prev.y.ptr, prev.y.cstride, prev.y.coff, prev.mean.data_ptr(),
                        prev.invstd.data_ptr(), prev.scale.data_ptr(), prev.shift.data_ptr(), partial.data_ptr(), s)
                if self.timer is None:
                    call("uh_conv_gemm_bnbwd", *args)
                else:
                    with self.timer.launch("conv_band", 2.0 * B * H * W * C * cin * 9):
                        call("uh_conv_gemm_bnbwd", *args)
                self._fused_stats[stats_for] = (partial, parts)
                return out
            self._gemm(B, H, W, dy.ptr, dy.cstride, dy.coff, C, wd.data_ptr(), cin, 9, EPI_MASKED,
                       out.ptr, out.cstride, out.coff, None, None, None, None, self.masks[lvl].data_ptr(), s,
                       whole_tiles=str(lvl) not in dgrad_cut_levels())
            return out
        if first:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())         # dy is complete here
            dx = dgrad()                                   # enqueued first: takes the CUs first
            side.wait_event(ev)
        elif side is not None:
            side.wait_stream(torch.cuda.current_stream())
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            sw = side.cuda_stream if side is not None else s
            self._wgrad(B, H, W, dy.ptr, dy.cstride, dy.coff, C, a_in.ptr, a_in.cstride, a_in.coff, cin, 9,
                        wpart.data_ptr(), sw)
            # partial [n][t][k] -> OIHW: grad[n*cin*9 + k*9 + t]
            call("uh_wgrad_reduce", wpart.data_ptr(), S, C, 9, cin, 9 * cin, cin, cin * 9, 1, 9, dw.data_ptr(), 0, sw)
        grads[name + ".weight"] = dw
        self._ready(*done)
        if want_dgrad and dx is None:
            dx = dgrad()
        return dx

    def backward(self, dlogits, grad_alloc=None, on_ready=None):
        """Returns {parameter name: fp32 gradient} for every parameter of the module.

        grad_alloc(name, shape) -> tensor lets the caller place gradients (e.g. inside flat all-reduce
        buckets); on_ready(names) is called as soon as the kernels producing those gradients have been
        enqueued on the current stream, in backward order (head first, conv1_1 last)."""
        self.grad_alloc, self.on_ready = grad_alloc, on_ready
        if not self.training:
            raise RuntimeError("backward through an eval-mode (running-statistics) forward is not built; "
                               "call model.train() before a forward pass that needs gradients")
        m, L, s = self.m, self.lease, _stream()
        self.join_side_stream()                            # the repack issued beside the forward pass
        self.wb = self.packer.get("bwd", s)                # rotated / transposed operands of the input gradients
        grads = {}
        dl = dlogits.contiguous().float()
        oc = m.out_channels
        B, H, W = self.geo[0]
        a = self.head_in
        fc = m.final_conv
        da = L.act(B, H, W, 64)
        parts = _lib.lib.uh_head_bwd_parts(B, H, W, 64)
        hp = L.f32(parts * oc * 65)
        if fuse_pool_stats():
            rec = self.recs["conv_up4_2"]
            bp = L.f32(parts * 2 * 64)
            call("uh_head_bwd_bnstat", B, H, W, 64, oc, a.ptr, a.cstride, a.coff, fc.weight.data_ptr(), dl.data_ptr(),
                 da.ptr, da.cstride, da.coff, hp.data_ptr(), rec.y.ptr, rec.y.cstride, rec.y.coff,
                 rec.mean.data_ptr(), rec.invstd.data_ptr(), rec.scale.data_ptr(), rec.shift.data_ptr(),
                 bp.data_ptr(), s)
            self._fused_stats["conv_up4_2"] = (bp, parts)
        else:
            call("uh_head_bwd", B, H, W, 64, oc, a.ptr, a.cstride, a.coff, fc.weight.data_ptr(), dl.data_ptr(),
                 da.ptr, da.cstride, da.coff, hp.data_ptr(), s)
        hw = L.f32(oc * 65)
        call("uh_colsum_finalize", hp.data_ptr(), parts, oc * 65, 1, hw.data_ptr(), 0, s)
        hw = hw.view(oc, 65)
        gw, gb = self._galloc("final_conv.weight", (oc, 64, 1, 1)), self._galloc("final_conv.bias", (oc,))
        gw.copy_(hw[:, :64].reshape(oc, 64, 1, 1))
        gb.copy_(hw[:, 64])
        grads["final_conv.weight"], grads["final_conv.bias"] = gw, gb
        self._ready("final_conv.weight", "final_conv.bias")

        skip_grads = [None] * 4
        for k in range(4, 0, -1):                            # decoder, last stage first
            l = 4 - k
            w = WIDTHS[l]
            B, H, W = self.geo[l]
            da = self._cbr_backward(f"conv_up{k}_2", l, da, grads, stats_for=f"conv_up{k}_1")
            dcat = self._cbr_backward(f"conv_up{k}_1", l, da, grads)     # [Mp][2w]
            skip_grads[l] = dcat.window(0, w)
            dup = dcat.window(w, w)
            up = getattr(m, f"upconv{k}")
            # bias gradient of the transposed conv: column sums of its output gradient
            parts = _lib.lib.uh_colsum_parts(dup.Mp, w)
            cp = L.f32(parts * w)
            db = self._galloc(f"upconv{k}.bias", (w,))
            side = self._side_stream()
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                sw = side.cuda_stream if side is not None else s
                call("uh_colsum", dup.Mp, w, dup.ptr, dup.cstride, dup.coff, cp.data_ptr(), sw)
                call("uh_colsum_finalize", cp.data_ptr(), parts, w, 1, db.data_ptr(), 0, sw)
            grads[f"upconv{k}.bias"] = db
            # gather the 2x2 output positions of every input position: d4[m][t*w + co]
            Bs, Hs, Ws = self.geo[l + 1]
            d4 = L.act(Bs, Hs, Ws, 4 * w)
            call("uh_space_to_depth2", Bs, Hs, Ws, w, dup.ptr, dup.cstride, dup.coff, d4.ptr, d4.cstride, d4.coff, s)
            a_in = self.acts[f"upconv{k}_in"]
            cin = 2 * w
            S = _lib.lib.uh_conv_wgrad_splits(Bs, Hs, Ws, 4 * w, cin, 1)
            wpart = L.f32(S * 4 * w * cin)
            dw = self._galloc(f"upconv{k}.weight", tuple(up.weight.shape))     # [cin][w][2][2]
            wd = self.wb[f"upconv{k}"]
            da = L.act(Bs, Hs, Ws, cin)

            def dgrad():
                # input gradient: dx[m][ci] = sum_{t,co} d4[m][t*w+co] * W[ci][co][t]
                nxt = f"conv_up{k - 1}_2" if k > 1 else "conv5_2"          # the layer whose output gradient this is
                prev = self.recs[nxt]
                if fuse_t1_stats() and prev.cout == cin:
                    parts = _lib.lib.uh_conv_gemm_mtiles(Bs, Hs, Ws, cin)
                    partial = L.f32(parts * 2 * cin)
                    args = (Bs, Hs, Ws, d4.ptr, d4.cstride, d4.coff, 4 * w, wd.data_ptr(), cin, da.ptr, da.cstride,
                            da.coff, self.masks[l + 1].data_ptr(), prev.y.ptr, prev.y.cstride, prev.y.coff,
                            prev.mean.data_ptr(), prev.invstd.data_ptr(), prev.scale.data_ptr(),
                            prev.shift.data_ptr(), partial.data_ptr(), s)
                    if self.timer is None:
                        call("uh_conv_gemm1_bnbwd", *args)
                    else:
                        with self.timer.launch("conv_gemm_t1", 2.0 * Bs * Hs * Ws * 4 * w * cin):
                            call("uh_conv_gemm1_bnbwd", *args)
                    self._fused_stats[nxt] = (partial, parts)
                    return
                self._gemm(Bs, Hs, Ws, d4.ptr, d4.cstride, d4.coff, 4 * w, wd.data_ptr(), cin, 1, EPI_MASKED,
                           da.ptr, da.cstride, da.coff, None, None, None, None, self.masks[l + 1].data_ptr(), s)
            side = self._side_stream()
            if side is not None:                             # as in _cbr_backward: the input gradient goes first
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                dgrad()
                side.wait_event(ev)
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                sw = side.cuda_stream if side is not None else s
                self._wgrad(Bs, Hs, Ws, d4.ptr, d4.cstride, d4.coff, 4 * w, a_in.ptr, a_in.cstride, a_in.coff,
                            cin, 1, wpart.data_ptr(), sw)
                call("uh_wgrad_reduce", wpart.data_ptr(), S, w, 4, cin, cin, w * cin, 4, 1, w * 4, dw.data_ptr(), 0, sw)
            grads[f"upconv{k}.weight"] = dw
            self._ready(f"upconv{k}.weight", f"upconv{k}.bias")
            if side is None:
                dgrad()
        for l in range(4, -1, -1):                           # bridge, then encoder levels 3..0
            w = WIDTHS[l]
            B, H, W = self.geo[l]
            if l < 4:
                a2 = self.acts[f"conv{l + 1}_2"]
                sg = skip_grads[l]
                dsum = L.act(B, H, W, w)
                if fuse_pool_stats():
                    rec = self.recs[f"conv{l + 1}_2"]
                    bparts = _lib.lib.uh_maxpool2_bwd_bnstat_parts(B, H, W, w)
                    bp = L.f32(bparts * 2 * w)
                    call("uh_maxpool2_bwd_bnstat", B, H, W, w, a2.ptr, a2.cstride, a2.coff, da.ptr, da.cstride,
                         da.coff, sg.ptr, sg.cstride, sg.coff, dsum.ptr, dsum.cstride, dsum.coff, rec.y.ptr,
                         rec.y.cstride, rec.y.coff, rec.mean.data_ptr(), rec.invstd.data_ptr(), rec.scale.data_ptr(),
                         rec.shift.data_ptr(), bp.data_ptr(), s)
                    self._fused_stats[f"conv{l + 1}_2"] = (bp, bparts)
                else:
                    call("uh_maxpool2_bwd", B, H, W, w, a2.ptr, a2.cstride, a2.coff, da.ptr, da.cstride, da.coff,
                         sg.ptr, sg.cstride, sg.coff, dsum.ptr, dsum.cstride, dsum.coff, s)
                da = dsum
            da = self._cbr_backward(f"conv{l + 1}_2", l, da, grads, stats_for=f"conv{l + 1}_1")
            da = self._cbr_backward(f"conv{l + 1}_1", l, da, grads, want_dgrad=(l > 0))
        self.join_side_stream()
        return grads

    def release(self):
        self.recs = {}
        self.acts = {}
        self.lease.release()
