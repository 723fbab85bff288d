"""Single-node data-parallel training step for the HIP U-Net: one process per GPU, batches sharded
across ranks, gradients all-reduced in a few large flat buckets on RCCL while backward is still
running.

The reference trains on one device (trainer/training_utils.py:32,125-183: Adam, BCEWithLogits or
clDice loss, per-step loss.item()); this is its multi-GPU counterpart.  Semantics are those of
DistributedDataParallel: every rank runs the full model on its shard with per-replica BatchNorm
statistics, the loss is the mean of the per-shard losses, gradients are averaged.

Overlap: the backward pass produces parameter gradients last-layer-first, straight into flat fp32
buckets laid out in that order.  As soon as the kernels that fill a bucket have been enqueued, its
all-reduce is issued asynchronously; RCCL runs it on its own stream behind an event, so it overlaps
with the remaining backward kernels.  xGMI is point-to-point, so buckets are few and large
(>= 25 MB): per-link bandwidth, not latency, should set the cost.
"""
import contextlib
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib
from .engine import UNetRun, side_stream_for
from .unet import _Features, FEATURE_LAYERS


class KernelTimer:
    """Times selected kernel launches with events recorded on the launching stream."""

    def __init__(self):
        self.records = {}      # tag -> list of (flops, start_event, end_event)

    @contextlib.contextmanager
    def launch(self, tag, flops):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        yield
        e1.record()
        self.records.setdefault(tag, []).append((flops, e0, e1))

    def summary(self):
        """tag -> dict(launches, flops, ms); call after torch.cuda.synchronize()."""
        out = {}
        for tag, recs in self.records.items():
            ms = sum(e0.elapsed_time(e1) for _, e0, e1 in recs)
            out[tag] = dict(launches=len(recs), flops=sum(f for f, _, _ in recs), ms=ms)
        return out


def backward_param_order(model):
    """Parameter names in the order the backward pass produces their gradients."""
    order = ["final_conv.weight", "final_conv.bias"]

    def layer(conv):
        bn = conv.replace("conv", "bn", 1)
        return [bn + ".weight", bn + ".bias", conv + ".weight", conv + ".bias"]

    for k in (4, 3, 2, 1):
        order += layer(f"conv_up{k}_2") + layer(f"conv_up{k}_1") + [f"upconv{k}.weight", f"upconv{k}.bias"]
    for l in (5, 4, 3, 2, 1):
        order += layer(f"conv{l}_2") + layer(f"conv{l}_1")
    names = {n for n, _ in model.named_parameters()}
    assert set(order) == names, (set(order) ^ names)
    return order


class GradBuckets:
    """Flat fp32 gradient storage in backward order, cut into buckets of >= bucket_bytes."""

    def __init__(self, model, bucket_bytes=25 << 20):
        params = dict(model.named_parameters())
        order = backward_param_order(model)
        dev = next(model.parameters()).device
        total = sum(params[n].numel() for n in order)
        self.total = total
        self.order = order
        self.flat = torch.zeros((total + 3) // 4 * 4, dtype=torch.float32, device=dev)   # padded for 16-byte vectors
        self.views, self.bucket_of, self.buckets = {}, {}, []
        off = start = 0
        names = []
        for n in order:
            p = params[n]
            self.views[n] = self.flat[off:off + p.numel()].view(p.shape)
            names.append(n)
            off += p.numel()
            if (off - start) * 4 >= bucket_bytes:
                self._close(start, off, names)
                start, names = off, []
        if names:
            self._close(start, off, names)
        for n, p in params.items():
            p.grad = self.views[n]          # the optimizer reads gradients straight from the buckets

    def _close(self, start, end, names):
        idx = len(self.buckets)
        self.buckets.append(dict(tensor=self.flat[start:end], names=list(names), pending=set(names), start=start, end=end))
        for n in names:
            self.bucket_of[n] = idx

    def reset(self):
        for b in self.buckets:
            b["pending"] = set(b["names"])
            b.pop("launched", None)
            b.pop("adam", None)


class FlatGrads:
    """All parameter gradients of a model in ONE flat fp32 buffer, in a given order of parameter names: `views[name]`
    is the gradient tensor of a parameter (also installed as its .grad), `span(name, n)` the n floats that start where
    that parameter's gradient starts.  What FlatAdam needs of GradBuckets (`flat`, `order`), without the buckets."""

    def __init__(self, model, order=None, aligned=None):
        """aligned(name) -> True for parameters that must start on a 16-byte boundary (kernels that read them with
        vector loads): up to three floats of padding go in front of those; all others follow their predecessor
        directly."""
        params = dict(model.named_parameters())
        order = list(order) if order is not None else list(params)
        assert sorted(order) == sorted(params), "order must name every parameter once"
        dev = next(model.parameters()).device
        self.order = order
        self.offset = {}
        off = 0
        for n in order:
            if aligned is not None and aligned(n):
                off = (off + 3) // 4 * 4
            self.offset[n] = off
            off += params[n].numel()
        self.total = off
        self.flat = torch.zeros((self.total + 3) // 4 * 4, dtype=torch.float32, device=dev)
        self.views = {}
        for n in order:
            p = params[n]
            self.views[n] = self.flat[self.offset[n]:self.offset[n] + p.numel()].view(p.shape)
            p.grad = self.views[n]

    def span(self, name, n):
        off = self.offset[name]
        if off + n > self.total:
            raise ValueError(f"{n} floats from {name} run past the end of the gradient buffer")
        return self.flat[off:off + n]


class FlatAdam:
    """torch.optim.Adam(model.parameters(), lr) (training_utils.py:107) as one kernel launch: the parameters are
    re-pointed into one flat fp32 buffer laid out like the gradient buckets, so p, g, m, v are four flat arrays.

    Under torch.amp.GradScaler (training_utils.py:117,153-155) `scaler.step(optimizer)` works as it does with torch's
    own fused optimizers: the scaler sees ONE parameter (the flat buffer, whose .grad is the flat gradient), so its
    inf check is one launch, and because of _step_supports_amp_scaling it leaves unscaling and the skipped step to
    step(), which folds both into the Adam kernel (uh_adam_step_amp): no unscale pass over the gradients."""

    _step_supports_amp_scaling = True

    def __init__(self, model, buckets, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.g = buckets.flat
        self.p = torch.zeros_like(self.g)
        self.m = torch.zeros_like(self.g)
        self.v = torch.zeros_like(self.g)
        self.steps = 0
        self._steps_dev = None             # device-side count of the steps actually taken (AMP: skipped ones excluded)
        params = dict(model.named_parameters())
        offsets = getattr(buckets, "offset", None)     # FlatGrads may pad in front of some parameters
        off = 0
        with torch.no_grad():
            for n in buckets.order:
                q = params[n]
                if offsets is not None:
                    off = offsets[n]
                view = self.p[off:off + q.numel()].view(q.shape)
                view.copy_(q.data)
                q.data = view                      # the module's parameter now lives in the flat buffer
                off += q.numel()
        flat = nn.Parameter(self.p, requires_grad=False)
        flat.grad = self.g
        self.param_groups = [{"params": [flat], "lr": lr}]     # what GradScaler walks for its inf check

    def step(self):
        """One Adam step over the whole buffer.  When a GradScaler has left `grad_scale` / `found_inf` on this object
        (it does so around the call, for optimizers with _step_supports_amp_scaling), the gradient is unscaled inside
        the kernel and the step does nothing if an inf / NaN was found."""
        grad_scale, found_inf = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)
        stream = torch.cuda.current_stream(self.p.device).cuda_stream
        if grad_scale is None and found_inf is None:
            self.begin_step()
            self.step_range(0, self.p.numel(), stream)
        else:
            self.step_amp(grad_scale, found_inf, stream)
        self.finish_step()

    def step_amp(self, grad_scale, found_inf, stream):
        """grad_scale, found_inf: fp32 scalars on the device (either may be None)"""
        if self._steps_dev is None:
            self._steps_dev = torch.full((1,), float(self.steps), dtype=torch.float32, device=self.p.device)
        if found_inf is not None:
            self._steps_dev.add_(1.0 - found_inf.reshape(1).to(self._steps_dev))     # a skipped step is not counted
        else:
            self._steps_dev.add_(1.0)
        self.steps += 1                            # host-side count: upper bound, only used without a device count
        _lib.call("uh_adam_step_amp", self.p.numel(), self.p.data_ptr(), self.g.data_ptr(), self.m.data_ptr(),
                  self.v.data_ptr(), self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.steps,
                  None if grad_scale is None else grad_scale.data_ptr(),
                  None if found_inf is None else found_inf.data_ptr(), self._steps_dev.data_ptr(), stream)

    # the same step in pieces (a bucket at a time, as its gradients arrive): begin_step, step_range..., finish_step
    def begin_step(self):
        self.steps += 1
        if self._steps_dev is not None:
            self._steps_dev.add_(1.0)

    def step_range(self, lo, hi, stream):
        """elements [lo, hi) of the flat buffers; lo and hi multiples of 4"""
        if hi > lo:
            if self._steps_dev is not None:        # steps have been skipped before: the device count is the true one
                _lib.call("uh_adam_step_amp", hi - lo, self.p.data_ptr() + 4 * lo, self.g.data_ptr() + 4 * lo,
                          self.m.data_ptr() + 4 * lo, self.v.data_ptr() + 4 * lo, self.lr, self.betas[0],
                          self.betas[1], self.eps, self.weight_decay, self.steps, None, None,
                          self._steps_dev.data_ptr(), stream)
                return
            _lib.call("uh_adam_step", hi - lo, self.p.data_ptr() + 4 * lo, self.g.data_ptr() + 4 * lo,
                      self.m.data_ptr() + 4 * lo, self.v.data_ptr() + 4 * lo, self.lr, self.betas[0], self.betas[1],
                      self.eps, self.weight_decay, self.steps, stream)

    def finish_step(self):
        # the kernel wrote through raw pointers: autograd's version counters did not move, and the packed bf16
        # operands key on them
        pk = getattr(self.model, "_packer", None)
        if pk is not None:
            pk.mark_dirty()

    def zero_grad(self, set_to_none=False):
        pass                                       # every gradient is overwritten by the next backward pass


def bce_with_logits(logits, targets, grad_scale=1.0):
    """(nn.BCEWithLogitsLoss()(logits, targets), grad_scale * its gradient with respect to the logits) from one
    kernel pass (uh_bce_logits); fp32 CUDA tensors of equal shape."""
    x = logits.contiguous().float()
    t = targets.contiguous().float()
    n = x.numel()
    dev = x.device
    ws_bytes = _lib.lib.uh_bce_logits_workspace(n)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dx = torch.empty_like(x)
    _lib.call("uh_bce_logits", n, x.data_ptr(), t.data_ptr(), float(grad_scale), loss.data_ptr(), dx.data_ptr(),
              ws.data_ptr(), ws_bytes, torch.cuda.current_stream().cuda_stream)
    return loss.view(()), dx


class DataParallelTrainer:
    """fwd + loss + bwd (+ bucketed all-reduce) + Adam for one shard per rank."""

    def __init__(self, model, lr=1e-3, criterion=None, bucket_bytes=25 << 20, process_group=None, standalone=False):
        """standalone: train alone even inside an initialised process group (independent replicas, e.g. one
        cross-validation fold per GPU)."""
        self.model = model
        joined = dist.is_initialized() and not standalone
        self.world = dist.get_world_size(process_group) if joined else 1
        self.rank = dist.get_rank(process_group) if joined else 0
        self.group = process_group
        self.criterion = criterion if criterion is not None else nn.BCEWithLogitsLoss()
        c = self.criterion
        # the trainer's default loss (training_utils.py:105) has a one-pass kernel for value and gradient
        self._fused_bce = (type(c) is nn.BCEWithLogitsLoss and c.reduction == "mean" and c.weight is None
                           and c.pos_weight is None)
        self.buckets = GradBuckets(model, bucket_bytes)
        self.optimizer = FlatAdam(model, self.buckets, lr=lr)                      # training_utils.py:107
        self.timer = None
        self._run = None
        self._zeroed = False
        self._eager, self._adam_begun = True, False
        # UNET_AMD_FORCE_COLLECTIVES=1 runs the broadcasts and all-reduces with a single rank too (a rehearsal
        # of the RCCL calls on a one-GPU machine)
        self.collectives = joined and (self.world > 1 or os.environ.get("UNET_AMD_FORCE_COLLECTIVES") == "1")
        # the second stream takes its hardware queue before the process group's streams take theirs
        dev = next(model.parameters()).device
        self._opt_stream = None
        if dev.type == "cuda":
            side_stream_for(dev)
            if self.collectives:
                self._opt_stream = side_stream_for(dev, "opt")
        if self.collectives:
            # same initial weights on every rank
            for p in model.parameters():
                dist.broadcast(p.data, src=0, group=process_group)
            for b in model.buffers():
                dist.broadcast(b.data, src=0, group=process_group)

    @staticmethod
    def shard(n_items, epoch, rank, world, seed=0):
        """indices of this rank's shard of a shuffled epoch (same permutation on every rank).  Every rank gets the same
        number of indices: when n_items is not a multiple of world the permutation is extended with its own first
        entries (as torch's DistributedSampler does), so that all ranks run the same number of steps with equal batch
        sizes and every collective of a step has all its participants."""
        g = torch.Generator().manual_seed(seed + epoch)
        perm = torch.randperm(n_items, generator=g)
        pad = (-n_items) % world
        if pad and n_items > 0:
            perm = torch.cat([perm, perm[torch.arange(pad) % n_items]])
        return perm[rank::world]

    def _on_ready(self, names):
        for n in names:
            i = self.buckets.bucket_of[n]
            b = self.buckets.buckets[i]
            b["pending"].discard(n)
            if b["pending"] or b.get("launched"):
                continue
            b["launched"] = True
            if self.collectives:
                # The bucket's gradients were written on this stream (BatchNorm, biases) and on the engine's second
                # stream (weight gradients).  The collective is issued from the second stream once that has caught up
                # with this one, so it is ordered after both and this stream never waits for a weight gradient.
                side = self._run._side_stream() if self._run is not None else None
                if side is not None:
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        b["work"] = dist.all_reduce(b["tensor"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                else:
                    b["work"] = dist.all_reduce(b["tensor"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._adam_for_bucket(i, b)

    def _adam_for_bucket(self, i, b):
        """Adam for a bucket as soon as its (averaged) gradients exist, beside the rest of the backward pass.  Nothing
        that backward still has to run reads fp32 parameters whose gradients are complete: input gradients use the bf16
        operands packed before, BatchNorm backward its saved scale.

        Without collectives the update runs on the engine's second stream, behind the weight gradients that fill the
        bucket.  With collectives it runs on a third stream, which is the one that waits for the bucket's all-reduce:
        the second stream carries every later weight gradient and must never sit behind a collective."""
        side = self._run._side_stream() if (self._eager and self._run is not None) else None
        if side is None:
            return
        last = i == len(self.buckets.buckets) - 1
        lo = b["start"] & ~3                           # bucket boundaries are parameter boundaries: round both ends
        hi = self.optimizer.p.numel() if last else (b["end"] & ~3)   # down; the 0-3 elements cut off go with the next
        w = b.get("work")
        if w is not None:
            st = self._opt_stream                      # ordered behind the collective only (which is behind both
        else:                                          # streams' writes: see _on_ready)
            st = side
            side.wait_stream(torch.cuda.current_stream())  # gradients written on this stream (BatchNorm, biases)
        with torch.cuda.stream(st):
            if w is not None:
                w.wait()                               # the third stream waits for the collective
            if not self._adam_begun:
                self.optimizer.begin_step()
                self._adam_begun = True
            self.optimizer.step_range(lo, hi, st.cuda_stream)
        b["adam"] = True

    def train_step(self, images, masks, step_optimizer=True, scaler=None):
        """images [b,C,H,W] fp32, masks [b,1,H,W] on this rank's device.  Returns the local loss (0-dim tensor).

        scaler: a torch.amp.GradScaler, used as the reference's loop uses it (training_utils.py:153-155:
        scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()): the backward pass is seeded with the
        scaled loss gradient, the optimizer steps once after the last bucket (a step that finds an inf must be skipped
        as a whole, so no bucket may be updated early), unscaling happens inside the Adam kernel."""
        model = self.model
        model.train()
        self.buckets.reset()
        amp = scaler is not None and scaler.is_enabled()
        self._eager, self._adam_begun = step_optimizer and not amp, False
        run = self._run = UNetRun(model, images.float().contiguous(), training=True)
        run.timer = self.timer
        run.zero_grads_kept = self._zeroed          # bucket memory: zero since the first step, summed with zeros since
        self._zeroed = True
        logits = run.forward().requires_grad_(True)
        # what model(images) would leave behind (a criterion may read it: models/losses.py via training_utils.py:143)
        model.intermediate_features = _Features([run.acts[n] for n in FEATURE_LAYERS], run.lease.pool)
        # d(mean over ranks of loss)/dlogits: the 1/world of gradient averaging is folded into the seed
        if self._fused_bce and masks.shape == logits.shape:
            loss, dlogits = bce_with_logits(logits.detach(), masks, 1.0 / self.world)
        else:
            loss = self.criterion(logits, masks)
            (dlogits,) = torch.autograd.grad(loss / self.world, logits)
        if amp:
            dlogits = scaler.scale(dlogits)          # d(scale * loss)/dlogits: the scale stays a device scalar
        run.backward(dlogits, grad_alloc=lambda n, shape: self.buckets.views[n], on_ready=self._on_ready)
        run.release()
        self._run = None
        for b in self.buckets.buckets:
            w = b.pop("work", None)
            if w is not None:
                w.wait()                     # current stream waits for the collective
        if self._opt_stream is not None:     # and for the bucket-wise optimizer updates behind the collectives
            torch.cuda.current_stream().wait_stream(self._opt_stream)
        if step_optimizer and amp:
            scaler.step(self.optimizer)          # inf check over the flat gradient, then FlatAdam.step with the scale
            scaler.update()
        elif step_optimizer:
            if self._adam_begun:                 # done bucket by bucket beside backward (joined at its end)
                missing = [b for b in self.buckets.buckets if not b.get("adam")]
                assert not missing, "a bucket never became ready"
                self.optimizer.finish_step()
            else:
                self.optimizer.step()
        return loss.detach()


class GradAverager:
    """Gradient averaging for a model whose backward pass hands over all gradients at once (LUNeXt, whose custom
    autograd function returns them together): after loss.backward(), copy every gradient into a few large flat
    fp32 buckets with a fixed layout, all-reduce the buckets, copy the means back.  Parameters without a gradient
    (LUNeXt's four unused att*.output_conv layers, lunext.py:74) occupy zeros in the buckets, so the layout never
    depends on which gradients exist, and they are left without a gradient afterwards."""

    def __init__(self, params, bucket_bytes=25 << 20, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        dev = self.params[0].device
        self.buckets = []                       # (flat tensor, [(param, view)])
        cur, size = [], 0
        for p in self.params:
            cur.append(p)
            size += p.numel()
            if size * 4 >= bucket_bytes:
                self._close(cur, size, dev)
                cur, size = [], 0
        if cur:
            self._close(cur, size, dev)

    def _close(self, plist, size, dev):
        flat = torch.zeros(size, dtype=torch.float32, device=dev)
        views, off = [], 0
        for p in plist:
            views.append((p, flat[off:off + p.numel()].view(p.shape)))
            off += p.numel()
        self.buckets.append((flat, views))

    def broadcast_from_rank0(self, module):
        if self.world > 1:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0, group=self.group)

    def average(self):
        """In place: every existing p.grad becomes the mean over ranks."""
        if self.world == 1:
            return
        works = []
        for flat, views in self.buckets:
            have = [(p, v) for p, v in views if p.grad is not None]
            if len(have) != len(views):
                flat.zero_()
            if have:
                torch._foreach_copy_([v for _, v in have], [p.grad for p, _ in have])
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for (flat, views), w in zip(self.buckets, works):
            w.wait()
            flat.mul_(1.0 / self.world)
            have = [(p, v) for p, v in views if p.grad is not None]
            if have:
                torch._foreach_copy_([p.grad for p, _ in have], [v for _, v in have])


class LUNeXtTrainer:
    """fwd + loss + bwd + Adam for LUNeXt (the loop body of trainer/training_utils.py:131-170 with model_type
    "LUNeXt") without autograd in the way: the backward pass writes every parameter gradient into one flat buffer
    (FlatGrads), the optimizer is one uh_adam_step launch over it (FlatAdam) instead of torch's multi-tensor sweeps,
    and with custom_loss "cl_dice" the glue around the loss (sigmoid, clamp and their backward) is two launches
    (cl_dice.cldice_with_logits).  Inside an initialised process group (unless standalone) the flat gradient is
    all-reduced once per step (14 MB: one collective) and 1/world is folded into the loss gradient.

    criterion: "cl_dice", "bce" or a callable (logits, masks) -> loss, differentiated by autograd at the logits."""

    def __init__(self, model, lr=1e-3, criterion="bce", process_group=None, standalone=False):
        from .lunext import flat_param_order
        self.model = model
        joined = dist.is_initialized() and not standalone
        self.world = dist.get_world_size(process_group) if joined else 1
        self.rank = dist.get_rank(process_group) if joined else 0
        self.group = process_group
        self.criterion = criterion
        if joined and self.world > 1:
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=0, group=process_group)
        # the depthwise kernels read their [C][3][3] filters with 16-byte loads
        self.grads = FlatGrads(model, flat_param_order(model), aligned=lambda n: n.endswith("depthwise.weight"))
        self.optimizer = FlatAdam(model, self.grads, lr=lr)

    def _loss(self, logits, masks, prepared=None):
        scale = 1.0 / self.world
        c = self.criterion
        if c == "cl_dice":
            from .cl_dice import cldice_with_logits
            return cldice_with_logits(masks, logits, grad_scale=scale, prepared=prepared)
        if c == "bce" and masks.shape == logits.shape:
            return bce_with_logits(logits, masks, scale)
        fn = nn.BCEWithLogitsLoss() if c == "bce" else c
        lg = logits.detach().requires_grad_(True)
        loss = fn(lg, masks)
        (dlogits,) = torch.autograd.grad(loss * scale, lg)
        return loss.detach(), dlogits

    def train_step(self, images, masks):
        """images [b,C,H,W], masks [b,1,H,W] on the model's device.  Returns the local loss (0-dim tensor)."""
        from .lunext import LUNeXtRun
        model = self.model
        model.train()
        prepared = None
        if self.criterion == "cl_dice" and os.environ.get("UNET_AMD_CLDICE_PREPARE", "1") != "0":
            # the labels' soft skeleton needs no prediction: on the second stream, beside the forward pass
            from .cl_dice import cldice_prepare
            two = os.environ.get("UNET_AMD_WGRAD_STREAM", "1") != "0"
            prepared = cldice_prepare(masks, stream=side_stream_for(masks.device) if two else None)
        run = LUNeXtRun(model, images.float().contiguous(), training=True)
        logits = run.forward()
        loss, dlogits = self._loss(logits, masks, prepared)
        run.backward(dlogits, gflat=self.grads)
        run.release()
        if self.world > 1:
            dist.all_reduce(self.grads.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.optimizer.step()
        return loss.detach()
