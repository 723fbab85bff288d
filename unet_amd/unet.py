"""Drop-in `UNet` module: the reference's constructor, parameters, buffers, state_dict keys and
`intermediate_features` attribute (models/unet.py:6-176), with forward and backward executed by
the gfx950 HIP kernels of libunet_hip.so.

The sub-modules (`nn.Conv2d`, `nn.BatchNorm2d`, `nn.ConvTranspose2d`) are kept only as parameter
and buffer containers, created in the reference's order so that the same `torch.manual_seed`
gives the same initial weights and `.pth` checkpoints interchange; their own `forward` is never
called.  There is no eager fallback: a CPU tensor raises.
"""
import torch
import torch.nn as nn

from .engine import UNetRun, WIDTHS

FEATURE_LAYERS = tuple(
    [f"conv{l}_{i}" for l in (1, 2, 3, 4) for i in (1, 2)]
    + [f"conv_up{k}_{i}" for k in (1, 2, 3) for i in (1, 2)]
    + ["conv_up4_2"])                      # the 15 tensors appended at unet.py:100-170


class _Features(list):
    """`intermediate_features`: materialised (NCHW fp32) on first access, from the bf16 NHWC buffers.

    The buffers belong to the forward pass that made the list and go back to the pool when it ends (after backward,
    or at once without gradients); they keep their contents until a later pass takes them again.  Reading the list
    after that raises instead of returning another pass's activations."""

    def __init__(self, acts=None, pool=None):
        super().__init__()
        self._acts, self._pool = acts, pool
        self._stamp = [pool.handed.get(id(a.buf), 0) for a in acts] if (pool is not None and acts) else None

    def acts(self):
        """The padded-linear bf16 views themselves, for kernels that read them in place; None once the list has
        been materialised."""
        if self._acts is not None and self._stamp is not None:
            now = [self._pool.handed.get(id(a.buf), 0) for a in self._acts]
            if now != self._stamp:
                raise RuntimeError("intermediate_features of an earlier forward pass: its activation buffers have "
                                   "been reused since (read them before the next forward/backward)")
        return self._acts

    def _fill(self):
        acts = self.acts()
        if acts is not None:
            self._acts = None
            super().extend(a.to_nchw() for a in acts)

    def __len__(self):
        self._fill()
        return super().__len__()

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def __getitem__(self, i):
        self._fill()
        return super().__getitem__(i)


class _UNetFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, module, x, *params):
        run = UNetRun(module, x, training=module.training)
        logits = run.forward()
        ctx.run = run
        ctx.names = [n for n, _ in module.named_parameters()]
        module._last_run = run
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
            out.append(grads[n] if ctx.needs_input_grad[2 + i] else None)
        return tuple(out)


class UNet(nn.Module):
    def __init__(self, in_channels=3, out_channels=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        cin = in_channels
        for lvl, w in enumerate(WIDTHS, start=1):                    # unet.py:19-45
            for i, ci in ((1, cin), (2, w)):
                setattr(self, f"conv{lvl}_{i}", nn.Conv2d(ci, w, kernel_size=3, padding=1))
                setattr(self, f"bn{lvl}_{i}", nn.BatchNorm2d(w))
            cin = w
        self.pool = nn.MaxPool2d(kernel_size=2, stride=2)            # unet.py:48 (no parameters)
        for k, w in enumerate((512, 256, 128, 64), start=1):         # unet.py:51-76
            setattr(self, f"upconv{k}", nn.ConvTranspose2d(2 * w, w, kernel_size=2, stride=2))
            for i, ci in ((1, 2 * w), (2, w)):
                setattr(self, f"conv_up{k}_{i}", nn.Conv2d(ci, w, kernel_size=3, padding=1))
                setattr(self, f"bn_up{k}_{i}", nn.BatchNorm2d(w))
        self.final_conv = nn.Conv2d(64, out_channels, kernel_size=1)  # unet.py:78
        self.relu = nn.ReLU(inplace=True)
        self.sigmoid = nn.Sigmoid()
        self.intermediate_features = []
        self._last_run = None
        self._packer = None               # packed bf16 operands (device buffers, not state)

    def __deepcopy__(self, memo):
        # copy.deepcopy(model) (k_fold.py:139): device buffers of the last pass are not state
        import copy
        run, feats, pk = self._last_run, self.intermediate_features, self._packer
        self._last_run, self.intermediate_features, self._packer = None, [], None
        try:
            cls = self.__class__
            new = cls.__new__(cls)
            memo[id(self)] = new
            for k, v in self.__dict__.items():
                setattr(new, k, copy.deepcopy(v, memo))
        finally:
            self._last_run, self.intermediate_features, self._packer = run, feats, pk
        return new

    def predict_counts(self, x, masks, counts, per_sample=False, return_logits=False):
        """Inference (no gradient) whose head launch also binarises its logits (sigmoid > 0.5) and adds tp, tn, fp, fn
        against `masks` to `counts` (int64 on the device: [4], or [B,4] with per_sample): what evaluate_model
        (k_fold.py:281-290) and save_results (test_helper/utils.py:124-131) compute per batch on the host.  Uses the
        module's current mode (call .eval() first, as those loops do).  Returns the logits if asked for."""
        if not x.is_cuda:
            raise RuntimeError("unet_amd.UNet runs on the HIP kernels only: move the model and the input to a "
                               "ROCm device (there is no CPU path)")
        B = x.shape[0]
        masks = masks.float().contiguous()
        if masks.numel() != B * x.shape[2] * x.shape[3]:
            raise ValueError(f"masks {tuple(masks.shape)} do not match the {B} one-channel predictions")
        if counts.dtype != torch.int64 or not counts.is_contiguous() or counts.numel() != (B if per_sample else 1) * 4:
            raise ValueError("counts must be a contiguous int64 tensor of 4 values per plane")
        with torch.no_grad():
            run = UNetRun(self, x.detach().float().contiguous(), training=self.training)
            logits = run.forward(count_into=(masks, counts, per_sample, return_logits))
        self.intermediate_features = _Features([run.acts[n] for n in FEATURE_LAYERS], run.lease.pool)
        run.release()
        return logits

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("unet_amd.UNet runs on the HIP kernels only: move the model and the input to a "
                               "ROCm device (there is no CPU path)")
        if x.dim() != 4:
            raise ValueError(f"expected a [B,C,H,W] tensor, got shape {tuple(x.shape)}")
        params = list(self.parameters())
        if params[0].device != x.device:
            raise RuntimeError(f"input is on {x.device} but the model is on {params[0].device}")
        need_grad = torch.is_grad_enabled() and self.training and any(p.requires_grad for p in params)
        if need_grad:
            x = x.float().contiguous()
            logits = _UNetFn.apply(self, x, *params)
            run = self._last_run
        else:
            with torch.no_grad():
                xf = x.detach().float().contiguous()
                run = UNetRun(self, xf, training=self.training)
                logits = run.forward()
        self.intermediate_features = _Features([run.acts[n] for n in FEATURE_LAYERS], run.lease.pool)
        if not need_grad:
            run.release()
        self._last_run = None
        return logits
