"""CPU oracle with the storage roundings of the HIP path.  TEST INFRASTRUCTURE ONLY.

`EmuUNet` is `ref_cpu.RefUNet` (same parameters, same maths, fp32 arithmetic) with a rounding
inserted at every point where the HIP path stores a tensor in 16 bits:

  * MFMA operands: conv / transposed-conv weights and every activation that feeds a conv -> bf16
  * the raw conv output that feeds BatchNorm                                              -> fp16
  * every activation gradient and conv-output gradient (dgrad / wgrad operands)           -> bf16

(the first conv and the 1x1 head run on fp32 weights in the HIP path and do so here).
Against the plain fp32 oracle the HIP path differs by what those roundings do to a 23-layer
network; against THIS oracle it differs only by fp32 summation order, so a layout or indexing
mistake cannot hide inside a "bf16 tolerance".  `tests/test_unet_gpu.py` checks both.
"""
import torch
import torch.nn.functional as F

from . import ref_cpu


class _Q(torch.autograd.Function):
    """forward: round to `dtype` (None = keep); backward: round the gradient to bf16."""

    @staticmethod
    def forward(ctx, x, dtype):
        return x if dtype is None else x.to(dtype).float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float(), None


def q(x, dtype):
    return _Q.apply(x, dtype)


def w_bf16(w):
    """bf16-rounded weight in the forward pass, straight-through fp32 gradient."""
    return w + (w.bfloat16().float() - w).detach()


class EmuUNet(ref_cpu.RefUNet):
    def _cbr(self, name, x, first=False):
        conv = getattr(self, name)
        bn = getattr(self, ref_cpu.bn_name(name))
        w = conv.weight if first else w_bf16(conv.weight)
        y = q(F.conv2d(x, w, conv.bias, padding=1), torch.float16)
        a = q(F.relu(bn(y)), torch.bfloat16)
        if name in ref_cpu.UNET_FEATURE_LAYERS:
            self.intermediate_features.append(a)
        return a

    def forward(self, x):
        self.intermediate_features = []
        skips = []
        for lvl in (1, 2, 3, 4):
            x = self._cbr(f"conv{lvl}_2", self._cbr(f"conv{lvl}_1", x, first=(lvl == 1)))
            skips.append(x)
            x = q(F.max_pool2d(x, 2, 2), None)
        x = self._cbr("conv5_2", self._cbr("conv5_1", x))
        for k in (1, 2, 3, 4):
            up = getattr(self, f"upconv{k}")
            u = q(F.conv_transpose2d(x, w_bf16(up.weight), up.bias, stride=2), torch.bfloat16)
            x = q(torch.cat([skips[4 - k], u], dim=1), None)
            x = self._cbr(f"conv_up{k}_2", self._cbr(f"conv_up{k}_1", x))
        return self.final_conv(x)


class EmuLUNeXt(ref_cpu.RefLUNeXt):
    """RefLUNeXt with the HIP path's storage roundings: bf16 activations and GEMM weights (depthwise,
    8-channel gate and head weights stay fp32), fp16 BatchNorm inputs, bf16 gradients."""

    def _dwb(self, blk, x):
        y = q(F.conv2d(x, blk.depthwise.weight, blk.depthwise.bias, padding=1, groups=x.shape[1]), torch.float16)
        return q(F.relu(blk.bn(y) + x), torch.bfloat16)

    def _gate(self, att, g, x):
        small = g.shape[1] < 64
        wq = (lambda w: w) if small else w_bf16
        rq = (lambda t: t) if small else (lambda t: q(t, torch.bfloat16))
        z1 = rq(F.conv2d(g, wq(att.sigmoid_conv.weight), att.sigmoid_conv.bias))
        z2 = rq(F.conv2d(g, wq(att.tanh_conv.weight), att.tanh_conv.bias))
        s = rq(g + x)
        z3 = rq(F.conv2d(s, wq(att.combine_conv.weight), att.combine_conv.bias))
        return q(torch.sigmoid(z3) * x + torch.sigmoid(z1) * torch.tanh(z2), torch.bfloat16)

    def forward(self, x):
        x = q(x, torch.bfloat16)
        skips = []
        for lvl in range(1, 5):
            x = self._dwb(getattr(self, f"conv{lvl}_2"), self._dwb(getattr(self, f"conv{lvl}_1"), x))
            skips.append(x)
            d = getattr(self, f"downconv{lvl}")
            x = q(F.conv2d(x, w_bf16(d.weight), d.bias, stride=2), torch.bfloat16)
        x = self._dwb(self.conv5_2, self._dwb(self.conv5_1, x))
        for k in range(1, 5):
            up = getattr(self, f"upconv{k}")
            u = q(F.conv_transpose2d(x, w_bf16(up.weight), up.bias, stride=2), torch.bfloat16)
            if u.shape != skips[4 - k].shape:                # lunext.py:226-227: stored again after the size fix
                u = q(F.interpolate(u, size=skips[4 - k].shape[2:], mode="bilinear", align_corners=False), torch.bfloat16)
            x = self._gate(getattr(self, f"att{k}"), u, skips[4 - k])
            x = self._dwb(getattr(self, f"conv_up{k}_2"), self._dwb(getattr(self, f"conv_up{k}_1"), x))
        return self.final_conv(x)
