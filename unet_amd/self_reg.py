"""Drop-in `SelfRegLoss` (models/losses.py:6-217; built at trainer/training_utils.py:100-101, called as
`criterion(outputs, masks, model.intermediate_features)` at :143,160,199).

    loss = 0.4 * CrossEntropy + 0.6 * Dice + 0.015 * (LSRC + LIFD)                      losses.py:72

What runs where: the Dice term (forward and backward) and both auxiliaries are HIP kernels of libunet_hip.so.
When the features are the list a `unet_amd.UNet` forward pass left behind, LSRC and LIFD read the bf16
activations where they lie; the 15 NCHW fp32 copies the reference's loss walks over (2.4 GB at batch 32, 256 x 256)
are never made.  A plain list of NCHW CUDA tensors is accepted too: each is repacked to the kernels' layout,
which rounds it to bf16 (the precision these models store activations in).

Differences from the reference, on purpose:
  * the auxiliaries stay on the device and are added there; the reference reads both back with .item() (:68-69);
  * the channel subsets of LSRC come from a private generator; the reference reseeds the global one (:129,136);
  * CrossEntropy over one-channel logits is identically zero with zero gradient (log_softmax over one class),
    so it is not computed; logits with more than one channel are refused.
"""
import torch
import torch.nn as nn

from . import _lib
from .layout import mp_rows, new_act


def reference_seed(idx):
    """losses.py:126.  Python salts string hashes per process unless PYTHONHASHSEED is set, so this value --
    like the reference's -- is fixed within a process only.  Pass it as `seed_of` to reproduce a reference run made
    under a fixed PYTHONHASHSEED."""
    return hash(str(idx) + "feature") % 10000


def stable_seed(idx):
    """The default: the same role as losses.py:126 (one seed in [0, 10000) per feature index) from a hash that does
    not depend on the process, so every rank and every run of a data-parallel job picks the same channel subsets."""
    import zlib
    return zlib.crc32((str(idx) + "feature").encode()) % 10000


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _DiceFn(torch.autograd.Function):
    """w_dice * Dice(sigmoid(logits), targets) + w_aux * (aux[0] + aux[1]); gradient to the logits only."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, logits, targets, aux, w_dice, w_aux):
        logits = logits.contiguous()
        targets = targets.contiguous()
        B = logits.shape[0]
        n = logits.numel() // B
        dev = logits.device
        ws_bytes = _lib.lib.uh_selfreg_loss_workspace(B, n)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
        out = torch.empty(1 + 2 * B, dtype=torch.float32, device=dev)
        loss, coef = out[:1], out[1:]
        _lib.call("uh_selfreg_loss_fwd", B, n, logits.data_ptr(), targets.data_ptr(),
                  aux.data_ptr() if aux is not None else None, w_dice, w_aux, loss.data_ptr(), coef.data_ptr(),
                  ws.data_ptr(), ws_bytes, _stream())
        ctx.save_for_backward(logits, targets, coef)
        ctx.w_dice = w_dice
        return loss.view(())

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gout):
        logits, targets, coef = ctx.saved_tensors
        B = logits.shape[0]
        g = gout.detach().to(torch.float32).reshape(1).contiguous()
        dlogits = torch.empty_like(logits)
        _lib.call("uh_selfreg_loss_bwd", B, logits.numel() // B, logits.data_ptr(), targets.data_ptr(),
                  coef.data_ptr(), g.data_ptr(), ctx.w_dice, dlogits.data_ptr(), _stream())
        return dlogits, None, None, None, None


def _log2_factor(big, small):
    f = big // small if small else 0
    if f < 1 or small * f != big or f & (f - 1):
        return None
    return f


class SelfRegLoss(nn.Module):
    """`SelfRegLoss()` as the reference builds it; `seed_of(idx)` (default: `stable_seed`; `reference_seed` is the reference's expression) gives the
    seed of the channel subset LSRC takes from feature idx."""

    W_CE, W_DICE, W_AUX = 0.4, 0.6, 0.015                  # losses.py:72

    def __init__(self, seed_of=None):
        super().__init__()
        self.seed_of = seed_of or stable_seed
        self._subsets = {}                                 # (device, idx, C, Cf, seed) -> (n, sel_f, sel_p)

    # ---- constants ------------------------------------------------------------------------------------------
    def _acts_of(self, features):
        acts = features.acts() if hasattr(features, "acts") else None
        if acts is not None:
            return acts
        out = []
        for i, f in enumerate(features):
            if not (torch.is_tensor(f) and f.is_cuda and f.dim() == 4):
                raise ValueError(f"intermediate_features[{i}]: expected a [B,C,H,W] tensor on a ROCm device")
            B, C, H, W = f.shape
            if C % 8 or C > 1024:
                raise ValueError(f"intermediate_features[{i}]: {C} channels (the kernels take multiples of 8, at "
                                 "most 1024)")
            a = new_act(B, H, W, C, f.device)
            a.fill_from_nchw(f.detach().float())
            out.append(a)
        return out

    def _subset(self, dev, idx, C, Cf):
        """losses.py:120-144: which channels of the feature and of the pooled last feature are compared."""
        seed = int(self.seed_of(idx)) if C != Cf else 0
        key = (dev, idx, C, Cf, seed)
        hit = self._subsets.get(key)
        if hit is None:
            n = min(C, Cf)
            sel_f = torch.arange(C)[:n]
            sel_p = torch.arange(Cf)[:n]
            if C > n:
                sel_f = torch.randperm(C, generator=torch.Generator().manual_seed(seed))[:n]
            if Cf > n:
                sel_p = torch.randperm(Cf, generator=torch.Generator().manual_seed(seed + 1))[:n]
            hit = (n, sel_f.to(torch.int32).to(dev), sel_p.to(torch.int32).to(dev))
            self._subsets[key] = hit
        return hit

    def auxiliaries(self, intermediate_features):
        """Device tensor [2] = (LSRC, LIFD) of losses.py:81-217 for a feature list, or None when the list has
        fewer than two entries (both are zero then, losses.py:93-96,168-170)."""
        acts = self._acts_of(intermediate_features)
        n = len(acts)
        if n <= 1:
            return None
        final = acts[-1]
        dev = final.buf.device
        B, Cf = final.B, final.C
        s = _stream()
        # pooled levels of the last feature, by power-of-two factor
        levels = {1: (final.ptr, 0, final.cstride, final.coff, None)}
        factors = []
        for i, a in enumerate(acts[:-1]):
            if a.B != B:
                raise ValueError(f"intermediate_features[{i}]: batch {a.B}, last feature has {B}")
            fh, fw = _log2_factor(final.H, a.H), _log2_factor(final.W, a.W)
            if fh is None or fh != fw:
                raise NotImplementedError(f"intermediate_features[{i}] is {a.H}x{a.W}, the last one {final.H}x"
                                          f"{final.W}: only power-of-two reductions of the last feature are built")
            factors.append(fh)
        f = 1
        while f < max(factors):
            src = levels[f]
            f *= 2
            Ho, Wo = final.H // f, final.W // f
            buf = torch.empty(mp_rows(B, Ho, Wo) * Cf, dtype=torch.float32, device=dev)
            _lib.call("uh_selfreg_pool2", B, Ho, Wo, Cf, src[0], src[1], src[2], src[3], buf.data_ptr(), s)
            levels[f] = (buf.data_ptr(), 1, Cf, 0, buf)
        aux = torch.empty(2, dtype=torch.float32, device=dev)
        for idx, (a, f) in enumerate(zip(acts[:-1], factors)):
            C, H, W = a.C, a.H, a.W
            sp = _lib.lib.uh_selfreg_stat_parts(B, H, W, C)
            pp = _lib.lib.uh_selfreg_pair_parts(B, H, W, C)
            if sp <= 0 or pp <= 0:
                raise ValueError(f"intermediate_features[{idx}]: shape [{B},{C},{H},{W}] is not supported")
            stats = torch.empty(B * sp * 2 * C, dtype=torch.float32, device=dev)
            order = torch.empty(C, dtype=torch.int32, device=dev)
            part = torch.empty(pp * 2, dtype=torch.float32, device=dev)
            nsel, sel_f, sel_p = self._subset(dev, idx, C, Cf)
            pooled = levels[f]
            _lib.call("uh_selfreg_plane_stats", B, H, W, C, a.ptr, a.cstride, a.coff, stats.data_ptr(), s)
            _lib.call("uh_selfreg_rank", B, H, W, C, stats.data_ptr(), order.data_ptr(), s)
            _lib.call("uh_selfreg_pair_sums", B, H, W, C, a.ptr, a.cstride, a.coff, order.data_ptr(),
                      pooled[0], pooled[1], Cf, pooled[2], pooled[3], nsel, sel_f.data_ptr(), sel_p.data_ptr(),
                      part.data_ptr(), s)
            pixels = float(B) * H * W
            _lib.call("uh_selfreg_accumulate", part.data_ptr(), pp, 1.0 / (pixels * nsel * (n - 1)),
                      1.0 / (pixels * (C // 2) * (n - 1)), 1 if idx == 0 else 0, aux.data_ptr(), s)
        return aux

    # ---- the loss ---------------------------------------------------------------------------------------------
    def forward(self, predictions, targets, intermediate_features):
        if not predictions.is_cuda:
            raise RuntimeError("unet_amd.SelfRegLoss runs on the HIP kernels only (there is no CPU path)")
        if predictions.dim() != 4 or predictions.shape[1] != 1:
            raise NotImplementedError("SelfRegLoss: one-channel logits [B,1,H,W] only (the CrossEntropy term of "
                                      "losses.py:58 is identically zero there)")
        if targets.shape != predictions.shape:
            raise ValueError(f"targets {tuple(targets.shape)} do not match predictions {tuple(predictions.shape)}")
        with torch.no_grad():
            aux = self.auxiliaries(intermediate_features)
        return _DiceFn.apply(predictions, targets.to(torch.float32), aux, self.W_DICE, self.W_AUX)
