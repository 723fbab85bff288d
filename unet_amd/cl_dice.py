"""Drop-in clDice loss: `soft_dice_cldice`, `soft_cldice`, `soft_dice`, `SoftSkeletonize` with the
reference's names, constructor arguments and call convention (losses/cl_dice.py), computed by the HIP
kernels of libunet_hip.so in fp32.  Call as the trainer does (training_utils.py:146-148):

    probs = torch.sigmoid(logits).clamp(1e-6, 1 - 1e-6)
    loss = criterion(masks, probs)            # argument order (y_true, y_pred)
"""
import torch
import torch.nn as nn

from ._lib import call, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _planes(t, what):
    if t.dim() == 5:
        raise NotImplementedError("the 3-D (5-D tensor) branch of the soft skeleton is not built")
    if t.dim() != 4:
        raise ValueError(f"{what}: expected [B,C,H,W], got {tuple(t.shape)}")
    if not t.is_cuda:
        raise RuntimeError("unet_amd clDice runs on the HIP kernels only: tensors must be on a ROCm device")
    return t.shape[0] * t.shape[1], t.shape[2], t.shape[3]


class _ClDiceFn(torch.autograd.Function):
    """forward(y_true, y_pred, alpha, smooth, mode) -> 0-dim loss.  mode 0: (1-alpha)*dice + alpha*clDice,
    mode 1: clDice only (soft_cldice), mode 2: dice only (soft_dice)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, y_true, y_pred, alpha, smooth):
        planes, H, W = _planes(y_pred, "y_pred")
        if y_true.shape != y_pred.shape:
            raise ValueError(f"y_true {tuple(y_true.shape)} and y_pred {tuple(y_pred.shape)} differ")
        yt = y_true.detach().float().contiguous()
        yp = y_pred.detach().float().contiguous()
        ws = torch.empty(lib.uh_cldice_workspace(planes, H, W) // 4, dtype=torch.float32, device=yp.device)
        loss = torch.empty((), dtype=torch.float32, device=yp.device)
        call("uh_cldice_forward", planes, H, W, yt.data_ptr(), yp.data_ptr(), float(alpha), float(smooth),
             ws.data_ptr(), loss.data_ptr(), _stream())
        ctx.save_for_backward(yt, yp, ws)
        ctx.dims = (planes, H, W)
        return loss

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gout):
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("gradient with respect to y_true is not built")
        yt, yp, ws = ctx.saved_tensors
        planes, H, W = ctx.dims
        g = gout.detach().float().contiguous()
        grad = torch.empty_like(yp)
        call("uh_cldice_backward", planes, H, W, yt.data_ptr(), yp.data_ptr(), ws.data_ptr(), g.data_ptr(),
             grad.data_ptr(), _stream())
        return None, grad, None, None


_SEEDS = {}


def _seed(value, dev):
    """A 0-dim device tensor holding `value`, made once per (device, value): the loss's backward reads its upstream
    gradient from device memory, and a fill launch per step for a constant is a launch too many."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device(), value)
    t = _SEEDS.get(key)
    if t is None:
        if len(_SEEDS) > 64:
            _SEEDS.clear()
        t = _SEEDS[key] = torch.full((), value, dtype=torch.float32, device=dev)
    return t


class PreparedLabels:
    """What cldice_prepare hands to cldice_with_logits: the labels as fp32 planes, the loss's workspace with skel(y_true)
    on its way into it, and the event that says when it is there (None: made on the caller's stream)."""
    __slots__ = ("yt", "shape", "ws", "event")


def cldice_prepare(y_true, stream=None):
    """The half of cldice_with_logits that needs no prediction: skel(y_true) (cl_dice.py:119).  A training loop has the
    labels when the network's forward pass starts (training_utils.py:125-148); started then on `stream` (a
    torch.cuda.Stream other than the current one) the skeleton is made beside the network instead of between its forward
    and backward passes.  Pass the result as cldice_with_logits(..., prepared=...)."""
    planes, H, W = _planes(y_true, "y_true")
    p = PreparedLabels()
    p.yt, p.shape = y_true.detach().float().contiguous(), tuple(y_true.shape)
    dev = p.yt.device
    p.ws = torch.empty(lib.uh_cldice_workspace(planes, H, W) // 4, dtype=torch.float32, device=dev)
    cur = torch.cuda.current_stream(dev)
    if stream is None or stream == cur:
        p.event = None
        call("uh_cldice_skel_true", planes, H, W, p.yt.data_ptr(), p.ws.data_ptr(), cur.cuda_stream)
    else:
        stream.wait_stream(cur)              # the labels are written, the workspace's last user is done
        call("uh_cldice_skel_true", planes, H, W, p.yt.data_ptr(), p.ws.data_ptr(), stream.cuda_stream)
        p.event = torch.cuda.Event()
        p.event.record(stream)
    return p


def cldice_with_logits(y_true, logits, alpha=0.5, smooth=1.0, lo=1e-6, hi=1 - 1e-6, grad_scale=1.0, prepared=None):
    """The trainer's clDice branch (training_utils.py:146-148) without autograd:
        probs = sigmoid(logits).clamp(lo, hi); loss = soft_dice_cldice(alpha, smooth)(y_true, probs)
    returns (loss, grad_scale * d loss / d logits), both from the HIP kernels: one launch for sigmoid + clamp and one for
    their backward instead of six element-wise launches around the loss.  prepared: cldice_prepare(y_true)'s result
    (same bits either way)."""
    planes, H, W = _planes(logits, "logits")
    if y_true.shape != logits.shape:
        raise ValueError(f"y_true {tuple(y_true.shape)} and logits {tuple(logits.shape)} differ")
    x = logits.detach().float().contiguous()
    dev, s, n = x.device, _stream(), x.numel()
    probs = torch.empty_like(x)
    call("uh_sigmoid_clamp", n, x.data_ptr(), float(lo), float(hi), probs.data_ptr(), s)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    if prepared is None:
        yt = y_true.detach().float().contiguous()
        ws = torch.empty(lib.uh_cldice_workspace(planes, H, W) // 4, dtype=torch.float32, device=dev)
        call("uh_cldice_forward", planes, H, W, yt.data_ptr(), probs.data_ptr(), float(alpha), float(smooth),
             ws.data_ptr(), loss.data_ptr(), s)
    else:
        if prepared.shape != tuple(y_true.shape) or prepared.yt.device != dev:
            raise ValueError(f"prepared labels {prepared.shape} on {prepared.yt.device} do not belong to this call")
        yt, ws = prepared.yt, prepared.ws
        if prepared.event is not None:
            torch.cuda.current_stream(dev).wait_event(prepared.event)
        call("uh_cldice_forward_pred", planes, H, W, yt.data_ptr(), probs.data_ptr(), float(alpha), float(smooth),
             ws.data_ptr(), loss.data_ptr(), s)
    seed = _seed(float(grad_scale), dev)
    dprobs = torch.empty_like(x)
    call("uh_cldice_backward", planes, H, W, yt.data_ptr(), probs.data_ptr(), ws.data_ptr(), seed.data_ptr(),
         dprobs.data_ptr(), s)
    dlogits = probs                                           # the probabilities are not needed any more
    call("uh_sigmoid_clamp_bwd", n, x.data_ptr(), float(lo), float(hi), dprobs.data_ptr(), dlogits.data_ptr(), s)
    return loss, dlogits


class SoftSkeletonize(nn.Module):
    """cl_dice.py:5-50.  The HIP kernel is built for num_iter = 10, the only value the reference instantiates
    (cl_dice.py:57,103); forward only (the differentiable path is soft_dice_cldice / soft_cldice)."""

    def __init__(self, num_iter=40, check_range=True):
        super().__init__()
        self.num_iter = num_iter
        self.check_range = check_range       # soft_skel validates its input's range (a device read-back) unless False

    def _morph(self, img, which):
        planes, H, W = _planes(img, "img")
        x = img.detach().float().contiguous()
        out = torch.empty_like(x)
        if which == "erode":
            call("uh_soft_erode", planes, H, W, x.data_ptr(), out.data_ptr(), 0, _stream())
        else:
            call("uh_soft_dilate", planes, H, W, x.data_ptr(), out.data_ptr(), _stream())
        return out

    def soft_erode(self, img):
        return self._morph(img, "erode")

    def soft_dilate(self, img):
        return self._morph(img, "dilate")

    def soft_open(self, img):
        return self.soft_dilate(self.soft_erode(img))

    def soft_skel(self, img):
        if self.num_iter != 10:
            raise NotImplementedError("the soft-skeleton kernel is built for num_iter=10")
        planes, H, W = _planes(img, "img")
        x = img.detach().float().contiguous()
        if self.check_range and x.numel():
            # The kernel clamps to [0, 1] like the loss (cl_dice.py:116-117); the reference's stand-alone module does
            # not, so values outside that range are refused rather than silently changed.  One fused reduction and one
            # read-back; SoftSkeletonize(check_range=False) skips both (the loss path never comes through here).
            lo, hi = torch.aminmax(x)
            if lo.item() < 0 or hi.item() > 1:
                raise NotImplementedError("the soft-skeleton kernel takes inputs in [0,1] (it clamps, as the loss does)")
        out = torch.empty_like(x)
        ws = torch.empty(lib.uh_soft_skel_workspace(planes, H, W) // 4, dtype=torch.float32, device=x.device)
        call("uh_soft_skel", planes, H, W, x.data_ptr(), out.data_ptr(), ws.data_ptr(), _stream())
        return out

    def forward(self, img):
        return self.soft_skel(img)


class soft_dice_cldice(nn.Module):
    """cl_dice.py:97-125: (1-alpha)*soft_dice + alpha*clDice, ten skeleton iterations (iter_ is ignored there too)."""

    def __init__(self, iter_=3, alpha=0.5, smooth=1., exclude_background=False):
        super().__init__()
        self.iter, self.smooth, self.alpha = iter_, smooth, alpha
        self.soft_skeletonize = SoftSkeletonize(num_iter=10)
        self.exclude_background = exclude_background     # a no-op slice in the reference (cl_dice.py:107-109)

    def forward(self, y_true, y_pred):
        return _ClDiceFn.apply(y_true, y_pred, self.alpha, self.smooth)


class soft_cldice(nn.Module):
    """cl_dice.py:52-70: clDice term only; unlike soft_dice_cldice it does not clamp its inputs, so they must
    already lie in [0,1] (the kernel clamps, which is then the identity)."""

    def __init__(self, iter_=3, smooth=1., exclude_background=False):
        super().__init__()
        self.iter, self.smooth = iter_, smooth
        self.soft_skeletonize = SoftSkeletonize(num_iter=10)
        self.exclude_background = exclude_background

    def forward(self, y_true, y_pred):
        return _ClDiceFn.apply(y_true, y_pred, 1.0, self.smooth)


def soft_dice(y_true, y_pred):
    """cl_dice.py:73-94 (dice term only)."""
    return _ClDiceFn.apply(y_true, y_pred, 0.0, 1.0)
