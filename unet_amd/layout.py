"""Device-memory plumbing for the HIP kernels: padded-linear NHWC activations and their pool.

Layout (see include/unet_hip.h): a logical [B,H,W,C] bf16 tensor is stored as
Mp = B*(H+1)*(W+1) rows of `cstride` channels with one zero pad column / pad row per image and
`guard` zero rows before and after.  Kernels never write guard rows, so a pooled buffer is
zero-filled once, when it is created.
"""
import torch

from . import _lib

_lib_fn = _lib.lib


def mp_rows(B, H, W):
    return B * (H + 1) * (W + 1)


class Act:
    """A channel window [coff, coff+C) of a padded-linear bf16 buffer of geometry (B,H,W)."""

    __slots__ = ("buf", "ptr", "B", "H", "W", "C", "cstride", "coff", "guard")

    def __init__(self, buf, B, H, W, C, cstride, coff, guard):
        self.buf, self.B, self.H, self.W = buf, B, H, W
        self.C, self.cstride, self.coff, self.guard = C, cstride, coff, guard
        self.ptr = buf.data_ptr() + guard * cstride * 2     # address of row 0

    @property
    def Mp(self):
        return mp_rows(self.B, self.H, self.W)

    def window(self, coff, C):
        assert coff + C <= self.cstride
        return Act(self.buf, self.B, self.H, self.W, C, self.cstride, coff, self.guard)

    # ---- host-side conversions (tests, intermediate_features) ------------------------------
    def to_nchw(self, half=False):
        """fp32 NCHW copy of the interior (half=True: the rows hold IEEE fp16, not bf16)."""
        B, H, W = self.B, self.H, self.W
        buf = self.buf.view(torch.float16) if half else self.buf
        rows = buf[self.guard:self.guard + self.Mp].view(B, H + 1, W + 1, self.cstride)
        return rows[:, :H, :W, self.coff:self.coff + self.C].permute(0, 3, 1, 2).float()

    def padded_rows(self):
        """[Mp, C] view including pad rows (tests)."""
        return self.buf[self.guard:self.guard + self.Mp, self.coff:self.coff + self.C]

    def fill_from_nchw(self, t, half=False):
        B, H, W = self.B, self.H, self.W
        buf = self.buf.view(torch.float16) if half else self.buf
        rows = buf[self.guard:self.guard + self.Mp].view(B, H + 1, W + 1, self.cstride)
        rows[:, :, :, self.coff:self.coff + self.C] = 0
        rows[:, :H, :W, self.coff:self.coff + self.C] = t.permute(0, 2, 3, 1).to(buf.dtype)


def new_act(B, H, W, C, device, cstride=None):
    """Freshly zeroed buffer (guards, pads and interior all zero)."""
    cstride = cstride or C
    guard = _lib_fn.uh_guard_rows(W)
    buf = torch.zeros(guard * 2 + mp_rows(B, H, W), cstride, dtype=torch.bfloat16, device=device)
    return Act(buf, B, H, W, C, cstride, 0, guard)


class BufferPool:
    """Per-device pool of activation buffers, float scratch and row masks.

    A Lease hands buffers out for one forward(+backward); releasing it returns them.  Training
    keeps the lease alive on the autograd context, so a second forward before the first
    backward gets different buffers instead of overwriting saved activations.
    """

    def __init__(self, device):
        self.device = device
        self.free = {}          # key -> list of tensors
        self.rowmasks = {}      # (B,H,W) -> uint8 tensor
        self.handed = {}        # id(tensor) -> times handed out (tells a reader that its buffer went to someone else)
        self.fin_counters = {}  # stream handle -> zeroed int32 tensor
        self.bytes_allocated = 0

    def rowmask(self, B, H, W):
        key = (B, H, W)
        m = self.rowmasks.get(key)
        if m is None:
            m = torch.empty(mp_rows(B, H, W), dtype=torch.uint8, device=self.device)
            _lib.call("uh_rowmask_build", B, H, W, m.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream)
            self.rowmasks[key] = m
        return m

    def counters(self, stream):
        """The arrival counters of the launches that finalise their own sums (uh_*_fin, uh_dwconv3x3_fwd_bn): one
        zeroed array per stream handle; every such launch leaves it zero, and launches on one stream are ordered."""
        c = self.fin_counters.get(stream)
        if c is None:
            c = self.fin_counters[stream] = torch.zeros(_lib_fn.uh_fin_counters(), dtype=torch.int32, device=self.device)
        return c

    def lease(self):
        return Lease(self)

    def clear(self):
        self.free.clear()
        self.rowmasks.clear()
        self.handed.clear()
        self.bytes_allocated = 0


class Lease:
    def __init__(self, pool):
        self.pool = pool
        self.held = []          # (key, tensor)

    def _take(self, key, make):
        lst = self.pool.free.get(key)
        t = lst.pop() if lst else make()
        self.held.append((key, t))
        self.pool.handed[id(t)] = self.pool.handed.get(id(t), 0) + 1
        return t

    def act(self, B, H, W, C, cstride=None, tag=None):
        """tag: a private sub-pool.  Buffers are zero when created and kernels only ever write the channels
        of their view, so a tagged buffer whose upper channels are never written keeps them zero."""
        cstride = cstride or C
        guard = _lib_fn.uh_guard_rows(W)
        rows = guard * 2 + mp_rows(B, H, W)
        key = ("act", tag, B, H, W, cstride)

        def make():
            self.pool.bytes_allocated += rows * cstride * 2
            return torch.zeros(rows, cstride, dtype=torch.bfloat16, device=self.pool.device)

        return Act(self._take(key, make), B, H, W, C, cstride, 0, guard)

    def f32(self, n):
        key = ("f32", n)

        def make():
            self.pool.bytes_allocated += n * 4
            return torch.empty(n, dtype=torch.float32, device=self.pool.device)

        return self._take(key, make)

    def bf16_flat(self, n, tag=None):
        """tag: a private, zero-initialised sub-pool (packed weights with zero padding)."""
        key = ("bf16", tag, n)

        def make():
            self.pool.bytes_allocated += n * 2
            f = torch.empty if tag is None else torch.zeros
            return f(n, dtype=torch.bfloat16, device=self.pool.device)

        return self._take(key, make)

    def release(self):
        for key, t in self.held:
            self.pool.free.setdefault(key, []).append(t)
        self.held = []

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


_POOLS = {}


def pool_for(device):
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    p = _POOLS.get(idx)
    if p is None:
        p = _POOLS[idx] = BufferPool(torch.device("cuda", idx))
    return p
