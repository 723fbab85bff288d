"""bf16 GEMM operands of all MFMA layers of a U-Net, repacked by one launch per direction.

The state dict keeps fp32 OIHW / [Cin][Cout][2][2] weights (models/unet.py:19-76); the MFMA kernels read
bf16 [N][tap][K] operands, the input-gradient kernels the same with rotated taps.  `WeightPacker` owns the
packed buffers and a device-resident descriptor table, and repacks a direction only when a parameter has
changed since the last repack (`Tensor._version`), so inference packs once.
"""
import torch

from . import _lib

WIDTHS = (64, 128, 256, 512, 1024)


def unet_pack_plan(module):
    """(direction, key, parameter, N, T, K, sn, st, sk, on, ot, reverse) for every packed operand."""
    plan = []
    names = [f"conv{l}_{i}" for l in range(1, 6) for i in (1, 2)] + \
            [f"conv_up{k}_{i}" for k in range(1, 5) for i in (1, 2)]
    for name in names:
        if name == "conv1_1":
            continue                      # the first layer reads fp32 weights directly
        conv = getattr(module, name)
        cout, cin = conv.out_channels, conv.in_channels
        w = conv.weight
        # forward: rows n = cout, taps t = ky*3+kx, k = cin  <-  OIHW
        plan.append(("fwd", name, w, cout, 9, cin, cin * 9, 1, 9, 9 * cin, cin, 0))
        # input gradient: rows n = cin, k = cout, taps rotated by 180 degrees
        plan.append(("bwd", name, w, cin, 9, cout, 9, 1, cin * 9, 9 * cout, cout, 1))
    for k in range(1, 5):
        up = getattr(module, f"upconv{k}")
        cin, w_ = up.in_channels, up.out_channels          # weight [cin][w][2][2]
        # forward: rows n = (a*2+b)*w + co  (N = w, T = 4 taps), k = cin
        plan.append(("fwd", f"upconv{k}", up.weight, w_, 4, cin, 4, 1, w_ * 4, cin, w_ * cin, 0))
        # input gradient: rows n = ci, k = co, taps as they are
        plan.append(("bwd", f"upconv{k}", up.weight, cin, 4, w_, w_ * 4, 1, 4, 4 * w_, w_, 0))
    return plan


class WeightPacker:
    def __init__(self, module, device):
        self.device = device
        self.groups = {}
        plan = unet_pack_plan(module)
        for direction in ("fwd", "bwd"):
            entries = [e for e in plan if e[0] == direction]
            total = sum(e[3] * e[4] * e[5] for e in entries)
            flat = torch.empty(total, dtype=torch.bfloat16, device=device)
            views, rows, off = {}, [], 0
            for (_, key, p, N, T, K, sn, st, sk, on, ot, rev) in entries:
                n = N * T * K
                views[key] = flat[off:off + n]
                rows.append([p.data_ptr(), flat.data_ptr() + 2 * off, N, T, K, rev, sn, st, sk, on, ot])
                off += n
            desc = torch.tensor(rows, dtype=torch.int64).to(device)
            self.groups[direction] = dict(flat=flat, views=views, desc=desc, params=[e[2] for e in entries],
                                          ptrs=[e[2].data_ptr() for e in entries], versions=None)

    def valid_for(self, module):
        """False once a parameter's storage moved (model.to(), a new tensor assigned)."""
        return all(p.data_ptr() == q for g in self.groups.values() for p, q in zip(g["params"], g["ptrs"]))

    def mark_dirty(self):
        """weights were changed by something that does not move Tensor._version (a kernel writing raw pointers)"""
        for g in self.groups.values():
            g["versions"] = None

    def get(self, direction, stream):
        """{layer name: packed bf16 operand}, repacked first if any weight changed since the last repack."""
        g = self.groups[direction]
        versions = [p._version for p in g["params"]]
        if versions != g["versions"]:
            _lib.call("uh_pack_weights_multi", g["desc"].data_ptr(), len(g["params"]), 256, stream)
            g["versions"] = versions
        return g["views"]


def packer_for(module, device):
    pk = getattr(module, "_packer", None)
    if pk is None or pk.device != device or not pk.valid_for(module):
        pk = WeightPacker(module, device)
        module._packer = pk
    return pk
