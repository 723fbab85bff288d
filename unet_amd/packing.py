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

    def get_both(self, stream):
        """(forward operands, input-gradient operands): both directions in ONE launch where both are stale and their
        descriptor tables lie one behind the other (a training step, whose optimizer changed every weight)."""
        f, b = self.groups["fwd"], self.groups["bwd"]
        vf, vb = [p._version for p in f["params"]], [p._version for p in b["params"]]
        joined = f["desc"].data_ptr() + f["desc"].numel() * 8 == b["desc"].data_ptr()
        if joined and vf != f["versions"] and vb != b["versions"]:
            _lib.call("uh_pack_weights_multi", f["desc"].data_ptr(), len(vf) + len(vb), 256, stream)
            f["versions"], b["versions"] = vf, vb
        return self.get("fwd", stream), self.get("bwd", stream)


def lunext_pack_plan(module):
    """Buffers and entries of every packed operand of a LUNeXt (models/lunext.py:139-185): the 2x2 stride-2
    convolutions, the transposed convolutions and the gates' 1x1 convolutions at 64 channels and more.

    Returns ({direction: {key: elements}}, [(direction, key, element offset, parameter, N, T, K, sn, st, sk, on, ot)]).
    Operands narrower than the 64 columns/rows the GEMM kernels work on are zero-padded: the buffers start as zeros
    and the repack writes the real entries only.  Keys ending in ".f32" are fp32 tables (the zero-padded parameters of
    the 8-channel vector kernels), written by the same launch."""
    chans = [module.in_channels, 64, 128, 256, 512]
    if chans[0] > 8:
        raise ValueError(f"LUNeXt's first level is carried as 8 channels: in_channels={chans[0]} is not supported")
    cp = [max(8, (c + 7) // 8 * 8) for c in chans]
    sizes = {"fwd": {}, "bwd": {}}
    entries = []
    for l in range(1, 5):
        w = getattr(module, f"downconv{l}").weight                   # [cout][cin][2][2]
        cin, cpi, cout = chans[l - 1], cp[l - 1], chans[l]
        K = max(64, 4 * cpi)
        sizes["fwd"][f"down{l}"] = cout * K
        entries.append(("fwd", f"down{l}", 0, w, cout, 4, cin, cin * 4, 1, 4, K, cpi))
        sizes["bwd"][f"down{l}"] = K * cout
        entries.append(("bwd", f"down{l}", 0, w, cin, 4, cout, 4, 1, cin * 4, cout, cpi * cout))
    for k in range(1, 5):
        l = 4 - k
        w = getattr(module, f"upconv{k}").weight                     # [cin][cout][2][2]
        cin, cout, cpo = chans[l + 1], chans[l], cp[l]
        N = max(64, 4 * cpo)
        sizes["fwd"][f"up{k}"] = N * cin
        entries.append(("fwd", f"up{k}", 0, w, cout, 4, cin, 4, 1, cout * 4, cin, cpo * cin))
        sizes["bwd"][f"up{k}"] = cin * N
        entries.append(("bwd", f"up{k}", 0, w, cin, 4, cout, cout * 4, 1, 4, N, cpo))
        C = chans[l]
        att = getattr(module, f"att{k}")
        if cpo != cout:                                               # few real channels carried as 8
            sizes["fwd"][f"up{k}.bias.f32"] = cpo
            entries.append(("fwd", f"up{k}.bias.f32", 0, getattr(module, f"upconv{k}").bias, 1, 1, cout, 0, 0, 1, cpo, 0))
        if C < 64:
            if cpo != 8:
                continue
            # the 8-channel gate is one fused vector kernel; its table: three [8][8] weights, then three [8] biases
            sizes["fwd"][f"att{k}.gate8.f32"] = 216
            for j, conv in enumerate((att.sigmoid_conv, att.tanh_conv, att.combine_conv)):
                entries.append(("fwd", f"att{k}.gate8.f32", 64 * j, conv.weight, C, 1, C, C, 0, 1, 8, 0))
                entries.append(("fwd", f"att{k}.gate8.f32", 192 + 8 * j, conv.bias, 1, 1, C, 0, 0, 1, 8, 0))
            continue
        sizes["fwd"][f"att{k}.w12"] = 2 * C * C                        # [sigmoid_conv; tanh_conv] rows
        entries.append(("fwd", f"att{k}.w12", 0, att.sigmoid_conv.weight, C, 1, C, C, 0, 1, C, 0))
        entries.append(("fwd", f"att{k}.w12", C * C, att.tanh_conv.weight, C, 1, C, C, 0, 1, C, 0))
        sizes["fwd"][f"att{k}.w3"] = C * C
        entries.append(("fwd", f"att{k}.w3", 0, att.combine_conv.weight, C, 1, C, C, 0, 1, C, 0))
        # the three 1x1 convolutions as ONE operand over the input [g | x] (2C channels side by side): rows
        # [W1 0; W2 0; W3 W3], so that W3 (g + x) needs no g + x (the zero blocks are never written)
        sizes["fwd"][f"att{k}.wcat"] = 3 * C * 2 * C
        entries.append(("fwd", f"att{k}.wcat", 0, att.sigmoid_conv.weight, C, 1, C, C, 0, 1, 2 * C, 0))
        entries.append(("fwd", f"att{k}.wcat", C * 2 * C, att.tanh_conv.weight, C, 1, C, C, 0, 1, 2 * C, 0))
        entries.append(("fwd", f"att{k}.wcat", 2 * C * 2 * C, att.combine_conv.weight, C, 1, C, C, 0, 1, 2 * C, 0))
        entries.append(("fwd", f"att{k}.wcat", 2 * C * 2 * C + C, att.combine_conv.weight, C, 1, C, C, 0, 1, 2 * C, 0))
        sizes["bwd"][f"att{k}.wd"] = C * 3 * C                         # transposed, side by side along K
        for j, conv in enumerate((att.sigmoid_conv, att.tanh_conv, att.combine_conv)):
            entries.append(("bwd", f"att{k}.wd", j * C, conv.weight, C, 1, C, 1, 0, C, 3 * C, 0))
        sizes["bwd"][f"att{k}.wd3"] = C * C
        entries.append(("bwd", f"att{k}.wd3", 0, att.combine_conv.weight, C, 1, C, 1, 0, C, C, 0))
    fc = module.final_conv                                            # the head's [oc][cin] weight as [oc][8]
    oc = fc.out_channels
    sizes["fwd"]["head.w.f32"] = oc * 8
    entries.append(("fwd", "head.w.f32", 0, fc.weight, oc, 1, chans[0], chans[0], 0, 1, 8, 0))
    return sizes, entries


class PlanPacker(WeightPacker):
    """WeightPacker over an explicit plan of zero-initialised buffers (several entries may share one)."""

    def __init__(self, module, device, plan):
        self.device = device
        self.groups = {}
        sizes, entries = plan
        tables = {}
        for direction in ("fwd", "bwd"):
            views, where = {}, {}
            for f32 in (False, True):
                start, off = {}, 0
                for key, n in sizes[direction].items():
                    if key.endswith(".f32") == f32:
                        start[key] = off
                        off += (n + 7) // 8 * 8                        # 16-byte aligned operands
                flat = torch.zeros(off, dtype=torch.float32 if f32 else torch.bfloat16, device=device)
                for key, o in start.items():
                    views[key] = flat[o:o + sizes[direction][key]]
                    where[key] = flat.data_ptr() + o * flat.element_size()
                if not f32:
                    main = flat
            mine = [e for e in entries if e[0] == direction]
            rows = [[p.data_ptr(), where[key] + eoff * views[key].element_size(), N, T, K,
                     2 if key.endswith(".f32") else 0, sn, st, sk, on, ot]
                    for (_, key, eoff, p, N, T, K, sn, st, sk, on, ot) in mine]
            tables[direction] = rows
            self.groups[direction] = dict(flat=main, views=views, params=[e[3] for e in mine],
                                          ptrs=[e[3].data_ptr() for e in mine], versions=None)
        # one device table, the forward descriptors first: a training step repacks both directions in one launch
        desc = torch.tensor(tables["fwd"] + tables["bwd"], dtype=torch.int64).to(device)
        nf = len(tables["fwd"])
        self.groups["fwd"]["desc"], self.groups["bwd"]["desc"] = desc[:nf], desc[nf:]


def lunext_packer_for(module, device):
    pk = getattr(module, "_packer", None)
    if pk is None or pk.device != device or not pk.valid_for(module):
        pk = PlanPacker(module, device, lunext_pack_plan(module))
        module._packer = pk
    return pk


def packer_for(module, device):
    pk = getattr(module, "_packer", None)
    if pk is None or pk.device != device or not pk.valid_for(module):
        pk = WeightPacker(module, device)
        module._packer = pk
    return pk
