"""What another 16-bit storage format for the ACTIVATION GRADIENTS would do to gradient parity (CPU only).

oracle/emu_bf16.py rounds every stored activation gradient to bf16, as the HIP path does; against the fp32 oracle the
per-tensor cosine then falls to 0.92-0.95 at the bridge.  This script swaps that one rounding for
  * fp16 with the loss scaled by 2^16 (what torch.amp.GradScaler starts with), 11 bits of mantissa instead of 8,
  * bf16 with stochastic rounding (unbiased, same 8 bits),
and, to tell which rounding the loss of cosine comes from at all, for two changes of the FORWARD storage instead:
  * activations stored as fp16 instead of bf16 (gradients bf16 as shipped),
  * MFMA weight operands left in fp32 (everything else as shipped),
and prints the cosine of selected weight gradients against the fp32 oracle, batch 4 at 3 x 256 x 256."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from oracle import ref_cpu, emu_bf16

SCALE = 65536.0
MODE = "bf16"


def _round(g):
    if MODE == "bf16":
        return g.bfloat16().float()
    if MODE == "fp16":          # the backward pass runs on scale * loss, so g is already scaled here
        return g.half().float()
    if MODE == "bf16_sr":       # stochastic rounding: add uniform noise below the kept bits, then truncate
        bits = g.view(torch.int32)
        noise = torch.randint(0, 1 << 16, bits.shape, dtype=torch.int32)
        return ((bits + noise) & ~0xFFFF).view(torch.float32)
    raise ValueError(MODE)


emu_bf16._Q.backward = staticmethod(lambda ctx, g: (_round(g), None))
ACT = torch.bfloat16
_q_orig = emu_bf16.q
emu_bf16.q = lambda x, dtype: _q_orig(x, ACT if dtype is torch.bfloat16 else dtype)
W_ROUND = True
_w_orig = emu_bf16.w_bf16
emu_bf16.w_bf16 = lambda w: _w_orig(w) if W_ROUND else w


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


torch.manual_seed(0)
torch.set_num_threads(8)
ref = ref_cpu.RefUNet(3, 1).train()
x, m = ref_cpu.synth_batch(4, 3, 256, 256)
crit = nn.BCEWithLogitsLoss()
crit(ref(x), m).backward()
refs = {n: p.grad.clone() for n, p in ref.named_parameters()}
names = ["conv_up4_2.weight", "conv_up3_1.weight", "conv_up1_1.weight", "conv5_2.weight", "conv5_1.weight",
         "conv4_1.weight", "conv3_1.weight", "conv2_1.weight", "conv1_1.weight"]
MODES = ("bf16", "fp16", "bf16_sr", "act_fp16", "w_fp32")
print(f"{'tensor':20s}" + "".join(f"{k:>10s}" for k in MODES))
rows = {n: [] for n in names}
worst = {}
for mode in MODES:
    MODE = mode if mode in ("bf16", "fp16", "bf16_sr") else "bf16"
    ACT = torch.float16 if mode == "act_fp16" else torch.bfloat16
    W_ROUND = mode != "w_fp32"
    torch.manual_seed(1)
    emu = emu_bf16.EmuUNet(3, 1).train()
    emu.load_state_dict(ref.state_dict())
    scale = SCALE if mode == "fp16" else 1.0
    (crit(emu(x), m) * scale).backward()
    cs = {n: cosine(p.grad / scale, refs[n]) for n, p in emu.named_parameters()
          if not (n.endswith(".bias") and n.startswith("conv"))}
    worst[mode] = min(cs.items(), key=lambda kv: kv[1])
    for n in names:
        rows[n].append(cs[n])
for n in names:
    print(f"{n:20s}" + "".join(f"{v:10.4f}" for v in rows[n]))
print("worst tensor per mode:", {k: (v[0], round(v[1], 4)) for k, v in worst.items()})
