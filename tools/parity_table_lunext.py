import sys, os, torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from oracle import ref_cpu, emu_bf16
DEV = torch.device("cuda", 0)
def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(0)
ref = ref_cpu.RefLUNeXt(3, 1).train()
model = unet_amd.LUNeXt(3, 1); model.load_state_dict(ref.state_dict()); model = model.to(DEV).train()
emu = emu_bf16.EmuLUNeXt(3, 1).train(); emu.load_state_dict(ref.state_dict())
x, m = ref_cpu.synth_batch(B, 3, 256, 256)
L = ref_cpu.RefSoftDiceClDice()
L(m, ref_cpu.probs_from_logits(ref(x))).backward()
L(m, ref_cpu.probs_from_logits(emu(x))).backward()
unet_amd.soft_dice_cldice()(m.to(DEV), torch.sigmoid(model(x.to(DEV))).clamp(1e-6, 1 - 1e-6)).backward()
torch.cuda.synchronize()
refs = dict(ref.named_parameters()); emus = dict(emu.named_parameters())
for n, p in model.named_parameters():
    if p.grad is None or n.endswith("depthwise.bias"): continue
    g, r, e = p.grad.float().cpu(), refs[n].grad, emus[n].grad
    extra = ""
    if g.numel() <= 9: extra = f" hip {[round(v, 6) for v in g.flatten().tolist()]} ref {[round(v, 6) for v in r.flatten().tolist()]} emu {[round(v, 6) for v in e.flatten().tolist()]}"
    print(f"{n:28s} n={g.numel():7d} hip cos {cosine(g, r):.4f} norm {g.norm()/r.norm():.4f} | emu cos {cosine(e, r):.4f} norm {e.norm()/r.norm():.4f}{extra}")
