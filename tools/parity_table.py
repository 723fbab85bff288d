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
ref = ref_cpu.RefUNet(3, 1).train()
model = unet_amd.UNet(3, 1); model.load_state_dict(ref.state_dict()); model = model.to(DEV).train()
emu = emu_bf16.EmuUNet(3, 1).train(); emu.load_state_dict(ref.state_dict())
x, m = ref_cpu.synth_batch(B, 3, 256, 256)
crit = nn.BCEWithLogitsLoss()
crit(ref(x), m).backward()
crit(emu(x), m).backward()
crit(model(x.to(DEV)), m.to(DEV)).backward()
torch.cuda.synchronize()
refs = dict(ref.named_parameters()); emus = dict(emu.named_parameters())
for n, p in model.named_parameters():
    if n.endswith(".bias") and n.startswith("conv"): continue
    g, r, e = p.grad.float().cpu(), refs[n].grad, emus[n].grad
    print(f"{n:22s} hip cos {cosine(g, r):.4f} norm {g.norm()/r.norm():.4f} | emu cos {cosine(e, r):.4f} norm {e.norm()/r.norm():.4f} | hip-vs-emu cos {cosine(g, e):.4f}")
