import sys, torch, torch.nn as nn
sys.path.insert(0, '.')
import unet_amd
from oracle import ref_cpu, emu_bf16
DEV = torch.device('cuda', 0)
def cosine(a,b):
    a,b=a.double().flatten(),b.double().flatten(); return (a@b/(a.norm()*b.norm()+1e-30)).item()
B,C,H,W = [int(v) for v in sys.argv[1:5]]
torch.manual_seed(4)
ref = ref_cpu.RefLUNeXt(C,1)
emu = emu_bf16.EmuLUNeXt(C,1); emu.load_state_dict(ref.state_dict())
model = unet_amd.LUNeXt(C,1); model.load_state_dict(ref.state_dict()); model=model.to(DEV)
x,m = ref_cpu.synth_batch(B,C,H,W)
ref.train(); model.train(); emu.train()
lo = ref(x); lr = nn.BCEWithLogitsLoss()(lo,m); lr.backward()
le_ = emu(x); le = nn.BCEWithLogitsLoss()(le_,m); le.backward()
lg_ = model(x.to(DEV)); lg = nn.BCEWithLogitsLoss()(lg_, m.to(DEV)); lg.backward()
print('loss hip', lg.item(), 'fp32', lr.item(), 'emu', le.item())
print('logit err hip-fp32', (lg_.detach().cpu()-lo.detach()).abs().max().item(), 'emu-fp32', (le_.detach()-lo.detach()).abs().max().item(), 'hip-emu', (lg_.detach().cpu()-le_.detach()).abs().max().item(), 'max', lo.abs().max().item())
for (n,p),(n2,p2),(n3,p3) in zip(model.named_parameters(), ref.named_parameters(), emu.named_parameters()):
    if p2.grad is None: continue
    g, r, e = p.grad.cpu(), p2.grad, p3.grad
    print(f"{n:28s} cos hip-fp32 {cosine(g,r):.5f} emu-fp32 {cosine(e,r):.5f} hip-emu {cosine(g,e):.6f} norm {g.norm().item():.4e} ref {r.norm().item():.4e} emu {e.norm().item():.4e}")
