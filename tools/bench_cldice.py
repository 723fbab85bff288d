"""clDice loss alone (forward of both skeletons + backward) at batch 32, 256 x 256: the strip-walk backward kernel
against the LDS-tile kernel (UNET_AMD_SKEL_TILES=1), bits compared, and a sweep of UNET_AMD_SKEL_R."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_amd
DEV = torch.device("cuda", 0)
B, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (32, 256, 256)))
g = torch.Generator().manual_seed(3)
m = (torch.rand(B, 1, H, W, generator=g) > 0.5).float().to(DEV)
p = torch.rand(B, 1, H, W, generator=g).clamp(1e-6, 1 - 1e-6).to(DEV).requires_grad_(True)
crit = unet_amd.soft_dice_cldice()


def run(env):
    for k in ("UNET_AMD_SKEL_TILES", "UNET_AMD_SKEL_R"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(3):
        (grad,) = torch.autograd.grad(crit(m, p), p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        (grad,) = torch.autograd.grad(crit(m, p), p)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20, grad.clone()


t_ref, g_ref = run({"UNET_AMD_SKEL_TILES": "1"})
print(f"tiles kernel: {t_ref * 1e3:7.1f} us per loss call", flush=True)
for R in ("0", "4", "6", "8", "10", "12", "16"):
    t, gr = run({"UNET_AMD_SKEL_R": R} if R != "0" else {})
    print(f"strip walk R={R:>2}: {t * 1e3:7.1f} us  bit-identical: {torch.equal(gr, g_ref)}  "
          f"max |diff| {(gr - g_ref).abs().max().item():.3g}", flush=True)
