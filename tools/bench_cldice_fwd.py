"""clDice loss forward + backward at batch 32, 256 x 256: the one-launch strip walk over all eleven levels against one
launch per level (UNET_AMD_SKEL_FWD=0): loss and gradient bits compared, and a sweep of UNET_AMD_SKEL_FWD_R."""
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
sk = unet_amd.SoftSkeletonize(num_iter=10, check_range=False)


def run(env):
    for k in ("UNET_AMD_SKEL_FWD", "UNET_AMD_SKEL_FWD_R"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(3):
        loss = crit(m, p)
        (grad,) = torch.autograd.grad(loss, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        loss = crit(m, p)
        (grad,) = torch.autograd.grad(loss, p)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20, loss.detach().clone(), grad.clone(), sk(p.detach()).clone()


t_ref, l_ref, g_ref, s_ref = run({"UNET_AMD_SKEL_FWD": "0"})
print(f"one launch per level: {t_ref * 1e3:7.1f} us per loss call", flush=True)
for R in ("0", "8", "16", "24", "32", "48", "64", "128"):
    t, l, gr, s = run({"UNET_AMD_SKEL_FWD_R": R} if R != "0" else {})
    print(f"strip walk R={R:>3}: {t * 1e3:7.1f} us  loss equal {torch.equal(l, l_ref)}  grad equal {torch.equal(gr, g_ref)}  "
          f"skeleton equal {torch.equal(s, s_ref)}  max |dgrad| {(gr - g_ref).abs().max().item():.3g}", flush=True)
