import sys, os, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_lunext_gpu as T
fx = np.load('tests/golden/lunext_s0_b2_32.npz', allow_pickle=False)
x, m = torch.from_numpy(fx["x"]), torch.from_numpy(fx["mask"])
o = T.three_way(2, 3, 32, 32, "cldice", seed=0, x=x, m=m)
ref_grads = {k[5:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("grad/")}
tot = max(g.norm().item() for g in o["ref"]["grads"].values() if g is not None)
for n, g in o["ref"]["grads"].items():
    if g is None or g.numel() < 64: continue
    gh, ge = o["hip"]["grads"][n], o["emu"]["grads"][n]
    ch, ce = T.cosine(gh, g), T.cosine(ge, g)
    if ch < 0.95 or ce < 0.95:
        fxs = f" fixture-cos {T.cosine(gh, ref_grads[n]):.4f}" if n in ref_grads else ""
        print(f"{n:32s} numel {g.numel():7d} rel-norm {g.norm().item()/tot:.2e} cos hip {ch:.4f} emu {ce:.4f}{fxs}")
