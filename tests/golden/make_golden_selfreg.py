#!/usr/bin/env python3
"""Write selfreg_cases.npz by IMPORTING THE REFERENCE's SelfRegLoss (build container only):

    python tests/golden/make_golden_selfreg.py

The reference seeds its channel subsets from Python's string hash (models/losses.py:126), so the script restarts
itself under PYTHONHASHSEED=0 and stores the seeds that interpreter produced next to the results.  Each case stores
the feature list (fp16: every value is k/64, exact in fp16 and in bf16), logits, targets, and the reference's
LSRC, LIFD, Dice, loss and d(loss)/d(logits).
"""
import os
import subprocess
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

UNET_CHANNELS = [64, 64, 128, 128, 256, 256, 512, 512, 512, 512, 256, 256, 128, 128, 64]
UNET_SCALE = [1, 1, 2, 2, 4, 4, 8, 8, 8, 8, 4, 4, 2, 2, 1]


def relu_like(g, *shape):
    return torch.clamp(torch.randint(-128, 256, shape, generator=g), min=0).float() / 64.0


def cases():
    g = torch.Generator().manual_seed(2024)
    feats = [relu_like(g, 2, c, 16 // s, 16 // s) for c, s in zip(UNET_CHANNELS, UNET_SCALE)]
    yield "unet_like", feats, 16
    feats = [relu_like(g, 2, c, 16 // s, 16 // s) for c, s in zip(UNET_CHANNELS, UNET_SCALE)]
    for f in feats[:-1]:
        f[:, 1::5] = 0.0                                  # dead channels: importance exactly 0, ties among them
    feats[3][:, 7] = feats[3][:, 9]                       # two identical live channels: an exact tie
    yield "dead_channels", feats, 16
    feats = [relu_like(g, 3, c, s, s) for c, s in ((8, 8), (16, 8), (40, 4), (24, 2), (32, 8))]
    yield "final_wider", feats, 8                         # the pooled side is the one cut down (seed + 1)
    yield "single_feature", [relu_like(g, 2, 8, 4, 4)], 4
    yield "two_channels", [relu_like(g, 2, 2, 4, 4), relu_like(g, 2, 1, 4, 4), relu_like(g, 2, 2, 4, 4)], 4


def main():
    if os.environ.get("PYTHONHASHSEED") != "0":
        env = dict(os.environ, PYTHONHASHSEED="0")
        sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)], env=env))
    sys.path.insert(0, REF)
    from models.losses import SelfRegLoss                 # the reference
    crit = SelfRegLoss()
    out = {}
    names = []
    g = torch.Generator().manual_seed(99)
    for name, feats, size in cases():
        b = feats[0].shape[0]
        logits = (torch.randn(b, 1, size, size, generator=g) * 2).requires_grad_(True)
        targets = (torch.rand(b, 1, size, size, generator=g) > 0.6).float()
        seeds = [hash(str(idx) + "feature") % 10000 for idx in range(len(feats))]     # losses.py:126
        with torch.no_grad():
            lsrc = float(crit.LSRC(feats, feats[-1]))
            lifd = float(crit.LIFD(feats))
            dice = float(crit.dice_loss(logits, targets))
            ce = float(crit.ce_loss_fn(logits, targets))
        loss = crit(logits, targets, feats)
        loss.backward()
        names.append(name)
        out[f"{name}.n"] = np.array(len(feats))
        for i, f in enumerate(feats):
            out[f"{name}.f{i}"] = f.numpy().astype(np.float16)
        out[f"{name}.logits"] = logits.detach().numpy()
        out[f"{name}.targets"] = targets.numpy()
        out[f"{name}.seeds"] = np.array(seeds, dtype=np.int64)
        out[f"{name}.scalars"] = np.array([lsrc, lifd, dice, ce, float(loss)], dtype=np.float64)
        out[f"{name}.dlogits"] = logits.grad.numpy()
        print(f"{name}: lsrc {lsrc:.6f} lifd {lifd:.6f} dice {dice:.6f} ce {ce:.3g} loss {float(loss):.6f}")
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "selfreg_cases.npz"), **out)
    print("wrote selfreg_cases.npz")


if __name__ == "__main__":
    main()
