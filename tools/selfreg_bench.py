"""Cost of the self-regularisation loss at the bench size (U-Net, batch 32, 256 x 256): the auxiliaries alone, and
the training step with it against the BCE step.  python tools/selfreg_bench.py"""
import json
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd.dp import DataParallelTrainer
from unet_amd.self_reg import SelfRegLoss
import torch.nn as nn

dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, S = 32, 256
x = torch.rand(B, 3, S, S, device=dev)
m = (torch.rand(B, 1, S, S, device=dev) > 0.5).float()
model = unet_amd.create_unet(3, 1).to(dev)
crit = SelfRegLoss(seed_of=lambda i: 7 * i)


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


out = {}
model.train()
with torch.no_grad():
    model.eval()
    model(x)
    feats = model.intermediate_features
    out["aux_ms"] = timed(lambda: crit.auxiliaries(feats), 20)
    acts = feats.acts()
    fbytes = sum(a.B * a.H * a.W * a.C * 2 for a in acts[:-1])
    out["feature_bytes"] = fbytes
    out["aux_read_twice_TBps"] = 2 * fbytes / (out["aux_ms"] * 1e-3) / 1e12
for name, c in (("bce", nn.BCEWithLogitsLoss()),
                ("self_reg", lambda lg, mk: crit(lg, mk, model.intermediate_features))):
    tr = DataParallelTrainer(model, lr=1e-3, criterion=c, standalone=True)
    out[f"step_{name}_ms"] = timed(lambda: tr.train_step(x, m), 20)
print(json.dumps(out))
