"""Evaluation path throughput (eval-mode U-Net forward + on-device confusion counts), the reference's inference
shapes: batch 1 at 1024^2 (test_helper/utils.py:116-125) and batch 8 at 256^2 (k_fold.py:281-290).
python tools/infer_bench.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_amd
from unet_amd import trainer

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = unet_amd.create_unet(3, 1).to(dev).eval()
out = {}
for name, B, S in (("b1_1024", 1, 1024), ("b8_256", 8, 256), ("b32_256", 32, 256), ("b1_256", 1, 256)):
    x = torch.rand(B, 3, S, S, device=dev)
    m = (torch.rand(B, 1, S, S, device=dev) > 0.5).float()
    counts = torch.zeros(4, dtype=torch.int64, device=dev)
    def step():
        with torch.no_grad():
            trainer.confusion_counts(model(x), m, counts)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    flops = 96.33e9 * (S / 256) ** 2 * B
    out[name] = {"ms_per_batch": round(dt * 1e3, 3), "img_per_s": round(B / dt, 1), "TFLOP/s": round(flops / dt / 1e12, 1)}
print(json.dumps(out))
