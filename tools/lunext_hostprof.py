"""host-side cost of one LUNeXt forward + backward (the step is dispatch-bound): cProfile of the launch sequence itself,
driven without autograd so that the backward pass runs on the profiled thread"""
import sys, os, time, cProfile, pstats, io, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd.lunext import LUNeXtRun
from oracle import ref_cpu
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = unet_amd.LUNeXt(3, 1).to(dev).train()
x, m = ref_cpu.synth_batch(32, 3, 256, 256); x, m = x.to(dev), m.to(dev)
dl = torch.randn(32, 1, 256, 256, device=dev)
def step():
    run = LUNeXtRun(model, x, training=True)
    run.forward(); g = run.backward(dl); run.release(); return g
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue {(t1-t0)/20*1e3:.2f} ms/step, total {(t2-t0)/20*1e3:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(22); print(s.getvalue()[:5000])
