"""2-rank rehearsal of the data-parallel step on ONE GPU (gloo backend, device tensors)."""
import os, sys, time, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(45, exit=True)
import torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
def say(*a): print(f"[rank {rank}]", *a, flush=True)
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("gloo"); say("pg up")
import unet_amd
from unet_amd.dp import DataParallelTrainer
from oracle import ref_cpu
torch.manual_seed(rank)               # different init per rank: the broadcast must make them equal
model = unet_amd.UNet(3, 1).to(dev)
tr = DataParallelTrainer(model, lr=1e-3); say("trainer up, buckets", len(tr.buckets.buckets))
x, m = ref_cpu.synth_batch(4, 3, 64, 64, seed_img=1234 + rank, seed_mask=4321 + rank)
x, m = x.to(dev), m.to(dev)
for i in range(3):
    loss = tr.train_step(x, m); torch.cuda.synchronize(); say("step", i, "loss", loss.item())
# parameters must be identical on both ranks after the steps
cs = torch.tensor([sum(p.double().sum().item() for p in model.parameters())], dtype=torch.float64)
lst = [torch.zeros_like(cs) for _ in range(world)]
dist.all_gather(lst, cs); say("param checksums", [v.item() for v in lst])
assert abs(lst[0].item() - lst[1].item()) < 1e-6 * abs(lst[0].item())
dist.barrier(); dist.destroy_process_group(); say("done")
