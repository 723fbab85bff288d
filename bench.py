#!/usr/bin/env python3
"""Benchmark of the hot path: U-Net 3x256x256 training steps (forward + BCE loss + backward +
gradient all-reduce + Adam) on N MI355X GPUs, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = images/s over all ranks (weak scaling: 32 images per GPU),
inputs resident in HBM before the timed region.  `roofline` prices the dominant kernel (conv_band_kernel,
the MFMA implicit-GEMM 3x3 convolution, forward and input-gradient launches) as algorithmic
FLOPs / event-timed duration against the dense bf16 MFMA peak; `cpu_baseline` is the fp32 CPU oracle
(oracle/ref_cpu.py, validated against the reference) on a bounded sample on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_DENSE = 2.5e15      # MI355X dense bf16 MFMA, FLOP/s
BATCH_PER_GPU = 32
IMG = 256
FLOP_PER_IMG_FWD_BWD = 288.8e9   # 3 x 96.33 GFLOP forward (2*MAC, all convs) - conv1_1 input gradient, at 256x256


def synth_batch(b, c, h, w, seed_img=1234, seed_mask=4321):
    """Synthetic batch of the loaders' value ranges (data_helpers/utils.py:58-62,106): images uniform in [0, 1),
    masks in {0, 1}; fp32 NCHW, on the host."""
    import torch
    gi = torch.Generator().manual_seed(seed_img)
    gm = torch.Generator().manual_seed(seed_mask)
    x = torch.rand(b, c, h, w, generator=gi)
    m = (torch.rand(b, 1, h, w, generator=gm) > 0.5).float()
    return x, m


def launch_ranks(n, argv, script=None):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as a child job (one process per GPU) and relay
    rank 0's JSON line and the job's exit code.  Called before this process has imported torch or touched the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + argv
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    return proc.returncode if lines or proc.returncode else 1


def cpu_baseline(threads):
    """fp32 CPU oracle, fwd + BCE + bwd + Adam, on 4 images of the same 3x256x256 workload."""
    import torch
    import torch.nn as nn
    from oracle import ref_cpu
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    model = ref_cpu.RefUNet(3, 1).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    x, m = ref_cpu.synth_batch(4, 3, IMG, IMG)
    crit = nn.BCEWithLogitsLoss()
    times = []
    for i in range(4):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = crit(model(x), m)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[1]                 # median of the 3 timed steps (first is warm-up)
    return {"value": round(4 / t, 3), "unit": "img/s", "cores": threads, "kind": "port",
            "sample": "4 images 3x256x256 per step (batch 4, 1/8 of the GPU batch), fp32, 1 warm-up + 3 timed steps, median"}


TRAFFIC_FILE = "r03_traffic.json"   # PMC passes of the kernels in this tree (tools/prof_round.sh writes it)
CLDICE_TRAFFIC_FILE = "r03_cldice_traffic.json"   # the same for the clDice loss alone (tools/prof_cldice.sh)
PEAK_HBM = 8.0e12             # MI355X HBM3E, B/s
CLDICE_BYTES_PER_PIXEL = 592   # both skeletons forward + the prediction's backward + the reductions, fp32, each plane once per pass


def lunext_result(args, with_cpu_baseline):
    """BASELINE configs[2]: LUNeXt 3x256x256, batch 32, soft_dice_cldice loss, one GPU.  The model's kernels are
    HBM-bound; `roofline` prices the clDice loss (forward + backward) against the HBM peak."""
    import torch
    import unet_amd
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    model = unet_amd.LUNeXt(3, 1).to(dev).train()
    from unet_amd.dp import LUNeXtTrainer
    # the loop body of training_utils.py:131-170 with model_type "LUNeXt", custom_loss "cl_dice":
    # probs = sigmoid(model(x)).clamp(1e-6, 1 - 1e-6); loss = soft_dice_cldice()(m, probs); backward; Adam
    trainer = LUNeXtTrainer(model, lr=1e-3, criterion="cl_dice", standalone=True)
    crit = unet_amd.soft_dice_cldice()
    x, m = synth_batch(args.batch, 3, args.size, args.size)
    x, m = x.to(dev), m.to(dev)

    def step():
        return trainer.train_step(x, m)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the loss alone, forward + backward with respect to the probabilities
    probs = torch.rand(args.batch, 1, args.size, args.size, device=dev).clamp(1e-6, 1 - 1e-6).requires_grad_(True)
    for _ in range(3):
        torch.autograd.grad(crit(m, probs), probs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        torch.autograd.grad(crit(m, probs), probs)
    e1.record()
    torch.cuda.synchronize()
    t_loss = e0.elapsed_time(e1) * 1e-3 / reps
    pixels = args.batch * args.size * args.size
    achieved = CLDICE_BYTES_PER_PIXEL * pixels / t_loss
    # L2 memory-side bytes of one loss call, from the committed PMC passes of these kernels (a profiler cannot run
    # inside this process); only valid for the default workload size
    traffic = None
    tpath = os.path.join(ROOT, "profiles", CLDICE_TRAFFIC_FILE)
    if os.path.exists(tpath) and args.size == IMG and args.batch == BATCH_PER_GPU:
        with open(tpath) as fh:
            traffic = json.load(fh)["loss_kernels_traffic_bytes_per_call"]
    out = {"metric": "train imgs/sec, LUNeXt 256^2 + clDice soft-skeleton loss, 1 MI355X", "value": round(args.batch * args.steps / dt, 2),
           "unit": "img/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "bf16 storage / fp32 loss", "data": "synthetic",
           "config": {"workload": f"LUNeXt 3x{args.size}x{args.size}, batch {args.batch}, soft_dice_cldice, fwd+bwd+Adam "
                                  "(BASELINE configs[2])", "loss_last_step": round(loss.item(), 6)},
           "roofline": {"bound": "hbm", "kernel": "clDice loss (soft-skeleton forward of both inputs + backward)",
                        "achieved": round(achieved / 1e9, 1), "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                        "frac": round(achieved / PEAK_HBM, 4), "traffic": traffic,
                        "traffic_unit": "bytes per loss call (L2 memory-side reads + writes, Infinity-Cache hits included: "
                                        "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/" + CLDICE_TRAFFIC_FILE + ")",
                        "ms_per_call": round(t_loss * 1e3, 3), "bytes_per_pixel": CLDICE_BYTES_PER_PIXEL}}
    if with_cpu_baseline:
        from oracle import ref_cpu
        threads = int(os.environ.get("OMP_NUM_THREADS", min(16, os.cpu_count() or 1)))
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        ref = ref_cpu.RefLUNeXt(3, 1).train()
        ropt = torch.optim.Adam(ref.parameters(), lr=1e-3)
        rc = ref_cpu.RefSoftDiceClDice()
        xs, ms = ref_cpu.synth_batch(8, 3, args.size, args.size)
        times = []
        for _ in range(3):
            t1 = time.perf_counter()
            ropt.zero_grad()
            rc(ms, torch.sigmoid(ref(xs)).clamp(1e-6, 1 - 1e-6)).backward()
            ropt.step()
            times.append(time.perf_counter() - t1)
        out["cpu_baseline"] = {"value": round(8 / sorted(times[1:])[0], 3), "unit": "img/s", "cores": threads, "kind": "port",
                               "sample": "8 images 3x256x256 per step (batch 8, 1/4 of the GPU batch), fp32, "
                                         "1 warm-up + 2 timed steps, best"}
    return out


def config3_per_gpu(dev, steps=8, warmup=3):
    """The per-GPU workload of BASELINE configs[3] (UNet 3x512x512, batch 16 per GPU, data-parallel on 2/4/8 GPUs) on
    this one GPU: forward + BCE + backward + Adam, no collective (a one-rank job has none)."""
    import torch
    import unet_amd
    from unet_amd.dp import DataParallelTrainer
    size, batch = 512, 16
    torch.manual_seed(0)
    model = unet_amd.UNet(3, 1).to(dev)
    trainer = DataParallelTrainer(model, lr=1e-3, standalone=True)
    x, m = synth_batch(batch, 3, size, size)
    x, m = x.to(dev), m.to(dev)
    for _ in range(warmup):
        trainer.train_step(x, m)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = trainer.train_step(x, m)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    value = batch * steps / dt
    return {"metric": "train imgs/sec on one GPU, UNet 512^2 bf16 fwd+bwd+Adam, batch 16 (the per-GPU workload of BASELINE "
                      "configs[3])", "value": round(value, 2), "unit": "img/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 3), "dtype": "bf16", "data": "synthetic",
            "step_mfma_frac": round(value * FLOP_PER_IMG_FWD_BWD * (size / IMG) ** 2 / PEAK_BF16_DENSE, 4),
            "loss_last_step": round(loss.item(), 6)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="images per GPU")
    ap.add_argument("--size", type=int, default=IMG)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-workload", action="store_true",
                    help="leave BASELINE configs[2] (LUNeXt + clDice) out of the line a one-GPU run prints")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank path where RCCL cannot run, e.g. several ranks sharing one GPU)")
    ap.add_argument("--workload", choices=["unet", "lunext"], default="unet",
                    help="unet: BASELINE configs[1] (the contract line); lunext: configs[2], LUNeXt + clDice loss")
    args = ap.parse_args()
    if args.workload == "lunext":
        print(json.dumps(lunext_result(args, not args.no_cpu_baseline)), flush=True)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    grouped = world > 1 or os.environ.get("UNET_AMD_FORCE_COLLECTIVES") == "1"
    stdout_fd = None
    if grouped:
        # RCCL's streams take hardware queues too: with the default of 4 per process the step's two streams can end up
        # on one queue (measured with a one-rank group: 14.9 ms per step against 14.1 ms with 8 queues).  Must be set
        # before the HIP runtime starts.
        if os.environ.get("UNET_AMD_BENCH_HWQ8", "1") == "1" and int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 8:
            os.environ["GPU_MAX_HW_QUEUES"] = "8"
        # RCCL prints a version banner on stdout when its communicator comes up; the contract is ONE JSON line there
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import unet_amd
    from unet_amd.dp import DataParallelTrainer, KernelTimer

    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.backend != "nccl":
        local = local % torch.cuda.device_count()          # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if grouped:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    torch.manual_seed(0)
    model = unet_amd.UNet(3, 1).to(dev)
    trainer = DataParallelTrainer(model, lr=1e-3)
    # synthetic data of the ISIC2018 shape, a different shard per rank, resident in HBM
    x, m = synth_batch(args.batch, 3, args.size, args.size, seed_img=1234 + rank, seed_mask=4321 + rank)
    x, m = x.to(dev), m.to(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.train_step(x, m)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = trainer.train_step(x, m)
    barrier()
    dt = time.perf_counter() - t0
    # After the timed region: per-kernel event timing.  It brackets every MFMA launch with two event records (~11 us of
    # stream bubbles per launch, ~100 launches per step), so the steps that carry it are not the ones `value` is from.
    timer = KernelTimer()
    instrumented = 5
    trainer.timer = timer
    for _ in range(instrumented):
        trainer.train_step(x, m)
    torch.cuda.synchronize()
    trainer.timer = None
    # Then the same kernels with everything on one stream.  In the step the weight gradients run
    # on a second stream beside the input-gradient and BatchNorm kernels, which shortens the step but stretches every
    # overlapped launch; these extra steps give each kernel's rate when it has the chip to itself.
    serial = KernelTimer()
    prev = os.environ.get("UNET_AMD_WGRAD_STREAM")
    os.environ["UNET_AMD_WGRAD_STREAM"] = "0"
    trainer.timer = serial
    for _ in range(3):
        trainer.train_step(x, m)
    torch.cuda.synchronize()
    trainer.timer = None
    if prev is None:
        os.environ.pop("UNET_AMD_WGRAD_STREAM")
    else:
        os.environ["UNET_AMD_WGRAD_STREAM"] = prev
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    if stdout_fd is not None:
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        os.close(stdout_fd)
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.batch * args.steps / dt
        ks = timer.summary()
        ss = serial.summary()
        g = ks["conv_band"]
        achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12            # TFLOP/s over all conv_band_kernel launches
        scale = (args.size / IMG) ** 2
        config_name = {(IMG, BATCH_PER_GPU): "BASELINE configs[1]", (512, 16): "BASELINE configs[3]"}.get(
            (args.size, args.batch), "not a BASELINE configuration")
        # HBM-side bytes per launch of the dominant kernel: measured separately with rocprofv3 PMC passes (the
        # profiler cannot run inside this process); only valid for the default workload size
        traffic = None
        tpath = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
        if os.path.exists(tpath) and args.size == IMG and args.batch == BATCH_PER_GPU:
            with open(tpath) as fh:
                traffic = json.load(fh)["kernels"]["conv_band_kernel"]["traffic_bytes_per_launch"]
        out = {
            "metric": "train imgs/sec, UNet 256^2 bf16 fwd+bwd, 1/2/4/8 MI355X; % MFMA peak",
            "value": round(value, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"UNet 3x{args.size}x{args.size}, batch {args.batch}/GPU, BCEWithLogits, "
                                   "fwd+bwd+grad all-reduce+Adam (" + config_name + ")",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}",
                       "loss_last_step": round(loss.item(), 6)},
            "step_mfma_frac": round(value / world * FLOP_PER_IMG_FWD_BWD * scale / PEAK_BF16_DENSE, 4),
            "roofline": {"bound": "mfma", "kernel": "conv_band_kernel (all 3x3 convolutions: 17 forward + 17 input-gradient launches per step)",
                         "achieved": round(achieved, 1), "peak": PEAK_BF16_DENSE / 1e12, "unit": "TFLOP/s",
                         "frac": round(achieved * 1e12 / PEAK_BF16_DENSE, 4), "traffic": traffic,
                         "traffic_unit": "bytes per launch (L2 memory-side reads + writes: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                         "passes of this command, profiles/" + TRAFFIC_FILE + ")",
                         "launches_per_step": g["launches"] // instrumented, "event_timed_steps": instrumented,
                         "avg_launch_us": round(g["ms"] * 1e3 / g["launches"], 1),
                         "achieved_one_stream": round(ss["conv_band"]["flops"] / (ss["conv_band"]["ms"] * 1e-3) / 1e12, 1),
                         "note": "achieved/frac: launches as they run in the step (input-gradient launches overlap with "
                                 "weight-gradient kernels on a second stream); achieved_one_stream: same launches alone. "
                                 "A launch is conv_band_kernel plus, where its tiles are cut evenly over the CUs (6 forward "
                                 "layers of the 34 launches), the conv_band_fixup_kernel that finishes the cut tiles: "
                                 "avg_launch_us compares with rocprofv3's conv_band_kernel average + 6/34 of its "
                                 "conv_band_fixup_kernel average",
                         "flop_per_launch": round(g["flops"] / g["launches"] / 1e9, 2)},
            "kernels": {k: {"TFLOP/s": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1),
                            "ms_per_step": round(v["ms"] / instrumented, 3),
                            "launches_per_step": v["launches"] // instrumented} for k, v in ks.items()},
        }
        if not args.no_cpu_baseline and world == 1:         # timed on the host cores of a one-GPU run only
            out["cpu_baseline"] = cpu_baseline(min(16, os.cpu_count() or 1) if "OMP_NUM_THREADS" not in os.environ
                                               else int(os.environ["OMP_NUM_THREADS"]))
        if world == 1 and not args.no_second_workload and args.size == IMG and args.batch == BATCH_PER_GPU:
            # BASELINE configs[2] rides in the same line: LUNeXt + clDice loss on the same GPU, measured right here
            del trainer, model
            torch.cuda.empty_cache()
            out["config2_lunext_cldice"] = lunext_result(args, not args.no_cpu_baseline)
            torch.cuda.empty_cache()
            out["config3_per_gpu"] = config3_per_gpu(dev)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
