#!/usr/bin/env python3
"""Benchmark of the UAPS training step (BASELINE.json metric: training images/sec, labelled +
unlabelled, NEU-Seg-shaped 256x256 4-class, K=3 auxiliary decoders = 4 heads).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A step = UAPSTrainer.train_step on one synthetic batch pair resident in HBM (16 labelled + 16
unlabelled images per GPU, config[1] of BASELINE.json): two forwards, fused HIP loss block, backward,
RCCL gradient average (N>1), Adam.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def loss_kernel_bytes(name, D, C, npix):
    """Algorithmic HBM bytes per launch (SURVEY.md section 8d, fp32 logits, int64 labels)."""
    per_px = {"uaps_unsup_fwd": 4 * D * C + 8,            # var maps not stored in the training step
              "uaps_unsup_bwd": 8 * D * C + 8,
              "uaps_sup_fwd": 4 * D * C + 8,
              "uaps_sup_bwd": 8 * D * C + 8}[name]
    return per_px * npix


def cpu_baseline(batch, H, W, steps=3):
    """The oracle's whole step (oracle/uaps_oracle.py CpuStep: same net, unfused loss, autograd, Adam)
    timed on this box's host cores on a bounded sample.  Baseline only, never the product path."""
    import numpy as np
    import torch
    import uaps_amd
    from oracle import uaps_oracle as O
    torch.manual_seed(1337); np.random.seed(1337)
    net = uaps_amd.UNet_UAPS(3, 4)
    st = O.CpuStep(net.state_dict())
    data = uaps_amd.data.SyntheticBatches(batch, H=H, W=W, n_batches=1, device="cpu")
    xl, yl, xu = data.next()
    st.step(xl, yl, xu)                                    # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        st.step(xl, yl, xu)
    dt = time.perf_counter() - t0
    return {"value": round(2 * batch * steps / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} steps of {batch}+{batch} images {H}x{W} D=4 C=4 (oracle CpuStep, torch CPU fp32, 1 warm-up)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="labelled images per GPU (+ as many unlabelled)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--classes", type=int, default=4)
    ap.add_argument("--aux", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import uaps_amd
    from uaps_amd import losses

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    torch.manual_seed(1337)
    D, C, H, W, b = args.aux + 1, args.classes, args.size, args.size, args.batch
    model = uaps_amd.net_factory("unet_uaps", 3, C, n_aux=args.aux)
    uaps_amd.dist.broadcast_model(model)
    trainer = uaps_amd.UAPSTrainer(model, seed=1337)
    data = uaps_amd.data.SyntheticBatches(b, 3, C, H, W, n_batches=2, seed=1337 + rank, device=dev)

    for _ in range(args.warmup):
        trainer.train_step(*data.next())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    losses.KERNEL_EVENTS = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.train_step(*data.next())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ev, losses.KERNEL_EVENTS = losses.KERNEL_EVENTS, None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    last_loss = float(trainer.last["loss"])

    if rank == 0:
        npix = b * H * W
        kern = {}
        for name, pairs in ev.items():
            ms = [s.elapsed_time(e) for s, e in pairs]
            avg_s = float(np.mean(ms)) * 1e-3
            by = loss_kernel_bytes(name, D, C, npix)
            kern[name] = {"avg_us": avg_s * 1e6, "GBps": by / avg_s / 1e9, "bytes": by}
        dom = max(kern, key=lambda k: kern[k]["avg_us"])
        roof = {"kernel": dom, "bound": "hbm", "achieved": round(kern[dom]["GBps"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(kern[dom]["GBps"] / HBM_PEAK_GBS, 4), "traffic": None,
                "note": "dominant hand-written kernel; the conv stack still runs through MIOpen this round and dominates the step"}
        res = {"metric": "training images/sec (labeled+unlabeled) NEU-Seg 256x256 K=3", "value": round(2 * b * world * args.steps / dt, 2),
               "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"UAPS K={args.aux} decoders, NEU-Seg-shaped {H}x{W} {C}-class, batch {b}+{b} per GPU (BASELINE.json configs[1])",
                          "heads": D, "per_gpu_batch": f"{b} labelled + {b} unlabelled", "parallelism": f"dp{world}", "final_loss": round(last_loss, 5)},
               "roofline": roof,
               "kernels": {k: {"avg_us": round(v["avg_us"], 2), "GBps": round(v["GBps"], 1)} for k, v in kern.items()}}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(4, H, W)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
