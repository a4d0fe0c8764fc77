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

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3  # dense fp32 matrix peak (v_mfma_f32_16x16x4_f32, same guide)


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
    ap.add_argument("--net", default="unet_uaps", help="unet_uaps (BASELINE.json configs[1], the reported metric) or resnet50_uaps (configs[4] shape study)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decoder-streams", action="store_true",
                    help="run the auxiliary decoders on their own HIP streams (UAPS_DECODER_STREAMS=1): higher images/s, but launches of "
                         "different decoders overlap, so the per-launch roofline figures no longer describe one kernel")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import uaps_amd
    from uaps_amd import losses

    if args.decoder_streams:
        import uaps_amd.unet as _unet
        _unet._DECODER_STREAMS = True
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torch.distributed.run (RANK set) the process group is created even for one rank, so the barrier /
    # max-over-ranks plumbing below is the same code at every N
    distributed = world > 1 or "RANK" in os.environ
    saved_stdout = None
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints its version banner (NCCL_DEBUG=VERSION on the GPU boxes) on stdout when the communicator is created:
        # point fd 1 at stderr until the first collective has run, so that stdout carries the one JSON line only
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        dist.init_process_group("nccl", device_id=dev)

    torch.manual_seed(1337)
    D, C, H, W, b = args.aux + 1, args.classes, args.size, args.size, args.batch
    model = uaps_amd.net_factory(args.net, 3, C, n_aux=args.aux)
    uaps_amd.dist.broadcast_model(model)
    if saved_stdout is not None:
        dist.barrier()
        torch.cuda.synchronize()
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    trainer = uaps_amd.UAPSTrainer(model, seed=1337)
    data = uaps_amd.data.SyntheticBatches(b, 3, C, H, W, n_batches=2, seed=1337 + rank, device=dev)

    from uaps_amd import conv

    def summarize(ev):
        """{kernel: [(start, end, work)]} -> {kernel: {calls, avg_us, total_us, work}} (needs a prior synchronize)."""
        out = {}
        for name, recs in ev.items():
            us = [s.elapsed_time(e) * 1e3 for s, e, *_ in recs]
            out[name] = {"calls": len(us), "avg_us": float(np.mean(us)), "total_us": float(np.sum(us)),
                         "work": float(sum(r[2] for r in recs if len(r) > 2))}
        return out

    # Warm-up.  The first warm-up step also times EVERY hand-written conv/loss launch once with HIP events to find
    # the kernel instantiation that dominates the step; the timed region then brackets only that kernel's launches
    # (and the four loss kernels), so the event records do not perturb the measured step.
    discover = None
    for i in range(args.warmup):
        if i == args.warmup - 1:
            torch.cuda.synchronize()
            conv.KERNEL_EVENTS, conv.EVENT_FILTER, losses.KERNEL_EVENTS = {}, None, {}
        trainer.train_step(*data.next())
        if i == args.warmup - 1:
            torch.cuda.synchronize()
            discover = summarize(conv.KERNEL_EVENTS)
            for k, pairs in losses.KERNEL_EVENTS.items():
                us = [s.elapsed_time(e) * 1e3 for s, e in pairs]
                discover[k] = {"calls": len(us), "avg_us": float(np.mean(us)), "total_us": float(np.sum(us)), "work": 0.0}
            conv.KERNEL_EVENTS = losses.KERNEL_EVENTS = None
    dominant = max(discover, key=lambda k: discover[k]["total_us"]) if discover else None
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    losses.KERNEL_EVENTS = {}
    conv.KERNEL_EVENTS, conv.EVENT_FILTER = {}, ({dominant} if dominant else None)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # per-step spread (median, p10 / p90)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        trainer.train_step(*data.next())
        marks[i + 1].record()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ev, losses.KERNEL_EVENTS = losses.KERNEL_EVENTS, None
    cev, conv.KERNEL_EVENTS, conv.EVENT_FILTER = conv.KERNEL_EVENTS, None, None
    if distributed:
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
            kern[name] = {"calls_per_step": len(ms) / args.steps, "avg_us": round(avg_s * 1e6, 2), "GBps": round(by / avg_s / 1e9, 1)}
        timed = summarize(cev)
        if dominant is None and timed:
            dominant = max(timed, key=lambda k: timed[k]["total_us"])
        for name, v in timed.items():
            kern[name] = {"calls_per_step": v["calls"] / args.steps, "avg_us": round(v["avg_us"], 2),
                          "TFLOPs": round(v["work"] / v["total_us"] / 1e6, 2)}
        for name, v in (discover or {}).items():          # one warm-up step's sample of the other instantiations
            if name not in kern:
                kern[name] = {"calls_per_step": v["calls"], "avg_us": round(v["avg_us"], 2), "sample": "1 warm-up step"}
                if v["work"]:
                    kern[name]["TFLOPs"] = round(v["work"] / v["total_us"] / 1e6, 2)
        traffic = None
        try:   # HBM bytes per launch from the rocprofv3 PMC passes (profiles/README.md), when they have been collected
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                traffic = json.load(f).get(dominant)
        except (OSError, ValueError):
            pass
        if dominant in timed:
            v = timed[dominant]
            ach = v["work"] / v["total_us"] / 1e6                     # TFLOP/s, algorithmic 2*B*H*W*Cin*Cout*k*k per launch
            roof = {"kernel": dominant, "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_F32_PEAK_TF, 4), "traffic": traffic, "avg_us": round(v["avg_us"], 2),
                    "launches_per_step": v["calls"] / args.steps,
                    "note": "kernel instantiation with the largest share of the step (found in the last warm-up step); algorithmic "
                            "flops = 2*B*H*W*Cin*Cout*k*k per launch, summed over its launches / summed HIP-event time; "
                            "traffic = HBM bytes per launch from profiles/pmc_traffic.json (rocprofv3 --pmc passes); peak is the "
                            "nominal 2.4 GHz figure -- under this kernel the shader clock measured 2.03-2.14 GHz (DESIGN.md section 5)"}
            if args.decoder_streams or os.environ.get("UAPS_DECODER_STREAMS", "0") != "0":
                roof["note"] += ("; DECODER STREAMS ON: launches of the four decoders overlap on the GPU, a launch's event-to-event time "
                                 "includes other kernels' share of the CUs, so achieved / frac are NOT standalone-kernel figures in this run")
        else:
            dom = max(ev, key=lambda k: kern[k]["avg_us"])
            roof = {"kernel": dom, "bound": "hbm", "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(kern[dom]["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic}
        step_ms = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)])
        res = {"metric": "training images/sec (labeled+unlabeled) NEU-Seg 256x256 K=3", "value": round(2 * b * world * args.steps / dt, 2),
               "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"UAPS K={args.aux} decoders, NEU-Seg-shaped {H}x{W} {C}-class, batch {b}+{b} per GPU (BASELINE.json configs[1])",
                          "heads": D, "per_gpu_batch": f"{b} labelled + {b} unlabelled", "parallelism": f"dp{world}", "final_loss": round(last_loss, 5)},
               "roofline": roof, "kernels": kern}
        res["step_ms"] = {"p10": round(float(np.percentile(step_ms, 10)), 3), "p50": round(float(np.percentile(step_ms, 50)), 3),
                          "p90": round(float(np.percentile(step_ms, 90)), 3), "note": "GPU time between per-step HIP events on rank 0"}
        if args.net != "unet_uaps":
            res["config"]["workload"] = f"{args.net} K={args.aux}, {H}x{W} {C}-class, batch {b}+{b} per GPU (not the BASELINE metric config)"
        if world == 1 and not args.no_cpu_baseline and args.net == "unet_uaps":
            res["cpu_baseline"] = cpu_baseline(4, H, W)
        print(json.dumps(res), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
