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
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 matrix peak (same guide); the split kernels issue 6 bf16 MFMA flops per fp32 flop


def mfma_peak_for(kernel_name):
    """Peak in ALGORITHMIC (fp32) TFLOP/s of the matrix pipe a conv kernel runs on: the fp16-split kernels (conv_h*) evaluate every
    fp32 multiply as three fp16 partial products, the bf16-split kernels (conv_s*) as six bf16 ones: their ceilings are the
    dense 16-bit matrix peak / 3 and / 6."""
    if kernel_name.startswith("conv_h") or kernel_name.startswith("conv_g1h") or kernel_name.startswith("conv_gw1h"):
        return MFMA_BF16_PEAK_TF / 3.0
    if (kernel_name.startswith("conv_s") and not kernel_name.startswith("conv_small")) or kernel_name.startswith("conv_g1s") or kernel_name.startswith("conv_gw1s"):
        return MFMA_BF16_PEAK_TF / 6.0
    return MFMA_F32_PEAK_TF


def loss_kernel_bytes(name, D, C, npix):
    """Algorithmic HBM bytes per launch (SURVEY.md section 8d, fp32 logits, int64 labels)."""
    per_px = {"uaps_unsup_fwd": 4 * D * C + 8,            # var maps not stored in the training step
              "uaps_unsup_bwd": 8 * D * C + 8,
              "uaps_sup_fwd": 4 * D * C + 8,
              "uaps_sup_bwd": 8 * D * C + 8,
              "uaps_pair_fwd": 2 * (4 * D * C + 8),        # both branches in one launch (pair_fwd_kernel; the one-block finalize is not in the time)
              "uaps_pair_bwd": 2 * (8 * D * C + 8)}[name]
    return per_px * npix


def host_cpu_info():
    """What the CPU baseline ran on: lscpu's socket / core / thread counts and the cores this process may use."""
    import subprocess
    info = {}
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        want = {"CPU(s)": "cpus", "Socket(s)": "sockets", "Core(s) per socket": "cores_per_socket", "Thread(s) per core": "threads_per_core",
                "Model name": "model"}
        for line in txt.splitlines():
            k, _, v = line.partition(":")
            if k.strip() in want and want[k.strip()] not in info:
                v = v.strip()
                info[want[k.strip()]] = int(v) if v.isdigit() else v
    except Exception as e:                                  # lscpu missing: report what Python knows
        info["lscpu_error"] = str(e)
    info["usable_cpus"] = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    try:                                                    # a cgroup CPU quota caps the cores a container really gets
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            info["cgroup_quota_cpus"] = round(int(quota) / int(period), 2)
    except Exception:
        pass
    return info


def cpu_baseline(batch, H, W, budget_s=25.0):
    """The oracle's whole step (oracle/uaps_oracle.py CpuStep: same net, unfused loss, autograd, Adam) timed on this box's
    host cores at the metric's own batch, with torch.set_num_threads at (a) the physical cores of one socket and (b) every
    usable core.  A bounded sample: one warm-up step, then steps until two are done or `budget_s`
    seconds have passed, per setting.  Baseline only, never the product path."""
    import numpy as np
    import torch
    import uaps_amd
    from oracle import uaps_oracle as O
    info = host_cpu_info()
    usable = int(info.get("cgroup_quota_cpus") or info["usable_cpus"])
    usable = max(1, min(usable, info["usable_cpus"]))
    one_socket = int(info.get("cores_per_socket") or usable)
    settings = sorted({max(1, min(one_socket, usable)), usable})      # one socket's cores and all usable cores: ONE setting when the
    # container's CPU share is smaller than a socket (the GPU boxes give 16 CPUs of a 2 x 64-core host)
    torch.manual_seed(1337); np.random.seed(1337)
    net = uaps_amd.UNet_UAPS(3, 4)
    data = uaps_amd.data.SyntheticBatches(batch, H=H, W=W, n_batches=1, device="cpu")
    xl, yl, xu = data.next()
    runs = []
    prev = torch.get_num_threads()
    for n in settings:
        torch.set_num_threads(n)
        st = O.CpuStep(net.state_dict())
        t0 = time.perf_counter()
        st.step(xl, yl, xu)                                # warm-up
        warm = time.perf_counter() - t0
        done, t0 = 0, time.perf_counter()
        while done < 3 and (done == 0 or time.perf_counter() - t0 < budget_s):
            st.step(xl, yl, xu)
            done += 1
        dt = time.perf_counter() - t0
        runs.append({"threads": n, "images_per_s": round(2 * batch * done / dt, 3), "s_per_step": round(dt / done, 2),
                     "steps": done, "warmup_step_s": round(warm, 2)})
    torch.set_num_threads(prev)
    best = max(runs, key=lambda r: r["images_per_s"])
    return {"value": best["images_per_s"], "unit": "images/s", "cores": best["threads"], "kind": "port",
            "sample": f"{best['steps']} step(s) of {batch}+{batch} images {H}x{W} D=4 C=4 after 1 warm-up step (oracle CpuStep: the "
                      "reference-structured unfused PyTorch-CPU fp32 step), best of the thread settings in `runs` "
                      f"({len(settings)} setting(s): one socket's cores / every usable core, capped by the container's CPU share of {usable})",
            "runs": runs, "host": info}


def other_configs(steps, dev):
    """BASELINE.json configs[3] (K = 5 decoders, DAGM-shaped 1 x 512 x 512, 2 classes, 8 + 8 images) and the per-GPU shape of
    configs[4] (ResNet-50 encoder, K = 3, 640 x 640, 2 classes, 8 + 8 images): `steps` training steps each in the headline launch
    mode (configs[3]: captured hipGraph + streams; the ResNet net steps eagerly), HIP events around the timed steps, then one
    single-stream eager step with dispatch events on every convolution launch for the step's algorithmic flops, the dominant
    kernel and the step-level roofline (t_min = max(flops at each kernel's matrix peak, HBM bytes / 8 TB/s); the HBM bytes come
    from profiles/pmc_traffic_<tag>.json when the PMC passes of that configuration are committed)."""
    import gc
    import numpy as np
    import torch
    import uaps_amd
    import uaps_amd.unet as _unet
    from uaps_amd import conv
    out = []
    for tag, name, net, in_chns, classes, aux, batch, size in (
            ("configs3", "configs[3]: UAPS K=5 decoders, DAGM-shaped 1x512x512 2-class, batch 8+8", "unet_uaps", 1, 2, 5, 8, 512),
            ("configs4", "configs[4] per-GPU shape: UAPS ResNet-50 encoder K=3, KoSDD2-shaped 3x640x640 2-class, batch 8+8", "resnet50_uaps", 3, 2, 3, 8, 640)):
        streams = _unet._DECODER_STREAMS
        try:
            torch.manual_seed(1337)
            model = uaps_amd.net_factory(net, in_chns, classes, n_aux=aux)
            use_graph = net == "unet_uaps"
            trainer = uaps_amd.UAPSTrainer(model, seed=1337, use_graph=use_graph)
            data = uaps_amd.data.SyntheticBatches(batch, in_chns, classes, size, size, n_batches=2, seed=1337, device=dev)
            for _ in range(4 if use_graph else 2):           # two eager steps, the capture and a first replay
                trainer.train_step(*data.next())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                trainer.train_step(*data.next())
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / steps
            loss = float(trainer.last["loss"])
            trainer.check_errors()
            graph_used = trainer.step_graph is not None and trainer.step_graph.graph is not None
            rec = {"workload": name, "images_per_s": round(2 * batch / ms * 1e3, 1), "ms_per_step": round(ms, 2), "steps": steps,
                   "final_loss": round(loss, 5),
                   "launch_mode": ("captured hipGraph replayed per step, " if graph_used else "eager, ") + "one HIP stream per auxiliary decoder"}
            # ---- one single-stream eager step with events on every conv launch: flops, dominant kernel, roofline ----
            _unet._DECODER_STREAMS = False
            trainer.step_graph, trainer.optimizer.from_step_state = None, False
            trainer.train_step(*data.next())
            torch.cuda.synchronize()
            conv.KERNEL_EVENTS, conv.EVENT_FILTER = {}, None
            trainer.train_step(*data.next())
            torch.cuda.synchronize()
            ev, conv.KERNEL_EVENTS = conv.KERNEL_EVENTS, None
            per = {k: {"calls": len(v), "us": float(sum(s.elapsed_time(e) for s, e, _ in v)) * 1e3, "work": float(sum(r[2] for r in v))}
                   for k, v in ev.items()}
            flops = sum(v["work"] for v in per.values())
            t_mfma = sum(v["work"] / (mfma_peak_for(k) * 1e12) for k, v in per.items()) * 1e3
            pmc = {}
            try:
                with open(os.path.join(ROOT, "profiles", f"pmc_traffic_{tag}.json")) as f:
                    pmc = json.load(f)
            except (OSError, ValueError):
                pass
            hbm = pmc.get("__step_total_bytes")
            t_hbm = hbm / (HBM_PEAK_GBS * 1e9) * 1e3 if hbm else None
            t_min = max(t_mfma, t_hbm or 0.0)
            dom = max(per, key=lambda k: per[k]["us"]) if per else None
            roof = {"step_flops": flops, "step_hbm_bytes": hbm, "t_min_mfma_ms": round(t_mfma, 3),
                    "t_min_hbm_ms": round(t_hbm, 3) if t_hbm else None, "t_min_ms": round(t_min, 3), "frac_step": round(t_min / ms, 4),
                    "step_TFLOPs": round(flops / ms / 1e9, 2),
                    "note": "flops: 2*B*H*W*Cin*Cout*k*k summed over every stride-1 convolution launch of one step (the three strided "
                            "convolutions of the ResNet stem / layer2 are not counted); t_min_mfma: each launch at the peak of the matrix "
                            "instruction it runs on; HBM bytes: rocprofv3 --pmc passes of this configuration (profiles/), null when not collected"}
            if dom:
                d = per[dom]
                ach = d["work"] / d["us"] / 1e6
                roof["dominant_kernel"] = {"kernel": dom, "launches_per_step": d["calls"], "avg_us": round(d["us"] / d["calls"], 2),
                                           "achieved": round(ach, 2), "peak": round(mfma_peak_for(dom), 1), "unit": "TFLOP/s",
                                           "frac": round(ach / mfma_peak_for(dom), 4), "traffic": pmc.get(dom),
                                           "note": "single-stream eager step, dispatch events (1 step)"}
            rec["roofline"] = roof
            out.append(rec)
            del trainer, model, data
        except Exception as e:                               # never lose the headline line to a side measurement
            out.append({"workload": name, "error": f"{type(e).__name__}: {e}"[:300]})
        finally:
            conv.KERNEL_EVENTS = None
            _unet._DECODER_STREAMS = streams
        gc.collect()
        torch.cuda.empty_cache()
    return out


def inference_block(dev, size=256, classes=4, aux=3):
    """Row f-2 (the reference's only published performance figures are inference: README.md:107-111 / fig_data/decoder-effect.jpg,
    notebook cells 11-19): ms per image of (a) the main head alone -- encoder + main decoder + arg-max, what the paper's
    "main decoder only" row times -- and (b) the all-heads ensemble -- every decoder + the mixing kernel's arg-max of the mean
    softmax -- at batch 1 and 16, eval mode, eager launches on one stream, inputs resident; median of `reps` timed calls;
    `main_head_graph`: (a) as a replayed hipGraph (inference.CapturedMainHead), the copy of the input into the graph's buffer included."""
    import numpy as np
    import torch
    import uaps_amd
    import uaps_amd.unet as _unet
    from uaps_amd import inference
    streams = _unet._DECODER_STREAMS
    out = {"note": "eval mode, eager launches, one HIP stream, inputs resident in HBM, fp32; median over timed calls of HIP-event time; "
                   "the paper's figures (hardware not stated in the repository): main head 4.48 ms / image, ensemble of 4 heads 29.47 ms / image",
           "paper_ms_per_image": {"main_head": 4.48, "ensemble_4_heads": 29.47}}
    try:
        _unet._DECODER_STREAMS = False
        torch.manual_seed(1337)
        model = uaps_amd.net_factory("unet_uaps", 3, classes, n_aux=aux)
        model.eval()
        for B in (1, 16):
            x = torch.randn(B, 3, size, size, device=dev)
            rec = {}
            for name, fn in (("main_head", lambda: inference.predict_main(model, x)), ("ensemble", lambda: inference.predict_ensemble(model, x))):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                ts = []
                for _ in range(10):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); fn(); e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                rec[name] = {"ms_per_call": round(float(np.median(ts)), 3), "ms_per_image": round(float(np.median(ts)) / B, 3)}
            # the main head as a replayed hipGraph (inference.CapturedMainHead): the input is copied into the graph's buffer inside the timed call
            run = inference.CapturedMainHead(model, x)
            same = bool(torch.equal(run(x), inference.predict_main(model, x)))
            torch.cuda.synchronize()
            ts = []
            for _ in range(10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(x); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            rec["main_head_graph"] = {"ms_per_call": round(float(np.median(ts)), 3), "ms_per_image": round(float(np.median(ts)) / B, 3),
                                      "equals_eager": same}
            del run
            out[f"batch_{B}"] = rec
    except Exception as e:                                   # never lose the headline line to a side measurement
        out["error"] = f"{type(e).__name__}: {e}"[:300]
    finally:
        _unet._DECODER_STREAMS = streams
    return out


class PowerLog:
    """Clock and socket power across the timed regions (round 6): tools/power_sampler.py runs as a child process started BEFORE this
    process touches the GPU (it makes no HIP call itself: amdsmi's metrics table, or the hwmon files), sampling every 20 ms; the
    regions are host time stamps taken next to the synchronizes that bracket them.  Never fails the bench: without a sampler the
    fields are null."""

    def __init__(self, period_s=0.02):
        import subprocess
        import tempfile
        self.proc, self.path, self.regions = None, None, {}
        try:
            fd, self.path = tempfile.mkstemp(prefix="uaps_power_", suffix=".jsonl")
            os.close(fd)
            self.proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "power_sampler.py"), self.path, str(period_s)],
                                         stdin=subprocess.PIPE, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except Exception:
            self.proc = None

    def mark(self, name, t0, t1):
        self.regions[name] = (t0, t1)

    def finish(self, bdf=None):
        """{region: {sclk_mhz mean / min / max, power_w mean / max, samples, ppt_limited_frac}} + power_cap_w, source."""
        if self.proc is None:
            return None
        try:
            time.sleep(0.05)
            self.proc.stdin.close()
            self.proc.wait(timeout=5)
        except Exception:
            try:
                self.proc.kill()
            except Exception:
                pass
        try:
            lines = [json.loads(l) for l in open(self.path) if l.strip()]
            os.unlink(self.path)
        except Exception:
            return None
        if not lines or not lines[0].get("header"):
            return None
        head, recs = lines[0], lines[1:]
        gpus = head.get("gpus") or []
        idx = next((i for i, g in enumerate(gpus) if bdf and g.get("bdf") == bdf), 0 if len(gpus) == 1 else None)
        if idx is None:      # several GPUs and no BDF match: the one that drew the most power over the run
            tot = [sum((r["gpus"][i].get("power_w") or 0) for r in recs if i < len(r["gpus"])) for i in range(len(gpus))]
            idx = max(range(len(gpus)), key=lambda i: tot[i]) if gpus else None
        if idx is None:
            return None
        out = {"source": head.get("source"), "period_s": head.get("period_s"), "gpu_bdf": gpus[idx].get("bdf"),
               "power_cap_w": gpus[idx].get("power_cap_w"), "regions": {}}
        for name, (t0, t1) in self.regions.items():
            rs = [r["gpus"][idx] for r in recs if t0 <= r["t"] <= t1 and idx < len(r["gpus"]) and "error" not in r["gpus"][idx]]
            clk = [g["sclk_mhz"] for g in rs if g.get("sclk_mhz")]
            pw = [g["power_w"] for g in rs if g.get("power_w")]
            reg = {"samples": len(rs), "seconds": round(t1 - t0, 3),
                   "sclk_mhz": round(sum(clk) / len(clk), 1) if clk else None, "sclk_mhz_min": round(min(clk), 1) if clk else None,
                   "sclk_mhz_max": round(max(clk), 1) if clk else None,
                   "power_w": round(sum(pw) / len(pw), 1) if pw else None, "power_w_max": round(max(pw), 1) if pw else None}
            acc = [(g.get("ppt_acc"), g.get("acc")) for g in rs if g.get("ppt_acc") is not None and g.get("acc") is not None]
            if len(acc) >= 2 and acc[-1][1] > acc[0][1]:      # share of the driver's accumulation ticks in which the PPT (package power) limiter was active
                reg["ppt_limited_frac"] = round((acc[-1][0] - acc[0][0]) / (acc[-1][1] - acc[0][1]), 4)
            out["regions"][name] = reg
        return out


def plan_ranks(usable_cpus, world, local_rank, graph_multi_env=""):
    """What a rank of an N-rank run on ONE host does about the host's cores (pure function: tests/test_host_cpu.py drives it with a
    mocked 16-CPU quota and 8 ranks).  Returns (affinity, cores_per_rank, graph_multi):
      affinity        the slice [r n, (r + 1) n) of the usable cores this rank pins itself to when n = cores / ranks >= 4 (launch thread +
                      autograd thread + RCCL's threads), else None: with fewer the scheduler is left alone;
      cores_per_rank  n as above (the pinned slice's size, or usable cores // ranks);
      graph_multi     step with the two-graph form of uaps_amd/graph.py instead of eager launches: forced by UAPS_GRAPH_MULTI=1 / =0, else
                      chosen when a rank has fewer than 2 cores (the eager step needs the host: 14.0 ms with two cores per rank, 18.3 ms
                      with one -- DESIGN.md section 6)."""
    cpus = sorted(usable_cpus)
    affinity = None
    if world > 1:
        per = len(cpus) // world
        if per >= 4:
            affinity = cpus[local_rank * per:(local_rank + 1) * per]
    cores_per_rank = len(affinity) if affinity is not None else len(cpus) // max(world, 1)
    graph_multi = graph_multi_env == "1" or (graph_multi_env != "0" and world > 1 and cores_per_rank < 2)
    return affinity, cores_per_rank, graph_multi


def spawn_ranks(n):
    """`python bench.py --gpus N ...` without torch.distributed.run: start N fresh copies of this command, one rank per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as the launcher would), wait for them and return the job's exit code.
    Called before anything has touched the GPU; the children are new processes, never an exec of this one.  Rank 0 inherits
    stdout (the one JSON line); the other ranks' stdout goes to stderr."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", str(port)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    codes = []
    try:
        # one rank failing must not leave the others waiting in a collective for ever
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is not None:
                    pending.discard(r)
                    codes.append(rc)
                    if rc != 0:
                        for q in pending:
                            procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p in procs:
        p.wait()
    bad = [c for c in (p.returncode for p in procs) if c != 0]
    return 0 if not bad else (bad[0] if bad[0] > 0 else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="labelled images per GPU (+ as many unlabelled)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--classes", type=int, default=4)
    ap.add_argument("--aux", type=int, default=3)
    ap.add_argument("--in-chns", type=int, default=3, help="image channels (1 for the DAGM-shaped configs[3])")
    ap.add_argument("--net", default="unet_uaps", help="unet_uaps (BASELINE.json configs[1], the reported metric) or resnet50_uaps (configs[4] shape study)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-stream", action="store_true",
                    help="time the headline steps with all launches on one HIP stream (default: one stream per auxiliary decoder, "
                         "same kernels and results, launches of different decoders overlap)")
    ap.add_argument("--no-graph", action="store_true",
                    help="N=1 only: run the timed steps eagerly instead of replaying the captured hipGraph of the step (uaps_amd/graph.py)")
    ap.add_argument("--exact-steps", type=int, default=5,
                    help="steps timed with every convolution on the fp32 matrix instruction after the headline (0 = skip)")
    ap.add_argument("--analysis-steps", type=int, default=6, help="single-stream steps after the timed region for the per-kernel figures")
    ap.add_argument("--no-inference", action="store_true", help="skip the inference block (main head / ensemble ms per image, N = 1 only)")
    ap.add_argument("--no-power-log", action="store_true", help="do not start tools/power_sampler.py (clock / socket power across the timed regions)")
    ap.add_argument("--other-configs", type=int, default=4,
                    help="N = 1 only: steps timed of each of BASELINE.json configs[3] and the per-GPU shape of configs[4] after everything else (0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # not under a launcher: this process becomes one (it has touched no GPU and imports no torch) and exits with the job's code
        raise SystemExit(spawn_ranks(args.gpus))

    # rank 0 samples clock and socket power for the whole run: the child process starts before anything here touches the GPU
    plog = PowerLog(0.01) if int(os.environ.get("RANK", "0")) == 0 and not args.no_power_log else None

    import numpy as np
    import torch
    import torch.distributed as dist
    import uaps_amd
    from uaps_amd import losses

    import uaps_amd.unet as _unet
    _unet._DECODER_STREAMS = not args.single_stream
    if os.environ.get("UAPS_BENCH_CPUS"):                   # experiment hook: this process on its first n usable cores (what one of 8 ranks gets)
        os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:int(os.environ["UAPS_BENCH_CPUS"])])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch --nproc-per-node {args.gpus} ranks "
                         "(or run without a launcher: this script then starts the ranks itself)")
    # test hooks (tests/test_gpu_two_ranks.py runs the N = 2 code path on a one-GPU box): every rank on one device, gloo
    # instead of RCCL, which refuses two ranks on a device
    backend = os.environ.get("UAPS_BENCH_BACKEND", "nccl")
    if "UAPS_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["UAPS_BENCH_DEVICE"])
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
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    ranks_seen, affinity = 1, None
    if distributed:
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)                                # the process group itself counts the ranks (not WORLD_SIZE)
        ranks_seen = int(one.item())
        if world > 1 and hasattr(os, "sched_setaffinity"):  # every rank its own slice of the usable cores: 8 launching processes + RCCL threads share the host
            affinity = plan_ranks(os.sched_getaffinity(0), world, local_rank)[0]
            if affinity is not None:
                try:
                    os.sched_setaffinity(0, affinity)
                except OSError:
                    affinity = None
    torch.manual_seed(1337)
    D, C, H, W, b = args.aux + 1, args.classes, args.size, args.size, args.batch
    model = uaps_amd.net_factory(args.net, args.in_chns, C, n_aux=args.aux)
    uaps_amd.dist.broadcast_model(model)
    if saved_stdout is not None:
        dist.barrier()
        torch.cuda.synchronize()
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    # one process: the step is captured as a hipGraph after two eager steps (needs --warmup >= 3 to stay out of the timed region)
    # and replayed (the eager step's kernels and arithmetic)
    # N > 1 steps eagerly (decoder streams, bucket all-reduces overlapped with the backward).  UAPS_GRAPH_MULTI=1 selects the two-graph
    # form of uaps_amd/graph.py instead (two gloo ranks on one card: bit-equal to the eager step; over RCCL only ever run with one
    # rank here, where tools/diag/rccl_split_graph.py once aborted behind other process groups of the same process -- so not the default)
    # Round 5: the eager step needs the host -- 14.0 ms with two cores per rank, 18.3 ms with one -- so when a rank
    # has fewer than 2 usable cores, the two-graph form is selected by itself (UAPS_GRAPH_MULTI=0 forces eager, =1 forces the graphs).
    # Not already below 4, as the round-4 review suggested: with 2-3 cores the eager step is within 7 % of the replayed one, and the
    # two-graph form has never run over RCCL with more than one rank -- 7 % is not worth the first such run being the scaling run.
    # The abort seen once behind it happened in a process that had created and destroyed other process groups with captures in
    # between; this script creates exactly one group per process, before any capture.
    usable = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else range(os.cpu_count() or 1)
    if affinity is not None:
        cores_per_rank, graph_multi = len(affinity), os.environ.get("UAPS_GRAPH_MULTI", "") == "1"
    else:
        _, cores_per_rank, graph_multi = plan_ranks(usable, world, local_rank, os.environ.get("UAPS_GRAPH_MULTI", ""))
    use_graph = not args.no_graph and args.net == "unet_uaps" and (world == 1 or graph_multi)
    trainer = uaps_amd.UAPSTrainer(model, seed=1337, use_graph=use_graph)
    data = uaps_amd.data.SyntheticBatches(b, args.in_chns, C, H, W, n_batches=2, seed=1337 + rank, device=dev)

    from uaps_amd import conv

    def summarize(ev):
        """{kernel: [(start, end, work)]} -> {kernel: {calls, avg_us, total_us, work}} (needs a prior synchronize)."""
        out = {}
        for name, recs in ev.items():
            us = [s.elapsed_time(e) * 1e3 for s, e, *_ in recs]
            out[name] = {"calls": len(us), "avg_us": float(np.mean(us)), "total_us": float(np.sum(us)),
                         "work": float(sum(r[2] for r in recs if len(r) > 2))}
        return out

    # ---- headline: W warm-up steps, then exactly K timed steps, in the fastest launch mode of the same kernels ----
    for i in range(args.warmup):
        trainer.train_step(*data.next())
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # per-step spread (median, p10 / p90)
    t0 = time.perf_counter()
    w0 = time.time()
    marks[0].record()
    for i in range(args.steps):
        trainer.train_step(*data.next())
        marks[i + 1].record()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if plog is not None:
        plog.mark("timed_region", w0, time.time())
    if distributed:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    last_loss = float(trainer.last["loss"])
    sg = trainer.step_graph
    graph_used = sg is not None and sg.graph is not None        # False: no capture asked for, too few warm-up steps, or a failed capture
    graph_split = graph_used and sg.split

    # ---- per-kernel analysis pass, same process, SINGLE stream (a kernel then has the chip to itself, as in the rocprofv3
    # profiles): one step with HIP events attached to the dispatch of every hand-written conv / loss kernel (finds the dominant
    # instantiation and sums the step's algorithmic flops), then `analysis_steps` steps with events around the dominant
    # kernel's launches and the loss kernels only, timed as a whole for the single-stream ms/step ----
    discover, ev, cev, single_ms, alg_bytes = None, {}, {}, None, None
    if args.analysis_steps > 0:                              # every rank steps (the gradient exchange is collective); rank 0's events are reported
        _unet._DECODER_STREAMS = False
        trainer.step_graph, trainer.optimizer.from_step_state = None, False     # eager launches: events can bracket each kernel
        trainer.train_step(*data.next())                   # re-warm in the new mode
        torch.cuda.synchronize()
        conv.KERNEL_EVENTS, conv.EVENT_FILTER, losses.KERNEL_EVENTS = {}, None, {}
        from uaps_amd import _lib as _ulib
        _ulib.lib().uaps_account(1)                        # the library tallies the algorithmic bytes of every launch of this one step
        trainer.train_step(*data.next())
        torch.cuda.synchronize()
        _ulib.lib().uaps_account(0)
        alg_bytes = float(_ulib.lib().uaps_accounted_bytes())
        discover = summarize(conv.KERNEL_EVENTS)
        for k, pairs in losses.KERNEL_EVENTS.items():
            us = [s.elapsed_time(e) * 1e3 for s, e in pairs]
            discover[k] = {"calls": len(us), "avg_us": float(np.mean(us)), "total_us": float(np.sum(us)), "work": 0.0}
        conv_names = [k for k in discover if discover[k]["work"] > 0]
        dominant = max(conv_names, key=lambda k: discover[k]["total_us"]) if conv_names else None
        losses.KERNEL_EVENTS = {}
        conv.KERNEL_EVENTS, conv.EVENT_FILTER = {}, ({dominant} if dominant else None)
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0 = time.time()
        a0.record()
        for i in range(args.analysis_steps):
            trainer.train_step(*data.next())
        a1.record()
        torch.cuda.synchronize()
        if plog is not None:
            plog.mark("analysis_pass", w0, time.time())
        single_ms = a0.elapsed_time(a1) / args.analysis_steps
        ev, losses.KERNEL_EVENTS = losses.KERNEL_EVENTS, None
        cev, conv.KERNEL_EVENTS, conv.EVENT_FILTER = conv.KERNEL_EVENTS, None, None
        _unet._DECODER_STREAMS = not args.single_stream
    else:
        dominant = None

    # ---- the same step on the fp32 matrix instruction everywhere (UAPS_CONV_MODE=0), a few eager steps in the headline stream
    # mode: the figure to hold against `value` for anyone who does not accept split arithmetic as fp32 ----
    def eager_leg(n_steps):
        """ms per step of `n_steps` eager steps in the headline stream mode after two warm-up steps (rank-local HIP events)."""
        for i in range(2):
            trainer.train_step(*data.next())
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x0.record()
        for i in range(n_steps):
            trainer.train_step(*data.next())
        x1.record()
        torch.cuda.synchronize()
        return x0.elapsed_time(x1) / n_steps

    exact_ms = split_ms = nocomm_ms = eager_ms = None
    comm = None
    if args.exact_steps > 0:
        prev_mode = conv.get_mode()
        trainer.step_graph, trainer.optimizer.from_step_state = None, False
        if world > 1 and trainer.buckets is not None:
            # the exchange, measured after the timed region: every bucket's all-reduce alone (events on the step's stream), and
            # the eager step with and without the exchange -- their difference is what the backward does not hide
            eager_ms = eager_leg(args.exact_steps)
            per_bucket = []
            for name, flat in zip(trainer.buckets.names, trainer.buckets._flat):
                scratch = flat.clone()
                dist.all_reduce(scratch)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    dist.all_reduce(scratch)
                e1.record()
                torch.cuda.synchronize()
                per_bucket.append({"bucket": name, "bytes": flat.numel() * 4, "allreduce_ms": round(e0.elapsed_time(e1) / 5, 4)})
            kept, trainer.buckets = trainer.buckets, None
            kept.muted = True                                # the post-accumulate hooks stay registered: they must neither count nor launch
            nocomm_ms = eager_leg(args.exact_steps)          # replicas drift apart from here on: nothing below compares ranks
            kept.muted = False
            kept.reset()
            trainer.buckets = kept
            comm = {"backend": backend, "buckets": per_bucket, "bytes_per_step": sum(p["bytes"] for p in per_bucket),
                    "eager_step_ms": round(eager_ms, 3), "eager_step_without_exchange_ms": round(nocomm_ms, 3),
                    "exposed_ms": round(max(0.0, eager_ms - nocomm_ms), 3),
                    "note": "rank 0; all-reduces issued from post-accumulate hooks during the backward (uaps_amd/dist.py)"}
        if prev_mode != "exact":
            conv.set_mode("exact")
            exact_ms = eager_leg(args.exact_steps)
        if prev_mode != "split":
            conv.set_mode("split")
            split_ms = eager_leg(args.exact_steps)
        conv.set_mode(prev_mode)

    if rank == 0:
        npix = b * H * W
        n_an = max(args.analysis_steps, 1)
        kern = {}
        for name, pairs in ev.items():
            ms = [s.elapsed_time(e) for s, e in pairs]
            avg_s = float(np.mean(ms)) * 1e-3
            by = loss_kernel_bytes(name, D, C, npix)
            kern[name] = {"calls_per_step": len(ms) / n_an, "avg_us": round(avg_s * 1e6, 2), "GBps": round(by / avg_s / 1e9, 1)}
        timed = summarize(cev)
        for name, v in timed.items():
            kern[name] = {"calls_per_step": v["calls"] / n_an, "avg_us": round(v["avg_us"], 2),
                          "TFLOPs": round(v["work"] / v["total_us"] / 1e6, 2)}
        for name, v in (discover or {}).items():          # one single-stream step's sample of the other instantiations
            if name not in kern:
                kern[name] = {"calls_per_step": v["calls"], "avg_us": round(v["avg_us"], 2), "sample": "1 step"}
                if v["work"]:
                    kern[name]["TFLOPs"] = round(v["work"] / v["total_us"] / 1e6, 2)
        pmc = {}
        try:   # HBM bytes per launch / per step from the rocprofv3 PMC passes (profiles/README.md), when they have been collected
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f)
        except (OSError, ValueError):
            pass
        traffic = pmc.get(dominant)
        ms_per_step = dt / args.steps * 1e3
        if dominant in timed:
            v = timed[dominant]
            ach = v["work"] / v["total_us"] / 1e6                     # TFLOP/s, algorithmic 2*B*H*W*Cin*Cout*k*k per launch
            peak = mfma_peak_for(dominant)
            roof = {"kernel": dominant, "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic, "avg_us": round(v["avg_us"], 2),
                    "launches_per_step": v["calls"] / n_an,
                    "note": "kernel instantiation with the largest share of the step; measured in the single-stream analysis pass "
                            f"of this process ({args.analysis_steps} steps after the timed region; launches of different decoders "
                            "overlap in the headline mode, so a kernel only has the chip to itself when single-stream): "
                            "algorithmic flops = 2*B*H*W*Cin*Cout*k*k per launch, summed over its launches / summed time between the two "
                            "HIP events attached to each kernel's dispatch on its launch stream (uaps_next_launch_events: the kernel's "
                            "execution time, as rocprofv3's kernel trace reports it); traffic = HBM bytes per launch from profiles/pmc_traffic.json (rocprofv3 --pmc passes); "
                            "peak: fp32 flops per second the kernel's matrix instruction allows at the nominal 2.4 GHz -- 157.3 for the "
                            "fp32 MFMA kernels, 2500 / 3 = 833.3 for the fp16-split kernels (conv_h*: three fp16 partial products per "
                            "fp32 multiply, fp32 accumulation), 2500 / 6 = 416.7 for the bf16-split kernels (conv_s*, six partial "
                            "products; operands without a magnitude bound); DESIGN.md section 5"}
            # What the chip has been SEEN to sustain with the forward kernels' arithmetic (tools/split_gemm_ceiling.hip, committed
            # measurement): the same fp16-split contraction -- fp32 activations split in staging, pre-split weights, three products, fp32
            # accumulate -- as a nine-tap GEMM without the convolution's geometry (no halo, no tap shifts), on large per-wave register
            # blocks, for the M x N x K of the four layers the forward / input-gradient tile kernels spend their time on; time-weighted.
            # Attached to the forward / input-gradient tile kernel with the largest share of the step (the dominant kernel itself, or
            # `forward_tile_kernel` beside it when a weight-gradient kernel leads on this box: the two are within 5 % of each other).
            try:
                with open(os.path.join(ROOT, "profiles", "r06_split_gemm_ceiling.json")) as f:
                    ceil = json.load(f)
                gf = sum(l["gflop"] for l in ceil["layers"])
                tw = gf / sum(l["gflop"] / l["best_tflops_reread"] for l in ceil["layers"])
                fwd_fam = lambda n: n.startswith(("conv_h32", "conv_hfwd", "conv_hg"))

                def practical(ach_tf):
                    return {"tflops": round(tw, 1), "frac": round(ach_tf / tw, 4),
                            "per_layer_tflops": {l["layer"]: l["best_tflops_reread"] for l in ceil["layers"]},
                            "source": "profiles/r06_split_gemm_ceiling.json",
                            "note": "same-arithmetic tap-reuse GEMM (fp32 activations split into two fp16 pieces in staging, three products on "
                                    "v_mfma_f32_16x16x32_f16, fp32 accumulation) at B = 32 for 64->64 and 128->64 @64^2, 128->128 and 256->128 "
                                    "@32^2: best form per layer with the A fragments re-read per tap (what a 3x3 kernel must do), time-weighted; "
                                    "the committed file is the round-6 closing box's measurement, not this run's box (DESIGN.md section 3.2)"}
                if fwd_fam(dominant):
                    roof["practical_peak"] = practical(ach)
                else:
                    cands = {k: v2 for k, v2 in (discover or {}).items() if fwd_fam(k) and v2.get("work")}
                    if cands:
                        k = max(cands, key=lambda n: cands[n]["total_us"])
                        a2 = cands[k]["work"] / cands[k]["total_us"] / 1e6
                        roof["forward_tile_kernel"] = {"kernel": k, "achieved": round(a2, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                                                       "frac": round(a2 / peak, 4), "avg_us": round(cands[k]["avg_us"], 2),
                                                       "launches_per_step": cands[k]["calls"], "sample": "1 single-stream step (dispatch events)",
                                                       "practical_peak": practical(a2)}
            except (OSError, ValueError, KeyError, ZeroDivisionError):
                pass
            # The same for the weight-gradient tile kernels (tools/split_wrw_ceiling.hip): BOTH operands are fp32 activations split while
            # staged, M = Cout, N = Cin x 9 taps, K = pixels split over the workgroups; the probe takes the geometry out (no halo rows, no
            # column shifts, no borders) and keeps the shipped blocking among its forms.
            try:
                if dominant.startswith("conv_hwrw"):
                    with open(os.path.join(ROOT, "profiles", "r06_split_wrw_ceiling.json")) as f:
                        wceil = json.load(f)
                    gfw = sum(l["gflop"] for l in wceil["layers"])
                    tww = gfw / sum(l["gflop"] / l["best_tflops"] for l in wceil["layers"])
                    roof["practical_peak"] = {
                        "tflops": round(tww, 1), "frac": round(ach / tww, 4),
                        "per_layer_tflops": {l["layer"]: l["best_tflops"] for l in wceil["layers"]},
                        "source": "profiles/r06_split_wrw_ceiling.json",
                        "note": "same-arithmetic weight-gradient contraction without the convolution's geometry (both operands fp32 in HBM, split "
                                "into two fp16 pieces in staging, three products on v_mfma_f32_16x16x32_f16, nine taps on one staged fragment, "
                                "per-split partial sums written once) at B = 32 on the four dominant layers, best form per layer, time-weighted, "
                                "back-to-back launches; the shipped kernel measured the same way reaches 248-291 TFLOP/s (0.72-0.76 of it, "
                                "profiles/r06_wrw_ablation.txt) -- inside the step it runs ~10 % below its back-to-back figure; committed "
                                "measurement of the round-6 closing box, not this run's (DESIGN.md section 3.3)"}
            except (OSError, ValueError, KeyError, ZeroDivisionError):
                pass
        elif ev:
            dom = max(ev, key=lambda k: kern[k]["avg_us"])
            roof = {"kernel": dom, "bound": "hbm", "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(kern[dom]["GBps"] / HBM_PEAK_GBS, 4), "traffic": pmc.get(dom)}
        else:
            roof = None
        if roof is not None and discover:
            # step level: t_min = max(flops / fp32 matrix peak, HBM bytes / HBM peak) against the measured step
            step_flops = float(sum(v["work"] for v in discover.values()))
            step_bytes = pmc.get("__step_total_bytes")
            # every conv launch at the peak of the matrix instruction it runs on (fp32 MFMA or the bf16-split form)
            t_mfma = float(sum(v["work"] / (mfma_peak_for(k) * 1e12) for k, v in discover.items() if v["work"])) * 1e3
            t_mfma_f32 = step_flops / (MFMA_F32_PEAK_TF * 1e12) * 1e3
            t_hbm = step_bytes / (HBM_PEAK_GBS * 1e9) * 1e3 if step_bytes else None
            t_min = max(t_mfma, t_hbm or 0.0)
            # the same against the ALGORITHMIC bytes of the step (uaps_account: every launch's operands read once and results written
            # once in this fused design) -- measured bytes in t_min_hbm raise the floor with every wasted byte, these do not
            t_alg = alg_bytes / (HBM_PEAK_GBS * 1e9) * 1e3 if alg_bytes else None
            roof.update({"step_algorithmic_bytes": alg_bytes, "t_min_alg_hbm_ms": round(t_alg, 3) if t_alg else None,
                         "t_min_alg_ms": round(max(t_mfma, t_alg or 0.0), 3),
                         "frac_step_alg": round(max(t_mfma, t_alg or 0.0) / ms_per_step, 4),
                         "hbm_bytes_over_algorithmic": round(step_bytes / alg_bytes, 3) if (step_bytes and alg_bytes) else None,
                         "step_algorithmic_bytes_note": "sum over the launches of one step of (operand tensors read once + results written once), fp32 / int64 as "
                                                        "the reference holds them, tallied by the library's entry points (uaps_account; DESIGN.md section 5): the "
                                                        "traffic floor of THIS fused design, against which step_hbm_bytes (PMC) shows the re-reads"})
            roof.update({"step_flops": step_flops, "step_hbm_bytes": step_bytes,
                         "step_hbm_bytes_note": "sum over all kernels of launches x (2*FETCH_SIZE + WRITE_SIZE) from the committed rocprofv3 "
                                                "--pmc passes of this bench command, per step; null when no PMC summary is committed",
                         "t_min_ms": round(t_min, 3), "t_min_mfma_ms": round(t_mfma, 3), "t_min_hbm_ms": round(t_hbm, 3) if t_hbm else None,
                         "t_min_if_all_fp32_mfma_ms": round(t_mfma_f32, 3),
                         "frac_step": round(t_min / ms_per_step, 4),
                         "frac_step_single_stream": round(t_min / single_ms, 4) if single_ms else None,
                         "step_TFLOPs": round(step_flops / ms_per_step / 1e9, 2),
                         "step_hbm_GBps": round(step_bytes / ms_per_step / 1e6, 1) if step_bytes else None,
                         "step_hbm_frac": round(step_bytes / ms_per_step / 1e6 / HBM_PEAK_GBS, 4) if step_bytes else None})
        step_ms = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)])
        mode = "single stream" if args.single_stream else ("one HIP stream per auxiliary decoder + side streams for work off the critical path (the decoders' "
                                                           "weight packing and the perturbed feature copies beside the encoder's forward; the decoders' "
                                                           "weight-gradient reductions and Adam step beside the encoder's backward); same kernels and results as single-stream")
        if graph_used and not graph_split:
            mode = "captured hipGraph of the whole step, replayed once per step (the eager step's kernels and arithmetic; DESIGN.md section 4 on bit reproducibility); " + mode
        elif graph_split:
            mode = ("two captured hipGraphs per step (forward + loss + backward | Adam + metrics) replayed around the eager RCCL all-reduce of "
                    "the flat gradient buckets (the eager step's kernels and arithmetic); ") + mode
        elif use_graph:
            mode = "eager launches (the capture was requested but did not take place); " + mode
        res = {"metric": "training images/sec (labeled+unlabeled) NEU-Seg 256x256 K=3", "value": round(2 * b * world * args.steps / dt, 2),
               "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "arithmetic": {"h16": "fp32 tensors and fp32 accumulation throughout; 3x3 convolutions with > 8 channels scale both fp32 operands by "
                                     "powers of two (from device-resident magnitude bounds) and split them into two fp16 pieces (22 significant bits), "
                                     "three partial products per multiply on the fp16 matrix pipe; measured error vs float64 at or below the fp32 "
                                     "matrix instruction's (tests/test_gpu_conv.py); UAPS_CONV_MODE=1: exact three-piece bf16 split, =0: fp32 MFMA",
                              "split": "fp32 tensors and fp32 accumulation throughout; 3x3 convolutions with > 8 channels split both fp32 operands exactly "
                                       "into three bf16 pieces and sum the six leading partial products on the bf16 matrix pipe",
                              "exact": "fp32 matrix instruction (v_mfma_f32_16x16x4_f32) everywhere"}[conv.get_mode()],
               "config": {"workload": f"UAPS K={args.aux} decoders, NEU-Seg-shaped {H}x{W} {C}-class, batch {b}+{b} per GPU (BASELINE.json configs[1])",
                          "heads": D, "per_gpu_batch": f"{b} labelled + {b} unlabelled", "parallelism": f"dp{world}", "final_loss": round(last_loss, 5),
                          "launch_mode": mode,
                          # every UAPS_* switch whose value is not the package default (uaps_amd/config.py) -- {} = the shipped configuration;
                          # UAPS_DECODER_STREAMS is set by this script itself for the headline (--single-stream turns it off)
                          "switches": uaps_amd.config.non_default()},
               "roofline": roof, "kernels": kern}
        # clock and socket power across the timed region and the analysis pass (tools/power_sampler.py)
        power = None
        if plog is not None:
            try:
                pr = torch.cuda.get_device_properties(dev)
                bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            except Exception:
                bdf = None
            power = plog.finish(bdf)
        if power is not None:
            res["power"] = power
            tr_, an_ = power["regions"].get("timed_region") or {}, power["regions"].get("analysis_pass") or {}
            if roof is not None:
                roof["sclk_mhz_under_load"] = tr_.get("sclk_mhz")
                roof["power_w"] = tr_.get("power_w")
                roof["power_cap_w"] = power.get("power_cap_w")
                roof["ppt_limited_frac"] = tr_.get("ppt_limited_frac")
                roof["sclk_mhz_analysis_pass"] = an_.get("sclk_mhz")
                clk = an_.get("sclk_mhz") or tr_.get("sclk_mhz")
                if clk and roof.get("bound") == "mfma":
                    # the dominant kernel is measured in the analysis pass: its fraction of the matrix peak at the clock the chip held there
                    roof["frac_at_measured_clock"] = round(roof["achieved"] / (roof["peak"] * clk / 2400.0), 4)
                    roof["power_note"] = ("sclk = mean over samples of the mean of the 8 XCDs' current_gfxclks (amdsmi metrics table, sampled by a child "
                                          "process every 10 ms); peak is quoted at the nominal 2400 MHz, frac_at_measured_clock rescales it to the sampled "
                                          "clock of the single-stream analysis pass; ppt_limited_frac = share of the driver's accumulation ticks with the "
                                          "package-power limiter active (ppt_residency_acc)")
        res["step_ms"] = {"p10": round(float(np.percentile(step_ms, 10)), 3), "p50": round(float(np.percentile(step_ms, 50)), 3),
                          "p90": round(float(np.percentile(step_ms, 90)), 3), "note": "GPU time between per-step HIP events on rank 0 (headline mode)"}
        if single_ms:
            res["single_stream"] = {"ms_per_step": round(single_ms, 3), "images_per_s": round(2 * b / single_ms * 1e3, 1),
                                    "steps": args.analysis_steps, "note": "the analysis pass the per-kernel figures come from (rank 0, HIP events)"}
        if exact_ms:
            res["fp32_mfma_everywhere"] = {"ms_per_step": round(exact_ms, 3), "images_per_s": round(2 * b / exact_ms * 1e3, 1),
                                           "steps": args.exact_steps,
                                           "note": "UAPS_CONV_MODE=0: v_mfma_f32_16x16x4_f32 for every convolution, eager launches, headline stream mode, rank 0"}
        if split_ms:
            res["bf16x3_exact"] = {"ms_per_step": round(split_ms, 3), "images_per_s": round(2 * b / split_ms * 1e3, 1), "steps": args.exact_steps,
                                   "note": "UAPS_CONV_MODE=1: both operands of every 3x3 convolution split EXACTLY into three bf16 pieces (24 bits), six "
                                           "partial products with fp32 accumulation; eager launches, headline stream mode, rank 0"}
        # north_star asks for the HBM fraction of the loss kernels it names: softmax + KL maps + mixing + arg-max + CE / Dice sums
        # (forward, with its one-block finalize) and the closed-form gradient (backward), algorithmic bytes / dispatch-event time
        rl = {}
        for kname in ("uaps_pair_fwd", "uaps_pair_bwd"):
            if kname in kern and "GBps" in kern[kname]:
                rl[kname] = {"avg_us": kern[kname]["avg_us"], "algorithmic_bytes": loss_kernel_bytes(kname, D, C, npix),
                             "achieved_GBps": kern[kname]["GBps"], "peak_GBps": HBM_PEAK_GBS, "frac": round(kern[kname]["GBps"] / HBM_PEAK_GBS, 4),
                             "traffic": next((v for k, v in pmc.items() if k.startswith({"uaps_pair_fwd": "pair_fwd_kernel", "uaps_pair_bwd": "pair_bwd_kernel"}[kname] + "<")), None)}
        if rl:
            res["roofline_loss"] = rl
        res["ranks_seen"] = ranks_seen
        if affinity is not None:
            res["config"]["cpu_affinity_rank0"] = f"{len(affinity)} cores"
        if comm is not None:
            res["comm"] = comm
        if args.net != "unet_uaps":
            res["config"]["workload"] = f"{args.net} K={args.aux}, {H}x{W} {C}-class, batch {b}+{b} per GPU (not the BASELINE metric config)"
        if world == 1 and args.other_configs > 0 and args.net == "unet_uaps" and (b, H, C, args.aux) == (16, 256, 4, 3):
            # the other single-GPU shapes of BASELINE.json, a few eager steps each (decoder streams as in the headline): not the
            # metric, but a number the driver sees for them
            res["other_configs"] = other_configs(args.other_configs, dev)
        if world == 1 and not args.no_inference and args.net == "unet_uaps" and (H, C, args.aux) == (256, 4, 3):
            res["inference"] = inference_block(dev, H, C, args.aux)
        if world > 1:
            res["config"]["cores_per_rank"] = cores_per_rank
        if world == 1 and not args.no_cpu_baseline and args.net == "unet_uaps":
            res["cpu_baseline"] = cpu_baseline(b, H, W)
        print(json.dumps(res), flush=True)
    if distributed:
        dist.destroy_process_group()
    if ranks_seen != world:
        # the process group did not see every rank the job was launched with: the line above says so (ranks_seen), and the
        # run fails rather than pass for an N-GPU measurement
        print(f"bench.py: ranks_seen = {ranks_seen} but n_gpus = {world}", file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
