#!/usr/bin/env python3
"""Samples shader clock, socket power and the power-limiter residency counter of every visible AMD GPU until it is told to stop.

    python tools/power_sampler.py OUT.jsonl [period_s]

bench.py starts this as a CHILD PROCESS before it touches the GPU (the sampler itself never makes a HIP call: it reads the
driver's metrics table through amdsmi, or the hwmon files when amdsmi is not importable) and stops it by closing its stdin.
One JSON object per line: {"t": unix seconds, "gpus": [{"bdf", "sclk_mhz" (mean over the XCDs' current_gfxclks), "sclk_min_mhz",
"power_w", "ppt_acc" (PPT-limiter residency accumulator), "acc" (the accumulators' tick counter), "energy" (energy accumulator)}]}.
The first line is a header with the power cap per GPU.  Round 6: VERDICT item 3 (power and clock in the driver-run record)."""
import glob
import json
import os
import select
import sys
import time


def _amdsmi():
    try:
        import amdsmi
        return amdsmi
    except Exception:
        sys.path.insert(0, "/opt/rocm/share/amd_smi")
        try:
            import amdsmi
            return amdsmi
        except Exception:
            return None


def _num(v):
    return v if isinstance(v, (int, float)) else None


def main():
    out_path = sys.argv[1]
    period = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
    smi = _amdsmi()
    handles, bdfs, caps = [], [], []
    if smi is not None:
        try:
            smi.amdsmi_init()
            handles = list(smi.amdsmi_get_processor_handles())
            for h in handles:
                try:
                    bdfs.append(str(smi.amdsmi_get_gpu_device_bdf(h)).lower())
                except Exception:
                    bdfs.append(None)
                try:
                    caps.append(_num(smi.amdsmi_get_power_info(h).get("power_limit")))      # micro-watts
                except Exception:
                    caps.append(None)
        except Exception:
            handles = []
    hw = []
    if not handles:      # fall-back: the hwmon files of every amdgpu card (power1_input in micro-watts, freq1_input in Hz)
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            if os.path.exists(os.path.join(d, "power1_input")):
                bdf = os.path.basename(os.path.realpath(os.path.join(d, "..", "..")))
                cap = None
                try:
                    cap = int(open(os.path.join(d, "power1_cap")).read())
                except Exception:
                    pass
                hw.append((d, bdf.lower(), cap))
    with open(out_path, "w") as f:
        f.write(json.dumps({"header": True, "source": "amdsmi" if handles else ("hwmon" if hw else "none"), "period_s": period,
                            "gpus": [{"bdf": b, "power_cap_w": (c / 1e6 if c else None)} for b, c in
                                     (zip(bdfs, caps) if handles else [(b, c) for _, b, c in hw])]}) + "\n")
        f.flush()
        while True:
            t = time.time()
            rec = {"t": t, "gpus": []}
            for i, h in enumerate(handles):
                try:
                    m = smi.amdsmi_get_gpu_metrics_info(h)
                    clks = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float))]
                    rec["gpus"].append({"bdf": bdfs[i], "sclk_mhz": (sum(clks) / len(clks)) if clks else _num(m.get("current_gfxclk")),
                                        "sclk_min_mhz": min(clks) if clks else None,
                                        "power_w": _num(m.get("current_socket_power")), "ppt_acc": _num(m.get("ppt_residency_acc")),
                                        "thm_acc": _num(m.get("socket_thm_residency_acc")), "acc": _num(m.get("accumulation_counter")),
                                        "energy": _num(m.get("energy_accumulator")), "busy": _num(m.get("average_gfx_activity"))})
                except Exception as e:
                    rec["gpus"].append({"bdf": bdfs[i], "error": str(e)[:80]})
            for d, bdf, _cap in hw:
                try:
                    rec["gpus"].append({"bdf": bdf, "sclk_mhz": int(open(os.path.join(d, "freq1_input")).read()) / 1e6,
                                        "power_w": int(open(os.path.join(d, "power1_input")).read()) / 1e6})
                except Exception as e:
                    rec["gpus"].append({"bdf": bdf, "error": str(e)[:80]})
            f.write(json.dumps(rec) + "\n")
            f.flush()
            wait = max(0.0, period - (time.time() - t))
            r, _, _ = select.select([sys.stdin], [], [], wait)      # the parent closes our stdin (or dies): stop
            if r and not sys.stdin.buffer.read(1):
                break


if __name__ == "__main__":
    main()
