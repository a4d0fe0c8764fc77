#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/gpu_pmc.sh).

Units and gfx950 corrections follow /opt/skills/guides/MI355X_MICROARCH.md, section HBM: both counters are in
KiB; WRITE_SIZE reads exactly for 16-byte-per-lane streaming stores; FETCH_SIZE reports exactly half of the
bytes of a wide (16 B/lane) coalesced streaming read, so the read side of kernels that stream with
buffer_load_dwordx4 (the convolution and loss kernels here) is doubled.  Writes
gpurun_out/pmc/pmc_traffic.json = {kernel name fragment: corrected HBM bytes per launch}; copy it to
profiles/pmc_traffic.json for bench.py to report as roofline.traffic."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def per_kernel(root, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(root, counter, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                a = acc[r["Kernel_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return acc


def short(name):
    """`void uaps::conv_h32_kernel<64>(uaps::ConvFwdArgs)` -> `conv_h32_kernel<64>`; `(anonymous namespace)::fanin_perturbed_kernel((anonymous
    namespace)::FanInArgs, ...)` -> `fanin_perturbed_kernel` (round 4 returned '' for these: the name starts with a parenthesis)."""
    s = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("uaps::", "").strip()
    depth, end = 0, len(s)
    for i, ch in enumerate(s):                 # cut at the argument list: the first '(' outside the template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            end = i
            break
    return s[:end].strip()


def main():
    root = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # training steps the profiled command ran (warm-up included)
    fetch, write = per_kernel(root, "FETCH_SIZE"), per_kernel(root, "WRITE_SIZE")
    out, rows = {}, []
    for k in set(fetch) | set(write):
        f, nf = fetch.get(k, [0.0, 0])
        w, nw = write.get(k, [0.0, 0])
        n = max(nf, nw, 1)
        fetch_b = f / max(nf, 1) * 1024.0
        write_b = w / max(nw, 1) * 1024.0
        corrected = 2.0 * fetch_b + write_b
        out[short(k)] = round(corrected)
        rows.append((corrected * n, short(k), n, fetch_b, write_b, corrected))
    rows.sort(reverse=True)
    if steps > 0:      # whole-step HBM traffic: every launch of every kernel of the run, per step (bench.py: roofline.step_hbm_bytes)
        out["__step_total_bytes"] = round(sum(r[0] for r in rows) / steps)
        out["__steps_profiled"] = steps
        print(f"HBM bytes per step over all kernels ({steps} steps profiled): {out['__step_total_bytes'] / 1e9:.2f} GB")
    print(f"{'kernel':70s} {'launches':>8s} {'FETCH_SIZE B/launch (raw)':>26s} {'WRITE_SIZE B/launch':>20s} {'HBM B/launch (2*F+W)':>22s}")
    for _, k, n, fb, wb, c in rows[:40]:
        print(f"{k[:70]:70s} {n:8d} {fb:26.0f} {wb:20.0f} {c:22.0f}")
    with open(os.path.join(root, "pmc_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
