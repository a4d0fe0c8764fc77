#!/usr/bin/env python3
"""Two rocprofv3 kernel_stats.csv files side by side: python tools/diag/prof_diff.py A.csv B.csv [steps] [name filter]
(us per step and average us per launch of every kernel in either file, and the totals)."""
import csv, re, sys

def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        n = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("uaps::", "")
        c, t = d.get(n, (0, 0.0))
        d[n] = (c + int(r["Calls"]), t + float(r["TotalDurationNs"]))
    return d

a, b = load(sys.argv[1]), load(sys.argv[2])
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 13.0
filt = sys.argv[4] if len(sys.argv) > 4 else ""
names = sorted(set(a) | set(b), key=lambda n: -(a.get(n, (0, 0))[1] + b.get(n, (0, 0))[1]))
print(f"{'kernel':58s} {'A calls':>7s} {'A us/step':>10s} {'A avg':>8s} | {'B calls':>7s} {'B us/step':>10s} {'B avg':>8s}")
ta = tb = 0.0
for n in names:
    ca, xa = a.get(n, (0, 0.0)); cb, xb = b.get(n, (0, 0.0))
    ta += xa; tb += xb
    if filt and filt not in n:
        continue
    print(f"{n[:58]:58s} {ca / steps:7.1f} {xa / steps / 1e3:10.1f} {xa / max(ca, 1) / 1e3:8.1f} | {cb / steps:7.1f} {xb / steps / 1e3:10.1f} {xb / max(cb, 1) / 1e3:8.1f}")
print(f"total kernel time per step: A {ta / steps / 1e6:.3f} ms, B {tb / steps / 1e6:.3f} ms")
