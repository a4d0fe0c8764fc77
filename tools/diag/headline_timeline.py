#!/usr/bin/env python3
"""Timeline of the HEADLINE step (hipGraph replay, decoder streams) from a rocprofv3 kernel trace: per step, the wall time between its
first and last kernel, the time no kernel runs, the time exactly one kernel runs, and which kernels run alone the longest.
  python tools/diag/headline_timeline.py <kernel_trace.csv> [steps to skip]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("uaps::", "")) for r in rows))
# a step starts at cat2_amax_kernel (the first kernel of forward_pair)
starts = [i for i, k in enumerate(ks) if k[2].startswith("cat2_amax_kernel")]
print(f"{len(ks)} kernels, {len(starts)} steps in the trace")
for si in range(skip, min(len(starts) - 1, skip + 3)):
    seg = ks[starts[si]:starts[si + 1]]
    t0, t1 = seg[0][0], max(e for _, e, _ in seg)
    ev = sorted([(s, 1, n) for s, e, n in seg] + [(e, -1, n) for s, e, n in seg])
    active, last = 0, t0
    idle = alone = 0
    alone_by = collections.Counter()
    cur = []
    for t, d, n in ev:
        dt = t - last
        if active == 0: idle += dt
        elif active == 1: alone += dt; alone_by[cur[0]] += dt
        last = t
        if d == 1: active += 1; cur.append(n)
        else: active -= 1; cur.remove(n)
    busy = sum(e - s for s, e, _ in seg)
    print(f"step {si}: wall {(t1 - t0) / 1e6:.3f} ms, {len(seg)} kernels, sum of kernel durations {busy / 1e6:.3f} ms, idle {idle / 1e6:.3f} ms, exactly one kernel running {alone / 1e6:.3f} ms")
    print("   longest alone: " + ", ".join(f"{k[:40]} {v / 1e3:.0f} us" for k, v in alone_by.most_common(12)))
    # phases by marker kernels: encoder forward (.. last fan-out), decoders forward (.. loss forward), decoders backward (loss backward
    # .. first fan-in), encoder backward + optimizer (first fan-in .. end)
    def first(name, lo=0): return next((k for k in seg if k[2].startswith(name) and k[0] >= lo), None)
    def last(name): return next((k for k in reversed(seg) if k[2].startswith(name)), None)
    # the decoders begin with the 1x1 projection of the deepest feature (the fan-out kernels run beside the encoder since round 5)
    fo, pf, pb, fi = first("conv_fwd_kernel<1, 16, 16"), first("pair_fwd"), first("pair_bwd"), first("fanin_perturbed")
    if fo is None:
        fo = last("fanout_perturbed")
    else:
        fo = (fo[0], fo[0], fo[2])
    if fo and pf and pb and fi:
        marks = [("encoder forward", t0, fo[1]), ("decoders forward", fo[1], pf[0]), ("loss", pf[0], pb[1]), ("decoders backward", pb[1], fi[0]),
                 ("encoder backward + optimizer", fi[0], t1)]
        out = []
        for name, a, b in marks:
            inside = [(max(s_, a), min(e_, b)) for s_, e_, _ in seg if e_ > a and s_ < b]
            out.append(f"{name} {(b - a) / 1e3:.0f} us ({len([1 for s_, e_, _ in seg if a <= s_ < b])} launches, kernel time {sum(e_ - s_ for s_, e_ in inside) / 1e3:.0f})")
        print("   phases: " + "; ".join(out))
    # how much of the step is a short kernel (< 15 us) running alone, or nothing running: the latency-bound part
    short_alone = sum(v for k, v in alone_by.items() if False) if False else 0
    evs = sorted([(s_, 1, i) for i, (s_, e_, n_) in enumerate(seg)] + [(e_, -1, i) for i, (s_, e_, n_) in enumerate(seg)])
    act, last_t, by_len = set(), t0, collections.Counter()
    for t, d_, i in evs:
        if len(act) == 1:
            j = next(iter(act)); dur = seg[j][1] - seg[j][0]
            by_len["< 8 us" if dur < 8e3 else "8-20 us" if dur < 20e3 else "20-50 us" if dur < 50e3 else ">= 50 us"] += t - last_t
        last_t = t
        if d_ == 1: act.add(i)
        else: act.discard(i)
    print("   alone time by the lone kernel's duration: " + ", ".join(f"{k} {v / 1e3:.0f} us" for k, v in sorted(by_len.items())))
    # the first 70 launches in start order (encoder forward and the head of the decoders): start offset, duration, kernel
    if si == skip:
        for s_, e_, n_ in seg[:70]:
            print(f"      +{(s_ - t0) / 1e3:8.1f} us {(e_ - s_) / 1e3:7.1f} us  {n_[:60]}")
