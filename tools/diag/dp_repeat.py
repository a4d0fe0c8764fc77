#!/usr/bin/env python3
"""Repeatability of the two-rank (gloo, one GPU) data-parallel runs of tests/test_gpu_two_ranks.py: eager state-mode twice and
the split-graph run, N rounds; prints which pairs differ.   python tools/diag/dp_repeat.py [rounds]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.multiprocessing as mp
import test_gpu_two_ranks as T
os.environ.setdefault("UAPS_TEST_MODE", {"0": "exact", "1": "split"}.get(os.environ.get("UAPS_CONV_MODE", "2"), "h16"))      # the workers of the tests run in exact mode; this tool in the environment's


def run(tmp, use_graph, tag):
    port = T._free_port()
    mp.spawn(T._graph_worker, args=(2, port, tmp, use_graph, STEPS), nprocs=2, join=True)
    out = []
    for r in range(2):
        src = os.path.join(tmp, f"rank{r}_{int(use_graph)}_1.pt")
        out.append(torch.load(src, weights_only=False))
        os.remove(src)
    return out


def diff(a, b):
    bad = []
    for r in range(2):
        ha, hb = a[r].get("hashes"), b[r].get("hashes")
        if ha is not None and hb is not None and not torch.equal(ha, hb):
            step = int((ha != hb).any(dim=1).nonzero()[0])
            cols = (ha[step] != hb[step]).nonzero().flatten().tolist()
            names = a[r]["names"]
            what = [f"{'grad ' if c % 2 else ''}{names[c // 2]}" for c in cols]
            import collections
            tops = collections.Counter(("grad " if c % 2 else "param ") + names[c // 2].split(".")[0] for c in cols)
            prev = "none" if step == 0 else ("params equal" if torch.equal(ha[step - 1][0::2], hb[step - 1][0::2]) else "params differ")
            bad.append(f"rank{r}: first bit difference at step {step} in {len(cols)} tensors {dict(tops)} (step before: {prev})")
        if a[r]["losses"] != b[r]["losses"]:
            first = next(i for i, (x, y) in enumerate(zip(a[r]["losses"], b[r]["losses"])) if x != y)
            rel = [abs(x - y) / max(abs(x), 1e-30) for x, y in zip(a[r]["losses"], b[r]["losses"])]
            bad.append(f"rank{r}: losses differ from step {first} (rel diff there {rel[first]:.2e}, max {max(rel):.2e})")
        n = sum(not torch.equal(v, b[r]["params"][k]) for k, v in a[r]["params"].items())
        if n:
            worst = max(float((v.double() - b[r]["params"][k].double()).abs().max() / (v.double().abs().max() + 1e-30)) for k, v in a[r]["params"].items())
            bad.append(f"rank{r}: {n} of {len(a[r]['params'])} state tensors differ, worst rel {worst:.2e}")
    return bad


STEPS = 8
if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(rounds):
            if os.environ.get("UAPS_DIAG_EAGER_ONLY"):
                e1, e2, e3 = run(tmp, False, "e1"), run(tmp, False, "e2"), run(tmp, False, "e3")
                print(f"round {i}: e1 vs e2: {diff(e1, e2) or 'equal'} | e1 vs e3: {diff(e1, e3) or 'equal'}", flush=True)
                continue
            e1, e2, g = run(tmp, False, "e1"), run(tmp, False, "e2"), run(tmp, True, "g")
            print(f"round {i}: eager vs eager: {diff(e1, e2) or 'equal'} | eager vs graph: {diff(e1, g) or 'equal'} | captured {[x['captured'] for x in g]}", flush=True)
