#!/usr/bin/env python3
"""Does time-sharing the GPU with another process change results?  `ref`: run the 20 steps of tools/diag/sp_repeat.py once, alone,
and save the bit hashes; `check N`: run them N times and compare (start two of these at once to share the card).
  python tools/diag/share_repeat.py ref /tmp/ref.pt ; python tools/diag/share_repeat.py check 8 /tmp/ref.pt & (x2)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools", "diag"))
import torch
import sp_repeat as S

if __name__ == "__main__":
    kw = {"step_state": True} if os.environ.get("UAPS_DIAG_KW", "state") == "state" else {}
    if sys.argv[1] == "ref":
        torch.save(S.run(kw, 20), sys.argv[2])
    else:
        n, ref = int(sys.argv[2]), torch.load(sys.argv[3])
        bad = []
        for i in range(n):
            h = S.run(kw, 20)
            if not torch.equal(ref, h):
                bad.append((i, int((ref != h).any(dim=1).nonzero()[0])))
        print(f"pid {os.getpid()}: {n} runs while sharing the GPU, differing from the solo reference: {bad or 'none'}", flush=True)
