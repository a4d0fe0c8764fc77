#!/usr/bin/env python3
"""Outline of a kernel's instruction stream: runs of global loads / stores, MFMAs, LDS stores, barriers and every s_waitcnt vmcnt,
in program order -- shows at a glance whether a prefetch is really left in flight across the matrix loop.
  python tools/diag/isa_outline.py <mangled-name substring> [lib.so]"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isa_lint import disassemble
pat = sys.argv[1]
so = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "uaps_amd", "lib", "libuaps_hip.so")
KIND = [("gload", re.compile(r"\b(buffer_load|global_load|flat_load)")), ("gstore", re.compile(r"\b(buffer_store|global_store|flat_store)")),
        ("mfma", re.compile(r"\bv_mfma")), ("ds_write", re.compile(r"\bds_write|\bds_store")), ("barrier", re.compile(r"\bs_barrier")),
        ("vmcnt", re.compile(r"s_waitcnt.*vmcnt\((\d+)\)")), ("branch", re.compile(r"\bs_cbranch|\bs_branch")), ("readfirstlane", re.compile(r"v_readfirstlane"))]
cur, runs, n = None, [], 0
for fn, line in disassemble(so):
    if pat not in fn:
        continue
    if fn != cur:
        if runs: print("  " + " | ".join(runs)); runs = []
        print(f"== {fn}"); cur = fn; last = None
    for k, rx in KIND:
        m = rx.search(line)
        if m:
            lab = f"vmcnt({m.group(1)})" if k == "vmcnt" else k
            if runs and runs[-1].split(" x")[0] == lab:
                c = int(runs[-1].split(" x")[1]) if " x" in runs[-1] else 1
                runs[-1] = f"{lab} x{c + 1}"
            else:
                runs.append(lab)
            break
if runs: print("  " + " | ".join(runs))
