#!/usr/bin/env python3
"""Disassembly of one kernel of the built library: python tools/diag/disasm.py <substring of the mangled name> [lib.so] > out.s
(also prints the .vgpr_count / LDS notes found in the code object's metadata)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isa_lint import disassemble
pat = sys.argv[1]
so = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "uaps_amd", "lib", "libuaps_hip.so")
cur = None
for fn, line in disassemble(so):
    if pat in fn:
        if fn != cur:
            print(f"\n;;;; {fn}")
            cur = fn
        print(line)
