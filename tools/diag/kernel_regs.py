#!/usr/bin/env python3
"""Register / LDS / scratch use of the kernels of a built library, from the code objects' metadata notes:
python tools/diag/kernel_regs.py [substring] [lib.so]"""
import glob, os, re, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
pat = sys.argv[1] if len(sys.argv) > 1 else ""
so = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "uaps_amd", "lib", "libuaps_hip.so")
tmp = tempfile.mkdtemp(prefix="uaps_regs_")
try:
    shutil.copy(so, os.path.join(tmp, "lib.so"))
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = []
    for o in sorted(glob.glob(os.path.join(tmp, "lib.so.*gfx950*"))):
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", o], stdout=subprocess.PIPE, text=True).stdout
        for b in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
            b = ".agpr_count:" + b
            g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", b) or [None, "?"])[1]
            rows.append((g("name"), g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), stdout=subprocess.PIPE, text=True).stdout.splitlines()
    for (name, v, a, s, l, p), d in zip(rows, names):
        if pat in d:
            print(f"{d.split('(')[0][-78:]:78s} vgpr {v:>3} agpr {a:>3} sgpr {s:>3} lds {l:>6} scratch {p}")
finally:
    shutil.rmtree(tmp, ignore_errors=True)
