import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from uaps_amd import conv, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
L.uaps_diag_read.argtypes = [C.c_void_p, C.c_size_t]
for (Cin, Cout, HW) in [(16, 16, 256), (32, 16, 256), (64, 64, 64), (128, 128, 32)]:
    x = torch.randn(32, Cin, HW, HW, device=dev); dy = torch.randn(32, Cout, HW, HW, device=dev)
    for _ in range(3): conv.conv_bwd_weight_raw(dy, x, 3, False)
    torch.cuda.synchronize()
    n = 1024 * 4 * 8
    buf = np.zeros(n, dtype=np.uint64)
    rc = L.uaps_diag_read(buf.ctypes.data, n)
    d = buf.reshape(-1, 8)
    d = d[d[:, 5] > 0]
    names = ["issue_loads", "mfma", "barrier1", "store", "barrier2", "total", "tiles"]
    tot = d[:, 5].astype(np.float64)
    print(f"wrw {Cin}->{Cout}@{HW}: waves {len(d)}, tiles/wave {d[:,6].mean():.1f}, total cycles/wave median {np.median(tot):.0f}")
    for i in range(5):
        print(f"    {names[i]:12s} {100 * np.median(d[:, i] / tot):5.1f} %   ({np.median(d[:, i] / np.maximum(d[:, 6], 1)):.0f} cycles/tile)")
