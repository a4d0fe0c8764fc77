import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from uaps_amd import _lib, conv
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
dev = torch.device("cuda:0"); L = _lib.lib(); st = _lib.current_stream(dev); B = 32
# warm the clocks
a = torch.randn(4096, 4096, device=dev)
for _ in range(50): a @ a
for Cin, Cout, HW, ks in [(16,16,256,3),(32,16,256,3),(32,32,128,3),(64,32,128,3)]:
    x = torch.randn(B, Cin, HW, HW, device=dev); dy = torch.randn(B, Cout, HW, HW, device=dev)
    res = []
    for rep in range(2):
        for ns in (0, 256, 512, 768, 1024, 2048, 4096):
            n = C.c_size_t(); L.uaps_conv_wrw_workspace_bytes(B, Cin, Cout, HW, HW, ks, ns, C.byref(n))
            ws = torch.empty(n.value, dtype=torch.uint8, device=dev)
            t = timeit(lambda: L.uaps_conv_bwd_weight_partial(dy.data_ptr(), x.data_ptr(), 0, B, Cin, Cout, HW, HW, ks, ns, ws.data_ptr(), ws.numel(), st))
            res.append(f"{ns}:{t:.1f}")
    print(f"{Cin}->{Cout}@{HW}: " + " ".join(res), flush=True)
