"""Round-5 glue A/B on one box: (a) uaps_up_cat_bwd at the decoder's three tall levels (run the script twice: with and without
UAPS_DIAG_UPBWD_NARROW=1 for the 32-column tiles), (b) the class head's input gradient 4 -> 16 @ 256 x 256 on the row kernel
against the fp32-instruction tile kernel (tuning bit NO_ROW16).  GPU box: python3 tools/diag/glue_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from uaps_amd import _lib, bounds, conv

dev = torch.device("cuda:0")
L = _lib.lib()
st = _lib.current_stream(dev)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print("up_cat_bwd tiles:", "32 columns (UAPS_DIAG_UPBWD_NARROW)" if os.environ.get("UAPS_DIAG_UPBWD_NARROW") else "64 columns where w % 64 == 0")
for B, Cl, h in ((32, 16, 128), (32, 32, 64), (32, 64, 32)):
    dout = torch.randn(B, Cl, 2 * h, 2 * h, device=dev)
    dlow = torch.empty(B, Cl, h, h, device=dev)
    t = timeit(lambda: L.uaps_up_cat_bwd(dout.data_ptr(), None, dlow.data_ptr(), B, 0, Cl, h, h, st))
    mb = (dout.numel() + dlow.numel()) * 4 / 1e6
    print(f"  up_cat_bwd B={B} Cl={Cl} {h}^2 -> {2 * h}^2: {t:7.1f} us  {mb / t / 1e3 * 1e3:6.2f} GB/s x1e3 ({mb:.0f} MB)")

B, Cin, Cout, H, W = 32, 16, 4, 256, 256          # the class head; its input gradient is the 4 -> 16 direction
dy = torch.randn(B, Cout, H, W, device=dev)
w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.1
wf, wb = conv.pack_weights(w)
dyb = (bounds.from_value(dy.abs().max()), 1.0)
prev = L.uaps_conv_get_tuning()
for name, flags in (("row kernel", prev), ("fp32-instruction tile kernel (NO_ROW16)", prev | 128)):
    L.uaps_conv_set_tuning(flags)
    t = timeit(lambda: conv.conv_bwd_data_raw(dy, wb, Cin, 3, 0, dyb=dyb))
    mb = (dy.numel() + B * Cin * H * W) * 4 / 1e6
    print(f"class head input gradient 4 -> 16 @256^2 B=32, {name}: {t:7.1f} us  {mb / t:6.2f} TB/s ({mb:.0f} MB)")
L.uaps_conv_set_tuning(prev)
