"""Do an HBM-bound and a matrix-bound convolution launch overlap when they run on two streams?  T(A || B) against T(A) + T(B) and
max(T(A), T(B)) for pairs of the step's layers at B = 32 (python tools/diag/corun_probe.py).  The headline mode runs the four
decoders on four streams in lock-step: all of them are in their HBM-bound 256^2 layers, or all in their matrix-bound mid-level
layers, at the same time; this probe measures what staggering them could give."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from uaps_amd import bounds, conv  # noqa: E402

dev = torch.device("cuda:0")
B = 32
torch.manual_seed(0)


def layer(Cin, Cout, HW, reps):
    x = torch.randn(B, Cin, HW, HW, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    wf, _ = conv.pack_weights(w)
    xb = (bounds.from_value(x.abs().max()), 1.0)

    def run():
        for _ in range(reps):
            conv.conv_fwd_raw(x, wf, None, Cout, 3, 0, xb=xb)
    return run


def timed(fns, streams, iters=5):
    for f, s in zip(fns, streams):
        with torch.cuda.stream(s):
            f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for s in streams:
            s.wait_event(e0)
        for f, s in zip(fns, streams):
            with torch.cuda.stream(s):
                f()
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
R = 8
hbm = {"16->16 @256 (hr16<2>)": layer(16, 16, 256, R), "32->16 @256 (hr16<4>)": layer(32, 16, 256, R)}
mfma = {"64->64 @64 (h32<64>)": layer(64, 64, 64, 2 * R), "128->128 @32 (h32<64>)": layer(128, 128, 32, 2 * R), "32->32 @128 (h32t<32>)": layer(32, 32, 128, R)}
print(f"{'pair':60s} {'T(A)':>8s} {'T(B)':>8s} {'A||B':>8s} {'sum':>8s} {'max':>8s}  overlap")
for na, fa in hbm.items():
    for nb, fb in list(mfma.items()) + [(k, v) for k, v in hbm.items() if k != na]:
        ta, tb = timed([fa], [s1]), timed([fb], [s2])
        tab = timed([fa, fb], [s1, s2])
        ov = (ta + tb - tab) / min(ta, tb)
        print(f"{na + ' || ' + nb:60s} {ta:8.0f} {tb:8.0f} {tab:8.0f} {ta + tb:8.0f} {max(ta, tb):8.0f}  {ov:5.2f}")
for na, fa in mfma.items():
    ta = timed([fa], [s1])
    tab = timed([fa, fa], [s1, s2])
    print(f"{na + ' || itself':60s} {ta:8.0f} {ta:8.0f} {tab:8.0f} {2 * ta:8.0f} {ta:8.0f}  {(2 * ta - tab) / ta:5.2f}")
