"""Host cost of one training step and small-batch throughput: eager (by-value arguments), state mode (uaps_amd/graph.py, eager)
and the captured hipGraph.  `python tools/host_small.py` on the GPU box; prints one line per mode and shape."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, uaps_amd
from uaps_amd import unet

dev = torch.device("cuda:0")


def run(mode, B, H, streams, steps=40):
    unet._DECODER_STREAMS = streams
    torch.manual_seed(0)
    model = uaps_amd.net_factory("unet_uaps", 3, 4).to(dev)
    kw = {"eager": {}, "state": {"step_state": True}, "graph": {"use_graph": True}}[mode]
    tr = uaps_amd.UAPSTrainer(model, seed=1337, **kw)
    data = uaps_amd.data.SyntheticBatches(B, 3, 4, H, H, n_batches=2, device=dev)
    for _ in range(6):
        tr.train_step(*data.next())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(steps):
        h0 = time.perf_counter()
        tr.train_step(*data.next())
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{mode:6s} streams={int(streams)} {B}+{B} @ {H}x{H}: {1e3 * dt:7.2f} ms/step  host {1e3 * host / steps:6.2f} ms/step  "
          f"{2 * B / dt:8.1f} img/s  loss {float(tr.last['loss']):.4f}", flush=True)


for B, H in ((2, 32), (4, 256), (16, 256)):
    for mode, streams in (("eager", False), ("eager", True), ("state", False), ("graph", False), ("graph", True)):
        run(mode, B, H, streams)
