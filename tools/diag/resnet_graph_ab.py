"""configs[4] per-GPU shape (ResNet-50 encoder, K = 3, 640 x 640, 8 + 8): eager against the captured hipGraph of the step.
GPU box: python3 tools/diag/resnet_graph_ab.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import uaps_amd
import uaps_amd.unet as _unet


def run(use_graph, steps, streams=True):
    _unet._DECODER_STREAMS = streams
    dev = torch.device("cuda", 0)
    torch.manual_seed(1337)
    model = uaps_amd.net_factory("resnet50_uaps", 3, 2, n_aux=3)
    tr = uaps_amd.UAPSTrainer(model, seed=1337, use_graph=use_graph)
    data = uaps_amd.data.SyntheticBatches(8, 3, 2, 640, 640, n_batches=2, seed=1337, device=dev)
    for _ in range(4 if use_graph else 2):
        tr.train_step(*data.next())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        tr.train_step(*data.next())
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    tr.check_errors()
    g = tr.step_graph is not None and tr.step_graph.graph is not None
    print(f"use_graph={use_graph} streams={streams} captured={g}: {e0.elapsed_time(e1) / steps:.2f} ms/step (wall {wall:.2f}), loss {float(tr.last['loss']):.5f}", flush=True)
    del tr, model, data
    torch.cuda.empty_cache()


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    for ug, st in ((False, True), (True, True), (False, False), (True, False)):
        try:
            run(ug, steps, st)
        except Exception as exc:       # the experiment reports, it does not decide
            print(f"use_graph={ug} streams={st}: FAILED {type(exc).__name__}: {exc}", flush=True)
