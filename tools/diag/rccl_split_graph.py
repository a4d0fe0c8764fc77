#!/usr/bin/env python3
"""Stand-alone check (NOT part of the test suite): the two-graph data-parallel step over the real RCCL backend with one rank and
the buckets' divisor forced to 2.  It passed when run alone (pytest -k) and once ABORTED the whole pytest process when it ran
behind the two other RCCL tests of tests/test_gpu_parity.py (a third init / destroy of a process group in one process, with
captures in between) -- which is why bench.py keeps the eager step at N > 1 unless UAPS_GRAPH_MULTI=1.
  python -m pytest tools/diag/rccl_split_graph.py -q"""
import numpy as np
import pytest
import torch

DEV = "cuda:0"
pytestmark = pytest.mark.gpu


def test_split_graph_over_rccl_single_rank():
    """The data-parallel captured step (uaps_amd/graph.py: forward + loss + backward and Adam + metrics as two hipGraphs, the
    bucket all-reduces issued eagerly between their replays) over the real RCCL backend with one rank and the buckets'
    divisor forced to 2: the process group, its watchdog thread and real collectives are alive around and between the
    captures.  Seven steps (two eager warm-ups, capture, four replays) must leave the parameters bit for bit where the eager
    state-mode step with the same buckets leaves them (fp32 matrix instruction: DESIGN.md section 4 on bit comparisons)."""
    import torch.distributed as dist
    import uaps_amd
    import uaps_amd.unet as unet_mod
    from uaps_amd import conv, dist as udist, perturb
    from uaps_amd.graph import StepGraph
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29657", rank=0, world_size=1, device_id=torch.device(DEV))
    prev_mode = conv.get_mode()
    conv.set_mode("exact")
    unet_mod._DECODER_STREAMS = True
    try:
        t = torch.ones(8, device=DEV)
        dist.all_reduce(t)                                          # communicator and watchdog are up before anything is captured
        torch.cuda.synchronize()
        data = uaps_amd.data.SyntheticBatches(2, 3, 4, 64, 64, n_batches=3, seed=9, device=DEV)
        batches = [data.next() for _ in range(3)]

        def run(capture):
            torch.manual_seed(9)
            model = uaps_amd.net_factory("unet_uaps", 3, 4).to(DEV)
            tr = uaps_amd.UAPSTrainer(model, seed=9)
            tr.buckets = udist.GradBuckets(model)
            tr.buckets.world = 2                                    # divisor of the average; one rank contributes the sum
            for bi, params in enumerate(tr.buckets.buckets):        # the hooks the world > 1 constructor registers
                for p in params:
                    tr.buckets._hooks.append(p.register_post_accumulate_grad_hook(tr.buckets._make_hook(bi)))
            tr.buckets.reset()
            tr.step_graph = StepGraph(tr, capture=capture)
            assert tr.step_graph.split
            np.random.seed(9); perturb.manual_seed(9)
            for i in range(7):
                tr.train_step(*batches[i % 3])
            torch.cuda.synchronize()
            assert (tr.step_graph.graph is not None and tr.step_graph.graph_tail is not None) == capture
            params = [p.detach().clone() for p in model.parameters()]
            tr.buckets.remove()
            return params

        eager, graph = run(False), run(True)
        for a, b in zip(eager, graph):
            assert torch.equal(a, b)
    finally:
        unet_mod._DECODER_STREAMS = False
        conv.set_mode(prev_mode)
        dist.destroy_process_group()
