#!/usr/bin/env python3
"""Child process of tests/test_gpu_parity.py::test_grad_buckets_over_rccl_single_rank (not collected by pytest: no test_ prefix).
  python tests/rccl_single_rank_check.py <0|1 decoder streams>   -> prints RCCL_CHECK_OK, exit code 0"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

DEV = "cuda:0"


class _Skip(Exception):
    pass


class pytest:                                   # the body below was a pytest test: its one pytest call
    @staticmethod
    def skip(msg):
        raise _Skip(msg)


def check(streams):
    import torch.distributed as dist
    import uaps_amd
    import uaps_amd.unet as unet_mod
    from uaps_amd import dist as udist, perturb
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    # with decoder streams the bucket hooks fire on the auxiliary decoders' side streams: the all-reduce must order itself
    # behind the stream that produced the bucket's gradients and the main stream behind the all-reduce
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{29653 + int(streams)}", rank=0, world_size=1, device_id=torch.device(DEV))
    unet_mod._DECODER_STREAMS = streams
    try:
        def grads(with_buckets):
            torch.manual_seed(9); np.random.seed(9); perturb.manual_seed(9)
            model = uaps_amd.net_factory("unet_uaps", 3, 4).to(DEV)
            data = uaps_amd.data.SyntheticBatches(2, 3, 4, 32, 32, n_batches=1, seed=9, device=DEV)
            x_l, y_l, x_u = data.next()
            buckets = None
            if with_buckets:
                buckets = udist.GradBuckets(model)
                buckets.world = 2                                   # divisor of the average; one rank contributes the sum
                for bi, params in enumerate(buckets.buckets):       # register the hooks the world > 1 constructor would
                    for p in params:
                        buckets._hooks.append(p.register_post_accumulate_grad_hook(buckets._make_hook(bi)))
                buckets.reset()
            both = model.forward_pair(x_l, x_u)
            out = uaps_amd.uaps_pair_loss(both, y_l, np.full(4, 0.25), 0.1, 0.1)
            out.loss.backward()
            if buckets is not None:
                buckets.finish()
                for bi, params in enumerate(buckets.buckets):       # the kernels wrote into the flat buffers, RCCL reduced in place
                    for k, p in enumerate(params):
                        assert p.grad.data_ptr() == buckets._view(bi, k).data_ptr()
            torch.cuda.synchronize()
            out = [p.grad.clone() for p in model.parameters()]
            if buckets is not None:
                buckets.remove()
            return out

        ref, got = grads(False), grads(True)
        assert len(ref) == len(got) == 208
        for a, b in zip(ref, got):
            assert torch.equal(b, a * 0.5)
    finally:
        unet_mod._DECODER_STREAMS = False
        dist.destroy_process_group()



if __name__ == "__main__":
    check(bool(int(sys.argv[1])))
    print("RCCL_CHECK_OK")
