"""Which step of the headline protocol triggers PyTorch's "AccumulateGrad node's stream does not match" warning?  (round 6)
Runs the bench's trainer (decoder streams, captured graph) with that warning turned into an error and reports the step index."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import uaps_amd
import uaps_amd.unet as _unet

_unet._DECODER_STREAMS = os.environ.get("PROBE_STREAMS", "1") != "0"
dev = torch.device("cuda:0")
torch.manual_seed(1337)
model = uaps_amd.net_factory("unet_uaps", 3, 4, n_aux=3)
tr = uaps_amd.UAPSTrainer(model, seed=1337, use_graph=os.environ.get("PROBE_GRAPH", "1") != "0")
data = uaps_amd.data.SyntheticBatches(4, 3, 4, 64, 64, n_batches=2, seed=1337, device=dev)
warnings.filterwarnings("error", message=".*AccumulateGrad node's stream.*")
for i in range(6):
    try:
        tr.train_step(*data.next())
        torch.cuda.synchronize()
        print("step", i, "ok", "graph" if (tr.step_graph is not None and tr.step_graph.graph is not None) else "eager", flush=True)
    except Exception as e:
        print("step", i, "RAISED", type(e).__name__, str(e)[:300].replace("\n", " "), flush=True)
        break
