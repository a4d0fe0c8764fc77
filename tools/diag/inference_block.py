"""bench.py's inference block alone (row f-2: main head / ensemble ms per image, eager and as a replayed hipGraph).  GPU box."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench

print(json.dumps(bench.inference_block(torch.device("cuda:0")), indent=1))
