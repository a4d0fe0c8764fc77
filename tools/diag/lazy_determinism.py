import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch
import test_gpu_lazy_bn as T
for shape in ((32, 32, 32, 64, 4, True), (32, 32, 128, 128, 8, False), (64, 64, 64, 64, 8, True), (16, 16, 32, 256, 4, True), (32, 32, 128, 128, 32, True)):
    ref = None
    for rep in range(4):
        g = T._block(7, *shape[:5], True, shape[5])
        torch.cuda.synchronize()
        if ref is None:
            ref = [t.clone() for t in g]
        else:
            bad = [i for i, (a, b) in enumerate(zip(g, ref)) if not torch.equal(a, b)]
            print(shape, "rep", rep, "differs in", bad, [float((g[i] - ref[i]).abs().max() / ref[i].abs().max()) for i in bad])
print("done")
