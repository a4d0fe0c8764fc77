import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, uaps_amd
DEV = "cuda:0"
def batches(n, B, H, W, seed):
    rng = np.random.default_rng(seed); out = []
    for _ in range(n):
        xl = torch.tensor(rng.standard_normal((B, 3, H, W)).astype(np.float32)).to(DEV)
        xu = torch.tensor(rng.standard_normal((B, 3, H, W)).astype(np.float32)).to(DEV)
        y = torch.tensor(uaps_amd.data.synthetic_masks(rng, B, 4, H, W)).to(DEV)
        out.append((xl, y, xu))
    return out
torch.manual_seed(8)
m0 = uaps_amd.UNet_UAPS(3, 4, feature_chns=[8, 16, 16, 32, 32]); m1 = copy.deepcopy(m0)
m0.to(DEV); m1.to(DEV)
eager = uaps_amd.UAPSTrainer(m0, base_lr=1e-3, seed=5, step_state=True)
graph = uaps_amd.UAPSTrainer(m1, base_lr=1e-3, seed=5, use_graph=True)
val = [(b[0], b[1]) for b in batches(2, 2, 64, 64, 22)]
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
for i, (xl, y, xu) in enumerate(batches(7, 2, 64, 64, 21)):
    for tr in (eager, graph):
        uaps_amd.perturb.manual_seed(5, 0); np.random.seed(5)
        tr.train_step(xl, y, xu)
    torch.cuda.synchronize()
    nd = sum(int(not torch.equal(a, b)) for a, b in zip(m0.parameters(), m1.parameters()))
    nb = sum(int(not torch.equal(a, b)) for a, b in zip(m0.buffers(), m1.buffers()))
    se, sg = eager.step_graph.state, graph.step_graph.state
    same_state = torch.equal(se.dev, sg.dev)
    host_ok = torch.equal(sg.dev.cpu(), sg.hosts[sg.slot])
    print(f"step {i}: differing params {nd} buffers {nb}  loss {float(eager.last['loss']):.6f} {float(graph.last['loss']):.6f}  state equal {same_state} dev==host {host_ok} sup {float(eager.last['sup']):.6f} {float(graph.last['sup']):.6f} unsup {float(eager.last['unsup']):.6f} {float(graph.last['unsup']):.6f}", flush=True)
    if nd and not globals().get("_shown"):
        globals()["_shown"] = True
        names = [n for (n, a), b in zip(m0.named_parameters(), m1.parameters()) if not torch.equal(a, b)]
        import collections
        print("   first differing step", i, collections.Counter(n.split(".")[0] + "." + n.split(".")[1] for n in names))
        bn = [n for (n, a), b in zip(m0.named_buffers(), m1.buffers()) if not torch.equal(a, b)]
        print("   buffers:", bn[:12])
    if i in (3, 5):
        if mode in ("both", "eager"): eager.validate(val)
        if mode in ("both", "graph"): graph.validate(val)
        nd = sum(int(not torch.equal(a, b)) for a, b in zip(m0.parameters(), m1.parameters()))
        print(f"   after validate: differing params {nd}")
import hashlib
def digest(m):
    h = hashlib.sha1()
    for p_ in m.parameters(): h.update(p_.detach().cpu().numpy().tobytes())
    return h.hexdigest()[:12]
print("digest eager", digest(m0), "graph", digest(m1))
