import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, ctypes as C
from uaps_amd import conv, _lib
dev = torch.device("cuda:0")
for (Cout, Cin) in ((16, 3), (16, 16), (64, 32)):
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.2
    wf, wb = conv.pack_weights(w)
    torch.cuda.synchronize()
    L = _lib.lib()
    # offsets
    def pad_k(c): return 4 if c <= 4 else (c + 7) // 8 * 8
    def pad_n(c): return (c + 15) // 16 * 16
    def cg(c): return ((c + 7) // 8 + 3) // 4 * 4
    nf = 9 * pad_k(Cin) * pad_n(Cout); nsf = 9 * cg(Cin) * pad_n(Cout) * 4
    nbk = 9 * pad_k(Cout) * pad_n(Cin); nsb = 9 * cg(Cout) * pad_n(Cin) * 4
    print(Cout, Cin, "amax", float(w.abs().max()), "wf numel", wf.numel(), nf + 5 * nsf + 32, "wb numel", wb.numel(), nbk + 5 * nsb + 32)
    print("  fwd hdr", wf[nf + 3 * nsf: nf + 3 * nsf + 20].tolist())
    print("  bwd hdr", wb[nbk + 3 * nsb: nbk + 3 * nsb + 20].tolist())
