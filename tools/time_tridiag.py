import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
import torch
from basd_amd import ops, _lib
for n, batch, k in ((384, 6, 48), (768, 52, 80)):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(1, 4 * n, n, generator=g)
    x[:, :, :32] *= 5.0
    G0 = (x.transpose(1, 2) @ x).cuda().repeat(batch, 1, 1).contiguous()
    for rep in range(3):
        G = G0.clone()
        _lib.timing = {}; _lib.timed_names = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ts = ops.tridiag_eigenvalues(G)
        vecs = ops.tridiag_eigenvectors(ts, k)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        tm = {kk: sum(a.elapsed_time(b) for a, b in v) for kk, v in _lib.timing.items()}
        _lib.timing = None
    print(n, batch, k, "total %.2f ms" % (1e3 * (t1 - t0)), {kk: round(v, 3) for kk, v in tm.items()})
