"""Stand-alone timing of basd_tridiag (HIP events on the launch stream): shared stage only vs shared + tail stage."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
import torch
from basd_amd import ops, _lib

dev = torch.device("cuda", 0)
for n, batch in [(384, 2), (384, 6), (768, 28), (192, 6)]:
    g = torch.Generator().manual_seed(n)
    x = torch.randn(batch, 4 * n, n, generator=g)
    G0 = (x.transpose(1, 2) @ x).to(dev)
    for tail in (1, 2):
        _lib.call("basd_tridiag_tuning", -1, -1, -1, -1, tail, 1)
        copies = [G0.clone() for _ in range(12)]
        for c in copies[:2]:
            ops.tridiagonalise(c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for c in copies[2:]:
            ts = ops.tridiagonalise(c)
        e1.record()
        torch.cuda.synchronize()
        print(f"n={n} batch={batch} tail={tail}: {e0.elapsed_time(e1) / 10:.3f} ms per factorisation, err {ts.err.tolist()[:2]}", flush=True)
_lib.call("basd_tridiag_tuning", -1, -1, -1, -1, -1, 1)
