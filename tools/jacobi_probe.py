"""Probe: LDS Jacobi with 4 vs 16 lanes per pair on the same batch: orthogonality reached, sweeps, time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
import torch
from basd_amd import ops, _lib
dev = "cuda:0"
for n, rd, rt, batch, graded in [(49, 49, 98, 1024, False), (49, 49, 98, 1024, True), (36, 36, 72, 512, False)]:
    g = torch.Generator().manual_seed(n)
    w0 = torch.randn(batch, n, rt, generator=g)
    if graded:
        w0[:, :, :rd] *= torch.logspace(0, -3, n).view(1, n, 1)
    for lanes in (16, 8, 4):
        _lib.call("basd_jacobi_tuning", lanes)
        W = w0.clone().to(dev)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sigma, sweeps = ops.jacobi_onesided(W, rd, want_sweeps=True)
        e1.record()
        torch.cuda.synchronize()
        top = W[:, :, :rd].double().cpu()
        gram = top @ top.transpose(1, 2)
        off = gram - torch.diag_embed(torch.diagonal(gram, dim1=1, dim2=2))
        nrm = torch.diagonal(gram, dim1=1, dim2=2).sqrt()
        cos = (off.abs() / (nrm.unsqueeze(2) * nrm.unsqueeze(1)).clamp_min(1e-30)).amax(dim=(1, 2))
        sv = torch.linalg.svdvals(w0[:, :, :rd].double())
        got = sigma.double().cpu().sort(dim=1, descending=True).values
        print(f"n={n} batch={batch} graded={graded} lanes={lanes}: {e0.elapsed_time(e1):.3f} ms, sweeps mean {sweeps.float().mean():.2f} max {int(sweeps.max())}, "
              f"cos max {cos.max():.2e} (#>5e-6: {(cos > 5e-6).sum().item()}), sv err {((got - sv).abs().amax(1) / sv[:, 0]).max():.2e}", flush=True)
_lib.call("basd_jacobi_tuning", 0)
