"""Diagnostic: sweeps / accuracy of the symmetric eigen-solve versus the stopping tolerance."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
import torch
from basd_amd import ops, synth
shape = synth.CONFIGS["cfg2"]
inp = synth.make_inputs(shape, 1234, device="cuda:0", strided=True)
x = inp.student[0]
mean = ops.colmean(x)
g = ops.gemm_tn(x, x, mean_a=mean, mean_b=mean).unsqueeze(0)
ref = torch.linalg.eigvalsh(g[0].double().cpu()).flip(0)
refv = torch.linalg.eigh(g[0].double().cpu())[1].flip(1)[:, :48]
for tol in (0.0, 1e-5, 1e-4, 1e-3):
    gg = g.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cn, sw = ops.jacobi_onesided(gg, gg.shape[1], want_sweeps=True, tol=tol)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    vals, vecs = ops.sort_extract(gg, cn, 48)
    err = ((vals[0].double().cpu() - ref).abs() / ref[0]).max().item()
    v = vecs[0].double().cpu().T
    perr = (v @ v.T - refv @ refv.T).abs().max().item()
    print(f"tol {tol:g}: {1e3*(t1-t0):.2f} ms, eig err/lmax {err:.2e}, top-48 projector err {perr:.2e}")
