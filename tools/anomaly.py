import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
import torch
from basd_amd import ops, synth
torch.manual_seed(0)
x = torch.randn(2000, 384, device="cuda:0")
g = (x.T @ x).unsqueeze(0)
for rep in range(8):
    gg = g.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cn = ops.jacobi_onesided(gg, 384)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"rep {rep}: enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms")
