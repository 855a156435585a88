"""One launch of the Procrustes-shaped LDS Jacobi (1024 x (98 x 49)) for PMC collection."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
import torch
from basd_amd import ops, _lib
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = torch.Generator().manual_seed(49)
w0 = torch.randn(1024, 49, 98, generator=g)
_lib.call("basd_jacobi_tuning", lanes)
for _ in range(3):
    W = w0.clone().to("cuda:0")
    ops.jacobi_onesided(W, 49)
torch.cuda.synchronize()
