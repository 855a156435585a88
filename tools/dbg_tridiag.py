import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
import torch
from basd_amd import ops, _lib
n, k = 100, 100
g = torch.Generator().manual_seed(n + k)
x = torch.randn(1, 6 * n, n, generator=g)
x[:, :, :24] *= torch.linspace(8.0, 3.0, 24)
G0 = (x.transpose(1, 2) @ x).cuda()
ts = ops.tridiag_eigenvalues(G0.clone())
z = torch.empty((1, k, n), device="cuda"); vecs = torch.empty((1, k, n), device="cuda")
_lib.call("basd_tridiag_eigenvectors", ts.d.data_ptr(), ts.e.data_ptr(), ts.tau.data_ptr(), ts.vh.data_ptr(),
          ts.vals.data_ptr(), n, k, 1, z.data_ptr(), vecs.data_ptr(), k, ops._stream())
torch.cuda.synchronize()
Z = z[0].double().cpu(); V = vecs[0].double().cpu()
T = torch.diag(ts.d[0].double().cpu()) + torch.diag(ts.e[0, :n-1].double().cpu(), 1) + torch.diag(ts.e[0, :n-1].double().cpu(), -1)
print("T eig match:", (torch.linalg.eigvalsh(T).flip(0) - ts.vals[0].double().cpu()).abs().max().item())
zz = Z @ Z.T
print("Z norms min/max", zz.diag().min().item(), zz.diag().max().item())
off = (zz - torch.diag(zz.diag())).abs()
print("Z max offdiag", off.max().item(), "argmax", divmod(off.argmax().item(), k))
res = (T @ Z.T - Z.T * ts.vals[0, :k].double().cpu()).norm(dim=0)
print("T residual max", (res / ts.vals[0, 0].item()).max().item(), "at", res.argmax().item())
vv = V @ V.T
print("V norms min/max", vv.diag().min().item(), vv.diag().max().item(), "max offdiag", (vv - torch.diag(vv.diag())).abs().max().item())
Q = torch.eye(n, dtype=torch.float64)
vh = ts.vh[0].double().cpu(); tau = ts.tau[0].double().cpu()
for j in range(n - 1):
    H = torch.eye(n, dtype=torch.float64) - tau[j] * torch.outer(vh[j], vh[j])
    Q = Q @ H
print("Q orth err", (Q.T @ Q - torch.eye(n, dtype=torch.float64)).abs().max().item())
print("Q^T G Q - T", (Q.T @ G0[0].double().cpu() @ Q - T).abs().max().item() / G0.abs().max().item())
print("V vs Q Z", (V - (Q @ Z.T).T).abs().max().item())
print("vals head", ts.vals[0, :5].tolist(), "tail", ts.vals[0, -5:].tolist())
