"""Diagnostic: stand-alone timings of the selector's contraction kernels at the cfg-2 shapes."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "vit-inductive-bias-distillation_amd"))
from basd_amd import ops, _lib

dev = torch.device("cuda:0")
B, N, D, DT, NT = 256, 196, 384, 2048, 49
xs = [torch.randn(B, N + 1, D, device=dev)[:, 1:, :] for _ in range(4)]
t = torch.randn(B, DT, NT, device=dev).transpose(1, 2)          # channel-major teacher
proj = torch.randn(D, DT, device=dev) / 45.0
z = ops.gemm_nt(t, proj)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print("student centered_grams x4 : %.1f us" % timeit(lambda: ops.centered_grams(xs)))
print("teacher centered_grams x2 : %.1f us" % timeit(lambda: ops.centered_grams([z, z], centered=[False, True], scales=[1.0 / z.shape[0], 1.0])))
print("teacher projection gemm_nt: %.1f us" % timeit(lambda: ops.gemm_nt(t, proj)))
tc = t.contiguous()
print("  (token-major teacher)   : %.1f us" % timeit(lambda: ops.gemm_nt(tc, proj)))
print("gemm_tn one student layer : %.1f us" % timeit(lambda: ops.gemm_tn(xs[0], xs[0])))
print("colmean one student layer : %.1f us" % timeit(lambda: ops.colmean(xs[0])))

for s in (8, 16, 24, 32, 40, 48, 64):
    print("student syrk x4 splits=%d : %.1f us" % (s, timeit(lambda: ops.centered_grams(xs, splits=s))))
print("default splits:", _lib.query("basd_syrk_splits", B * N, D, 4), _lib.query("basd_syrk_splits", B * NT, D, 2),
      _lib.query("basd_syrk_splits", 128 * 197, 768, 48))
