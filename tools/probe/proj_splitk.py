"""Probe: teacher projection (12544 x 2048 -> 384, channel-major A) as one GEMM vs split over K through the batch
dimension (2 / 4 slabs + a sum)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "vit-inductive-bias-distillation_amd"))
from basd_amd import ops
dev = torch.device("cuda:0")
B, D, DT, NT = 256, 384, 2048, 49
t = torch.randn(B, DT, NT, device=dev).transpose(1, 2)
proj = (torch.randn(D, DT, device=dev) / 45.0).contiguous()


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


ref = ops.gemm_nt(t, proj)
print("one launch: %.1f us" % timeit(lambda: ops.gemm_nt(t, proj)))
for S in (2, 4):
    kc = DT // S
    def split():
        out = ops.gemm_nt(t[:, :, :kc], proj[:, :kc], batch=S, a_batch_stride=kc * t.stride(2), b_batch_stride=kc,
                          rows=B * NT)
        return out.sum(0)
    z = split()
    err = ((z - ref).norm() / ref.norm()).item()
    print("split-K %d via batch + sum: %.1f us (rel diff %.1e)" % (S, timeit(split), err))
    print("   gemm alone: %.1f us" % timeit(lambda: ops.gemm_nt(t[:, :, :kc], proj[:, :kc], batch=S,
                                                                 a_batch_stride=kc * t.stride(2), b_batch_stride=kc,
                                                                 rows=B * NT)))
