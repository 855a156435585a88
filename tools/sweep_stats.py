"""Diagnostic: Jacobi sweep counts / timings at a BASELINE config (run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
import torch
from basd_amd import ops, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
shape = synth.CONFIGS[cfg]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else shape.batch
inp = synth.make_inputs(shape, 1234, batch=batch, device="cuda:0", strided=True, attn_on_device=True)
students = list(inp.student.values())
teachers = [inp.teacher[k] for k in sorted(inp.teacher)]
attns = [inp.attn[k] for k in sorted(inp.attn)]
mix = torch.full((len(students), len(teachers)), 1.0 / len(teachers), device="cuda:0")
pc = ops.procrustes_forward(students, teachers, attns, mix, shape.has_cls, want_sweeps=True)
torch.cuda.synchronize()
sw = pc.sweeps.float()
print("procrustes core sweeps: min %d mean %.2f max %d" % (sw.min(), sw.mean(), sw.max()))
# symmetric eigen-solve of a student Gram
x = students[0]
mean = ops.colmean(x)
g = ops.gemm_tn(x, x, mean_a=mean, mean_b=mean).unsqueeze(0)
for rep in range(2):
    gg = g.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cn, sweeps = ops.jacobi_onesided(gg, gg.shape[1], want_sweeps=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("sym eig n=%d: %.2f ms" % (gg.shape[1], 1e3 * (t1 - t0)))
flags = None
