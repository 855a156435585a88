"""A/B: LDS-resident vs tiled teacher-side factor at a given core order (default: cfg-4's 196 tokens, 512 cores)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
import torch
from basd_amd import ops, synth
n, B, E = [int(a) for a in sys.argv[1:4]] + [196, 128, 4][len(sys.argv) - 1:]
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(3)
students = [synth.structured(gen, B, n, 256, 32).to(dev) for _ in range(E)]
teachers = [synth.structured(gen, B, n, 320, 24).to(dev) for _ in range(2)]
attns = [torch.softmax(torch.randn(B, 2, n + 1, n + 1, generator=gen), dim=-1).to(dev) for _ in range(2)]
mix = torch.softmax(torch.randn(E, 2, generator=gen), dim=-1).to(dev)
pc = ops.procrustes_forward(students, teachers, attns, mix, True, need_mix_grad=True)
mg = pc.mixgrad
kt, tn = torch.empty((E * B, n, n), device=dev), torch.empty((E, B, n), device=dev)
scratch = torch.empty_like(kt)
common = (mg["W"].data_ptr(), 2 * n * n, mg["sigma"].data_ptr(), n, n, E * B, mg["l_a"].data_ptr(), mg["g_b"].data_ptr(),
          n * n, mg["omega_e"].data_ptr(), None, None, None, None, None, kt.data_ptr(), tn.data_ptr())
for name, extra in (("basd_teacher_factor", ()), ("basd_teacher_factor_tiled", (scratch.data_ptr(),))):
    ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops._lib.call(name, *common, *extra, ops._stream())
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"n={n} cores={E * B} {name}: {min(ts):.3f} ms (checksum {kt.double().sum().item():.6e})")
