"""Sweeps and time of the Procrustes Jacobi on the cores of a BASELINE configuration (bench.py's synthetic inputs).
usage: procrustes_sweeps.py [cfg2] [reps=5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
import torch
from basd_amd import ops, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
shape = synth.CONFIGS[cfg]
dev = torch.device("cuda:0")
inp = synth.make_inputs(shape, 1234, device=dev, strided=True, attn_on_device=shape.layers_t > 1)
students = [inp.student[k] for k in sorted(inp.student)]
keys = sorted(inp.teacher)
E, L = len(students), len(keys)
mix = torch.full((E, L), 1.0 / L, device=dev)
import basd_amd._lib as _lib
_lib.timing, _lib.timed_names = {}, None
for r in range(reps):
    ctx = ops.procrustes_forward(students, [inp.teacher[k] for k in keys], [inp.attn[k] for k in keys], mix,
                                 shape.has_cls, want_sweeps=True)
torch.cuda.synchronize()
sw = ctx.sweeps.float()
print(f"{cfg}: {sw.numel()} cores, sweeps mean {sw.mean():.2f} min {int(sw.min())} max {int(sw.max())}; "
      f"loss mean {ctx.loss_b.mean().item():.6f}")
for name, evs in _lib.timing.items():
    ts = sorted(a.elapsed_time(b) for a, b in evs[1:])
    print(f"  {name}: median {ts[len(ts) // 2]:.3f} ms over {len(ts)} calls")
