"""Time of the Procrustes forward + student gradients on cores past LDS (576 tokens: a ViT teacher at 384 x 384), with
the per-entry-point split.  usage: large_core_bench.py [B=16] [E=4] [n=576] [d_s=768] [d_t=1024]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
import torch
from basd_amd import ops, synth, _lib

B, E, n, d_s, d_t = [int(a) for a in sys.argv[1:6]] + [16, 4, 576, 768, 1024][len(sys.argv) - 1:]
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(1)
students = [synth.structured(gen, B, n, d_s, 32).to(dev) for _ in range(E)]
teacher = synth.structured(gen, B, n, d_t, 48).to(dev)
attn = torch.softmax(torch.randn(B, 4, n + 1, n + 1, generator=gen), dim=-1).to(dev)
mix = torch.ones(E, 1, device=dev)
gl = torch.ones(E, device=dev)
for rep in range(3):
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    ctx = ops.procrustes_forward(students, [teacher], [attn], mix, True, want_sweeps=True)
    e1.record()
    grads = ops.procrustes_student_grads(students, ctx, gl)
    e2.record()
    torch.cuda.synchronize()
    print(f"rep {rep}: forward {e0.elapsed_time(e1):.1f} ms, student grads {e1.elapsed_time(e2):.1f} ms, sweeps mean "
          f"{ctx.sweeps.float().mean():.1f} max {int(ctx.sweeps.max())}, loss {ctx.loss_b.mean().item():.4f}", flush=True)
