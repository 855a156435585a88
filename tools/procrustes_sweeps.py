import os, sys
sys.path.insert(0, "/root/repo/vit-inductive-bias-distillation_amd"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/vit-inductive-bias-distillation_amd")
import torch
from basd_amd import ops, synth
shape = synth.CONFIGS["cfg2"]
inp = synth.make_inputs(shape, 1234, device="cuda:0", strided=True)
students = [inp.student[k] for k in sorted(inp.student)]
teachers = [inp.teacher[k] for k in sorted(inp.teacher)]
attns = [inp.attn[k].cuda() for k in sorted(inp.attn)]
mix = torch.ones(4, 1, device="cuda")
pc = ops.procrustes_forward(students, teachers, attns, mix, False, want_sweeps=True)
sw = pc.sweeps.cpu()
print("sweeps: min %d max %d mean %.2f" % (sw.min(), sw.max(), sw.float().mean()), torch.bincount(sw).tolist())
