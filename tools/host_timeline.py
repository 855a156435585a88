"""Diagnostic: where the HOST spends a bench step (mean over steps, microseconds since the step began)."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
import torch
import bench
from basd_amd import ops, synth, ddp

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
shape = synth.CONFIGS[cfg]
device = torch.device("cuda", 0)
mod = bench.build(shape, cfg, device)
inp = synth.make_inputs(shape, 1234, batch=shape.batch, device=device, strided=True, attn_on_device=shape.layers_t > 1)
leaves = {k: v.detach().requires_grad_(True) for k, v in inp.student.items()}
logits = inp.logits.detach().requires_grad_(True)
bucket = ddp.FlatGradBucket(bench.STUDENT_PARAMS[cfg], list(mod.parameters()), device)
bucket.attach_grads([])
for _ in range(5):
    bench.one_step(mod, inp, leaves, logits, bucket)
torch.cuda.synchronize()
acc = collections.OrderedDict()
N = 50
t_all = time.perf_counter()
for _ in range(N):
    ops.HOST_TRACE = []
    t0 = time.perf_counter()
    loss = mod(logits, inp.targets, leaves, inp.teacher, inp.attn)
    ops.trace("fwd_returned")
    loss.backward()
    ops.trace("bwd_queued")
    grads = [leaves[l].grad for l in mod.token_layers]
    rows = grads[0].shape[0] * grads[0].shape[1]
    sums = ops.column_means(grads)
    bucket.student_view[: sums.numel()].copy_(sums.reshape(-1))
    bucket.student_view[: sums.numel()].mul_(rows)
    bucket.pack_loss_grads()
    bucket.all_reduce_mean()
    ops.trace("step_out")
    for v in leaves.values():
        v.grad = None
    for label, t in ops.HOST_TRACE:
        acc[label] = acc.get(label, 0.0) + (t - t0)
torch.cuda.synchronize()
print("ms/step %.3f" % ((time.perf_counter() - t_all) / N * 1e3))
if ops.CHAIN_EVENTS:
    ch = sorted(a.elapsed_time(b) for a, b in ops.CHAIN_EVENTS)
    print("teacher chain on the GPU (start -> ranks): median %.3f ms, min %.3f, max %.3f" % (ch[len(ch) // 2], ch[0], ch[-1]))
prev = 0.0
for label, s in acc.items():
    print("%-20s %8.1f us   (+%.1f)" % (label, s / N * 1e6, s / N * 1e6 - prev))
    prev = s / N * 1e6
