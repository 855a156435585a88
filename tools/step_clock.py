"""One clock for host and GPU: where a bench step's time goes, in the STEADY STATE (consecutive steps, no synchronisation
between them).  Host labels (ops.trace), GPU marks on the caller's stream (ops.gpu_mark: timing events recorded on the
stream that is current at the mark) and the marks the selector chain records on its own streams (chain.TIMING) are all
reported in microseconds since the step's forward entry on the host, averaged over the steps; marks of a step that
land after the next step's entry show how far the steps overlap."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
import torch
import bench
from basd_amd import ops, synth, ddp, chain, _lib

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
shape = synth.CONFIGS[cfg]
device = torch.device("cuda", 0)
mod = bench.build(shape, cfg, device)
inp = synth.make_inputs(shape, 1234, batch=shape.batch, device=device, strided=True, attn_on_device=shape.layers_t > 1)
leaves = {k: v.detach().requires_grad_(True) for k, v in inp.student.items()}
logits = inp.logits.detach().requires_grad_(True)
bucket = ddp.FlatGradBucket(bench.STUDENT_PARAMS[cfg], list(mod.parameters()), device)
bucket.attach_grads([])
for _ in range(8):
    bench.one_step(mod, inp, leaves, logits, bucket)
torch.cuda.synchronize()
ref_raw = chain._event(True)
_lib.call("basd_event_record", ref_raw, torch.cuda.current_stream().cuda_stream)
ref = torch.cuda.Event(enable_timing=True)
ref.record()
torch.cuda.synchronize()
t_ref = time.perf_counter()
N = 30
chain.TIMING = []
host, gpu = [], []
for it in range(N):
    ops.HOST_TRACE, ops.GPU_MARKS = [], []
    t0 = time.perf_counter()
    bench.one_step(mod, inp, leaves, logits, bucket)
    ops.gpu_mark("step_end_main")
    ops.trace("step_out")
    host.append((t0, ops.HOST_TRACE))
    gpu.append(ops.GPU_MARKS)
mod.layer_selector.finish_pending()
torch.cuda.synchronize()
t_all = time.perf_counter() - t_ref
ops.HOST_TRACE = ops.GPU_MARKS = None
raw, chain.TIMING = chain.TIMING, None
print("ms/step %.3f (marks cost a little: compare with bench.py)" % (t_all / N * 1e3))
acc = collections.OrderedDict()
for i, ((t0, tr), marks) in enumerate(zip(host, gpu)):
    if i < 5:
        continue
    base = (t0 - t_ref) * 1e6
    for label, t in tr:
        acc.setdefault("host " + label, []).append((t - t_ref) * 1e6 - base)
    for label, ev in marks:
        acc.setdefault("GPU  " + label, []).append(ref.elapsed_time(ev) * 1e3 - base)
    if i < len(raw):
        for label, ev in raw[i].items():
            try:
                t_us = chain.elapsed_us(ref_raw, ev) - base
            except RuntimeError:
                continue                  # a mark of a branch that did not run this step
            acc.setdefault("GPU  chain " + label, []).append(t_us)
if mod._chain_plans:
    for sl in list(mod._chain_plans.values())[0].slots:
        w = sl.mirror.tolist()
        if w[-2] > 0:                  # BASD_TRIDIAG_CLOCKS=1 only
            print("factorisation kernel of the last step in this slot: %.1f us on the 100 MHz clock, %.0f shader cycles -> %.2f GHz"
                  % (w[-2] / 100.0, w[-1] * 16.0, w[-1] * 16.0 / max(w[-2] * 10.0, 1)))
period = (host[-1][0] - host[5][0]) / (len(host) - 6) * 1e6
print("%9.1f us  == next step's forward entry (the period)" % period)
rows = sorted(((sum(v) / len(v), k) for k, v in acc.items()))
for t, k in rows:
    print("%9.1f us  %s" % (t, k))
