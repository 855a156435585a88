"""One clock for host and GPU: where a bench step's time goes.  Host labels (ops.trace) and GPU marks (timing events
recorded on the stream that is current at the mark) are both reported in microseconds since the step's forward entry
on the host; GPU marks via a reference event recorded right after a device synchronisation."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
import torch
import bench
from basd_amd import ops, synth, ddp, chain, _lib
chain.TIMED_EVENTS = True
chain.TIMING = []

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
shape = synth.CONFIGS[cfg]
device = torch.device("cuda", 0)
mod = bench.build(shape, cfg, device)
inp = synth.make_inputs(shape, 1234, batch=shape.batch, device=device, strided=True, attn_on_device=shape.layers_t > 1)
leaves = {k: v.detach().requires_grad_(True) for k, v in inp.student.items()}
logits = inp.logits.detach().requires_grad_(True)
bucket = ddp.FlatGradBucket(bench.STUDENT_PARAMS[cfg], list(mod.parameters()), device, slots=2)
for _ in range(8):
    bench.one_step(mod, inp, leaves, logits, bucket)
torch.cuda.synchronize()
ref = torch.cuda.Event(enable_timing=True)
ref.record()
ref_raw = chain._event(True)
_lib.call("basd_event_record", ref_raw, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
t_ref = time.perf_counter()
N = 30
host, gpu, raw = [], [], []
plans = lambda: list(mod._chain_plans.values())
for it in range(N):
    ops.HOST_TRACE, ops.GPU_MARKS = [], []
    t0 = time.perf_counter()
    bench.one_step(mod, inp, leaves, logits, bucket)
    ops.gpu_mark("step_end_main")
    ops.trace("step_out")
    host.append((t0, ops.HOST_TRACE))
    gpu.append(ops.GPU_MARKS)
    if plans():     # the slot this step used: its events are re-recorded two steps later, so read them now
        torch.cuda.synchronize()
        slot = plans()[0].slots[plans()[0].turn ^ 1]
        raw.append({k: chain.elapsed_us(ref_raw, getattr(slot, k)) for k in ("ev_fork", "ev_ranks", "ev_tail")
                    if k != "ev_tail" or slot.used})
        for k, ev in chain.TIMING[-1].items():
            try:
                raw[-1][k] = chain.elapsed_us(ref_raw, ev)
            except RuntimeError:
                pass                      # a mark of a branch that did not run this step
    else:
        raw.append({})
torch.cuda.synchronize()
t_all = time.perf_counter() - t_ref
ops.HOST_TRACE = ops.GPU_MARKS = None
print("ms/step %.3f" % (t_all / N * 1e3))
acc = collections.OrderedDict()
for (t0, tr), marks, rw in list(zip(host, gpu, raw))[5:]:
    base = (t0 - t_ref) * 1e6
    for label, t in rw.items():
        acc.setdefault("GPU  chain " + label, []).append(t - base)
    for label, t in tr:
        acc.setdefault("host " + label, []).append((t - t_ref) * 1e6 - base)
    for label, ev in marks:
        acc.setdefault("GPU  " + label, []).append(ref.elapsed_time(ev) * 1e3 - base)
rows = sorted(((sum(v) / len(v), k) for k, v in acc.items()))
for t, k in rows:
    print("%9.1f us  %s" % (t, k))
