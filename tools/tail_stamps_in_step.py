"""Phase stamps of the tridiagonalisation's tail stage INSIDE a training step (library built with
`make -C vit-inductive-bias-distillation_amd/csrc clean all EXTRA=-DBASD_TAIL_DBG`): the launch that delivers the
ranks (teacher chain, the one the host waits for) and the other one (student chain); cycles per phase at a few steps of
the factorisation.  Compare with tools/probe/tail_phase_probe.hip (the kernel alone on an idle chip)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
import torch
import bench
from basd_amd import synth, ddp

cfg = "cfg2"
shape = synth.CONFIGS[cfg]
device = torch.device("cuda", 0)
mod = bench.build(shape, cfg, device)
inp = synth.make_inputs(shape, 1234, batch=shape.batch, device=device, strided=True, attn_on_device=False)
leaves = {k: v.detach().requires_grad_(True) for k, v in inp.student.items()}
logits = inp.logits.detach().requires_grad_(True)
bucket = ddp.FlatGradBucket(bench.STUDENT_PARAMS[cfg], list(mod.parameters()), device)
bucket.attach_grads([])
for _ in range(12):
    bench.one_step(mod, inp, leaves, logits, bucket)
torch.cuda.synchronize()
print("ranks", dict(mod.layer_selector.subspace_ranks))
N = 2 * 8 * 2 * 1024
buf = (ctypes.c_longlong * N)()
lib = ctypes.CDLL(os.path.join(ROOT, "vit-inductive-bias-distillation_amd", "basd_amd", "libbasd_hip.so"))
rc = lib.basd_debug_tail_stamps(buf)
assert rc == 0, rc
names = ["pass", "A-wait", "sum", "scalar", "reflector", "B-wait"]
for label, off in (("ranked launch (teacher chain)", 0), ("plain launch (student chain)", 8 * 2 * 1024)):
    print(label)
    tot = buf[off + (254 * 8) * 2] - buf[off + 0]
    print(f"  steps 0..254: {tot} ticks of s_memtime")
    s_begin, s_end, t_begin, t_end = (buf[off + 8 * 2 * 1024 - 8 + i] for i in range(4))
    print(f"  100 MHz clock: shared stage {(s_end - s_begin) / 100.0:.0f} us, gap to the tail kernel's first instruction "
          f"{(t_begin - s_end) / 100.0:.0f} us, tail kernel {(t_end - t_begin) / 100.0:.0f} us (incl. rank)")
    c0, w0, c1, w1 = (buf[off + 8 * 2 * 1024 - 4 + i] for i in range(4))
    print(f"  whole loop: {c1 - c0} shader ticks in {(w1 - w0) / 100.0:.1f} us of the 100 MHz clock -> {(c1 - c0) / max(1, w1 - w0) * 100:.0f} MHz")
    for w, wname in ((0, "wave 0"), (1, "last wave")):
        for jl in (2, 64, 128, 200, 250):
            t = [buf[off + (jl * 8 + s) * 2 + w] for s in range(7)]
            nxt = buf[off + ((jl + 1) * 8) * 2 + w]
            cols = names if w == 0 else ["pass", "A-wait"]
            print(f"  {wname} step {jl:3d}: " + " ".join(f"{n} {t[i + 1] - t[i]}" for i, n in enumerate(cols)) + f" | step {nxt - t[0]}")
