"""Why does the tridiagonalisation's one-workgroup tail kernel start late inside a training step?  (Library built with
EXTRA=-DBASD_TAIL_DBG.)  A ranked 384-factorisation (2 matrices) on one stream, alone and beside the candidates on a
second stream: the Procrustes Jacobi (1024 stacked 98 x 49 cores) and the student Gram launch; the 100 MHz stamps give
the time between the end of the shared stage and the tail kernel's first instruction."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
import torch
from basd_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
x = torch.randn(2, 4 * 384, 384, generator=g)
G0 = (x.transpose(1, 2) @ x).to(dev)
cores = torch.randn(1024, 49, 98, generator=g).to(dev)
big = [torch.randn(256, 197, 384, generator=g).to(dev)[:, 1:, :] for _ in range(4)]
pin = torch.zeros(1 + 8, dtype=torch.int32).pin_memory()
lib = ctypes.CDLL(os.path.join(ROOT, "vit-inductive-bias-distillation_amd", "basd_amd", "libbasd_hip.so"))
buf = (ctypes.c_longlong * (2 * 8 * 2 * 1024))()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def run(label, beside, delay_us):
    gaps = []
    for rep in range(4):
        Gc, Wc = G0.clone(), cores.clone()
        torch.cuda.synchronize()
        with torch.cuda.stream(sa):
            ts = ops.tridiagonalise(Gc, mp_rank=(4 * 384, 384, 383, 1, pin, None))
        if beside is not None:
            with torch.cuda.stream(sb):
                torch.cuda._sleep(int(delay_us * 2400))          # start `delay_us` into the factorisation
                beside(Wc)
        torch.cuda.synchronize()
        assert lib.basd_debug_tail_stamps(buf) == 0
        s_begin, s_end, t_begin, t_end = (buf[8 * 2 * 1024 - 8 + i] for i in range(4))
        gaps.append(((s_end - s_begin) / 100.0, (t_begin - s_end) / 100.0, (t_end - t_begin) / 100.0))
    print(f"{label:46s} shared / gap / tail us: " + "  ".join(f"{a:.0f}/{b:.0f}/{c:.0f}" for a, b, c in gaps[1:]), flush=True)


run("alone", None, 0)
for d in (100, 350):
    run(f"Jacobi 1024 x (98 x 49) started at +{d} us", lambda W: ops.jacobi_onesided(W, 49), d)
    run(f"student Grams (4 x 50176 x 384) at +{d} us", lambda W: ops.centered_grams(big), d)
