"""Generate the golden fixtures in this directory by IMPORTING the reference.

Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/make_goldens.py [--skip-large]

The reference (``/root/reference/src/losses``) is imported unmodified on CPU
(torch 2.10.0, fp32, no autocast); inputs come from the build's own seeded
generator (``basd_amd.synth``) so tests can regenerate them anywhere.  Only
inputs/outputs are stored -- no reference source.
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

from basd_amd import synth  # noqa: E402

import src.losses.combined as ref_combined  # noqa: E402
import src.losses.layer_selector as ref_ls  # noqa: E402
import src.losses.relational as ref_rel  # noqa: E402

assert ref_combined.__file__.startswith("/root/reference"), ref_combined.__file__


def npf(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# --------------------------------------------------------------------------- #
def g_mp_rank():
    out = {}
    cases = [("m32_d192", 32, 192, 6), ("m128_d192", 128, 192, 10), ("m1000_d384", 1000, 384, 24),
             ("m12544_d384", 12544, 384, 48)]
    for i, (tag, m, d, r) in enumerate(cases):
        gen = torch.Generator().manual_seed(100 + i)
        x = synth.structured(gen, 1, m, d, r)[0]
        out[f"{tag}_rank"] = np.int64(ref_ls.marchenko_pastur_rank(x))
        out[f"{tag}_sum"] = np.float64(x.double().sum())
        if m * d <= 40000:
            out[f"{tag}_x"] = npf(x)
    gen = torch.Generator().manual_seed(200)
    noise = torch.randn(2000, 128, generator=gen)
    out["noise_rank"] = np.int64(ref_ls.marchenko_pastur_rank(noise))
    out["noise_sum"] = np.float64(noise.double().sum())
    save("mp_rank.npz", **out)


def g_subspace():
    out = {}
    for i, (m, d, r, k) in enumerate([(400, 96, 10, 10), (3000, 192, 20, 31), (150, 64, 5, 0)]):
        gen = torch.Generator().manual_seed(300 + i)
        z = synth.structured(gen, 1, m, d, r)[0] + 0.7
        basis, s = ref_ls._grassmann_subspace(z, k=k)
        out[f"c{i}_shape"] = np.array([m, d, r, k])
        out[f"c{i}_proj"] = npf(basis @ basis.T)
        out[f"c{i}_svals"] = npf(s)
        out[f"c{i}_basis_shape"] = np.array(basis.shape)
    save("subspace.npz", **out)


def g_align():
    out = {}
    for n_in, n_out in [(49, 196), (256, 196), (144, 576), (1, 64), (64, 64), (7, 3)]:
        gen = torch.Generator().manual_seed(400 + n_in)
        x = torch.randn(3, n_in, 5, generator=gen)
        y = ref_combined._align_token_count(x, n_out)
        out[f"{n_in}_{n_out}_x"] = npf(x)
        out[f"{n_in}_{n_out}_y"] = npf(y)
    save("align.npz", **out)


def g_relational():
    out = {}
    cases = [
        # tag, B, N_s, D_s, N_attn(tokens w/o CLS), D_t, H, cls
        ("cls_same", 4, 49, 64, 49, 96, 4, True),
        ("cls_interp", 4, 64, 48, 49, 80, 2, True),
        ("nocls_uniform", 3, 36, 64, 9, 128, 1, False),
        ("nocls_attn", 3, 25, 40, 25, 56, 3, False),
    ]
    for i, (tag, b, n_s, d_s, n_a, d_t, h, cls) in enumerate(cases):
        gen = torch.Generator().manual_seed(500 + i)
        s = synth.structured(gen, b, n_s, d_s, 6).requires_grad_(True)
        t = (synth.structured(gen, b, n_s, d_t, 5) + 0.3).requires_grad_(True)
        a = n_a + (1 if cls else 0)
        if tag == "nocls_uniform":
            attn = torch.ones(b, h, a, a) / a
        else:
            attn = torch.softmax(torch.randn(b, h, a, a, generator=gen), dim=-1)
        attn.requires_grad_(True)
        loss = ref_rel.geometric_relational_loss(s, t, attn, has_cls_token=cls)
        gs, gt, ga = torch.autograd.grad(loss, [s, t, attn])
        per = [ref_rel.geometric_relational_loss(s[j:j + 1], t[j:j + 1], attn[j:j + 1], has_cls_token=cls)
               for j in range(b)]
        out[f"{tag}_meta"] = np.array([b, n_s, d_s, n_a, d_t, h, int(cls)])
        out[f"{tag}_s"] = npf(s)
        out[f"{tag}_t"] = npf(t)
        out[f"{tag}_attn"] = npf(attn)
        out[f"{tag}_loss"] = npf(loss)
        out[f"{tag}_per_sample"] = npf(torch.stack(per))
        out[f"{tag}_grad_s"] = npf(gs)
        out[f"{tag}_grad_t"] = npf(gt)
        out[f"{tag}_grad_attn_cls_row_sum"] = np.float64(ga.double().abs().sum())
    save("relational.npz", **out)


# --------------------------------------------------------------------------- #
class _SoftmaxTap:
    """Records the argument of F.softmax inside the reference selector so the
    per-(student layer, teacher layer) distances can be stored (d = -arg * tau)."""

    def __init__(self):
        self.args = []
        self._orig = ref_ls.F.softmax

    def __enter__(self):
        def tap(x, dim=None, **kw):
            self.args.append(x.detach().clone())
            return self._orig(x, dim=dim, **kw)
        self._ns = SimpleNamespace(**{k: getattr(ref_ls.F, k) for k in dir(ref_ls.F) if not k.startswith("__")})
        self._ns.softmax = tap
        self._saved = ref_ls.F
        ref_ls.F = self._ns
        return self

    def __exit__(self, *exc):
        ref_ls.F = self._saved


def _build_ref(shape, label_smoothing, module_seed=42):
    torch.manual_seed(module_seed)                       # reference configs/config.yaml:8
    cfg = SimpleNamespace(num_extraction_points=shape.points)
    crit = torch.nn.CrossEntropyLoss(label_smoothing=label_smoothing)
    return ref_combined.BASDLoss(
        crit, shape.d_s, shape.d_t, shape.depth, shape.n_s, config=cfg,
        teacher_has_cls_token=shape.has_cls)


def _run_full(shape, seed, batch, label_smoothing, want_grads=True, store_grads=False):
    mod = _build_ref(shape, label_smoothing)
    inp = synth.make_inputs(shape, seed, batch=batch)
    for v in inp.student.values():
        v.requires_grad_(True)
    inp.logits.requires_grad_(True)
    t0 = time.time()
    with _SoftmaxTap() as tap:
        loss = mod(inp.logits, inp.targets, inp.student, inp.teacher, inp.attn)
    t_fwd = time.time() - t0
    layers = mod.token_layers
    tau = mod.layer_selector.temperatures.detach()
    d = torch.stack([-(a * tau[i]) for i, a in enumerate(tap.args)])      # (E, L)
    w = torch.stack([torch.softmax(a, dim=0) for a in tap.args])
    rec = {
        "loss": npf(loss), "token_layers": np.array(layers),
        "ranks": np.array([mod.layer_selector.subspace_ranks[k] for k in sorted(inp.teacher.keys())]),
        "d_grass_sq": npf(d), "mix_weights": npf(w),
        "proj_s_sum": np.float64(mod.layer_selector.proj_s.double().sum()),
        "proj_t_abs_sum": np.float64(mod.layer_selector.proj_t.double().abs().sum()),
        "t_fwd_s": np.float64(t_fwd),
    }
    if want_grads:
        t0 = time.time()
        loss.backward()
        rec["t_bwd_s"] = np.float64(time.time() - t0)
        rec["grad_logits_norm"] = np.float64(inp.logits.grad.double().norm())
        rec["grad_student_norms"] = np.array([inp.student[l].grad.double().norm().item() for l in layers])
        rec["grad_log_temperatures"] = npf(mod.layer_selector.log_temperatures.grad)
        if store_grads:
            rec["grad_logits"] = npf(inp.logits.grad)
            for l in layers:
                rec[f"grad_student_{l}"] = npf(inp.student[l].grad)
    return rec


sys.path.insert(0, HERE)
from make_goldens_shapes import TOY_VIT, TOY_CNN, TOY_VIT_SAME  # noqa: E402


def g_full_small():
    out = {}
    for tag, shape, seed in [("vit", TOY_VIT, 3), ("cnn", TOY_CNN, 5), ("vit_same", TOY_VIT_SAME, 9)]:
        rec = _run_full(shape, seed, None, 0.01, store_grads=True)
        for k, v in rec.items():
            out[f"{tag}_{k}"] = v
    save("full_small.npz", **out)


def g_selector_outputs():
    """mixed tokens / attention of the toy ViT case (slices + sums)."""
    shape, seed = TOY_VIT, 3
    mod = _build_ref(shape, 0.01)
    inp = synth.make_inputs(shape, seed)
    with torch.no_grad():
        mixed, mixed_attn = mod.layer_selector(inp.student, inp.teacher, inp.attn, mod.token_layers)
    out = {"token_layers": np.array(mod.token_layers)}
    for l in mod.token_layers:
        out[f"mixed_{l}_slice"] = npf(mixed[l][:, :5, :7])
        out[f"mixed_{l}_sum"] = np.float64(mixed[l].double().sum())
        out[f"attn_{l}_cls_row"] = npf(mixed_attn[l][:, :, 0, 1:])
    save("selector_outputs.npz", **out)


def g_structure():
    out = {}
    for depth in (12, 24):
        for n in (1, 2, 4):
            shape = synth.LossShape("s", 2, 4, 8, depth, 4, 8, 1, 1, False, 3, points=n)
            mod = _build_ref(shape, 0.0)
            out[f"layers_d{depth}_n{n}"] = np.array(mod.token_layers)
    mod = _build_ref(synth.LossShape("s", 2, 4, 8, 12, 4, 12, 1, 1, False, 3), 0.0)
    out["state_keys"] = np.array(sorted(mod.state_dict().keys()))
    out["param_names"] = np.array([n for n, _ in mod.named_parameters()])
    out["log_temperatures"] = npf(mod.layer_selector.log_temperatures)
    out["temperatures"] = npf(mod.layer_selector.temperatures)
    out["proj_s"] = npf(mod.layer_selector.proj_s)
    out["proj_t"] = npf(mod.layer_selector.proj_t)
    save("structure.npz", **out)


def g_checkpoint():
    """Checkpoint interchange (reference trainer.py:84,94-123: accelerate stores the registered BASDLoss through
    state_dict() / torch.save).  A reference-format state file with NON-default temperatures, the loss the reference
    computes from it, and the reverse direction: the build's state_dict loaded into the reference module."""
    shape, seed = TOY_VIT, 3
    mod = _build_ref(shape, 0.01)
    with torch.no_grad():
        mod.layer_selector.log_temperatures += torch.linspace(-0.4, 0.5, shape.points)
    torch.save(mod.state_dict(), os.path.join(HERE, "ref_basd_state.pth"))
    inp = synth.make_inputs(shape, seed)
    for v in inp.student.values():
        v.requires_grad_(True)
    with _SoftmaxTap() as tap:
        loss = mod(inp.logits, inp.targets, inp.student, inp.teacher, inp.attn)
    loss.backward()
    out = {"loss": npf(loss), "mix_weights": npf(torch.stack([torch.softmax(a, dim=0) for a in tap.args])),
           "grad_log_temperatures": npf(mod.layer_selector.log_temperatures.grad),
           "log_temperatures": npf(mod.layer_selector.log_temperatures)}
    # reverse: the build's module (constructed on CPU: no kernel needed) -> its state_dict -> the reference module
    from basd_amd.losses import BASDLoss as OurLoss
    torch.manual_seed(7)
    ours = OurLoss(torch.nn.CrossEntropyLoss(), shape.d_s, shape.d_t, shape.depth, shape.n_s,
                   config=SimpleNamespace(num_extraction_points=shape.points), teacher_has_cls_token=shape.has_cls)
    ref2 = _build_ref(shape, 0.01, module_seed=123)
    res = ref2.load_state_dict(ours.state_dict(), strict=True)
    same = all(torch.equal(a, b) for a, b in zip(ref2.state_dict().values(), ours.state_dict().values()))
    out["reverse_load_ok"] = np.int64(int(same and not res.missing_keys and not res.unexpected_keys))
    out["ours_seed7_proj_t_abs_sum"] = np.float64(ours.layer_selector.proj_t.double().abs().sum())
    save("checkpoint.npz", **out)


def g_baseline_scalars(skip_large):
    """Scalar goldens at the BASELINE.json shapes (inputs regenerated from seed)."""
    out = {}
    plan = [
        ("cfg1", 1234, None, 0.01),          # full size (B=32)
        ("cfg2", 1234, 8, 0.001),
        ("cfg2", 1235, 8, 0.001),            # rank-1 seed of the DDP config (cfg3)
        ("cfg4", 1234, 4, 0.001),
        ("cfg5", 1234, 4, 0.001),
    ]
    if not skip_large:
        plan.append(("cfg2", 1234, 256, 0.001))
    for name, seed, batch, ls in plan:
        shape = synth.CONFIGS[name]
        tag = f"{name}_s{seed}_b{batch or shape.batch}"
        print("running", tag, flush=True)
        rec = _run_full(shape, seed, batch, ls)
        print(f"  loss {rec['loss']}, ranks {rec['ranks'][:6]}..., fwd {rec['t_fwd_s']:.1f}s bwd {rec.get('t_bwd_s', 0):.1f}s")
        for k, v in rec.items():
            out[f"{tag}_{k}"] = v
    save("baseline_scalars.npz", **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-large", action="store_true")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.set_num_threads(8)
    steps = {
        "mp_rank": g_mp_rank, "subspace": g_subspace, "align": g_align, "relational": g_relational,
        "full_small": g_full_small, "selector_outputs": g_selector_outputs, "structure": g_structure, "checkpoint": g_checkpoint,
        "baseline": lambda: g_baseline_scalars(args.skip_large),
    }
    for name, fn in steps.items():
        if args.only and name not in args.only.split(","):
            continue
        fn()
