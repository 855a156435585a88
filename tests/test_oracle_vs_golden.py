"""Pins oracle/basd_oracle.py against fixtures generated from the imported
reference (tests/golden/make_goldens.py).  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from basd_amd import synth
from oracle import basd_oracle as O

import make_goldens_shapes as S


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_mp_rank(golden):
    g = golden("mp_rank.npz")
    cases = [("m32_d192", 32, 192, 6), ("m128_d192", 128, 192, 10), ("m1000_d384", 1000, 384, 24),
             ("m12544_d384", 12544, 384, 48)]
    for i, (tag, m, d, r) in enumerate(cases):
        gen = torch.Generator().manual_seed(100 + i)
        x = synth.structured(gen, 1, m, d, r)[0]
        assert abs(float(x.double().sum()) - float(g[f"{tag}_sum"])) < 1e-6 * m * d, "generator drift"
        if f"{tag}_x" in g:
            assert np.array_equal(g[f"{tag}_x"], x.numpy())
        assert O.mp_rank(x) == int(g[f"{tag}_rank"]), tag
    gen = torch.Generator().manual_seed(200)
    noise = torch.randn(2000, 128, generator=gen)
    assert O.mp_rank(noise) == int(g["noise_rank"])


def test_subspace(golden):
    g = golden("subspace.npz")
    for i in range(3):
        m, d, r, k = [int(v) for v in g[f"c{i}_shape"]]
        gen = torch.Generator().manual_seed(300 + i)
        z = synth.structured(gen, 1, m, d, r)[0] + 0.7
        basis, s = O.pca_subspace(z, k)
        assert list(basis.shape) == list(g[f"c{i}_basis_shape"])
        np.testing.assert_allclose((basis @ basis.T).numpy(), g[f"c{i}_proj"], atol=2e-5)
        np.testing.assert_allclose(s.numpy(), g[f"c{i}_svals"], rtol=1e-6)


def test_align(golden):
    g = golden("align.npz")
    for n_in, n_out in [(49, 196), (256, 196), (144, 576), (1, 64), (64, 64), (7, 3)]:
        y = O.resample_tokens(T(g[f"{n_in}_{n_out}_x"]), n_out)
        np.testing.assert_allclose(y.numpy(), g[f"{n_in}_{n_out}_y"], rtol=0, atol=5e-7)


@pytest.mark.parametrize("tag", ["cls_same", "cls_interp", "nocls_uniform", "nocls_attn"])
def test_relational(golden, tag):
    g = golden("relational.npz")
    cls = bool(g[f"{tag}_meta"][6])
    s = T(g[f"{tag}_s"]).requires_grad_(True)
    t = T(g[f"{tag}_t"]).requires_grad_(True)
    attn = T(g[f"{tag}_attn"])
    loss = O.geometric_relational_loss(s, t, attn, has_cls_token=cls)
    np.testing.assert_allclose(loss.item(), g[f"{tag}_loss"], rtol=2e-6)
    w = O.token_weights(attn, cls, s.shape[1])
    tr_s, tr_t, nuc = O.procrustes_terms(s, t, w)
    np.testing.assert_allclose((tr_s + tr_t - 2 * nuc).detach().numpy(), g[f"{tag}_per_sample"], rtol=2e-5)
    gs, gt = torch.autograd.grad(loss, [s, t])
    for mine, ref in ((gs, g[f"{tag}_grad_s"]), (gt, g[f"{tag}_grad_t"])):
        err = np.linalg.norm(mine.numpy() - ref) / np.linalg.norm(ref)
        assert err < 1e-5, (tag, err)


def _oracle_full(shape, seed, batch, ls):
    torch.manual_seed(42)
    state = O.SelectorState.create(shape.points, shape.d_s, shape.d_t)
    inp = synth.make_inputs(shape, seed, batch=batch)
    for v in inp.student.values():
        v.requires_grad_(True)
    inp.logits.requires_grad_(True)
    layers = O.extraction_layers(shape.depth, shape.points)
    crit = torch.nn.CrossEntropyLoss(label_smoothing=ls)
    loss, trace = O.basd_forward(state, crit, layers, shape.n_s, shape.has_cls, inp.logits, inp.targets,
                                 inp.student, inp.teacher, inp.attn)
    return state, inp, layers, loss, trace


@pytest.mark.parametrize("tag", ["vit", "cnn", "vit_same"])
def test_full_small(golden, tag):
    g = golden("full_small.npz")
    shape, seed = S.SMALL[tag]
    state, inp, layers, loss, trace = _oracle_full(shape, seed, None, 0.01)
    assert abs(float(state.proj_s.double().sum()) - float(g[f"{tag}_proj_s_sum"])) < 1e-9
    assert layers == list(g[f"{tag}_token_layers"])
    assert [trace.selector.ranks[k] for k in sorted(inp.teacher)] == list(g[f"{tag}_ranks"])
    np.testing.assert_allclose(loss.item(), g[f"{tag}_loss"], rtol=1e-6)
    d = torch.stack([trace.selector.d_grass_sq[l] for l in layers]).numpy()
    np.testing.assert_allclose(d, g[f"{tag}_d_grass_sq"], rtol=2e-4, atol=1e-6)
    w = torch.stack([trace.selector.mix_weights[l] for l in layers]).numpy()
    np.testing.assert_allclose(w, g[f"{tag}_mix_weights"], rtol=2e-4, atol=1e-6)
    loss.backward()
    np.testing.assert_allclose(inp.logits.grad.numpy(), g[f"{tag}_grad_logits"], rtol=1e-5, atol=1e-8)
    for l in layers:
        ref = g[f"{tag}_grad_student_{l}"]
        err = np.linalg.norm(inp.student[l].grad.numpy() - ref) / np.linalg.norm(ref)
        assert err < 1e-3, (tag, l, err)   # route (b) goes through eigenvector perturbation
    np.testing.assert_allclose(state.log_temperatures.grad.numpy(), g[f"{tag}_grad_log_temperatures"],
                               rtol=5e-3, atol=1e-7)


def test_selector_outputs(golden):
    g = golden("selector_outputs.npz")
    shape, seed = S.SMALL["vit"]
    torch.manual_seed(42)
    state = O.SelectorState.create(shape.points, shape.d_s, shape.d_t)
    inp = synth.make_inputs(shape, seed)
    layers = O.extraction_layers(shape.depth, shape.points)
    with torch.no_grad():
        mixed, mixed_attn, _ = O.selector_forward(state, inp.student, inp.teacher, inp.attn, layers)
    assert layers == list(g["token_layers"])
    for l in layers:
        np.testing.assert_allclose(mixed[l][:, :5, :7].numpy(), g[f"mixed_{l}_slice"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(mixed_attn[l][:, :, 0, 1:].numpy(), g[f"attn_{l}_cls_row"], rtol=1e-4, atol=1e-7)


def test_structure(golden):
    g = golden("structure.npz")
    for depth in (12, 24):
        for n in (1, 2, 4):
            assert O.extraction_layers(depth, n) == list(g[f"layers_d{depth}_n{n}"])
            assert synth.extraction_layers(depth, n) == list(g[f"layers_d{depth}_n{n}"])
    torch.manual_seed(42)
    st = O.SelectorState.create(4, 8, 12)
    assert np.array_equal(st.proj_s.numpy(), g["proj_s"])
    assert np.array_equal(st.proj_t.numpy(), g["proj_t"])
    np.testing.assert_allclose(st.log_temperatures.detach().numpy(), g["log_temperatures"], rtol=0, atol=0)


@pytest.mark.parametrize("name,seed,batch,ls", [
    ("cfg1", 1234, None, 0.01), ("cfg2", 1234, 8, 0.001), ("cfg2", 1235, 8, 0.001),
    ("cfg4", 1234, 4, 0.001), ("cfg5", 1234, 4, 0.001)])
def test_baseline_scalars(golden, name, seed, batch, ls):
    g = golden("baseline_scalars.npz")
    shape = synth.CONFIGS[name]
    tag = f"{name}_s{seed}_b{batch or shape.batch}"
    state, inp, layers, loss, trace = _oracle_full(shape, seed, batch, ls)
    assert [trace.selector.ranks[k] for k in sorted(inp.teacher)] == list(g[f"{tag}_ranks"])
    np.testing.assert_allclose(loss.item(), g[f"{tag}_loss"], rtol=2e-6)
    loss.backward()
    norms = np.array([inp.student[l].grad.double().norm().item() for l in layers])
    np.testing.assert_allclose(norms, g[f"{tag}_grad_student_norms"], rtol=1e-3)
