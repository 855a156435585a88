"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol the
header declares, and the module mirrors the reference's constructor / state_dict contract."""
import os
import re
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "basd_hip.h")).read()
    return sorted(set(re.findall(r"^(?:int|long) (basd_\w+)\(", text, flags=re.M)))


def test_library_exports_every_declared_symbol():
    from basd_amd import _lib
    lib = _lib.load()
    names = _header_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/basd_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"


def test_host_sizing_helpers_need_no_gpu():
    from basd_amd import _lib
    assert _lib.query("basd_gemm_tn_splits", 50176) >= 1
    assert _lib.query("basd_colmean_parts", 100) == 1
    assert _lib.query("basd_jacobi_workspace_ints", 4, 30) == 2 * 4 * 30   # sweep flags + column-norm maxima


def test_no_cpu_fallback():
    from basd_amd.losses import marchenko_pastur_rank
    with pytest.raises(RuntimeError):
        marchenko_pastur_rank(torch.randn(64, 16))


def test_module_contract(golden):
    from basd_amd.losses import BASDLoss
    g = golden("structure.npz")
    for depth in (12, 24):
        for n in (1, 2, 4):
            torch.manual_seed(42)
            mod = BASDLoss(torch.nn.CrossEntropyLoss(), 8, 8, depth, 4,
                           config=SimpleNamespace(num_extraction_points=n), teacher_has_cls_token=False)
            assert mod.token_layers == list(g[f"layers_d{depth}_n{n}"])
    torch.manual_seed(42)
    mod = BASDLoss(torch.nn.CrossEntropyLoss(), 8, 12, 12, 4, config=SimpleNamespace(num_extraction_points=4),
                   teacher_has_cls_token=False)
    assert sorted(mod.state_dict().keys()) == list(g["state_keys"])
    assert [n for n, _ in mod.named_parameters()] == list(g["param_names"])
    # same global-RNG consumption as the reference: identical projections for the same seed
    assert np.array_equal(mod.layer_selector.proj_s.numpy(), g["proj_s"])
    assert np.array_equal(mod.layer_selector.proj_t.numpy(), g["proj_t"])
    np.testing.assert_array_equal(mod.layer_selector.log_temperatures.detach().numpy(), g["log_temperatures"])
    np.testing.assert_allclose(mod.layer_selector.temperatures.detach().numpy(), g["temperatures"], rtol=1e-7)
    # round trip through a state dict (checkpoint compatibility: reference trainer.py:84)
    sd = {k: v.clone() for k, v in mod.state_dict().items()}
    mod2 = BASDLoss(torch.nn.CrossEntropyLoss(), 8, 12, 12, 4, config=SimpleNamespace(num_extraction_points=4),
                    teacher_has_cls_token=False)
    mod2.load_state_dict(sd)
    assert torch.equal(mod2.layer_selector.proj_t, mod.layer_selector.proj_t)


def test_drop_in_module_paths():
    import importlib
    combined = importlib.import_module("src.losses.combined")
    ls = importlib.import_module("src.losses.layer_selector")
    rel = importlib.import_module("src.losses.relational")
    assert combined.BASDLoss.__module__ == "basd_amd.losses"
    for name in ("GrassmannianLayerSelector", "marchenko_pastur_rank", "_grassmann_subspace"):
        assert hasattr(ls, name)
    assert hasattr(rel, "geometric_relational_loss") and hasattr(combined, "_align_token_count")
    # every method of the reference classes exists under the same name with the same parameters
    import inspect
    sel = ls.GrassmannianLayerSelector
    assert list(inspect.signature(sel._mix_for_student_layer).parameters) == [
        "self", "i", "s_tokens", "teacher_indices", "stacked_tokens", "stacked_attns", "subspaces", "spectral_weights"]
    assert list(inspect.signature(sel.forward).parameters) == [
        "self", "student_tokens_per_layer", "all_teacher_tokens", "all_teacher_attns", "extraction_indices"]
    assert list(inspect.signature(sel._estimate_ranks).parameters) == ["self", "all_teacher_tokens"]
    assert list(inspect.signature(combined.BASDLoss.forward).parameters) == [
        "self", "student_output", "targets", "student_intermediates", "all_teacher_tokens", "all_teacher_attns"]
    assert isinstance(sel.temperatures, property)


def test_interp_taps_match_oracle():
    from basd_amd import ops
    from oracle import basd_oracle as O
    for n_in, n_out in [(49, 196), (256, 196), (144, 576), (1, 64), (7, 3)]:
        i0, i1, lam, r0, r1 = ops._taps_cpu(n_in, n_out)
        o0, o1, ol = O.interp_taps(n_in, n_out)
        assert torch.equal(i0.long(), o0) and torch.equal(i1.long(), o1) and torch.equal(lam, ol)
        for j in range(n_in):
            touching = [s for s in range(n_out) if int(i0[s]) == j or int(i1[s]) == j]
            if touching:
                assert int(r0[j]) == touching[0] and int(r1[j]) == touching[-1] + 1
                assert touching == list(range(touching[0], touching[-1] + 1))
            else:
                assert int(r0[j]) >= int(r1[j])


def test_reference_state_file_loads_on_cpu(golden):
    """The reference-format state file (tests/golden/ref_basd_state.pth, written by the imported reference) has exactly
    the build's state_dict keys, shapes and dtypes, and loads strictly (no kernels involved: runs without a GPU)."""
    import os
    from types import SimpleNamespace

    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_goldens_shapes as S
    from basd_amd.losses import BASDLoss
    shape, _ = S.SMALL["vit"]
    torch.manual_seed(7)
    mod = BASDLoss(torch.nn.CrossEntropyLoss(), shape.d_s, shape.d_t, shape.depth, shape.n_s,
                   config=SimpleNamespace(num_extraction_points=shape.points), teacher_has_cls_token=shape.has_cls)
    g = golden("checkpoint.npz")
    # same RNG consumption as the reference's constructor: the seed-7 projections match what make_goldens recorded
    np.testing.assert_allclose(mod.layer_selector.proj_t.double().abs().sum().item(), g["ours_seed7_proj_t_abs_sum"], rtol=1e-12)
    state = torch.load(os.path.join(os.path.dirname(__file__), "golden", "ref_basd_state.pth"), map_location="cpu",
                       weights_only=True)
    own = mod.state_dict()
    assert list(state.keys()) == list(own.keys())
    for k in own:
        assert state[k].shape == own[k].shape and state[k].dtype == own[k].dtype, k
    mod.load_state_dict(state, strict=True)
    np.testing.assert_array_equal(mod.layer_selector.log_temperatures.detach().numpy(), g["log_temperatures"])
