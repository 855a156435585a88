"""Feature-capture adapters: hooks and sizing helpers either side of the loss path.
CPU tests cover the torch plumbing (hooks, views); GPU tests the calls into the library."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn

from basd_amd import capture
from oracle import basd_oracle as O


class _Attn(nn.Module):
    def __init__(self, dim, heads, bias=True):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, 3 * dim, bias=bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        a = ((qkv[0] @ qkv[1].transpose(-2, -1)) * (C // self.num_heads) ** -0.5).softmax(-1)
        return self.proj((a @ qkv[2]).transpose(1, 2).reshape(B, N, C))


class _Block(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.attn = _Attn(dim, heads)
        self.mlp = nn.Linear(dim, dim)

    def forward(self, x):
        x = x + self.attn(x)
        return x + torch.tanh(self.mlp(x))


class _ToyViT(nn.Module):
    def __init__(self, dim=32, heads=4, depth=3, tokens=10, classes=5):
        super().__init__()
        self.embed = nn.Linear(8, dim)
        self.cls = nn.Parameter(torch.zeros(1, 1, dim))
        self.blocks = nn.ModuleList([_Block(dim, heads) for _ in range(depth)])
        self.head = nn.Linear(dim, classes)

    def forward(self, x):
        x = torch.cat([self.cls.expand(x.shape[0], -1, -1), self.embed(x)], dim=1)
        for b in self.blocks:
            x = b(x)
        return self.head(x[:, 0])


def _teacher(model, depth=3):
    return SimpleNamespace(model=model, layer_paths=[f"blocks.{i}" for i in range(depth)], attn_subpath="attn",
                           has_cls_token=True, feature_format="token", embed_dim=32, heads_per_layer=[4] * depth,
                           depth=depth, mlp_ratio=4.0)


@pytest.mark.parametrize("bias", [True, False])
def test_cls_row_hook_matches_the_full_map(bias):
    torch.manual_seed(0)
    attn = _Attn(32, 4, bias=bias)
    x = torch.randn(3, 11, 32)
    full, row = {}, {}
    h1 = attn.register_forward_hook(capture.make_attn_capture_hook(full, 0))
    h2 = attn.register_forward_hook(capture.make_attn_capture_hook(row, 0, cls_row_only=True))
    attn(x)
    h1.remove(), h2.remove()
    assert row[0].shape == full[0].shape == (3, 4, 11, 11)
    assert row[0].stride(2) == 0                                   # nothing but the CLS row is stored
    torch.testing.assert_close(row[0][:, :, 0, :], full[0][:, :, 0, :], rtol=1e-5, atol=1e-6)
    # what the loss reads (relational.py:23-24) agrees
    torch.testing.assert_close(O.token_weights(row[0], True, 10), O.token_weights(full[0], True, 10), rtol=1e-5, atol=1e-7)


def test_extract_intermediates_and_student_views():
    torch.manual_seed(1)
    model = _ToyViT()
    x = torch.randn(2, 10, 8)
    toks, attns = capture.extract_intermediates(_teacher(model), x)
    assert sorted(toks) == [0, 1, 2] and sorted(attns) == [0, 1, 2]
    assert toks[0].shape == (2, 10, 32) and toks[0].storage_offset() == 32      # CLS-sliced view, not a copy
    assert attns[0].shape == (2, 4, 11, 11) and attns[0].stride(2) == 0
    full = capture.extract_intermediates(_teacher(model), x, cls_row_only=False)[1]
    torch.testing.assert_close(attns[1][:, :, 0, 1:], full[1][:, :, 0, 1:], rtol=1e-5, atol=1e-6)
    logits, s_tok = capture._extract_student(model, x, [0, 2], layer_paths=[f"blocks.{i}" for i in range(3)],
                                             has_cls_token=True)
    assert logits.shape == (2, 5) and sorted(s_tok) == [0, 2] and s_tok[2].shape == (2, 10, 32)
    # CNN-style teachers: channel-major view + constant attention of the reference's values
    cnn = SimpleNamespace(model=SimpleNamespace(forward_features=lambda im: torch.randn(2, 16, 3, 3)),
                          feature_format="nchw", has_cls_token=False)
    t, a = capture.extract_intermediates(cnn, None)
    assert t[0].shape == (2, 9, 16) and t[0].stride(1) == 1
    torch.testing.assert_close(a[0].contiguous(), torch.ones(2, 1, 9, 9) / 9)
    assert capture._to_token_format(torch.randn(2, 3, 3, 16), "nhwc", False).shape == (2, 9, 16)


def test_derive_from_teacher():
    t = SimpleNamespace(embed_dim=768, heads_per_layer=[12] * 12, depth=12, mlp_ratio=4.0)
    assert capture._derive_from_teacher(t, 100) == {"embed_dim": 128, "depth": 12, "num_heads": 2, "mlp_ratio": 4.0}
    assert capture._derive_from_teacher(t, 5000)["embed_dim"] == 768


@pytest.mark.gpu
def test_estimate_intrinsic_dim_on_the_library():
    """estimate_intrinsic_dim (teacher.py:161-177) = MP rank of the last layer's tokens, on the GPU kernels; the
    calibration shape of train.py:88-99 (about 10 D_t rows): a 2048 x 2048 Gram for a ResNet-50-wide teacher."""
    from basd_amd import synth
    dev = "cuda:0"
    gen = torch.Generator().manual_seed(9)
    D, rows = 2048, 20480
    feats = synth.structured(gen, 1, rows, D, 40)[0]                       # (rows, D): rank-40 signal + noise

    class _Last(nn.Module):
        def forward(self, x):
            return feats.to(x.device).reshape(1, rows, D)

    model = nn.Sequential()
    model.add_module("stage", _Last())
    teacher = SimpleNamespace(model=model, layer_paths=["stage"], feature_format="token", has_cls_token=False)
    got = capture.estimate_intrinsic_dim(teacher, torch.zeros(1, device=dev))
    ref = O.mp_rank(feats)
    assert got == ref and 40 <= ref < 64          # the planted rank plus the noise eigenvalues past the MP edge


@pytest.mark.gpu
def test_loss_on_expanded_cls_rows_equals_loss_on_full_maps():
    """BASDLoss fed the CLS-row-only captures (zero query stride) gives the value it gives on the full maps."""
    from basd_amd import synth
    from basd_amd.losses import BASDLoss
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_goldens_shapes as S
    dev = "cuda:0"
    shape, seed = S.SMALL["vit"]
    inp = synth.make_inputs(shape, seed, device=dev)
    vals = []
    for rows_only in (False, True):
        torch.manual_seed(42)
        mod = BASDLoss(nn.CrossEntropyLoss(label_smoothing=0.01), shape.d_s, shape.d_t, shape.depth, shape.n_s,
                       config=SimpleNamespace(num_extraction_points=shape.points), teacher_has_cls_token=True).to(dev)
        attn = inp.attn
        if rows_only:
            attn = {k: a[:, :, :1, :].contiguous().expand_as(a) for k, a in inp.attn.items()}
            assert all(a.stride(2) == 0 for a in attn.values())
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in inp.student.items()}
        loss = mod(inp.logits, inp.targets, leaves, inp.teacher, attn)
        loss.backward()
        vals.append((loss.item(), [leaves[l].grad.clone() for l in mod.token_layers]))
    np.testing.assert_allclose(vals[0][0], vals[1][0], rtol=1e-6)
    for a, b in zip(vals[0][1], vals[1][1]):
        assert ((a - b).norm() / a.norm()).item() < 1e-5
