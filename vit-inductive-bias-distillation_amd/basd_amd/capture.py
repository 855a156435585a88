"""Feature-capture adapters on either side of the loss path: the callers that hand
teacher / student tensors to ``BASDLoss`` and the start-up sizing pass that calls ``marchenko_pastur_rank``.

Same function names, arguments and return contracts as the reference's helpers
(``src/models/teacher.py:27-39, 151-177, 180-215``, ``src/training/trainer.py:16-37``, ``src/train.py:57-66``), so a
maintainer swaps the import and nothing else.  What differs is what is materialised:

* tokens are handed over as zero-copy strided views (CLS-sliced, channel-major): the kernels consume them in place;
* ``make_attn_capture_hook(..., cls_row_only=True)`` computes the CLS query's attention row only -- the loss reads
  ``attn[:, :, 0, 1:]`` and nothing else of a ViT teacher's maps (relational.py:23-24) -- and returns it as a
  (B, H, N, N) view with a zero stride along the query axis: B*H*N floats per layer instead of B*H*N^2
  (cfg-4: 38.5 MB instead of 7.6 GB per step);
* ``estimate_intrinsic_dim`` runs the Marchenko-Pastur rank on the GPU library (no CPU eigvalsh of a D_t x D_t Gram).

Only torch module plumbing lives here (hooks, views, one Linear slice per hooked attention block); no kernels.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .losses import marchenko_pastur_rank

__all__ = ["make_attn_capture_hook", "_to_token_format", "estimate_intrinsic_dim", "extract_intermediates",
           "_extract_student", "_derive_from_teacher"]


# reference src/models/teacher.py:27-39
def make_attn_capture_hook(capture_dict: dict, layer_idx: int, *, apply_softmax: bool = True,
                           cls_row_only: bool = False):
    """Forward hook for a timm-style attention module (``.qkv`` Linear, ``.num_heads``).
    ``cls_row_only=False``: the reference's hook (full (B, H, N, N) map).  ``True``: only query 0, expanded."""

    def hook(mod, inp, out):
        x_in = inp[0]
        B, N, C = x_in.shape
        nh = mod.num_heads
        hd = C // nh
        if not cls_row_only:
            qkv = mod.qkv(x_in).reshape(B, N, 3, nh, hd).permute(2, 0, 3, 1, 4)
            attn = (qkv[0] @ qkv[1].transpose(-2, -1)) * (hd ** -0.5)
            capture_dict[layer_idx] = attn.softmax(dim=-1) if apply_softmax else attn
            return
        w, b = mod.qkv.weight, mod.qkv.bias
        # q of the CLS token, k of every token: two slices of the fused projection (views of its weight)
        q = F.linear(x_in[:, 0, :], w[:C], None if b is None else b[:C]).reshape(B, nh, 1, hd)
        k = F.linear(x_in, w[C:2 * C], None if b is None else b[C:2 * C]).reshape(B, N, nh, hd).permute(0, 2, 1, 3)
        row = (q @ k.transpose(-2, -1)) * (hd ** -0.5)                     # (B, nh, 1, N)
        row = row.softmax(dim=-1) if apply_softmax else row
        capture_dict[layer_idx] = row.expand(B, nh, N, N)                  # zero stride along the query axis

    return hook


# reference src/models/teacher.py:151-158 (pure views: nothing is copied)
def _to_token_format(t: torch.Tensor, feature_format: str, has_cls_token: bool) -> torch.Tensor:
    if feature_format == "nhwc":
        t = t.permute(0, 3, 1, 2).flatten(2).transpose(1, 2)
    elif feature_format == "nchw":
        t = t.flatten(2).transpose(1, 2)                                   # channel-major view, consumed in place
    if has_cls_token:
        t = t[:, 1:, :]
    return t


# reference src/models/teacher.py:161-177
@torch.no_grad()
def estimate_intrinsic_dim(teacher, images: torch.Tensor) -> int:
    """Marchenko-Pastur rank of the teacher's last-layer token representations on calibration images
    (``teacher``: anything with ``.model``, ``.layer_paths``, ``.feature_format``, ``.has_cls_token``)."""
    captured = {}
    mod = teacher.model.get_submodule(teacher.layer_paths[-1])
    h = mod.register_forward_hook(lambda m, i, o: captured.update(out=o))
    teacher.model(images)
    h.remove()
    tokens = _to_token_format(captured["out"], teacher.feature_format, teacher.has_cls_token)
    flat = tokens.reshape(-1, tokens.shape[-1])          # a view where possible; the Gram kernel reads strided rows
    return marchenko_pastur_rank(flat)


# reference src/models/teacher.py:180-215
@torch.no_grad()
def extract_intermediates(teacher, x: torch.Tensor, *, cls_row_only: bool = True):
    """(tokens per layer, attention per layer) as ``BASDLoss.forward`` expects them."""
    if teacher.feature_format != "token":
        features = teacher.model.forward_features(x)
        features = _to_token_format(features, teacher.feature_format, teacher.has_cls_token)
        B, N, _ = features.shape
        # reference: ones(B, 1, N, N) / N; same values as an expanded (B, 1, 1, 1) constant -- N^2 times smaller
        uniform_attn = torch.full((B, 1, 1, 1), 1.0 / N, device=features.device, dtype=features.dtype).expand(B, 1, N, N)
        return {0: features}, {0: uniform_attn}
    hooks, captured_tokens, captured_attns = [], {}, {}
    for idx, path in enumerate(teacher.layer_paths):
        module = teacher.model.get_submodule(path)

        def make_token_hook(i):
            def hook(mod, inp, out):
                captured_tokens[i] = _to_token_format(out, teacher.feature_format, teacher.has_cls_token)
            return hook
        hooks.append(module.register_forward_hook(make_token_hook(idx)))
        if teacher.attn_subpath is not None:
            attn_mod = teacher.model.get_submodule(f"{path}.{teacher.attn_subpath}")
            hooks.append(attn_mod.register_forward_hook(make_attn_capture_hook(
                captured_attns, idx, apply_softmax=True, cls_row_only=cls_row_only and teacher.has_cls_token)))
    teacher.model(x)
    for h in hooks:
        h.remove()
    return captured_tokens, captured_attns


# reference src/training/trainer.py:16-37
def _extract_student(model: nn.Module, x: torch.Tensor, layer_indices: list[int], *, layer_paths: list[str],
                     has_cls_token: bool):
    hooks, captured_tokens = [], {}
    for idx in layer_indices:
        block = model.get_submodule(layer_paths[idx])

        def make_token_hook(i, _has_cls=has_cls_token):
            def hook(mod, inp, out):
                captured_tokens[i] = out[:, 1:, :] if _has_cls else out    # CLS-sliced view: consumed in place
            return hook
        hooks.append(block.register_forward_hook(make_token_hook(idx)))
    logits = model(x)
    for h in hooks:
        h.remove()
    return logits, captured_tokens


# reference src/train.py:57-66
def _derive_from_teacher(teacher, intrinsic_dim: int) -> dict:
    head_dim = teacher.embed_dim // teacher.heads_per_layer[0]
    D_s = math.ceil(intrinsic_dim / head_dim) * head_dim
    D_s = min(D_s, teacher.embed_dim)
    return {"embed_dim": D_s, "depth": teacher.depth, "num_heads": D_s // head_dim, "mlp_ratio": teacher.mlp_ratio}
