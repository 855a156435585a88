"""Seeded synthetic teacher/student feature stacks for the BASD loss path.

The reference needs timm + network access to produce real features, so the
benchmark, the parity tests and the golden-vector script all draw their inputs
from here (low-rank + noise, snr 4; pure iid Gaussian features
give Marchenko-Pastur rank 0 and a NaN loss in the reference).

Everything is generated with a CPU ``torch.Generator`` so that the same seed
gives the same tensors in the build container and on the GPU box (same torch
build); ``device=`` only moves the result.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch


@dataclass(frozen=True)
class LossShape:
    """Shapes of one BASELINE.json configuration, as the loss sees them."""
    name: str
    batch: int
    n_s: int            # student tokens (CLS stripped)
    d_s: int            # student width
    depth: int          # student depth (extraction layers derive from it)
    n_t: int            # teacher tokens (CLS stripped)
    d_t: int            # teacher width
    layers_t: int       # teacher layers seen by the selector
    heads: int          # teacher attention heads (1 for CNN teachers)
    has_cls: bool       # teacher attention carries a CLS row/col
    num_classes: int
    points: int = 4     # basd.num_extraction_points (reference configs/config.yaml:39)
    r_s: int = 32       # synthetic student signal rank
    r_t: int = 48       # synthetic teacher signal rank (ViT teachers: 16 + 2*l)


# The five BASELINE.json configs (SURVEY.md section 8 shape table).
CONFIGS = {
    "cfg1": LossShape("cfg1 DeiT-Tiny<-ResNet-18 32x32", 32, 64, 192, 12, 1, 512, 1, 1, False, 100, r_s=8, r_t=4),
    "cfg2": LossShape("cfg2 DeiT-S<-ResNet-50 224x224", 256, 196, 384, 12, 49, 2048, 1, 1, False, 1000),
    "cfg4": LossShape("cfg4 ViT-B/16<-ViT-L/16", 128, 196, 768, 12, 196, 1024, 24, 16, True, 1000),
    "cfg5": LossShape("cfg5 DeiT-B/384<-ConvNeXt-L", 64, 576, 768, 12, 144, 1536, 1, 1, False, 1000),
}


def extraction_layers(depth: int, points: int) -> list[int]:
    if points == 1:
        return [depth - 1]
    return [round(i * (depth - 1) / (points - 1)) for i in range(points)]


def structured(gen: torch.Generator, b: int, n: int, d: int, rank: int, snr: float = 4.0) -> torch.Tensor:
    """``snr * (U V) / sqrt(rank) + E`` with iid N(0,1) factors, shape (b, n, d)."""
    u = torch.randn(b * n, rank, generator=gen)
    v = torch.randn(rank, d, generator=gen)
    x = snr * (u @ v) / math.sqrt(rank) + torch.randn(b * n, d, generator=gen)
    return x.reshape(b, n, d)


@dataclass
class LossInputs:
    logits: torch.Tensor
    targets: torch.Tensor
    student: dict[int, torch.Tensor]
    teacher: dict[int, torch.Tensor]
    attn: dict[int, torch.Tensor]


def make_inputs(
    shape: LossShape,
    seed: int,
    *,
    batch: int | None = None,
    device: str | torch.device = "cpu",
    dtype: torch.dtype = torch.float32,
    strided: bool = False,
    attn_on_device: bool = False,
) -> LossInputs:
    """Draw one minibatch.  ``strided=True`` reproduces the layouts the reference's
    callers hand over: student / ViT-teacher tokens as
    ``full[:, 1:, :]`` views of a (B, N+1, D) buffer, CNN-teacher tokens as the
    transpose view of a channel-major (B, D, N) buffer."""
    b = shape.batch if batch is None else batch
    gen = torch.Generator().manual_seed(seed)
    logits = torch.randn(b, shape.num_classes, generator=gen)
    targets = torch.randint(0, shape.num_classes, (b,), generator=gen)
    layers = extraction_layers(shape.depth, shape.points)
    student = {}
    for layer in layers:
        x = structured(gen, b, shape.n_s, shape.d_s, shape.r_s).to(dtype)
        student[layer] = _as_cls_view(x, device) if strided else x.to(device)
    teacher = {}
    attn = {}
    for l in range(shape.layers_t):
        rank = shape.r_t if shape.layers_t == 1 else 16 + 2 * l
        t = structured(gen, b, shape.n_t, shape.d_t, rank).to(dtype)
        if strided:
            teacher[l] = _as_cls_view(t, device) if shape.has_cls else _as_channel_major_view(t, device)
        else:
            teacher[l] = t.to(device)
        a = shape.n_t + (1 if shape.has_cls else 0)
        if shape.heads == 1 and not shape.has_cls:
            # CNN teacher: uniform attention                 (reference teacher.py:188-191)
            attn[l] = (torch.ones(b, 1, a, a, dtype=dtype) / a).to(device)
        elif attn_on_device:
            dgen = torch.Generator(device=device).manual_seed(seed * 1000 + l)
            attn[l] = torch.softmax(
                torch.randn(b, shape.heads, a, a, generator=dgen, device=device), dim=-1).to(dtype)
        else:
            attn[l] = torch.softmax(torch.randn(b, shape.heads, a, a, generator=gen), dim=-1).to(dtype).to(device)
    return LossInputs(logits.to(device), targets.to(device), student, teacher, attn)


def _as_cls_view(x: torch.Tensor, device) -> torch.Tensor:
    b, n, d = x.shape
    full = torch.zeros(b, n + 1, d, dtype=x.dtype, device=device)
    full[:, 1:, :] = x.to(device)
    return full[:, 1:, :]


def _as_channel_major_view(x: torch.Tensor, device) -> torch.Tensor:
    chan = x.to(device).transpose(1, 2).contiguous()   # (B, D, N) physical
    return chan.transpose(1, 2)                        # (B, N, D) view
