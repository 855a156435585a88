"""Harness for the trainer-step tests and benchmarks -- NOT part of the product package.

Stand-ins for what the reference takes from ``timm`` / ``torchvision`` (absent from the image, no network): DeiT / ViT
and ResNet trunks with timm's attribute layout (``blocks.N.attn.qkv``, ``cls_token``, ``forward_features``), random
init, plus the model probing the reference's ``src/models/teacher.py:40-148`` does before it builds a ``Trainer``
(same discovery rules, so that the product's ``Trainer`` / ``capture`` see what they would see with the real models).
No kernels here: torch module plumbing only.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ["StockViT", "StockResNet", "TeacherModel", "probe_model", "make_teacher"]


# ----------------------------------------------------------------------------------------------------------------
# stock models (timm layout, random init)
# ----------------------------------------------------------------------------------------------------------------
class _Attention(nn.Module):
    def __init__(self, dim: int, num_heads: int) -> None:
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        y = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        return self.proj(y.transpose(1, 2).reshape(B, N, C))


class _Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int) -> None:
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.fc2(F.gelu(self.fc1(x)))


class _Block(nn.Module):
    def __init__(self, dim: int, num_heads: int, mlp_ratio: float) -> None:
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class StockViT(nn.Module):
    """DeiT / ViT with the attribute layout ``probe_model`` and the hooks rely on (reference teacher.py:40-110)."""

    def __init__(self, *, img_size: int = 224, patch_size: int = 16, embed_dim: int = 384, depth: int = 12,
                 num_heads: int = 6, mlp_ratio: float = 4.0, num_classes: int = 1000) -> None:
        super().__init__()
        self.embed_dim = embed_dim
        self.patch_embed = nn.Conv2d(3, embed_dim, patch_size, patch_size)
        n = (img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.randn(1, n + 1, embed_dim) * 0.02)
        self.blocks = nn.ModuleList([_Block(embed_dim, num_heads, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()

    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        x = self.patch_embed(x).flatten(2).transpose(1, 2)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed
        for blk in self.blocks:
            x = blk(x)
        return self.norm(x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.head(self.forward_features(x)[:, 0])


class _Bottleneck(nn.Module):
    def __init__(self, cin: int, mid: int, cout: int, stride: int) -> None:
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(cin, mid, 1, bias=False), nn.BatchNorm2d(mid)
        self.conv2, self.bn2 = nn.Conv2d(mid, mid, 3, stride, 1, bias=False), nn.BatchNorm2d(mid)
        self.conv3, self.bn3 = nn.Conv2d(mid, cout, 1, bias=False), nn.BatchNorm2d(cout)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + (x if self.down is None else self.down(x)))


class _Basic(nn.Module):
    def __init__(self, cin: int, cout: int, stride: int) -> None:
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False), nn.BatchNorm2d(cout)
        self.conv2, self.bn2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + (x if self.down is None else self.down(x)))


class StockResNet(nn.Module):
    """ResNet-18 / -50 trunk (``num_classes=0`` as the reference's ``load_teacher`` asks timm for).  The four stages
    sit under ``stages`` -- one of the container names ``probe_model`` looks for (timm's ``layer1..4`` are not:
    SURVEY.md Appendix C-8)."""

    def __init__(self, layers=(3, 4, 6, 3), bottleneck: bool = True, width: int = 64) -> None:
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, width, 7, 2, 3, bias=False), nn.BatchNorm2d(width), nn.ReLU(),
                                  nn.MaxPool2d(3, 2, 1))
        stages, cin = [], width
        for i, n in enumerate(layers):
            mid = width * 2 ** i
            cout = mid * 4 if bottleneck else mid
            blocks = []
            for j in range(n):
                stride = 2 if (j == 0 and i > 0) else 1
                blocks.append(_Bottleneck(cin, mid, cout, stride) if bottleneck else _Basic(cin, cout, stride))
                cin = cout
            stages.append(nn.Sequential(*blocks))
        self.stages = nn.Sequential(*stages)
        self.num_features = cin

    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        return self.stages(self.stem(x))                         # (B, C, H, W): "nchw"

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.forward_features(x).mean(dim=(2, 3))


# ----------------------------------------------------------------------------------------------------------------
# teacher wrapper / probing (reference src/models/teacher.py:9-24, 40-110)
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class TeacherModel:
    model: nn.Module
    embed_dim: int
    heads_per_layer: list
    depth: int
    mlp_ratio: float
    layer_paths: list
    attn_subpath: str | None
    has_cls_token: bool
    feature_format: str
    mean: tuple = (0.485, 0.456, 0.406)
    std: tuple = (0.229, 0.224, 0.225)
    extra: dict = field(default_factory=dict)


def probe_model(model: nn.Module, img_size: int) -> dict:
    """Same discovery rules as the reference's ``probe_model`` (teacher.py:40-110); the probe image is created on the
    model's own device."""
    embed_dim = getattr(model, "embed_dim", None) or getattr(model, "num_features", None)
    layer_paths = []
    for name in ("blocks", "layers", "stages"):
        container = getattr(model, name, None)
        if isinstance(container, (nn.Sequential, nn.ModuleList)):
            layer_paths = [f"{name}.{i}" for i in range(len(container))]
            break
    attn_subpath, heads_per_layer, mlp_ratio = None, [], 0.0
    for path in layer_paths:
        block = model.get_submodule(path)
        block_heads = 0
        for child_name, child in block.named_children():
            if hasattr(child, "num_heads"):
                attn_subpath = attn_subpath or child_name
                block_heads = child.num_heads
                break
        heads_per_layer.append(block_heads)
        if mlp_ratio == 0.0:
            for _, child in block.named_children():
                if hasattr(child, "fc1"):
                    mlp_ratio = child.fc1.out_features / embed_dim
                    break
    has_cls_token = any(n == "cls_token" for n, _ in model.named_parameters())
    dev = next(model.parameters()).device
    num_tokens, captured = 0, {}
    with torch.no_grad():
        mod = model.get_submodule(layer_paths[-1])
        h = mod.register_forward_hook(lambda m, i, o: captured.update(out=o))
        was_training = model.training
        model.eval()
        model(torch.zeros(1, 3, img_size, img_size, device=dev))
        model.train(was_training)
        h.remove()
    out = captured["out"]
    if out.dim() == 4:
        feature_format = "nchw" if out.shape[1] > out.shape[3] else "nhwc"
        heads_per_layer = [1]                          # CNN teachers: one synthetic head for the uniform attention
    else:
        feature_format = "token"
        num_tokens = out.shape[1] - int(has_cls_token)
    return {"embed_dim": embed_dim, "heads_per_layer": heads_per_layer, "depth": len(layer_paths),
            "mlp_ratio": mlp_ratio, "layer_paths": layer_paths, "attn_subpath": attn_subpath,
            "has_cls_token": has_cls_token, "feature_format": feature_format, "num_tokens": num_tokens}


def make_teacher(model: nn.Module, img_size: int) -> TeacherModel:
    """``load_teacher`` (teacher.py:113-148) for a model that is already in memory: eval mode, frozen, probed."""
    model.eval()
    for p in model.parameters():
        p.requires_grad = False
    info = probe_model(model, img_size)
    return TeacherModel(model=model, embed_dim=info["embed_dim"], heads_per_layer=info["heads_per_layer"],
                        depth=info["depth"], mlp_ratio=info["mlp_ratio"], layer_paths=info["layer_paths"],
                        attn_subpath=info["attn_subpath"], has_cls_token=info["has_cls_token"],
                        feature_format=info["feature_format"])
