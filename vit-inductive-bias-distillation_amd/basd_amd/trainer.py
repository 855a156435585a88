"""A real distillation step around the loss path: the reference's ``Trainer`` inner loop
(``src/training/trainer.py:40-169``) on ROCm with stock torch models, so that the synthetic benchmark's loss call can
be seen inside an actual DeiT <- ResNet / ViT step.

What is mirrored: constructor arguments and attribute names of the reference ``Trainer`` (``basd_loss``, ``optimizer``,
``model``, ``criterion``), ``_train_epoch`` / the per-batch order of operations (student forward with token hooks,
frozen teacher forward, loss, backward, optimizer step), ``probe_model`` / ``TeacherModel`` (``src/models/teacher.py``)
and the checkpoint contents.  What is fixed:

* the loss runs OUTSIDE autocast on fp32-accumulating kernels (the reference feeds bf16 into ``matrix_norm`` / ``eigvalsh``);
* ``BASDLoss.parameters()`` (the selector's temperatures) are reduced across ranks together with the student gradients
  in ONE flat RCCL all-reduce (``ddp.FlatGradBucket``): the reference leaves them out of ``accelerator.prepare``;
* loaders are sharded with ``DistributedSampler`` (``shard_loader``).

What is absent from the image and therefore replaced: ``timm`` / ``torchvision`` models (``StockViT`` / ``StockResNet``
below: same module layout -- ``blocks.N.attn.qkv``, ``cls_token``, ``forward_features`` -- random init),
``schedulefree.AdamWScheduleFree`` (``torch.optim.AdamW``), ``accelerate`` (plain ``torch.distributed``),
``torchvision.transforms.v2.MixUp / CutMix`` (``mixup_cutmix``).  No kernels here: torch module plumbing only.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import capture
from .ddp import FlatGradBucket

__all__ = ["StockViT", "StockResNet", "TeacherModel", "probe_model", "make_teacher", "mixup_cutmix", "shard_loader",
           "Trainer"]


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


# ----------------------------------------------------------------------------------------------------------------
# data side
# ----------------------------------------------------------------------------------------------------------------
def mixup_cutmix(images: torch.Tensor, targets: torch.Tensor, num_classes: int, *, alpha: float = 1.0):
    """``RandomChoice([MixUp(alpha), CutMix(alpha)])`` of the reference (trainer.py:90-93) on device tensors:
    one Beta(alpha, alpha) draw per batch (global torch RNG), partner = the batch rolled by one, soft
    (B, num_classes) targets out."""
    u = torch.rand(3)
    lam = float(torch.distributions.Beta(alpha, alpha).sample())
    onehot = F.one_hot(targets, num_classes).to(torch.float32)
    if float(u[0]) < 0.5:                                           # MixUp
        mixed = lam * images + (1.0 - lam) * images.roll(1, 0)
    else:                                                           # CutMix: a box of area (1 - lam)
        H, W = images.shape[-2:]
        rh, rw = int(H * math.sqrt(1.0 - lam)), int(W * math.sqrt(1.0 - lam))
        cy, cx = int(float(u[1]) * H), int(float(u[2]) * W)
        y0, y1, x0, x1 = max(cy - rh // 2, 0), min(cy + rh // 2, H), max(cx - rw // 2, 0), min(cx + rw // 2, W)
        mixed = images.clone()
        mixed[..., y0:y1, x0:x1] = images.roll(1, 0)[..., y0:y1, x0:x1]
        lam = 1.0 - (y1 - y0) * (x1 - x0) / float(H * W)
    return mixed, lam * onehot + (1.0 - lam) * onehot.roll(1, 0)


def shard_loader(dataset, batch_size: int, *, shuffle: bool = True, seed: int = 0, **kw):
    """A ``DataLoader`` over this rank's shard (the reference iterates the full loader on every rank)."""
    sampler = None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, shuffle=shuffle, seed=seed, drop_last=True)
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, sampler=sampler,
                                       shuffle=shuffle and sampler is None, drop_last=True, **kw)


# ----------------------------------------------------------------------------------------------------------------
# the trainer
# ----------------------------------------------------------------------------------------------------------------
class Trainer:
    """``Trainer(student_model, config, teacher, student_info=probe_model(student, img_size))``.

    ``config`` needs ``.training.{label_smoothing, learning_rate, weight_decay}``, ``.basd.num_extraction_points`` and
    ``.model.num_classes`` (the fields the reference constructor reads, trainer.py:52-93).  ``loss_cls`` defaults to
    the HIP-backed ``basd_amd.losses.BASDLoss``; tests on CPU hand in the oracle's class.
    ``autocast_dtype``: dtype of the model forward passes (``torch.bfloat16`` as ``Accelerator(mixed_precision="bf16")``,
    or ``None``); the loss always sees fp32 logits and computes in fp32."""

    def __init__(self, student_model: nn.Module, config, teacher: TeacherModel, *, student_info: dict, loss_cls=None,
                 autocast_dtype=None, mixup: bool = True) -> None:
        self.config = config
        self.device = next(student_model.parameters()).device
        self.criterion = nn.CrossEntropyLoss(label_smoothing=config.training.label_smoothing)
        self.model = student_model
        self._teacher = teacher
        self._student_layer_paths = student_info["layer_paths"]
        self._student_has_cls = student_info["has_cls_token"]
        if loss_cls is None:
            from .losses import BASDLoss as loss_cls
        self.basd_loss = loss_cls(
            base_criterion=self.criterion, student_dim=student_info["embed_dim"], teacher_dim=teacher.embed_dim,
            student_depth=student_info["depth"], num_student_tokens=student_info["num_tokens"], config=config.basd,
            teacher_has_cls_token=teacher.has_cls_token,
        ).to(self.device)
        self.optimizer = torch.optim.AdamW(student_model.parameters(), lr=config.training.learning_rate,
                                           weight_decay=config.training.weight_decay)
        self.optimizer.add_param_group({"params": list(self.basd_loss.parameters())})
        self.autocast_dtype = autocast_dtype
        self.mixup = mixup
        self.best_val_acc = 0.0
        self._params = [p for p in student_model.parameters() if p.requires_grad]
        self._bucket = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # one flat buffer: [student gradients | selector temperatures]; replicas start from rank 0's values
            for p in list(self.model.parameters()) + list(self.basd_loss.parameters()):
                dist.broadcast(p.data, src=0)
            for b in list(self.model.buffers()) + list(self.basd_loss.buffers()):
                dist.broadcast(b.data, src=0)
            self._bucket = FlatGradBucket(sum(p.numel() for p in self._params), list(self.basd_loss.parameters()),
                                          self.device)

    # -- one batch: the body of the reference's _train_epoch loop (trainer.py:133-164)
    def train_step(self, batch: dict) -> dict:
        dev = self.device
        clean = batch["clean"].to(dev, non_blocking=True)
        student_imgs = batch["augmented"].to(dev, non_blocking=True)
        targets = batch["label"].to(dev, non_blocking=True)
        mixed_targets = targets
        if self.mixup:
            student_imgs, mixed_targets = mixup_cutmix(student_imgs, targets, self.config.model.num_classes)
        ac = torch.autocast(dev.type, dtype=self.autocast_dtype, enabled=self.autocast_dtype is not None)
        with ac:
            logits, s_tokens = capture._extract_student(self.model, student_imgs, self.basd_loss.token_layers,
                                                        layer_paths=self._student_layer_paths,
                                                        has_cls_token=self._student_has_cls)
            teacher_tokens, teacher_attns = capture.extract_intermediates(self._teacher, clean)
        # the loss is computed outside autocast: fp32 logits, tokens consumed in their own dtype (fp32 internal)
        loss = self.basd_loss(logits.float(), mixed_targets, s_tokens, teacher_tokens, teacher_attns)
        loss.backward()
        self._all_reduce_gradients()
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=False)
        return {"loss": loss.detach(), "correct": logits.detach().argmax(1).eq(targets).sum(), "n": targets.size(0)}

    def _all_reduce_gradients(self) -> None:
        if self._bucket is None:
            return
        flat, off = self._bucket.student_view, 0
        for p in self._params:
            n = p.numel()
            if p.grad is None:
                flat[off:off + n].zero_()
            else:
                flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        self._bucket.pack_loss_grads()
        self._bucket.all_reduce_mean()
        off = 0
        for p in self._params:
            n = p.numel()
            if p.grad is not None:
                p.grad.copy_(flat[off:off + n].view_as(p))
            off += n
        self._bucket.unpack_loss_grads()

    def _train_epoch(self, train_loader) -> dict:
        total_loss = torch.zeros((), device=self.device)
        correct = torch.zeros((), device=self.device, dtype=torch.long)
        total = 0
        self.model.train()
        for batch in train_loader:
            out = self.train_step(batch)
            total_loss += out["loss"] * out["n"]
            correct += out["correct"]
            total += out["n"]
        return {"train_loss": (total_loss / total).item(), "train_acc": 100.0 * (correct / total).item()}

    # -- checkpoints: what accelerator.save_state + custom_state.pth hold in the reference (trainer.py:94-123)
    def state_dict(self, epoch: int) -> dict:
        return {"model": self.model.state_dict(), "optimizer": self.optimizer.state_dict(),
                "basd_loss": self.basd_loss.state_dict(), "epoch": epoch, "best_val_acc": self.best_val_acc}

    def save_checkpoint(self, path: str, epoch: int) -> None:
        torch.save(self.state_dict(epoch), path)

    def load_checkpoint(self, path: str) -> int:
        state = torch.load(path, map_location=self.device, weights_only=True)
        self.model.load_state_dict(state["model"])
        self.optimizer.load_state_dict(state["optimizer"])
        self.basd_loss.load_state_dict(state["basd_loss"])
        self.best_val_acc = state["best_val_acc"]
        return state["epoch"] + 1
