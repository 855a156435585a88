"""A real distillation step around the loss path: the reference's ``Trainer`` inner loop
(``src/training/trainer.py:40-169``) on ROCm with stock torch models, so that the synthetic benchmark's loss call can
be seen inside an actual DeiT <- ResNet / ViT step.

What is mirrored: constructor arguments and attribute names of the reference ``Trainer`` (``basd_loss``, ``optimizer``,
``model``, ``criterion``, ``metrics_history``), ``_train_epoch`` / ``train`` and the per-batch order of operations
(student forward with token hooks, frozen teacher forward, loss, backward, optimizer step) and the checkpoint contents.
The models and the probing that produce ``teacher`` / ``student_info`` are the caller's (the reference's
``src/models/teacher.py``; ``tools/stock_models.py`` for the tests here).  What is fixed:

* the loss runs OUTSIDE autocast on fp32-accumulating kernels (the reference feeds bf16 into ``matrix_norm`` / ``eigvalsh``);
* ``BASDLoss.parameters()`` (the selector's temperatures) are reduced across ranks together with the student gradients
  in ONE flat RCCL all-reduce (``ddp.FlatGradBucket``): the reference leaves them out of ``accelerator.prepare``;
* loaders are sharded with ``DistributedSampler`` (``shard_loader``).

What is absent from the image and therefore replaced: ``schedulefree.AdamWScheduleFree`` (``torch.optim.AdamW``), ``accelerate`` (plain ``torch.distributed``),
``torchvision.transforms.v2.MixUp / CutMix`` (``mixup_cutmix``).  No kernels here: torch module plumbing only.
"""
from __future__ import annotations

import math
from collections import defaultdict

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import capture
from .ddp import FlatGradBucket

__all__ = ["mixup_cutmix", "shard_loader", "Trainer"]


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
    """``Trainer(student_model, config, teacher, student_info=probe_model(student, img_size))`` -- ``teacher`` is the
    reference's ``TeacherModel`` record (``model``, ``embed_dim``, ``layer_paths``, ``attn_subpath``, ``has_cls_token``,
    ``feature_format``, ``heads_per_layer``), ``student_info`` what its ``probe_model`` returns.

    ``config`` needs ``.training.{label_smoothing, learning_rate, weight_decay}``, ``.basd.num_extraction_points`` and
    ``.model.num_classes`` (the fields the reference constructor reads, trainer.py:52-93).  ``loss_cls`` defaults to
    the HIP-backed ``basd_amd.losses.BASDLoss``; tests on CPU hand in the oracle's class.
    ``autocast_dtype``: dtype of the model forward passes (``torch.bfloat16`` as ``Accelerator(mixed_precision="bf16")``,
    or ``None``); the loss always sees fp32 logits and computes in fp32."""

    def __init__(self, student_model: nn.Module, config, teacher, *, student_info: dict, loss_cls=None,
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
        self.metrics_history = defaultdict(list)
        self._params = [p for p in student_model.parameters() if p.requires_grad]
        self._bucket = None
        self.reattached = 0
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # replicas start from rank 0's values
            for p in list(self.model.parameters()) + list(self.basd_loss.parameters()):
                dist.broadcast(p.data, src=0)
            for b in list(self.model.buffers()) + list(self.basd_loss.buffers()):
                dist.broadcast(b.data, src=0)
            # ONE flat buffer [student gradients | selector temperatures]; every parameter's ``.grad`` is a VIEW into it
            # (autograd accumulates in place, ``zero_grad(set_to_none=False)`` keeps the views), so the all-reduce needs
            # no per-parameter copy kernels (~300 small launches a step for DeiT-S when it packed and unpacked)
            self._bucket = FlatGradBucket(sum(p.numel() for p in self._params), list(self.basd_loss.parameters()),
                                          self.device)
            self._bucket.attach_grads(self._params)

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
        """The step's one exchange (reference trainer.py:157 leaves it to ``accelerator.backward``): mean over ranks of
        the flat gradient buffer, queued on the communicator's stream and joined by the current stream before the
        optimizer reads the gradients -- which ARE the buffer (``attach_grads``), so nothing is packed or unpacked."""
        if self._bucket is None:
            return
        # a gradient autograd replaced (or never produced) goes back into the buffer; 0 in the steady state
        self.reattached += self._bucket.reattach_missing()
        self._bucket.all_reduce_mean(async_op=True)
        self._bucket.wait()

    def _train_epoch(self, train_loader, epoch: int = 0) -> dict:
        # a sharded loader draws a different permutation every epoch only if it is told the epoch (every rank the same one)
        sampler = getattr(train_loader, "sampler", None)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)
        total_loss = torch.zeros((), device=self.device)
        correct = torch.zeros((), device=self.device, dtype=torch.long)
        total = 0
        self.model.train()
        for batch in train_loader:
            out = self.train_step(batch)
            total_loss += out["loss"] * out["n"]
            correct += out["correct"]
            total += out["n"]
        self._finish_loss()
        return {"train_loss": (total_loss / total).item(), "train_acc": 100.0 * (correct / total).item()}

    def _finish_loss(self) -> None:
        """Complete what the loss may have deferred past its last call (a rank read-back in ``sync_ranks = False`` mode,
        with the error it carries): nothing of a step may be left pending at an epoch end or in a checkpoint."""
        sel = getattr(self.basd_loss, "layer_selector", None)
        if sel is not None and hasattr(sel, "finish_pending"):
            sel.finish_pending()

    def train(self, train_loader, val_loader=None, start_epoch: int = 0, *, evaluate=None, on_epoch_end=None) -> dict:
        """The reference's epoch loop (trainer.py:171-216): ``_train_epoch``, validation, ``metrics_history``,
        ``best_val_acc``.  Validation is the caller's (``evaluate(model, val_loader) -> {"val_acc": ...}``; the
        reference's ``evaluate_model`` is outside this path); ``on_epoch_end(trainer, epoch, improved)`` is where a
        caller saves checkpoints."""
        for epoch in range(start_epoch, self.config.training.num_epochs):
            metrics = self._train_epoch(train_loader, epoch)
            if evaluate is not None and val_loader is not None:
                self.model.eval()
                metrics.update(evaluate(self.model, val_loader))
            for key, value in metrics.items():
                self.metrics_history[key].append(value)
            improved = metrics.get("val_acc", float("-inf")) > self.best_val_acc
            if improved:
                self.best_val_acc = metrics["val_acc"]
            if on_epoch_end is not None:
                on_epoch_end(self, epoch, improved)
        return self.metrics_history

    # -- checkpoints: what accelerator.save_state + custom_state.pth hold in the reference (trainer.py:94-123)
    def state_dict(self, epoch: int) -> dict:
        self._finish_loss()
        return {"model": self.model.state_dict(), "optimizer": self.optimizer.state_dict(),
                "basd_loss": self.basd_loss.state_dict(), "epoch": epoch, "best_val_acc": self.best_val_acc,
                "metrics_history": dict(self.metrics_history)}

    def save_checkpoint(self, path: str, epoch: int) -> None:
        torch.save(self.state_dict(epoch), path)

    def load_checkpoint(self, path: str) -> int:
        state = torch.load(path, map_location=self.device, weights_only=True)
        self.model.load_state_dict(state["model"])
        self.optimizer.load_state_dict(state["optimizer"])
        self.basd_loss.load_state_dict(state["basd_loss"])
        self.best_val_acc = state["best_val_acc"]
        self.metrics_history = defaultdict(list, state.get("metrics_history", {}))
        return state["epoch"] + 1
