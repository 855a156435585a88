"""Data-parallel glue for the loss path: one process per GPU, the only exchange of a step is the
all-reduce (mean) of the gradients -- student parameters AND the selector's `log_temperatures`, which
the reference leaves out of `accelerator.prepare` (defect 1).

The loss itself needs no collective: every statistic it uses is per-minibatch
(reference layer_selector.py:72,135; relational.py:36-50).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class FlatGradBucket:
    """One contiguous fp32 buffer holding [student gradients | loss-module gradients]; a single
    all-reduce per step over RCCL/xGMI (backend "nccl" on ROCm) or gloo (CPU tests)."""

    def __init__(self, student_numel: int, loss_params: list[torch.nn.Parameter], device) -> None:
        self.loss_params = list(loss_params)
        self.student_numel = int(student_numel)
        self.extra = sum(p.numel() for p in self.loss_params)
        self.buffer = torch.zeros(self.student_numel + self.extra, device=device, dtype=torch.float32)

    @property
    def student_view(self) -> torch.Tensor:
        return self.buffer[: self.student_numel]

    def pack_loss_grads(self) -> None:
        off = self.student_numel
        for p in self.loss_params:
            n = p.numel()
            if p.grad is None:
                self.buffer[off:off + n].zero_()
            else:
                self.buffer[off:off + n].copy_(p.grad.reshape(-1))
            off += n

    def all_reduce_mean(self) -> None:
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if dist.get_backend() == "nccl" and getattr(self, "_avg_ok", True):
                try:
                    dist.all_reduce(self.buffer, op=dist.ReduceOp.AVG)  # RCCL averages in the reduction
                    return
                except (RuntimeError, ValueError):                      # a build without ncclAvg: rejected before launch
                    self._avg_ok = False
            dist.all_reduce(self.buffer, op=dist.ReduceOp.SUM)          # gloo (CPU tests) has no AVG
            self.buffer.div_(dist.get_world_size())

    def unpack_loss_grads(self) -> None:
        off = self.student_numel
        for p in self.loss_params:
            n = p.numel()
            g = self.buffer[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n


def rank_seed(base: int = 1234) -> int:
    """Per-rank data seed (weak scaling: every rank draws its own minibatch)."""
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    return base + rank
