"""Data-parallel glue for the loss path: one process per GPU, the only exchange of a step is the
all-reduce (mean) of the gradients -- student parameters AND the selector's `log_temperatures`, which
the reference leaves out of `accelerator.prepare` (defect 1).

The loss itself needs no collective: every statistic it uses is per-minibatch
(reference layer_selector.py:72,135; relational.py:36-50).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


# test hook (BASD_FORCE_ALLREDUCE=1): run the collective at world size 1 as well -- what the N > 1 plumbing (communicator
# stream, its events) costs a step can then be measured on one GPU
import os as _os
_FORCE_AT_WORLD_1 = _os.environ.get("BASD_FORCE_ALLREDUCE") == "1"


class FlatGradBucket:
    """One contiguous fp32 buffer holding [student gradients | loss-module gradients]; a single
    all-reduce per step over RCCL/xGMI (backend "nccl" on ROCm) or gloo (CPU tests)."""

    def __init__(self, student_numel: int, loss_params: list[torch.nn.Parameter], device, slots: int = 1) -> None:
        """``slots`` > 1: a ring of buffers, so that the all-reduce of step i (queued with ``async_op``) runs on
        the communicator's stream underneath step i+1 while that step fills the next slot."""
        self.loss_params = list(loss_params)
        self.student_numel = int(student_numel)
        self.extra = sum(p.numel() for p in self.loss_params)
        self._slots = [torch.zeros(self.student_numel + self.extra, device=device, dtype=torch.float32)
                       for _ in range(max(1, int(slots)))]
        self._pending: list = [None] * len(self._slots)
        self._slot = 0
        self._attached: list = []
        self._avg_ok = self._probe_avg(device)

    @staticmethod
    def _probe_avg(device) -> bool:
        """Whether the backend reduces with ``ReduceOp.AVG`` (RCCL does, gloo does not): decided ONCE, here, with a
        one-element all-reduce that is waited for -- a rejection may only surface when the work is joined, which a
        try / except around the hot path's async call would never see."""
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return False
        if dist.get_backend() != "nccl":
            return False
        try:
            probe = torch.full((1,), float(dist.get_rank() + 1), device=device, dtype=torch.float32)
            dist.all_reduce(probe, op=dist.ReduceOp.AVG, async_op=True).wait()
            world = dist.get_world_size()
            return abs(float(probe.item()) - (world + 1) / 2.0) < 1e-6
        except (RuntimeError, ValueError):
            return False

    def attach_grads(self, student_params: list[torch.nn.Parameter]) -> None:
        """Make ``.grad`` of every student parameter and of every loss parameter a view into the (single) flat buffer:
        autograd then accumulates straight into it and the optimizer reads the reduced values out of it -- no
        per-parameter pack / unpack kernels.  Needs ``slots == 1`` and fp32 parameters on the buffer's device."""
        assert len(self._slots) == 1, "gradient views need one buffer: the optimizer reads the gradients where they are reduced"
        # an empty list: only the loss parameters are attached (behind the student part, which the caller fills itself)
        assert not student_params or sum(p.numel() for p in student_params) == self.student_numel
        self.buffer.zero_()
        self._attached = []
        off = 0
        if not student_params:
            off = self.student_numel
        for p in list(student_params) + self.loss_params:
            n = p.numel()
            assert p.dtype == torch.float32 and p.device == self.buffer.device
            view = self.buffer[off:off + n].view_as(p)
            p.grad = view
            self._attached.append((p, view))
            off += n

    def reattach_missing(self) -> int:
        """After a backward: a parameter whose ``.grad`` is no longer its view (set to None by somebody, or replaced
        instead of accumulated into) is copied back into the buffer and re-attached.  Returns how many needed it
        (0 in the steady state: the tests assert that)."""
        fixed = 0
        for p, view in self._attached:
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
            else:
                continue
            p.grad = view
            fixed += 1
        return fixed

    @property
    def buffer(self) -> torch.Tensor:
        return self._slots[self._slot]

    def next_slot(self) -> None:
        """Move on to the next buffer of the ring; its previous all-reduce (if still queued) is joined first."""
        self._slot = (self._slot + 1) % len(self._slots)
        self.wait()

    def wait(self, all_slots: bool = False) -> None:
        """Join the all-reduce queued on the current slot (every slot with ``all_slots``): the current stream waits
        for the collective (RCCL: no host block) and the mean is completed where the backend only sums."""
        for i in (range(len(self._slots)) if all_slots else (self._slot,)):
            pending, self._pending[i] = self._pending[i], None
            if pending is not None:
                work, divide = pending
                work.wait()
                if divide:
                    self._slots[i].div_(dist.get_world_size())

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

    def all_reduce_mean(self, async_op: bool = False) -> None:
        """Mean over ranks of the current slot.  ``async_op``: only queued (on the communicator's own stream, behind
        the work already on the current stream); ``wait`` / ``next_slot`` join it."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        if dist.get_world_size() <= 1 and not _FORCE_AT_WORLD_1:
            return
        if self._avg_ok:       # decided once by ``_probe_avg``
            work = dist.all_reduce(self.buffer, op=dist.ReduceOp.AVG, async_op=True)       # RCCL averages in the reduction
            divide = False
        else:
            work = dist.all_reduce(self.buffer, op=dist.ReduceOp.SUM, async_op=True)       # gloo (CPU tests) has no AVG
            divide = True
        self._pending[self._slot] = (work, divide)
        if not async_op:
            self.wait()

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
