"""Host side of ``basd_selector_chain`` (csrc/chain.hip): the selector of one loss step -- ranks
(layer_selector.py:69-74), teacher subspaces (:131-138), principal angles and d_grass_sq (:86-105) -- queued by ONE
library call into a persistent, shape-keyed workspace.

A plan owns a few slots of device buffers (``SLOTS``: the tail of step i may still read its factorisation while step i + 1 writes
the next one), the events that order the three streams, the pinned host words the rank kernel writes, and one
pre-filled argument block per slot; a step only patches the input pointers.  No ``torch.empty``, no Python event
objects and a single FFI call on the path in front of the chain the host waits for.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, ops

import os

RELEASE_DELAY = int(os.environ.get("BASD_CHAIN_RELEASE_DELAY", "0"))      # mode 3, rounds of ~3.4 us (BasdSelectorChain.release_delay)
# mode 3: queue the teacher's factorisation FIRST, on a stream of its own, gated by a device word behind the Grams
# (basd_tridiag_ranked_gated): its whole-CU workgroups hold their CUs before the step's throughput launches fill the chip
EARLY_LAUNCH = os.environ.get("BASD_CHAIN_EARLY", "0") == "1"      # measured: ranks 0.3 ms earlier, step 0.1 ms LONGER
EARLY_BUDGET = 1500        # polls of ~1.7 us before the gated kernel gives up (a profiler serialising kernels)
CERTIFICATE = os.environ.get("BASD_RANK_CERT", "1") != "0"      # the rank certificate kernel (BasdSelectorChain.cert_mirror)
# Workspace slots per plan: the chain of step i + SLOTS may only start once the tail of step i has finished.  With the rank
# read-back left to the next call the selector of a step is ~3.4 ms end to end (cfg-2) for a period of ~1.7: two slots made
# every chain wait ~0.15 ms for the tail of two steps back
SLOTS = max(2, int(os.environ.get("BASD_CHAIN_SLOTS", "3")))
# The tails of consecutive steps are independent.  A teacher of high rank makes the tail long (rank 160: 4 ms of latency-bound
# launches in four workgroups) and, on one stream, the period: alternate steps then queue theirs on a second stream (4.16 ->
# 2.95 ms per step).  At rank 48 the tail (1.2 ms) fits the period and the extra stream COSTS 0.3 ms (1.63 -> 1.92: one more
# stream among the step's six makes unrelated launches wait for each other), so it is only taken from TAIL_SPLIT_RANK on.
# BASD_TAIL_STREAMS = 1 / 2 pins the choice.
TAIL_STREAMS = int(os.environ.get("BASD_TAIL_STREAMS", "0"))
TAIL_SPLIT_RANK = 96
SPEC_MARGIN = 8            # eigenvectors computed beyond the previous step's largest rank (the rank may grow a little)


TIMED_EVENTS = False        # diagnostics (tools/step_clock.py): the plans' events carry time stamps
TM_NAMES = ("tm_proj", "tm_tgram", "tm_scol0", "tm_scol1", "tm_sgram", "tm_tri0", "tm_mid", "tm_spec")
# measurement (bench.py): while a list, every queued step appends {name: timed event} of its launches, ev_ranks included
# (created per step: the slots' own events are re-recorded two steps later)
TIMING: list | None = None


def _event(timed: bool | None = None) -> int:
    h = C.c_void_p()
    _lib.call("basd_event_create_timed" if (TIMED_EVENTS if timed is None else timed) else "basd_event_create",
              C.byref(h))
    return h.value


def elapsed_us(ev_from: int, ev_to: int) -> float:
    ms = C.c_float()
    _lib.call("basd_event_elapsed_ms", ev_from, ev_to, C.byref(ms))
    return 1e3 * ms.value


class _Slot:
    """One set of device buffers + events + the argument block that points at them."""

    def __init__(self, plan: "SelectorChainPlan"):
        p = plan
        dev, f32, i32 = p.device, torch.float32, torch.int32
        E, L, n, K = p.E, p.L, p.d_s, p.kmax_cap
        M_t, M_s = p.B * p.n_t, p.B * p.n_s
        tiles = (M_t + 127) // 128
        nmat = 2 * L + E

        def buf(count, dtype=f32):
            return torch.empty((int(count),), device=dev, dtype=dtype)

        self.t_splits = _lib.query("basd_syrk_splits", M_t, n, L)
        self.s_splits = _lib.query("basd_syrk_splits", M_s, n, E)
        self.s_parts = _lib.query("basd_colmean_parts", M_s)
        b = self.bufs = dict(
            z=buf(L * M_t * n), z_sums=buf(L * tiles * n), z_means=buf(L * n), t_slabs=buf(L * self.t_splits * n * n),
            s_partial=buf(E * self.s_parts * n), s_means=buf(E * n), s_slabs=buf(E * self.s_splits * n * n),
            grams=buf(nmat * n * n), d=buf(nmat * n), e=buf(nmat * n), tau=buf(nmat * n), vh=buf(nmat * n * n),
            vals=buf(nmat * n), ranks=buf(L, i32),
            zv=buf((L + E) * K * n), vecs=buf((L + E) * K * n), u_rot=buf(L * K * n), sw=buf(L * K),
            cos=buf(E * L * K * K), sigma=buf(E * L * K), k_arr=buf(E * L, i32),
            jflags=buf(_lib.query("basd_jacobi_workspace_ints", E * L, ops.MAX_SWEEPS), i32),
        )
        split = p.mode not in (0, 4)
        b["tri_work"] = buf(_lib.query("basd_tridiag_workspace_bytes", n, 2 * L if split else nmat), torch.uint8)
        b["tri_work_s"] = buf(_lib.query("basd_tridiag_workspace_bytes", n, E), torch.uint8) if split else None
        b["z_ptrs"] = torch.tensor([b["z"].data_ptr() + 4 * l * M_t * n for l in range(L)], dtype=torch.int64).to(dev)
        b["sw_index"] = torch.tensor(list(range(L)) * E, dtype=i32).to(dev)
        b["go_flag"] = torch.zeros((1,), device=dev, dtype=i32)
        self.go_value = 0
        self.mirror = torch.zeros((L + 8,), dtype=i32).pin_memory()
        self.student_mirror = torch.zeros((8,), dtype=i32).pin_memory() if split else None
        self.ev_fork, self.ev_student, self.ev_ranks, self.ev_tail, self.ev_tgram, self.ev_tg0 = (_event() for _ in range(6))
        self.ev_cert = _event()
        self.cert_mirror = torch.zeros((8,), dtype=i32).pin_memory()       # [0]: 1 = every rank of this step is >= 1, proven
        self.used = False                   # ev_tail has been recorded at least once
        self.student_status_pending = False
        self.teacher_ptrs = (C.c_void_p * L)()
        self.d_out = None                   # this step's (E, L) output tensor (fresh per step)

        a = self.args = _lib.SelectorChainArgs()
        a.teacher_host_ptrs = C.cast(self.teacher_ptrs, C.c_void_p)
        a.E, a.L, a.B, a.n_s, a.n_t, a.d_s, a.d_t = E, L, p.B, p.n_s, p.n_t, n, p.d_t
        a.mp_factor = (1 + (n / M_t) ** 0.5) ** 2          # float64 on the host, as layer_selector.py:11,18
        a.rank_cap, a.kmax_cap, a.mode = n - 1, K, p.mode
        a.t_splits, a.s_splits, a.s_parts = self.t_splits, self.s_splits, self.s_parts
        for name in ("z", "z_sums", "z_ptrs", "z_means", "t_slabs", "s_partial", "s_means", "s_slabs", "grams", "d", "e",
                     "tau", "vh", "vals", "tri_work", "tri_work_s", "ranks", "zv", "vecs", "u_rot", "sw", "cos", "sigma",
                     "k_arr", "sw_index", "jflags"):
            setattr(a, name, None if b[name] is None else b[name].data_ptr())
        a.host_mirror = self.mirror.data_ptr()
        a.student_status_mirror = None if self.student_mirror is None else self.student_mirror.data_ptr()
        a.ev_fork, a.ev_student, a.ev_ranks, a.ev_tail = self.ev_fork, self.ev_student, self.ev_ranks, self.ev_tail
        a.ev_tgram = self.ev_tgram
        a.ev_tg0 = self.ev_tg0
        a.release_delay = RELEASE_DELAY
        if p.cert_stream is not None:
            b["cert_scratch"] = torch.zeros((_lib.query("basd_rank_certificate_scratch_bytes", L),), device=dev,
                                            dtype=torch.uint8)
            a.cert_stream, a.cert_mirror, a.ev_cert = p.cert_stream.cuda_stream, self.cert_mirror.data_ptr(), self.ev_cert
            a.cert_scratch = b["cert_scratch"].data_ptr()
        a.go_budget = EARLY_BUDGET


class SelectorChainPlan:
    """Persistent workspace + argument blocks of ``basd_selector_chain`` for one (shapes, layouts, device) key."""

    @staticmethod
    def key(students, teachers, mode: int):
        s, t = students[0], teachers[0]
        return (len(students), len(teachers), tuple(s.shape), s.stride(), s.dtype, tuple(t.shape), t.stride(), t.dtype,
                str(s.device), mode)

    @staticmethod
    def supported(students, teachers) -> bool:
        s, t = students[0], teachers[0]
        if ops.EIG_SOLVER != "tridiag" or s.dim() != 3 or t.dim() != 3:
            return False
        B, n_t, _ = t.shape
        return B * n_t >= s.shape[2] and s.shape[2] >= 2 and s.shape[2] <= 1024

    def __init__(self, students, teachers, mode: int, streams, fact_stream=None):
        s, t = students[0], teachers[0]
        self.fact_stream = fact_stream
        # the early launch needs the one-kernel factorisation (orders 257..384) and mode 3
        self.early = bool(EARLY_LAUNCH and fact_stream is not None and mode == 3 and 256 < s.shape[2] <= 384)
        self.E, self.L = len(students), len(teachers)
        self.B, self.n_s, self.d_s = s.shape
        _, self.n_t, self.d_t = t.shape
        self.device, self.mode = s.device, mode
        self.kmax_cap = self.d_s - 1
        self.chain_stream, self.student_stream, self.tail_stream = streams
        # the rank certificate (BasdSelectorChain.cert_mirror) is queued on the chain's own stream, between the teacher Grams
        # and the factorisation (~10 us).  On a stream of its own it cost 0.4 ms per step (2.43 against 2.03-2.07 ms with the
        # read-back deferred): one more stream among the step's six made launches of unrelated streams wait for each other
        self.cert_stream = self.chain_stream if CERTIFICATE else None
        self.slots = [_Slot(self) for _ in range(SLOTS)]
        self.turn = 0
        self.hint = 0                      # previous step's largest rank (0: none yet)
        for slot in self.slots:
            a = slot.args
            a.t_dtype, (a.t_sb, a.t_sn, a.t_sd) = ops._dtype_code(t), t.stride()
            a.s_dtype, (a.s_sb, a.s_sn, a.s_sd) = ops._dtype_code(s), s.stride()
            a.chain_stream = self.chain_stream.cuda_stream
            a.student_stream = self.student_stream.cuda_stream
            slot.tail_stream = self.tail_stream
            a.tail_stream = self.tail_stream.cuda_stream

    def speculative_kmax(self) -> int:
        """Eigenvectors the tail computes before the host has seen this step's ranks (0: wait for them)."""
        if self.hint <= 0:
            return 0
        k = min(self.hint + SPEC_MARGIN, self.kmax_cap)
        if _lib.query("basd_jacobi_lds_square_fits", k) or (k >= 96 and _lib.query("basd_jacobi_plain4_fits", k)):
            return k
        return 0

    def fork(self, main_stream: int) -> None:
        """Mark the point of the caller's stream the NEXT ``queue`` may start behind (its inputs are ready there); work the
        caller queues after this call is not waited for by the chain."""
        _lib.call("basd_event_record", self.slots[self.turn].ev_fork, main_stream)
        self._forked = True

    def queue(self, students, teachers, proj_t: torch.Tensor, proj_s_t: torch.Tensor, main_stream: int) -> _Slot:
        """Queue the whole selector of this step; returns the slot whose ``ev_ranks`` / ``mirror`` the host reads."""
        slot = self.slots[self.turn]
        self.turn = (self.turn + 1) % len(self.slots)
        a = slot.args
        if getattr(self, "_forked", False):
            main_stream, self._forked = None, False
        for l, t in enumerate(teachers):
            slot.teacher_ptrs[l] = t.data_ptr()
        a.student_ptrs = ops._ptr_table(students).data_ptr()
        a.s_vec_ok = int(all(x.data_ptr() % 16 == 0 for x in students))
        a.proj_t, a.proj_s_t = proj_t.data_ptr(), proj_s_t.data_ptr()
        a.main_stream = main_stream
        a.ev_slot_free = slot.ev_tail if slot.used else None
        two = TAIL_STREAMS == 2 or (TAIL_STREAMS == 0 and self.hint >= TAIL_SPLIT_RANK)
        if two and self.fact_stream is not None and not self.early:
            self._tail_turn = getattr(self, "_tail_turn", 0) ^ 1
            slot.tail_stream = self.fact_stream if self._tail_turn else self.tail_stream
        else:
            slot.tail_stream = self.tail_stream
        a.tail_stream = slot.tail_stream.cuda_stream
        slot.d_out = torch.empty((self.E, self.L), device=self.device, dtype=torch.float32)
        slot.d_out.record_stream(slot.tail_stream)
        a.d_out = slot.d_out.data_ptr()
        slot.kmax = a.kmax = self.speculative_kmax()
        if self.early:
            slot.go_value = slot.go_value % 0x7FFFFFF0 + 1           # never 0, never the word's previous value
            a.fact_stream, a.go_flag, a.go_value = (self.fact_stream.cuda_stream, slot.bufs["go_flag"].data_ptr(),
                                                    slot.go_value)
        else:
            a.fact_stream, a.go_flag, a.go_value = None, None, 0
        marks = None
        if TIMING is not None:
            marks = {name: _event(True) for name in TM_NAMES}
            for name, ev in marks.items():
                setattr(a, name, ev)
        elif a.tm_proj:
            for name in TM_NAMES:
                setattr(a, name, None)
        _lib.call("basd_selector_chain", C.addressof(a))
        if marks is not None:
            marks["ranks"] = _event(True)           # behind the factorisation + rank kernel, on the stream it runs on
            _lib.call("basd_event_record", marks["ranks"],
                      (self.fact_stream if self.early else self.chain_stream).cuda_stream)
            marks["student_end"] = _event(True)
            _lib.call("basd_event_record", marks["student_end"], self.student_stream.cuda_stream)
            marks["tail_end"] = _event(True)        # behind whatever of the tail this call queued
            _lib.call("basd_event_record", marks["tail_end"], slot.tail_stream.cuda_stream)
            TIMING.append(marks)
        slot.used = slot.used or slot.kmax > 0
        slot.student_status_pending = self.mode not in (0, 4)
        return slot

    def read_ranks(self, slot: _Slot) -> tuple[list[int], list[int]]:
        """Block on the rank kernel of ``slot`` (nothing else); returns (ranks, the factorisation's 8 status words)."""
        _lib.call("basd_event_synchronize", slot.ev_ranks)
        host = slot.mirror.tolist()
        return host[:self.L], host[self.L:]

    def ranks_certified(self, slot: _Slot) -> bool:
        """Block on the certificate kernel of ``slot`` (queued behind the teacher Grams, ~0.6 ms into the chain): True when
        it has PROVEN that no teacher layer of this step has Marchenko-Pastur rank 0 -- the reference's forward cannot
        raise, and the read-back of the ranks themselves may be left to the next step."""
        if self.cert_stream is None:
            return False
        _lib.call("basd_event_synchronize", slot.ev_cert)
        return int(slot.cert_mirror[0]) == 1

    def finish_tail(self, slot: _Slot, ranks: list[int]) -> torch.Tensor:
        """Make sure the tail of ``slot`` has been queued with enough eigenvectors for ``ranks``; returns d_grass_sq
        (E, L), written on the tail stream."""
        need = max(ranks)
        if need > slot.kmax:
            exact = self.L == 1
            kmax = need if exact else min(max(need, self.hint) + SPEC_MARGIN, self.kmax_cap)
            _lib.call("basd_selector_chain_tail", C.addressof(slot.args), kmax, int(exact))
            slot.kmax, slot.used = kmax, True
        self.hint = need
        return slot.d_out
