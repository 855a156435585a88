"""The reference's loss API (``src/losses``) over the MI355X kernels.

Same class / function names, constructor arguments, state_dict keys, public
attributes and error behaviour as the reference modules
(``src/losses/combined.py``, ``layer_selector.py``, ``relational.py``); the
arithmetic runs in ``libbasd_hip.so``.  Importing this module does not need a
GPU; calling anything numerical without the HIP library or with CPU tensors
raises ``RuntimeError`` (no fallback path exists).
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .chain import SelectorChainPlan

__all__ = [
    "BASDLoss", "GrassmannianLayerSelector", "geometric_relational_loss", "marchenko_pastur_rank",
    "_grassmann_subspace", "_align_token_count",
]


# --------------------------------------------------------------------------- #
# reference src/losses/layer_selector.py:8-20
# --------------------------------------------------------------------------- #
def _uncentred_gram(features: torch.Tensor) -> torch.Tensor:
    """Gram on the smaller side, scaled by 1/M, not centred (layer_selector.py:12-15)."""
    M, D = features.shape
    if M >= D:
        return ops.gemm_tn(features, features, scale=1.0 / M)
    x = features if features.stride(1) == 1 and features.dtype == torch.float32 else features.float().contiguous()
    return ops.gemm_nt(x, x, scale=1.0 / M)


def _eigenvalues_desc(grams: torch.Tensor) -> torch.Tensor:
    """All eigenvalues (descending) of a batch of symmetric matrices (destroyed)."""
    if ops.EIG_SOLVER == "tridiag" and grams.shape[1] >= 2:
        return ops.tridiag_eigenvalues(grams).vals
    return ops.sym_eig(grams)[0]


@torch.no_grad()
def marchenko_pastur_rank(features: torch.Tensor) -> int:
    features = ops.as_supported(features)
    M, D = features.shape
    gram = _uncentred_gram(features).unsqueeze(0).contiguous()
    vals = _eigenvalues_desc(gram)
    rank = ops.mp_rank_device(vals, M, D, cap=1 << 30)
    return int(rank.item())


# --------------------------------------------------------------------------- #
# reference src/losses/layer_selector.py:23-37
# --------------------------------------------------------------------------- #
def _grassmann_subspace(z_flat: torch.Tensor, *, k: int) -> tuple[torch.Tensor, torch.Tensor]:
    """Top-k PCA subspace: (basis (D, k), singular values (k,)) of the column-centred input."""
    with torch.no_grad():
        z = ops.as_supported(z_flat)
        mean = ops.colmean(z)
        gram = ops.gemm_tn(z, z, mean_a=mean, mean_b=mean).unsqueeze(0)
        vals, vecs, _, _ = ops.sym_eig(gram, kmax=max(k, 1))
        k_eff = min(k, min(z.shape))
        basis = vecs[0, :k_eff].T.contiguous()
        svals = ops.sqrt_clamp(vals[0, :k_eff]) if k_eff > 0 else vals[0, :0]
    return basis, svals


# --------------------------------------------------------------------------- #
# reference src/losses/combined.py:9-14
# --------------------------------------------------------------------------- #
class _Resample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens: torch.Tensor, target_n: int):
        x = ops.as_supported(tokens)
        B, n_in, D = x.shape
        tp = ops.taps(n_in, target_n, x.device)
        out = torch.empty((B, target_n, D), device=x.device, dtype=torch.float32)
        ops._lib.call("basd_resample_tokens", x.data_ptr(), ops._dtype_code(x), x.stride(0), x.stride(1), x.stride(2),
                      B, n_in, target_n, D, tp.tap0.data_ptr(), tp.tap1.data_ptr(), tp.lam.data_ptr(),
                      out.data_ptr(), ops._stream())
        ctx.shape = (B, n_in, D, target_n)
        ctx.in_dtype = tokens.dtype
        return out.to(tokens.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        B, n_in, D, n_out = ctx.shape
        g = grad_out.float().contiguous()
        tp = ops.taps(n_in, n_out, g.device)
        dx = torch.empty((B, n_in, D), device=g.device, dtype=torch.float32)
        ops._lib.call("basd_resample_tokens_adjoint", g.data_ptr(), B, n_in, n_out, D, tp.tap0.data_ptr(),
                      tp.tap1.data_ptr(), tp.lam.data_ptr(), tp.range0.data_ptr(), tp.range1.data_ptr(),
                      dx.data_ptr(), ops._stream())
        return dx.to(ctx.in_dtype), None


def _align_token_count(tokens: torch.Tensor, target_n: int) -> torch.Tensor:
    if tokens.shape[1] == target_n:
        return tokens
    return _Resample.apply(tokens, target_n)


# --------------------------------------------------------------------------- #
# Procrustes loss over all extraction layers (autograd boundary)
# --------------------------------------------------------------------------- #
class TridiagGiveUp(RuntimeError):
    """Workgroups sharing a matrix in the tridiagonalisation's first stage lost each other (bounded spin): the eigen-
    solve results of this step are invalid.  The callers re-queue the step's selector with ONE workgroup per matrix
    (nothing waits for anything then) and keep that setting: a slower selector instead of a dead training run."""


def _single_member_mode(why: str) -> None:
    import warnings
    warnings.warn("basd_tridiag: " + why + "; switching the tridiagonalisation to one workgroup per matrix "
                  "(no inter-workgroup hand-off; slower first stage) for the rest of this process", RuntimeWarning)
    ops._lib.call("basd_tridiag_tuning", 1, -1, -1, -1, -1, 0)


def _record_stream(obj, stream) -> None:
    """Mark every tensor reachable from a (nested) state object as in use on `stream`."""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)
    elif hasattr(obj, "__dataclass_fields__"):
        for name in obj.__dataclass_fields__:
            _record_stream(getattr(obj, name), stream)


def ctx_zero(ctx, like: torch.Tensor):
    if ctx.zero_shape is None or not ctx.needs_input_grad[3]:
        return None
    return torch.zeros(ctx.zero_shape[0], device=like.device, dtype=ctx.zero_shape[1])


class _ProcrustesLayers(torch.autograd.Function):
    """(mix (E, L), has_cls, eager, zero_param, teachers, attns, *students) -> per-layer loss (E,).

    ``zero_param``: a parameter whose gradient through this loss is identically zero but must still be delivered
    as zeros (the temperatures with ONE teacher layer: softmax over a single logit); backward hands it zeros
    directly instead of pushing zeros through softmax / division / softplus nodes.

    ``eager``: queue the student-token gradient kernels (for a unit upstream gradient) right behind the forward
    kernels.  The gradient is linear in the upstream scalar, so backward only scales -- used where the caller is
    about to block on a read-back anyway, which takes the gradient launches off the host's critical path."""

    @staticmethod
    def forward(ctx, mix, has_cls, eager, zero_param, teachers, attns, *students):
        need_bwd = any(s.requires_grad for s in students)
        # one teacher layer: softmax over one logit is constant, d loss / d mix is exactly 0
        need_mix = bool(mix.requires_grad and mix.shape[1] > 1)
        pc = ops.procrustes_forward(list(students), teachers, attns, mix.detach(), has_cls,
                                    need_backward=need_bwd or need_mix, need_mix_grad=need_mix)
        ctx.pc = pc
        ctx.n_students = len(students)
        ctx.save_for_backward(*students)
        ctx.need_mix = need_mix
        ctx.mix_shape = tuple(mix.shape)
        ctx.zero_shape = None if zero_param is None else (tuple(zero_param.shape), zero_param.dtype)
        ctx.unit_grads = None
        if eager and need_bwd and not need_mix and pc.k_prime is not None:
            ones = ops._device_consts((1.0,) * len(students), torch.float32, pc.loss_b.device)
            ctx.unit_grads = ops.procrustes_student_grads(list(students), pc, ones)
        return pc.loss_b.mean(dim=1)

    @staticmethod
    def backward(ctx, grad_layers):
        ops.trace("bwd_procrustes_in")
        students = ctx.saved_tensors
        pc = ctx.pc
        g_mix = None
        if pc.k_prime is None:
            if ctx.needs_input_grad[0]:
                g_mix = torch.zeros(ctx.mix_shape, device=grad_layers.device)
            return (g_mix, None, None, ctx_zero(ctx, grad_layers), None, None) + (None,) * ctx.n_students
        if ctx.need_mix:
            kt, tnorm2 = ops.procrustes_teacher_factor(pc)
            grads, gomega = ops.procrustes_student_grads(list(students), pc, grad_layers, tnorm2)
            g_mix = ops.procrustes_mix_grads(pc, kt, gomega, grad_layers)
        else:
            if ctx.unit_grads is not None:
                gl = grad_layers.float()
                grads = [u * gl[e] for e, u in enumerate(ctx.unit_grads)]
            else:
                grads = ops.procrustes_student_grads(list(students), pc, grad_layers)
            if ctx.needs_input_grad[0]:
                g_mix = torch.zeros(ctx.mix_shape, device=grad_layers.device)
        grads = [g.to(s.dtype) if ctx.needs_input_grad[6 + i] else None
                 for i, (g, s) in enumerate(zip(grads, students))]
        ops.trace("bwd_procrustes_out")
        return (g_mix, None, None, ctx_zero(ctx, grad_layers), None, None, *grads)


class _FusedCrossEntropy(torch.autograd.Function):
    """torch.nn.CrossEntropyLoss (mean, no class weights) as ONE kernel that leaves the loss and the unit gradient of the
    logits; backward is one scaling.  (The stock module queues ~15 micro-kernels for forward + backward, and their
    host time sits between the rank read-back and the end of the step.)"""

    @staticmethod
    def forward(ctx, logits, targets, eps, ignore_index):
        x = logits if logits.stride(1) == 1 else logits.contiguous()
        B, C = x.shape
        row = torch.empty((B,), device=x.device, dtype=torch.float32)
        dlog = torch.empty((B, C), device=x.device, dtype=torch.float32)
        hard = not targets.is_floating_point()
        t = targets.contiguous() if hard else targets.float().contiguous()
        ops._lib.call("basd_cross_entropy", x.data_ptr(), ops._dtype_code(x), x.stride(0), B, C,
                      t.data_ptr() if hard else None, None if hard else t.data_ptr(), 0 if hard else t.stride(0),
                      float(eps), int(ignore_index), row.data_ptr(), dlog.data_ptr(), ops._stream())
        ctx.save_for_backward(dlog)
        ctx.dtype = logits.dtype
        return row.sum().to(logits.dtype)

    @staticmethod
    def backward(ctx, g):
        (dlog,) = ctx.saved_tensors
        return (dlog * g).to(ctx.dtype), None, None, None


def _base_loss(criterion: nn.Module, logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """``criterion(logits, targets)``; the stock cross entropy on CUDA tensors takes the fused kernel."""
    if (type(criterion) is nn.CrossEntropyLoss and criterion.weight is None and criterion.reduction == "mean"
            and logits.is_cuda and logits.dim() == 2 and logits.dtype in (torch.float32, torch.bfloat16)
            and ((targets.dim() == 1 and targets.dtype == torch.int64)
                 or (targets.dim() == 2 and targets.is_floating_point() and targets.shape == logits.shape))):
        return _FusedCrossEntropy.apply(logits, targets, criterion.label_smoothing, criterion.ignore_index)
    return criterion(logits, targets)


class _UWSOCombine(torch.autograd.Function):
    """(ce, geo_layers (E,)) -> w_ce * ce + w_geo * mean(geo_layers), UW-SO weights w_i = (1/L_i) / sum_j (1/L_j)
    computed from the detached loss values (combined.py:78-85).  One autograd node instead of the eight the
    elementwise expression builds: the weights carry no gradient, so the backward is two scalings."""

    @staticmethod
    def forward(ctx, ce, geo_layers):
        geo = geo_layers.mean()
        vals = torch.stack([ce, geo.to(ce.dtype)])
        eps = torch.finfo(ce.dtype).eps
        inv = 1.0 / vals.clamp(min=eps)
        w = inv / inv.sum()
        ctx.save_for_backward(w)
        ctx.n_layers = geo_layers.numel()
        ctx.geo_dtype = geo_layers.dtype
        return (w * vals).sum()

    @staticmethod
    def backward(ctx, g):
        ops.trace("bwd_uwso_in")
        (w,) = ctx.saved_tensors
        gw = g * w
        return gw[0], (gw[1] / ctx.n_layers).to(ctx.geo_dtype).expand(ctx.n_layers)


class _SingleTeacherTotal(torch.autograd.Function):
    """(ce, has_cls, zero_param, teachers, attns, *students) -> (total, geo_layers): the Procrustes loss of every
    extraction layer against ONE teacher layer (mixing weights exactly 1) and its UW-SO combination with the base
    loss, as a single autograd node.  The student gradients for a unit upstream gradient are queued by forward
    (see ``_ProcrustesLayers``); backward is one small product for the weights and one scaling per layer."""

    @staticmethod
    def forward(ctx, ce, has_cls, zero_param, teachers, attns, *students):
        E = len(students)
        need_bwd = any(s.requires_grad for s in students)
        ctx.n_students = E
        ctx.dtypes = [s.dtype for s in students]
        ctx.zero_shape = None if zero_param is None else (tuple(zero_param.shape), zero_param.dtype)
        ones = ops._device_consts((1.0,) * E, torch.float32, ce.device)
        # the UW-SO combination inside the library call when the base loss is one fp32 on the device: the student
        # gradients queued by forward are then FINAL for a unit upstream gradient (no 308 MB scaling pass in backward,
        # no dozen torch micro-kernels for the weights between the Procrustes kernels and the rank read-back)
        ctx.final = need_bwd and ce.dtype == torch.float32 and ce.numel() == 1 and ce.is_cuda
        if ctx.final:
            pc = ops.procrustes_forward(list(students), teachers, attns, ones.view(-1, 1), has_cls, need_backward=True,
                                        need_mix_grad=False, uwso_ce=ce.detach().reshape(1))
            uw = pc.uw
            geo_layers = uw[4 + E:4 + 2 * E]
            ctx.unit_grads = pc.dx      # (E, B, N_s, D_s): d total / d student tokens for a unit upstream gradient
            ctx.applied = None          # upstream gradient the buffer has been scaled by (None: 1)
            ctx.save_for_backward(uw[:2])
            ctx.mark_non_differentiable(geo_layers)
            return uw[2].reshape(()), geo_layers
        pc = ops.procrustes_forward(list(students), teachers, attns, ones.view(-1, 1), has_cls, need_backward=need_bwd,
                                    need_mix_grad=False, grad_layers=ones if need_bwd else None)
        geo_layers = pc.loss_b.mean(dim=1)
        ctx.unit_grads = pc.dx          # (E, B, N_s, D_s): gradients for a unit upstream gradient, one buffer
        # UW-SO (combined.py:78-85) on the detached values
        vals = torch.stack([ce.detach(), geo_layers.mean().to(ce.dtype)])
        inv = 1.0 / vals.clamp(min=torch.finfo(ce.dtype).eps)
        w = inv / inv.sum()
        ctx.save_for_backward(w)
        ctx.mark_non_differentiable(geo_layers)
        return (w * vals).sum(), geo_layers

    @staticmethod
    def backward(ctx, g, _g_geo):
        ops.trace("bwd_total_in")
        (w,) = ctx.saved_tensors
        gw = g * w
        grads = [None] * ctx.n_students
        if ctx.unit_grads is not None and ctx.final:
            gf = g.detach().float().reshape(1)
            if ctx.applied is None:
                # in place; a launch that returns at once for the usual upstream gradient of exactly 1
                ops.scale_unless_one(ctx.unit_grads, gf)
                ctx.applied = gf
                buf = ctx.unit_grads
            else:
                # a second backward through a retained graph: the buffer may be aliased by gradients handed out before
                buf = ctx.unit_grads * (gf / ctx.applied)
            grads = [buf[i].to(dt) if ctx.needs_input_grad[5 + i] else None for i, dt in enumerate(ctx.dtypes)]
        elif ctx.unit_grads is not None:
            scale = gw[1] / ctx.n_students
            scaled = ctx.unit_grads * scale         # ONE launch over all layers (they share the upstream scalar)
            grads = [scaled[i].to(dt) if ctx.needs_input_grad[5 + i] else None for i, dt in enumerate(ctx.dtypes)]
        zero = None
        if ctx.zero_shape is not None and ctx.needs_input_grad[2]:
            zero = torch.zeros(ctx.zero_shape[0], device=g.device, dtype=ctx.zero_shape[1])
        ops.trace("bwd_total_out")
        ops.gpu_mark("bwd_procrustes_end")
        return (gw[0] if ctx.needs_input_grad[0] else None, None, zero, None, None, *grads)


class _GrassmannDistance(torch.autograd.Function):
    """(selector, keys, teachers, *students) -> d_grass_sq (E, L), differentiable w.r.t. the student tokens
    (reference layer_selector.py:86-105; the backward is the eigenvector-perturbation route)."""

    @staticmethod
    def forward(ctx, selector, keys, teachers, *students):
        want_grad = any(s.requires_grad for s in students)
        # the student side (E Grams + eigen-solves) is independent of the teacher side (2L) and may take its own stream,
        # joined where the cosine matrices need both.  OFF by default: at cfg-4 the two shared tridiagonalisation stages
        # side by side cost more than queueing them behind each other (63.7 vs 58.3 ms per step)
        ss = selector._student_side_stream(students[0].device) if selector.overlap_student_side else None

        def run():
            if ss is not None:
                ss.wait_stream(torch.cuda.current_stream())
                for s in students:
                    s.record_stream(ss)
            spectra = selector._spectra_async(list(students), teachers, all_student_vectors=want_grad, student_stream=ss,
                                              gate_student=False)
            out = selector._angles_from_spectra(spectra, keys, want_grad=want_grad)
            if ss is not None:
                # allocated on the student stream, kept for the backward on this one
                _record_stream([spectra.get("s_ts"), spectra.get("means"), spectra.get("s_stack"),
                                spectra.get("s_colnorm"), out[1]], torch.cuda.current_stream())
            return out
        try:
            d, saved = run()
        except TridiagGiveUp as exc:       # degrade, do not die: once more with one workgroup per matrix
            _single_member_mode(str(exc))
            torch.cuda.synchronize(students[0].device)
            d, saved = run()
        ctx.saved = saved
        ctx.selector = selector
        ctx.n_students = len(students)
        ctx.save_for_backward(*students)
        return d

    @staticmethod
    def backward(ctx, gd):
        students = ctx.saved_tensors
        if ctx.saved is None:
            return (None, None, None) + (None,) * ctx.n_students
        grads = ctx.selector._distance_backward(ctx.saved, list(students), gd)
        grads = [g.to(s.dtype) if ctx.needs_input_grad[3 + i] else None
                 for i, (g, s) in enumerate(zip(grads, students))]
        return (None, None, None, *grads)


class _RelationalAllInputs(torch.autograd.Function):
    """(student, teacher, attn, has_cls) -> loss, differentiable w.r.t. ALL three tensors like the reference function
    (relational.py:18-50).  The trainer never needs the teacher side (teacher outputs carry no grad, teacher.py:180);
    the stand-alone function's contract does.  Teacher tokens arrive already on the student grid, so
        d loss / d T = (2 / B) Kt T_c                       (the centring's adjoint vanishes: 1^T Kt T_c = 0)
        d loss / d attn: d loss_b / d omega -> normalisation -> weight interpolation -> head / query mean."""

    @staticmethod
    def forward(ctx, student, teacher, attn, has_cls):
        pc = ops.procrustes_forward([student], [teacher], [attn],
                                    ops._device_consts((1.0,), torch.float32, student.device).view(1, 1), has_cls,
                                    need_backward=True, need_mix_grad=True)
        ctx.pc, ctx.has_cls = pc, has_cls
        ctx.save_for_backward(student)
        ctx.shapes = (teacher.shape, teacher.dtype, attn.shape, attn.dtype)
        return pc.loss_b[0].mean()

    @staticmethod
    def backward(ctx, g):
        (student,) = ctx.saved_tensors
        pc = ctx.pc
        t_shape, t_dtype, a_shape, a_dtype = ctx.shapes
        B = t_shape[0]
        gl = g.reshape(1).float()
        kt, tnorm2 = ops.procrustes_teacher_factor(pc)
        grads, gomega = ops.procrustes_student_grads([student], pc, gl, tnorm2)
        _, r, g_raw = ops.procrustes_mix_grads(pc, kt, gomega, gl, want_inputs=True)
        g_s = grads[0].to(student.dtype) if ctx.needs_input_grad[0] else None
        g_t = g_a = None
        if ctx.needs_input_grad[1]:
            g_t = (r.view(t_shape) * (2.0 / B * gl)).to(t_dtype)
        if ctx.needs_input_grad[2]:
            _, H, A, _ = a_shape
            g_a = torch.zeros(a_shape, device=g.device, dtype=torch.float32)
            w = g_raw[0] * (gl / B)                                     # (B, n_a)
            if ctx.has_cls:
                g_a[:, :, 0, 1:] = (w / H).unsqueeze(1)                 # relational.py:24: CLS row, head mean
            else:
                g_a[:] = (w / (H * A)).view(B, 1, 1, -1)                # relational.py:27: head and query mean
            g_a = g_a.to(a_dtype)
        return g_s, g_t, g_a, None


# --------------------------------------------------------------------------- #
# reference src/losses/relational.py:5-50
# --------------------------------------------------------------------------- #
def geometric_relational_loss(
    student_tokens: torch.Tensor,
    teacher_tokens: torch.Tensor,
    teacher_attn: torch.Tensor,
    *,
    has_cls_token: bool,
) -> torch.Tensor:
    """Attention-weighted Procrustes loss between (B, N_s, D_s) student tokens and (B, N_s, D_t)
    teacher tokens already on the student grid.  Differentiable w.r.t. all three tensors, like the reference."""
    if teacher_tokens.shape[1] != student_tokens.shape[1]:
        raise RuntimeError("teacher_tokens must already be aligned to the student token count")
    if teacher_tokens.requires_grad or teacher_attn.requires_grad:
        return _RelationalAllInputs.apply(student_tokens, teacher_tokens, teacher_attn, bool(has_cls_token))
    mix = torch.ones((1, 1), device=student_tokens.device, dtype=torch.float32)
    return _ProcrustesLayers.apply(mix, bool(has_cls_token), False, None, [teacher_tokens], [teacher_attn], student_tokens)[0]


# --------------------------------------------------------------------------- #
# reference src/losses/layer_selector.py:40-152
# --------------------------------------------------------------------------- #
class GrassmannianLayerSelector(nn.Module):
    def __init__(self, num_extraction_points: int, student_dim: int, teacher_dim: int):
        super().__init__()
        self.student_dim = student_dim
        self._subspace_ranks: dict[int, int] = {}
        self._pending_tail = None          # deferred rank read-back + selector tail of the latest forward
        self.gate_student_chain = os.environ.get("BASD_STUDENT_GATE", "1") != "0"
        self.teacher_space_gram = os.environ.get("BASD_TEACHER_SPACE_GRAM", "1") != "0"
        self.rank1_mp = os.environ.get("BASD_RANK1_MP", "1") != "0"
        self.merge_factorisations = os.environ.get("BASD_MERGE_FACTORISATIONS", "1") != "0"   # cfg-4: 46.4 -> ? ms
        self.overlap_student_side = os.environ.get("BASD_OVERLAP_STUDENT_SIDE", "0") != "0"    # measured: 63.7 vs 58.3 ms at cfg-4
        self._student_streams: dict = {}

        # Same global-RNG consumption order as the reference (proj_s, then proj_t), on CPU.
        proj_s = torch.empty(student_dim, student_dim)
        proj_t = torch.empty(student_dim, teacher_dim)
        nn.init.orthogonal_(proj_s)
        nn.init.orthogonal_(proj_t)
        self.register_buffer("proj_s", proj_s)
        self.register_buffer("proj_t", proj_t)
        self.log_temperatures = nn.Parameter(
            torch.full((num_extraction_points,), math.log(math.exp(1.0) - 1)))

    @property
    def temperatures(self) -> torch.Tensor:
        return F.softplus(self.log_temperatures)

    # ``subspace_ranks`` is the reference's public dict (layer_selector.py:48,74), refreshed by every forward.
    # With one teacher layer the ranks do not feed the loss, so ``BASDLoss.forward`` does not wait for them: the
    # read-back (and the rank-0 error the reference raises inside forward) is completed by the next forward, or
    # by whoever reads this attribute first -- reading it blocks exactly like the reference's ``.item()`` does.
    @property
    def subspace_ranks(self) -> dict[int, int]:
        self.finish_pending()
        return self._subspace_ranks

    @subspace_ranks.setter
    def subspace_ranks(self, value: dict[int, int]) -> None:
        self._subspace_ranks = value

    def finish_pending(self) -> None:
        """Complete the deferred part of the latest forward: wait for its rank kernel, refresh
        ``subspace_ranks``, raise what the reference would have raised, queue the rest of the selector."""
        pending, self._pending_tail = self._pending_tail, None
        if pending is not None:
            pending()

    # ---- teacher side: ranks + subspaces -------------------------------------------------
    @torch.no_grad()
    def _teacher_projections(self, teachers: list[torch.Tensor]) -> tuple[list[torch.Tensor], torch.Tensor]:
        """Per teacher layer z = tokens @ proj_t^T (M, d_s)   (layer_selector.py:72 / :135), plus the column sums of
        every 128-row tile of z from the projection kernel's epilogue (L, tiles, d_s): the Gram launch folds the
        column means of z (:35) from them, so no separate reduction sits on the teacher chain."""
        d_s, L = self.student_dim, len(teachers)
        M = teachers[0].shape[0] * teachers[0].shape[1]
        tiles = ops.gemm_nt_row_tiles(M, d_s, teachers[0].shape[2])
        sums = torch.empty((L, tiles, d_s), device=teachers[0].device, dtype=torch.float32)
        proj_t = self.proj_t.float().contiguous()
        zs = [ops.gemm_nt(ops.as_supported(t), proj_t, col_sums=sums[l]) for l, t in enumerate(teachers)]
        return zs, sums

    def _teacher_grams(self, teachers: list[torch.Tensor], projected: list[torch.Tensor] | None = None):
        """Per teacher layer: projected tokens -> uncentred Gram / M (for the MP rank, layer_selector.py:12-15)
        and centred Gram (for the subspace, :35).  ``projected``: projections already queued by the caller."""
        d_s = self.student_dim
        L = len(teachers)
        B, n_t, _ = teachers[0].shape
        M = B * n_t
        proj_t = self.proj_t.float().contiguous()
        n_u = d_s if M >= d_s else M
        if projected is None and self._teacher_space_route(teachers):
            return self._teacher_grams_in_teacher_space(teachers)
        zs, sums = projected if projected is not None else self._teacher_projections(teachers)
        if n_u == d_s:
            # uncentred / M and centred Grams of every layer's projected tokens: one symmetric launch,
            # laid out [uncentred 0..L-1, centred 0..L-1]
            stack, _ = ops.centered_grams(zs + zs, scales=[1.0 / M] * L + [1.0] * L, fold=(sums, L))
            return stack[:L], stack[L:], M, stack
        g_u = torch.empty((L, n_u, n_u), device=proj_t.device, dtype=torch.float32)
        for l, z in enumerate(zs):
            g_u[l] = _uncentred_gram(z)
        g_c, _ = ops.centered_grams(zs, fold=(sums, 0))
        return g_u, g_c, M, None

    def _teacher_space_route(self, teachers: list[torch.Tensor]) -> bool:
        B, n_t, d_t = teachers[0].shape
        return B * n_t >= self.student_dim and d_t <= 1.5 * self.student_dim and self.teacher_space_gram

    def _teacher_grams_in_teacher_space(self, teachers: list[torch.Tensor], centred_out: torch.Tensor | None = None):
        """The same two Grams WITHOUT the projected tokens, for teachers about as wide as the student (ViT teachers:
        D_t <= 1.5 D_s): z = t P^T is never formed.  One centred Gram per layer in the teacher's own space G_c = t_c^T t_c
        (lower tiles: 29.6 GF at 25088 x 1024), then P G_c P^T (2.8 GF) = the centred Gram of z, and its uncentred Gram
        / M = (P G_c P^T + M zbar zbar^T) / M with zbar = P tbar -- an addition, no cancellation.  32 GF per layer
        instead of 74 (projection 39.5 + two 768^2 Grams of z), and the reference's second projection of every layer
        (layer_selector.py:135) is not paid either."""
        d_s, L = self.student_dim, len(teachers)
        B, n_t, d_t = teachers[0].shape
        M = B * n_t
        proj_t = self.proj_t.float().contiguous()
        g_t, tbar = ops.centered_grams(teachers)                                   # (L, d_t, d_t), (L, d_t)
        wt = ops.gemm_nt(proj_t, g_t[0], batch=L, b_batch_stride=d_t * d_t, rows=d_s, n_cols=d_t)      # P G_c
        wt = wt.view(L, d_s, d_t)
        c = ops.gemm_nt(wt[0], proj_t, batch=L, a_batch_stride=d_s * d_t, rows=d_s, n_cols=d_s).view(L, d_s, d_s)
        zbar = ops.gemm_nt(tbar, proj_t)                                           # (L, d_s)
        self._teacher_zbar = zbar          # column means of the projected tokens (the rank-one route of the MP rank)
        if centred_out is not None:
            # only the centred Gram, where the caller wants it (one factorisation launch over teacher + student matrices;
            # the MP rank comes from basd_tridiag_mp_rank_rank1)
            assert centred_out.shape == (L, d_s, d_s) and centred_out.is_contiguous()
            ops._lib.call("basd_gram_finish", c.data_ptr(), zbar.data_ptr(), d_s, L, M, None, centred_out.data_ptr(),
                          ops._stream())
            return None, centred_out, M, None
        stack = torch.empty((2 * L, d_s, d_s), device=c.device, dtype=torch.float32)
        ops._lib.call("basd_gram_finish", c.data_ptr(), zbar.data_ptr(), d_s, L, M, stack.data_ptr(),
                      stack[L].data_ptr(), ops._stream())
        return stack[:L], stack[L:], M, stack

    @torch.no_grad()
    def _estimate_ranks(self, all_teacher_tokens: dict[int, torch.Tensor]) -> None:
        keys = list(all_teacher_tokens.keys())
        teachers = ops._check_common_layout([ops.as_supported(all_teacher_tokens[k]) for k in keys],
                                            "teacher token tensors")
        g_u, _, M, _ = self._teacher_grams(teachers)
        vals_u = _eigenvalues_desc(g_u)
        ranks_dev = ops.mp_rank_device(vals_u, M, self.student_dim, cap=self.student_dim - 1)   # :74
        self.finish_pending()
        for k, r in zip(keys, ranks_dev.tolist()):
            self._subspace_ranks[k] = int(r)

    def _student_side_stream(self, device) -> "torch.cuda.Stream":
        key = str(device)
        if key not in self._student_streams:
            self._student_streams[key] = torch.cuda.Stream(device=device)
        return self._student_streams[key]

    def _proj_s_transposed(self) -> torch.Tensor:
        """proj_s^T, fp32 contiguous; cached (the buffer only changes on load_state_dict / .to())."""
        key = (self.proj_s.data_ptr(), self.proj_s._version, self.proj_s.dtype)
        if getattr(self, "_proj_s_t_key", None) != key:
            self._proj_s_t = self.proj_s.float().t().contiguous()
            self._proj_s_t_key = key
        return self._proj_s_t

    # ---- distances + mixing weights ------------------------------------------------------
    @torch.no_grad()
    def _spectra_async(self, students: list[torch.Tensor], teachers: list[torch.Tensor],
                       all_student_vectors: bool = False, student_stream=None, defer_student: bool = False,
                       gate_student: bool = True) -> dict:
        """Queue every Gram matrix and eigen-solve of the step; no host sync.
        (layer_selector.py:69-74, :131-138, :86-92)

        Solver choice: the teacher matrices need eigenvalues (MP rank) and the leading k eigenvectors, the
        student matrices the leading k eigenvectors -- tridiagonalisation + bisection + inverse iteration.
        Only the backward of multi-layer teachers needs ALL student eigenvectors: those go through Jacobi.

        The teacher chain runs on the current stream; with ``student_stream`` the (independent) student chain
        is queued there, so the latency-bound eigen-solves of the two sides overlap.  ``defer_student``: when the student
        chain is gated behind the teacher's first tridiagonalisation stage anyway, leave its queueing to the caller
        (``st["queue_student"]()``): the host can queue the Procrustes kernels first."""
        d_s = self.student_dim
        E, L = len(students), len(teachers)
        dev = students[0].device
        tri = ops.EIG_SOLVER == "tridiag"
        # all student eigenvectors are only needed by the Jacobi-mode backward; the tridiagonal route's backward
        # works from the leading k and shifted solves (``_distance_backward``)
        stud_jacobi = not tri
        st = dict(E=E, L=L, tri=tri, stud_jacobi=stud_jacobi, student_stream=student_stream)

        cur = torch.cuda.current_stream()
        projected = None
        chain_t0 = None
        if ops.HOST_TRACE is not None:             # diagnostics (tools/host_timeline.py): GPU time of the teacher chain
            chain_t0 = torch.cuda.Event(enable_timing=True)
            chain_t0.record()
        ops.gpu_mark("chain_begin")
        if student_stream is not None:
            student_stream.wait_stream(cur)
            # two chains: give the teacher stream its first (large) launch before the host queues the
            # student chain, so both start together
            projected = self._teacher_projections(teachers)
            ops.gpu_mark("teacher_projected")

        def student_chain(gate=None):
            """centred Grams -> eigen-solve of the E student layers; ``gate``: event the chain waits for first"""
            with torch.cuda.stream(student_stream if student_stream is not None else cur):
                if isinstance(gate, torch.cuda.Event):
                    torch.cuda.current_stream().wait_event(gate)
                elif gate is not None:
                    ops.stream_wait_event(torch.cuda.current_stream(), gate)
                # proj_s is orthogonal: principal angles are unchanged if the teacher bases are rotated by
                # proj_s^T instead of the student tokens by proj_s (layer_selector.py:88 folded into :99)
                xs = ops._check_common_layout([ops.as_supported(x) for x in students], "student token tensors")
                s_stack, means = ops.centered_grams(xs)
                st["means"] = means
                if stud_jacobi:
                    st["s_stack"], st["s_colnorm"] = s_stack, ops.jacobi_onesided(s_stack, d_s)
                else:
                    s_ts = ops.tridiagonalise(s_stack)
                    if student_stream is not None and gate_student:
                        # status word of this factorisation: read by the host one step later (it never waits
                        # for the student chain)
                        st["student_status"] = self._queue_readback([s_ts.err], "student")
                    else:
                        # same stream as the teacher chain that follows: complete when the rank event is
                        st["student_status_now"] = self._queue_readback([s_ts.err], "student")
                    ops.tridiag_spectrum(s_ts)
                    st["s_ts"] = s_ts

        # Two chains: the student chain is not on the path the host waits for, and its Gram launch (the largest MFMA
        # launch of the step) next to the teacher's Grams / multi-workgroup tridiagonalisation stage costs the teacher
        # chain ~1 ms (rocprofv3: teacher Grams 0.46 ms instead of 0.06, shared stage 0.72 instead of 0.48).  So it is
        # queued BEHIND the teacher chain and gated on an event recorded after that stage -- from there on the teacher
        # factorisation sits in one CU per matrix and no longer cares.
        gated = student_stream is not None and tri and self.gate_student_chain and gate_student
        if (tri and self.rank1_mp and self.merge_factorisations and student_stream is None
                and self._teacher_space_route(teachers)):
            # ONE factorisation launch over the L centred teacher Grams and the E student Grams (multi-layer ViT teachers:
            # a launch over E = 4 matrices of order 768 takes 2.8 ms, over L = 24 5.0 -- both are bound by dependent
            # steps, not by the number of matrices), one bisection launch over all spectra.  The MP ranks of the
            # uncentred teacher Grams come from the centred factorisations (rank-one route, below).
            xs = ops._check_common_layout([ops.as_supported(x) for x in students], "student token tensors")
            g_all = torch.empty((L + E, d_s, d_s), device=dev, dtype=torch.float32)
            _, st["means"] = ops.centered_grams(xs, out=g_all[L:])
            _, _, M, _ = self._teacher_grams_in_teacher_space(teachers, centred_out=g_all[:L])
            ops.gpu_mark("teacher_grams")
            st["o_c"] = 0
            pin = self._pinned_ints("teacher", L + 8)
            ts = ops.tridiagonalise(g_all)
            t_ts = ops.tridiag_slice(ts, 0, L)
            w = ops.tridiag_apply_q(t_ts, self._teacher_zbar.view(L, 1, d_s).contiguous(), transpose=True)
            st["t_ts"], st["s_ts"] = t_ts, ops.tridiag_slice(ts, L, E)
            st["ranks_dev"] = ops.tridiag_mp_rank_rank1(t_ts, w.view(L, d_s), M, d_s, d_s - 1, pin)
            ready = torch.cuda.Event()
            ready.record()
            st["rank_ready"] = (pin, ready)
            ops.tridiag_spectrum(ts)
            return st
        if not gated:
            student_chain()

        # ---- teacher side: projection, Grams, eigen-solve, MP ranks ----
        self._teacher_zbar = None
        g_u, g_c, M, t_stack = self._teacher_grams(teachers, projected)
        ops.gpu_mark("teacher_grams")
        same = t_stack is not None
        o_c = L if same else 0
        if not same:
            t_stack = g_c
        st["o_c"] = o_c
        if tri and same and self.rank1_mp and self._teacher_zbar is not None and student_stream is None:
            # ONE factorisation per teacher layer (the centred Gram's); the MP rank of the uncentred Gram is counted in
            # its basis as a rank-one modification: z^T z = Q (T + M w w^T) Q^T, w = Q^T zbar (basd_tridiag_mp_rank_rank1)
            st["o_c"] = 0
            pin = self._pinned_ints("teacher", L + 8)
            ts = ops.tridiagonalise(g_c.contiguous())
            w = ops.tridiag_apply_q(ts, self._teacher_zbar.view(L, 1, d_s).contiguous(), transpose=True)
            st["t_ts"] = ts
            st["ranks_dev"] = ops.tridiag_mp_rank_rank1(ts, w.view(L, d_s), M, d_s, d_s - 1, pin)
            ready = torch.cuda.Event()
            ready.record()
            st["rank_ready"] = (pin, ready)
            ops.tridiag_spectrum(ts)
            if gated:
                student_chain()
            return st
        if tri and same:
            # The host waits for the ranks and for nothing else: they come straight from the tridiagonals of
            # the uncentred Grams (median by multisection + one Sturm count), are copied to pinned memory at
            # once, and only then are the centred spectra (needed by the eigenvector stage alone) queued.
            # ranks straight out of the factorisation's last kernel                          (:16-19, :74)
            pin = self._pinned_ints("teacher", L + 8)
            # Where the gate sits.  Student layers that feed this step's loss (several teacher layers): behind the
            # teacher's multi-workgroup stage, as early as the spinning members allow.  One teacher layer (the student
            # chain feeds nothing this step): behind the WHOLE teacher factorisation -- released at the hand-over, its
            # Gram launch floods the chip in the very moment the teacher's one-workgroup tail kernel looks for a CU
            # with 8 free wave slots and 2 x 192 VGPRs per SIMD (100 MHz stamps inside a step, tools/
            # tail_stamps_in_step.py: 474 us between the end of the shared stage and the tail kernel's first instruction).
            gate, gate_after_ranks = None, gated and defer_student
            if gated and not gate_after_ranks:
                if getattr(self, "_gate_event", None) is None:
                    self._gate_event = ops.new_event()
                gate = self._gate_event
            ts = ops.tridiagonalise(t_stack, mp_rank=(M, d_s, d_s - 1, L, pin, gate))
            st["t_ts"] = ts
            st["ranks_dev"] = ts.ranks
            ops.gpu_mark("ranks_ready")
            ready = torch.cuda.Event(enable_timing=chain_t0 is not None)
            ready.record()
            st["rank_ready"] = (pin, ready)
            if chain_t0 is not None:
                ops.CHAIN_EVENTS.append((chain_t0, ready))
            ops.tridiag_spectrum(ts, first=o_c, count=L)
            if gate_after_ranks:
                gate = self._gate_event = ready
            if gated and defer_student:
                st["queue_student"] = lambda: student_chain(gate)
            elif gated:
                student_chain(gate)
            return st
        if gated:
            student_chain()
        if tri:
            ts = ops.tridiag_eigenvalues(t_stack)
            st["t_ts"] = ts
            vals_u = ops.tridiag_eigenvalues(g_u).vals
        else:
            colnorm = ops.jacobi_onesided(t_stack, d_s)
            st["t_stack"], st["t_colnorm"] = t_stack, colnorm
            if same:
                vals_u, _ = ops.sort_extract(t_stack[:L], colnorm[:L], 0)
            else:
                vals_u, _, _, _ = ops.sym_eig(g_u)
        st["ranks_dev"] = ops.mp_rank_device(vals_u, M, d_s, cap=d_s - 1)    # :74
        return st

    def _pinned_ints(self, slot: str, count: int) -> torch.Tensor:
        """Pinned int32 host buffer; a ring of four per slot, so that a value may still be read two steps
        later (deferred read-back) while the following steps' values are being produced."""
        bufs = self.__dict__.setdefault("_pinned", {})
        flip = self.__dict__.setdefault("_pinned_flip", {})
        idx = flip.get(slot, 0)
        flip[slot] = (idx + 1) % 4
        key = (slot, idx)
        if key not in bufs or bufs[key].numel() != count:
            for i in range(4):       # the whole ring at once: a pinned allocation synchronises the device
                bufs[(slot, i)] = torch.empty((count,), dtype=torch.int32, pin_memory=True)
        return bufs[key]

    def _queue_readback(self, tensors: list[torch.Tensor], slot: str):
        """Async copy of small int32 device tensors into a pinned buffer on the current stream + an event."""
        pin, off = self._pinned_ints(slot, sum(t.numel() for t in tensors)), 0
        for t in tensors:
            pin[off:off + t.numel()].copy_(t.reshape(-1), non_blocking=True)
            off += t.numel()
        ev = torch.cuda.Event()
        ev.record()
        return pin, ev

    def _read_ranks(self, st: dict, keys: list[int]) -> list[int]:
        """The step's one host read-back: the MP ranks (and the status words of the eigen-solves).  Blocks on
        the teacher chain's rank kernel only.  Refreshes ``subspace_ranks``; raises like the reference on rank 0."""
        ranks_dev, L = st["ranks_dev"], st["L"]
        # the step's one read-back: the ranks, and behind them the status words of the eigen-solves
        if "rank_ready" in st:
            pin, ev = st["rank_ready"]
            ev.synchronize()                       # the teacher chain up to the rank kernel; nothing else
            host = pin.tolist()
            status = host[L:L + 1]
            pending = self.__dict__.pop("_pending_status", None)
            if pending is not None:                # student chain of the PREVIOUS step: long complete
                pending[1].synchronize()
                if int(pending[0][0]):
                    # its eigenvectors only fed that step's principal angles, which nothing observed (one teacher
                    # layer): nothing to redo, but do not let it happen again
                    _single_member_mode("the student-side factorisation of the previous step timed out")
            if "student_status" in st:
                self._pending_status = st["student_status"]
            if "student_status_now" in st:         # multi-layer teachers: the student eigenvectors feed this loss
                st["student_status_now"][1].synchronize()
                status.append(int(st["student_status_now"][0][0]))
        else:
            errs = [st[k].err for k in ("t_ts", "s_ts") if k in st and st[k].err is not None]
            host = torch.cat([ranks_dev.to(torch.int32), *errs]).tolist() if errs else ranks_dev.tolist()
            status = host[L::8]
        ops.trace("ranks_read")
        ranks = [int(r) for r in host[:L]]
        if any(status):
            raise TridiagGiveUp("basd_tridiag: workgroups sharing a matrix timed out waiting for each other "
                                f"(device oversubscribed?); eigen-solve results are invalid [{host[L:]}]")
        for k, r in zip(keys, ranks):
            self._subspace_ranks[k] = r
        if min(ranks) == 0:
            # reference: 0/0 distance -> NaN weights -> NaN tokens -> torch.linalg.svd raises
            raise torch.linalg.LinAlgError(
                "linalg.svd: The algorithm failed to converge because the input matrix contained "
                "non-finite values (a teacher layer has Marchenko-Pastur rank 0).")
        return ranks

    @torch.no_grad()
    def _angles_from_spectra(self, st: dict, keys: list[int], want_grad: bool = False,
                             ranks_host: list[int] | None = None):
        """The step's single D2H read-back (ranks), then principal angles -> d_grass_sq (E, L)
        (layer_selector.py:95-105).  Refreshes ``subspace_ranks``.  Returns (d, saved state for
        ``_distance_backward`` or None)."""
        d_s = self.student_dim
        ranks_dev, o_c, E, L = (st[k] for k in ("ranks_dev", "o_c", "E", "L"))
        dev = ranks_dev.device
        ranks = ranks_host if ranks_host is not None else self._read_ranks(st, keys)
        kmax = max(ranks)
        if (not want_grad and st["tri"] and not st["stud_jacobi"] and "t_ts" in st and "s_ts" in st
                and st["student_stream"] is None):
            # the common case: one library call queues the whole tail (host time is on the step's critical path)
            return ops.selector_tail(st["t_ts"], o_c, L, st["s_ts"], kmax, ranks_dev, self._proj_s_transposed()), None
        n_stud = d_s if want_grad else kmax
        # student eigenvectors (on the student chain's stream when there is one)
        ss = st["student_stream"]
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(ss if ss is not None else cur):
            if st["stud_jacobi"]:
                lam_s, v_all = ops.sort_extract(st["s_stack"], st["s_colnorm"], n_stud)
            else:
                lam_s = st["s_ts"].vals
                v_all = ops.tridiag_eigenvectors(st["s_ts"], kmax)
        # teacher eigenvectors
        if st["tri"]:
            ts = st["t_ts"]
            vals_c = ts.vals[o_c:o_c + L]
            u_t = ops.tridiag_eigenvectors(ts, kmax, first=o_c, count=L)                 # (L, kmax, d_s)
        else:
            vals_c, u_t = ops.sort_extract(st["t_stack"][o_c:o_c + L], st["t_colnorm"][o_c:o_c + L], kmax)
        if ss is not None:
            cur.wait_stream(ss)
            for t in (lam_s, v_all):          # allocated on the student stream, consumed on this one
                t.record_stream(cur)
        sw = ops.sqrt_clamp(vals_c[:, :kmax])                      # singular values S[:k]   (:36-37)
        v_s = v_all[:, :kmax]                                      # (E, kmax, d_s) rows = Vt_s[:kmax]
        u_rot = ops.gemm_nt(u_t.view(L * kmax, d_s), self._proj_s_transposed()).view(L, kmax, d_s)   # rows: (proj_s^T u)^T
        cos = torch.empty((E, L, kmax, kmax), device=dev, dtype=torch.float32)
        if L == 1 and v_s.is_contiguous():
            # one launch over the extraction layers: Vt_s[e][:k] @ U_t   (:99)
            ops.gemm_nt(v_s[0], u_rot[0], out=cos, batch=E, a_batch_stride=kmax * d_s, b_batch_stride=0,
                        rows=kmax, n_cols=kmax)
        else:
            for e in range(E):
                a = v_s[e] if v_s[e].is_contiguous() else v_s[e].contiguous()
                ops.gemm_nt(a, u_rot[0], out=cos[e], batch=L, a_batch_stride=0, b_batch_stride=kmax * d_s,
                            rows=kmax, n_cols=kmax)                # Vt_s[:k] @ U_t   (:99)
        k_arr = ranks_dev if E == 1 else (ranks_dev.expand(E) if L == 1 else ranks_dev.repeat(E)).contiguous()
        sw_index = ops._device_consts(tuple(range(L)) * E, torch.int32, dev)
        if not want_grad:
            sigma = ops.jacobi_onesided(cos.view(E * L, kmax, kmax), kmax, n_arr=k_arr)
            return ops.grassmann_distance(sigma, k_arr, sw, sw_index).view(E, L), None   # (:100-105)
        # with a backward: [cos ; I] stacks so that the solver also delivers the right singular vectors
        ang = ops.angle_stack(cos.view(E * L, kmax, kmax), k_arr)
        sigma = ops.jacobi_onesided(ang, kmax)
        d = ops.grassmann_distance(sigma, k_arr, sw, sw_index).view(E, L)
        saved = dict(ang=ang, sigma=sigma, k_arr=k_arr, sw=sw, sw_index=sw_index, u_rot=u_rot, v_all=v_all,
                     lam=lam_s, means=st["means"], kmax=kmax, E=E, L=L, s_ts=st.get("s_ts"))
        return d, saved

    @torch.no_grad()
    def _distance_backward(self, sv: dict, students: list[torch.Tensor], gd: torch.Tensor) -> list[torch.Tensor]:
        """d (sum gd * d_grass_sq) / d student tokens: autograd of layer_selector.py:86-105.
        svdvals -> (theta, weighted distance) -> Vt_s -> centred student Gram -> tokens."""
        d_s = self.student_dim
        E, L, kmax = sv["E"], sv["L"], sv["kmax"]
        gwt = ops.grassmann_distance_bwd(sv["ang"], sv["sigma"], sv["k_arr"], sv["sw"], sv["sw_index"],
                                         gd.reshape(E * L).float())                    # (E*L, kmax, kmax)
        u_flat = sv["u_rot"].view(L * kmax, d_s)
        # G_V^T (kmax x d_s) per student layer = sum_l gW_l U_l'^T : contraction over (l, j)
        gvt = torch.stack([ops.gemm_tn(gwt[e * L:(e + 1) * L].view(L * kmax, kmax), u_flat) for e in range(E)])
        if sv.get("s_ts") is not None:
            qs = self._eigvec_adjoint_leading(sv, gvt)
        else:
            m_all = torch.empty((E, d_s, kmax), device=gd.device, dtype=torch.float32)
            for e in range(E):
                m_all[e] = ops.gemm_nt(sv["v_all"][e], gvt[e])                          # M = V^T G_V  (d_s x kmax)
            k2 = ops.eigvec_k2(m_all, sv["lam"])                                        # (E, d_s, d_s), symmetric
            qs = []
            for e in range(E):
                v = sv["v_all"][e]
                qs.append(ops.gemm_tn(v, ops.gemm_tn(k2[e], v)))                        # Q = V K2 V^T (symmetric)
        grads = []
        for e, x in enumerate(students):
            q = qs[e]
            mu = sv["means"][e].view(1, d_s)
            bias = ops.gemm_nt(mu, q).view(d_s)                                         # mu Q
            x = ops.as_supported(x)
            dx = ops.gemm_nt(x, q, bias=bias)                                           # (X - 1 mu^T) Q
            grads.append(dx.view(x.shape[0], x.shape[1], d_s))
        return grads

    def _eigvec_adjoint_leading(self, sv: dict, gvt: torch.Tensor) -> list[torch.Tensor]:
        """Q = V K2 V^T (the adjoint of `leading k eigenvectors of G`, see ``eigvec_k2``) WITHOUT the trailing
        eigenvectors.  K2 couples a leading vector j with (a) the other leading vectors -- a k x k block from V_k
        alone -- and (b) every trailing vector i through v_i v_i^T g_j / (lam_j - lam_i), which summed over i is
        (lam_j I - G)^{-1} applied to the part of g_j orthogonal to the leading space: k shifted solves with the
        tridiagonal form G = Q_H T Q_H^T that the forward already has.
            Q = V_k^T (K2_kk V_k + Y) + Y^T V_k,    rows of Y:  y_j = P_perp (lam_j I - G)^{-1} P_perp g_j."""
        ts, E, kmax = sv["s_ts"], sv["E"], sv["kmax"]
        v_k = sv["v_all"][:, :kmax].contiguous()                                        # (E, kmax, d_s) rows v_j
        lam_k = sv["lam"][:, :kmax].contiguous()
        qs = []
        m_kk = torch.stack([ops.gemm_nt(v_k[e], gvt[e]) for e in range(E)])            # M_kk[i][j] = v_i . g_j
        k2 = ops.eigvec_k2(m_kk, lam_k)                                                 # (E, kmax, kmax), symmetric
        # R^T rows r_j = g_j - sum_i M_kk[i][j] v_i     (P_perp g_j)
        r = torch.stack([gvt[e] - ops.gemm_tn(m_kk[e], v_k[e]) for e in range(E)])
        z = ops.tridiag_shifted_solve(ts, lam_k, ops.tridiag_apply_q(ts, r.contiguous(), transpose=True))
        y = -ops.tridiag_apply_q(ts, z, transpose=False)                                # (lam_j I - G)^{-1} r_j
        for e in range(E):
            ye = y[e] - ops.gemm_tn(ops.gemm_nt(v_k[e], y[e]), v_k[e])                  # P_perp again: y - (y V_k^T) V_k
            w_e = ops.gemm_tn(k2[e], v_k[e]) + ye                                       # K2_kk V_k + Y   (kmax x d_s)
            qs.append(ops.gemm_tn(v_k[e], w_e) + ops.gemm_tn(ye, v_k[e]))               # V_k^T W + Y^T V_k
        return qs

    def _distances(self, students: list[torch.Tensor], keys: list[int], teachers: list[torch.Tensor]) -> torch.Tensor:
        """d_grass_sq (E, L) (layer_selector.py:86-105), differentiable w.r.t. the student tokens."""
        return _GrassmannDistance.apply(self, keys, teachers, *students)

    def mixing_weights(self, students: list[torch.Tensor], keys: list[int],
                       teachers: list[torch.Tensor]) -> torch.Tensor:
        """softmax(-d / tau) per extraction layer -> (E, L)   (layer_selector.py:107-108)."""
        d = self._distances(students, keys, teachers)
        self._last_d_grass_sq = d.detach()                  # layer_selector.py:105, kept for ``last_components``
        tau = self.temperatures.float()
        w = torch.softmax(-d / tau.unsqueeze(1), dim=1)
        if teachers[0].dtype != torch.float32:
            # layer_selector.py:110: the weights are cast to the token dtype before they mix the layers; autograd's adjoint of
            # a cast is the identity, so the rounding is applied to the VALUE only (the mixing itself stays in fp32 here)
            w = w + (w.to(teachers[0].dtype).float() - w).detach()
        return w

    @torch.no_grad()
    def _mix_for_student_layer(
        self,
        i: int,
        s_tokens: torch.Tensor,
        teacher_indices: list[int],
        stacked_tokens: torch.Tensor,
        stacked_attns: torch.Tensor,
        subspaces: dict[int, torch.Tensor],
        spectral_weights: dict[int, torch.Tensor],
    ) -> tuple[torch.Tensor, torch.Tensor]:
        """API parity with reference layer_selector.py:76-114 (same arguments, same return): teacher layers mixed
        for student layer ``i`` from PRE-COMPUTED teacher subspaces / spectral weights (``_grassmann_subspace``
        outputs in the projected space) and ``self.subspace_ranks``.  Values only: the differentiable path of
        the training loss is ``mixing_weights`` (which ``BASDLoss`` uses and which never materialises the mix)."""
        d_s = self.student_dim
        x = ops.as_supported(s_tokens)
        kmax = max(int(self.subspace_ranks[t]) for t in teacher_indices)
        if kmax < 1:
            raise torch.linalg.LinAlgError(
                "linalg.svd: The algorithm failed to converge because the input matrix contained "
                "non-finite values (a teacher layer has Marchenko-Pastur rank 0).")
        gram, _ = ops.centered_grams([x])                                     # :88-91 (proj_s folded below)
        ts = ops.tridiag_eigenvalues(gram)
        v_s = ops.tridiag_eigenvectors(ts, kmax)[0]                           # (kmax, d_s): Vt of the raw tokens
        proj_s_t = self._proj_s_transposed()
        L = len(teacher_indices)
        cos = torch.zeros((L, kmax, kmax), device=x.device, dtype=torch.float32)
        sw = torch.zeros((L, kmax), device=x.device, dtype=torch.float32)
        for j, t_idx in enumerate(teacher_indices):
            k = int(self.subspace_ranks[t_idx])
            u_t = subspaces[t_idx].float()[:, :k]                             # (d_s, k) in the projected space
            u_rot = ops.gemm_nt(u_t.t().contiguous(), proj_s_t)               # rows: (proj_s^T u)^T   (:99)
            cos[j, :k, :k] = ops.gemm_nt(v_s[:k].contiguous(), u_rot)         # Vt_s[:k] proj_s^T U_t
            sw[j, :k] = spectral_weights[t_idx].float()[:k]
        k_arr = torch.tensor([int(self.subspace_ranks[t]) for t in teacher_indices], dtype=torch.int32).to(x.device)
        sigma = ops.jacobi_onesided(cos, kmax, n_arr=k_arr)
        sw_index = torch.arange(L, device=x.device, dtype=torch.int32)
        d = ops.grassmann_distance(sigma, k_arr, sw, sw_index)                # :100-105
        weights = torch.softmax(-d / self.temperatures[i].float(), dim=0).to(stacked_tokens.dtype)    # :107-110
        mixed = (weights.view(-1, 1, 1, 1) * stacked_tokens).sum(dim=0)
        mixed_attn = (weights.view(-1, 1, 1, 1, 1) * stacked_attns).sum(dim=0)
        return mixed, mixed_attn

    def forward(
        self,
        student_tokens_per_layer: dict[int, torch.Tensor],
        all_teacher_tokens: dict[int, torch.Tensor],
        all_teacher_attns: dict[int, torch.Tensor],
        extraction_indices: list[int],
    ) -> tuple[dict[int, torch.Tensor], dict[int, torch.Tensor]]:
        """API-parity entry point: returns the materialised mixed teacher tokens / attention maps.
        (``BASDLoss`` does not call this -- it feeds the mixing weights straight to the fused loss.)"""
        keys = sorted(all_teacher_tokens.keys())
        teachers = ops._check_common_layout([ops.as_supported(all_teacher_tokens[k]) for k in keys],
                                            "teacher token tensors")
        students = [student_tokens_per_layer[s] for s in extraction_indices]
        mix = self.mixing_weights(students, keys, teachers)
        tok_stack = torch.stack([all_teacher_tokens[k] for k in keys])
        att_stack = torch.stack([all_teacher_attns[k] for k in keys])
        mixed_t, mixed_a = {}, {}
        for i, s_layer in enumerate(extraction_indices):
            w = mix[i].to(tok_stack.dtype)
            mixed_t[s_layer] = (w.view(-1, 1, 1, 1) * tok_stack).sum(dim=0)
            mixed_a[s_layer] = (w.view(-1, 1, 1, 1, 1) * att_stack).sum(dim=0)
        return mixed_t, mixed_a


# --------------------------------------------------------------------------- #
# reference src/losses/combined.py:17-85
# --------------------------------------------------------------------------- #
class BASDLoss(nn.Module):
    def __init__(
        self,
        base_criterion: nn.Module,
        student_dim: int,
        teacher_dim: int,
        student_depth: int,
        num_student_tokens: int,
        *,
        config,
        teacher_has_cls_token: bool,
    ):
        super().__init__()
        self.base_criterion = base_criterion
        self.teacher_has_cls_token = teacher_has_cls_token
        self.num_student_tokens = num_student_tokens

        if config.num_extraction_points == 1:
            self.token_layers = [student_depth - 1]
        else:
            self.token_layers = [
                round(i * (student_depth - 1) / (config.num_extraction_points - 1))
                for i in range(config.num_extraction_points)
            ]
        self.layer_selector = GrassmannianLayerSelector(
            num_extraction_points=len(self.token_layers),
            student_dim=student_dim,
            teacher_dim=teacher_dim,
        )
        self.last_components: dict[str, torch.Tensor] = {}
        self._side_streams: dict = {}
        # The reference reads the ranks (and raises on rank 0) inside forward; so do we -- but with one teacher layer the
        # ranks do not feed the loss, and waiting for them is waiting for the whole factorisation (~1.5 ms).
        #   "auto" (default): a small kernel behind the teacher Grams PROVES "no rank is 0" where the spectrum allows it
        #       (BasdSelectorChain.cert_mirror: largest eigenvalue >= ||G||_F^2 / tr G against median <= tr G / (n/2), with a
        #       factor 4 to spare); the host waits for that word (~0.65 ms into the step) and, when it is set, leaves the
        #       read-back of the ranks to the next forward / the first reader of ``subspace_ranks`` -- forward cannot have
        #       raised.  Not proven (a flat spectrum, NaNs, shapes the chain does not take): the read-back happens inside
        #       forward as in the reference.
        #   "sync": always inside forward.   "deferred" (``sync_ranks = False``): never -- a rank-0 layer then raises one
        #       step late.
        self.rank_readback = os.environ.get("BASD_RANK_READBACK", "auto")
        if self.rank_readback not in ("auto", "sync", "deferred"):
            raise ValueError(f"BASD_RANK_READBACK={self.rank_readback!r}: expected auto, sync or deferred")
        self.readback_deferred_steps = 0       # steps whose read-back was left to the next call (diagnostics)
        # single-teacher steps: the selector as ONE library call into a persistent workspace (basd_selector_chain);
        # BASD_SELECTOR_CHAIN=0 keeps the kernel-by-kernel layout.  chain_mode: see BasdSelectorChain.mode
        self.use_chain = os.environ.get("BASD_SELECTOR_CHAIN", "1") != "0"
        # (None = by situation: 0 -- ONE factorisation launch over teacher and student matrices, student Grams beside the
        # teacher's projection -- while the host does not wait for the ranks (the step is then bound by the GPU's total work:
        # 1.92-1.95 ms at cfg-2 against 2.02-2.06 in mode 3); 3 -- teacher matrices first, student side held back -- in steps
        # that follow one whose ranks had to be waited for (2.38 against 2.46))
        self._chain_mode_forced = int(os.environ["BASD_CHAIN_MODE"]) if "BASD_CHAIN_MODE" in os.environ else None
        self._last_step_waited = False
        # which of the step's two chains the host queues first: the Procrustes kernels of the caller's stream or the selector
        self.procrustes_first = os.environ.get("BASD_PROCRUSTES_FIRST", "0") == "1"
        self.student_low_priority = os.environ.get("BASD_STUDENT_LOW_PRIORITY", "1") == "1"
        self.tail_low_priority = os.environ.get("BASD_TAIL_LOW_PRIORITY", "0") != "0"
        self._chain_plans: dict = {}

    def _selector_stream(self, device, index: int = 0) -> "torch.cuda.Stream":
        key = (str(device), index)
        if key not in self._side_streams:
            # the teacher chain (0) gates the step's one host read-back (the ranks): high priority, so that its
            # kernels are dispatched ahead of the student chain's (1) and the main stream's when they compete
            mode = os.environ.get("BASD_CHAIN_PRIORITY", "2")
            prio = -1 if (mode == "1" and index % 3 < 2) or (mode == "2" and index % 3 == 0) else 0
            self._side_streams[key] = torch.cuda.Stream(device=device, priority=prio)
        return self._side_streams[key]

    @property
    def chain_mode(self) -> int:
        if self._chain_mode_forced is not None:
            return self._chain_mode_forced
        return 3 if (self.rank_readback == "sync" or self._last_step_waited) else 0

    @chain_mode.setter
    def chain_mode(self, value: int | None) -> None:
        self._chain_mode_forced = None if value is None else int(value)

    @property
    def sync_ranks(self) -> bool:
        """True unless the rank read-back is ALWAYS left to the next call (``rank_readback == "deferred"``)."""
        return self.rank_readback != "deferred"

    @sync_ranks.setter
    def sync_ranks(self, value: bool) -> None:
        self.rank_readback = "sync" if value else "deferred"

    def _forward_single_teacher_legacy(self, student_output, targets, students, keys, teachers, attns, comp):
        """One teacher layer, shapes ``basd_selector_chain`` does not take (fewer teacher tokens than student features:
        the token-side Gram of layer_selector.py:14-15; the Jacobi eigen-solver): the selector queued kernel by kernel
        over four streams, its tail one step late (round-2 layout)."""
        sel = self.layer_selector
        # One teacher layer (CNN teachers): softmax over a single distance is 1 whatever the distance, so
        # the Procrustes loss does not depend on the selector.  The selector's eigen-solves are latency-
        # bound chains of small launches; run them on a side stream underneath the Procrustes kernels.
        main = torch.cuda.current_stream()
        # two sets of chain streams, used by alternate steps: with the deferred read-back the chains of
        # consecutive steps overlap instead of queueing behind each other
        lane = 0
        if not self.sync_ranks:
            lane = self._chain_lane = (getattr(self, "_chain_lane", 1) + 1) % 2
        side = self._selector_stream(main.device, 3 * lane)
        side2 = self._selector_stream(main.device, 3 * lane + 1)
        side.wait_stream(main)
        # the borrowed inputs are read on the side streams after this call has returned (the tail of the
        # selector is not joined into the main stream: nothing downstream of it feeds the loss)
        for t in (*students, *teachers):
            t.record_stream(side)
            t.record_stream(side2)
        with torch.cuda.stream(side):
            spectra = sel._spectra_async(students, teachers, student_stream=side2, defer_student=True)
        ops.trace("chains_queued")
        ce_loss = _base_loss(self.base_criterion, student_output, targets)     # behind the chains' first launches
        # softmax over ONE logit: the mixing weights are exactly 1 and d loss / d temperature exactly 0
        mix = ops._device_consts((1.0,) * len(students), torch.float32, main.device).view(-1, 1)
        total, geo_layers = _SingleTeacherTotal.apply(ce_loss, bool(self.teacher_has_cls_token),
                                                      sel.log_temperatures, teachers, attns, *students)
        ops.trace("procrustes_queued")
        # the student chain waits for the teacher's first tridiagonalisation stage on the GPU anyway: queue it now,
        # behind the Procrustes kernels, instead of in front of them
        queue_student = spectra.pop("queue_student", None)
        if queue_student is not None:
            with torch.cuda.stream(side):
                queue_student()
        # what the selector tail of THIS step has to wait for (it may be queued after later steps' chains)
        chain_done = []
        for st_ in (side, side2):
            ev = torch.cuda.Event()
            ev.record(st_)
            chain_done.append(ev)
        tail = self._selector_stream(main.device, 3 * lane + 2)

        def read_ranks_once():
            # The host reads the ranks here (and raises on rank 0 like the reference).
            if "rank_ready" in spectra:
                return sel._read_ranks(spectra, keys)               # waits on the rank kernel's event
            with torch.cuda.stream(side):                           # plain read-back behind both chains
                side.wait_stream(side2)
                return sel._read_ranks(spectra, keys)

        def read_ranks():
            nonlocal spectra, chain_done
            try:
                return read_ranks_once()
            except TridiagGiveUp as exc:
                # degrade, do not die: the selector of THIS step once more, one workgroup per matrix
                _single_member_mode(str(exc))
                torch.cuda.synchronize(main.device)
                with torch.cuda.stream(side):
                    spectra = sel._spectra_async(students, teachers, student_stream=side2)
                chain_done = []
                for st_ in (side, side2):
                    ev = torch.cuda.Event()
                    ev.record(st_)
                    chain_done.append(ev)
                return read_ranks_once()

        def queue_tail(ranks, gate_tail=False):
            # The rest of the selector (eigenvectors, principal angles) goes to a third stream: the next step's
            # eigen-solve chains do not queue behind it, and nothing of it feeds this loss when there is one
            # teacher layer -- so it is not even queued in this call (see below).
            for ev in chain_done:
                tail.wait_event(ev)
            # queued one step later (see below): then also behind that step's multi-workgroup tridiagonalisation
            # stage, like its student chain -- the members of that stage must not queue for CUs behind these kernels
            if gate_tail and isinstance(getattr(sel, "_gate_event", None), torch.cuda.Event):
                tail.wait_event(sel._gate_event)
            elif gate_tail and getattr(sel, "_gate_event", None) is not None:
                ops.stream_wait_event(tail, sel._gate_event)
            ops.trace("tail_waits")
            # only what the tail reads needs marking (every marked block costs an event when it is freed)
            _record_stream([spectra.get("t_ts"), spectra.get("s_ts"), spectra["ranks_dev"],
                            spectra.get("t_stack"), spectra.get("t_colnorm"), spectra.get("s_stack"),
                            spectra.get("s_colnorm")], tail)
            ops.trace("tail_marked")
            spectra["student_stream"] = None
            with torch.cuda.stream(tail):
                # layer_selector.py:99-105; nothing of THIS loss reads it (softmax over one distance), so it is
                # only published: ``last_components["d_grass_sq"]`` of the step, valid after ``finish_pending()``
                # (written on the tail stream: synchronise the device or that stream before reading)
                comp["d_grass_sq"] = sel._angles_from_spectra(spectra, keys, ranks_host=ranks)[0]
            ops.trace("tail_queued")

        def selector_tail():
            queue_tail(read_ranks())
        return total, ce_loss, geo_layers, mix, (read_ranks, queue_tail, selector_tail)

    def _chain_plan(self, students, teachers, main) -> "SelectorChainPlan":
        key = SelectorChainPlan.key(students, teachers, self.chain_mode)
        plan = self._chain_plans.get(key)
        if plan is None:
            if len(self._chain_plans) >= 4:         # shapes changed for good (another resolution): drop the old workspaces
                torch.cuda.synchronize(main.device)
                self._chain_plans.clear()
            streams = [self._selector_stream(main.device, i) for i in range(3)]
            if self.student_low_priority:
                # the student side is throughput work nothing waits for: the device's LOWEST stream priority (torch only
                # offers high / default), so that the dispatcher serves the two chains the step waits for first
                key_s = (str(main.device), "student-low")
                if key_s not in self._side_streams:
                    import ctypes
                    h = ctypes.c_void_p()
                    with torch.cuda.device(main.device):
                        ops._lib.call("basd_stream_create_priority", ctypes.byref(h), 1)
                    self._side_streams[key_s] = torch.cuda.ExternalStream(h.value, device=main.device)
                streams[1] = self._side_streams[key_s]
            if self.tail_low_priority:
                # ... and so is the selector's tail (its d_grass_sq is only published)
                key_t = (str(main.device), "tail-low")
                if key_t not in self._side_streams:
                    import ctypes
                    h = ctypes.c_void_p()
                    with torch.cuda.device(main.device):
                        ops._lib.call("basd_stream_create_priority", ctypes.byref(h), 1)
                    self._side_streams[key_t] = torch.cuda.ExternalStream(h.value, device=main.device)
                streams[2] = self._side_streams[key_t]
            streams = tuple(streams)
            plan = self._chain_plans[key] = SelectorChainPlan(students, teachers, self.chain_mode, streams,
                                                              fact_stream=self._selector_stream(main.device, 3))
        return plan

    def _forward_single_teacher(self, student_output, targets, students, keys, teachers, attns, comp):
        """One teacher layer (every CNN teacher; softmax over a single distance is 1, so the loss does not wait for the
        selector): the whole selector is ONE library call over three side streams (``basd_selector_chain``), the base
        loss and the Procrustes loss + unit gradients follow on the caller's stream, then the host waits for the
        teacher ranks and for nothing else (the reference reads them, and raises on rank 0, inside forward)."""
        sel = self.layer_selector
        previous, sel._pending_tail = sel._pending_tail, None      # a read-back the previous step deferred (that mode)
        main = torch.cuda.current_stream()
        xs = ops._check_common_layout([ops.as_supported(x) for x in students], "student token tensors")
        plan = self._chain_plan(xs, teachers, main)
        # the borrowed inputs are read on the side streams after this call has returned
        for t in (*xs, *teachers):
            t.record_stream(plan.chain_stream)
            t.record_stream(plan.student_stream)
        proj_t = sel.proj_t if sel.proj_t.dtype == torch.float32 and sel.proj_t.is_contiguous() \
            else sel.proj_t.float().contiguous()
        ops.gpu_mark("chain_begin")
        slot = None
        if self.procrustes_first:
            plan.fork(main.cuda_stream)        # the chain starts behind THIS point, not behind the kernels queued next
        else:
            slot = plan.queue(xs, teachers, proj_t, sel._proj_s_transposed(), main.cuda_stream)
            ops.trace("chains_queued")
        ce_loss = _base_loss(self.base_criterion, student_output, targets)
        # softmax over ONE logit: the mixing weights are exactly 1 and d loss / d temperature exactly 0
        mix = ops._device_consts((1.0,) * len(students), torch.float32, main.device).view(-1, 1)
        total, geo_layers = _SingleTeacherTotal.apply(ce_loss, bool(self.teacher_has_cls_token),
                                                      sel.log_temperatures, teachers, attns, *students)
        ops.trace("procrustes_queued")
        if slot is None:
            slot = plan.queue(xs, teachers, proj_t, sel._proj_s_transposed(), main.cuda_stream)
            ops.trace("chains_queued")

        # the step's inputs, for the two redo paths below; dropped when the read-back is left to the next call, so that a
        # pending read-back does not keep a whole step's token tensors alive (a failed factorisation is then not redone:
        # nothing of that step's loss depended on it)
        held = [xs, teachers]
        n_layers = len(students)

        def lost_step(why: str) -> None:
            import warnings
            warnings.warn(f"basd_selector_chain: {why}; the selector of that step is not redone (its inputs are gone): "
                          "subspace_ranks keeps its previous values, d_grass_sq of that step is NaN", RuntimeWarning)
            comp["d_grass_sq"] = torch.full((n_layers, 1), float("nan"), device=main.device)

        def complete():
            nonlocal slot
            ranks, status = plan.read_ranks(slot)
            ops.trace("ranks_read")
            if status[0] == 2 and plan.early:
                # the factorisation that was queued ahead of its input gave up waiting for it (kernels serialised by a
                # profiler, or a device that cannot run it beside the producers): plain launches from now on
                import warnings
                warnings.warn("basd_selector_chain: the early-launched factorisation timed out waiting for its input; "
                              "queueing it behind its input for the rest of this process", RuntimeWarning)
                plan.early = False
                if held[0] is None:
                    return lost_step("an early-launched factorisation timed out")
                torch.cuda.synchronize(main.device)
                slot = plan.queue(held[0], held[1], proj_t, sel._proj_s_transposed(), main.cuda_stream)
                ranks, status = plan.read_ranks(slot)
            if status[0]:       # (words 6, 7 may carry clock readings: BASD_TRIDIAG_CLOCKS)
                # workgroups sharing a matrix lost each other (bounded spin): degrade, do not die -- the selector of
                # THIS step once more with one workgroup per matrix, and keep that setting
                _single_member_mode("workgroups sharing a matrix timed out waiting for each other "
                                    f"(device oversubscribed?) [{status}]")
                if held[0] is None:
                    return lost_step("workgroups sharing a matrix timed out waiting for each other")
                torch.cuda.synchronize(main.device)
                slot = plan.queue(held[0], held[1], proj_t, sel._proj_s_transposed(), main.cuda_stream)
                ranks, status = plan.read_ranks(slot)
                if status[0]:
                    raise TridiagGiveUp(f"basd_tridiag: the factorisation failed with one workgroup per matrix [{status}]")
            sel._subspace_ranks[keys[0]] = int(ranks[0])
            if min(ranks) == 0:
                # reference: 0/0 distance -> NaN weights -> NaN tokens -> torch.linalg.svd raises
                raise torch.linalg.LinAlgError(
                    "linalg.svd: The algorithm failed to converge because the input matrix contained "
                    "non-finite values (a teacher layer has Marchenko-Pastur rank 0).")
            # layer_selector.py:99-105: nothing of THIS loss reads it; published as ``last_components["d_grass_sq"]``
            # (written on the selector's tail stream: synchronise the device before reading it)
            comp["d_grass_sq"] = plan.finish_tail(slot, ranks)

        defer = self.rank_readback == "deferred" or (self.rank_readback == "auto" and plan.ranks_certified(slot))
        ops.trace("ranks_certified")
        self._last_step_waited = not defer
        if not defer:
            if previous is not None:
                previous()
            complete()
        else:
            self.readback_deferred_steps += 1
            held[0] = held[1] = None
            # deferred: this step's ranks are read by the next forward or by the first reader of ``subspace_ranks``;
            # the PREVIOUS step's now, with this step already queued (its rank kernel finished long ago -- if not,
            # this wait is the back-pressure that keeps the host at most one step ahead)
            sel._pending_tail = complete
            if previous is not None:
                previous()
        return total, ce_loss, geo_layers, mix

    # (no torch.compiler.disable wrapper: it costs ~40 us of host time per call on the step's critical path;
    # under torch.compile the ctypes launches graph-break by themselves)
    def forward(
        self,
        student_output: torch.Tensor,
        targets: torch.Tensor,
        student_intermediates: dict[int, torch.Tensor],
        all_teacher_tokens: dict[int, torch.Tensor],
        all_teacher_attns: dict[int, torch.Tensor],
    ) -> torch.Tensor:
        ops.trace("fwd_in")
        keys = sorted(all_teacher_tokens.keys())
        students = [student_intermediates[l] for l in self.token_layers]
        for s in students:
            if s.shape[1] != self.num_student_tokens:
                raise RuntimeError(
                    f"student tokens have {s.shape[1]} tokens, expected num_student_tokens={self.num_student_tokens}")
        teachers = ops._check_common_layout([ops.as_supported(all_teacher_tokens[k]) for k in keys],
                                            "teacher token tensors")
        attns = [all_teacher_attns[k] for k in keys]

        sel = self.layer_selector
        comp: dict[str, torch.Tensor] = {}          # this step's ``last_components`` (the selector tail fills d_grass_sq)
        selector_tail = None
        if len(keys) == 1 and self.use_chain and SelectorChainPlan.supported(students, teachers):
            total, ce_loss, geo_layers, mix = self._forward_single_teacher(student_output, targets, students, keys,
                                                                           teachers, attns, comp)
        elif len(keys) == 1:
            total, ce_loss, geo_layers, mix, (read_ranks, queue_tail, selector_tail) = \
                self._forward_single_teacher_legacy(student_output, targets, students, keys, teachers, attns, comp)
        else:
            sel.finish_pending()
            ce_loss = _base_loss(self.base_criterion, student_output, targets)
            mix = sel.mixing_weights(students, keys, teachers)
            geo_layers = _ProcrustesLayers.apply(mix, bool(self.teacher_has_cls_token), False, None, teachers, attns,
                                                 *students)
            total = _UWSOCombine.apply(ce_loss, geo_layers)
            comp["d_grass_sq"] = sel._last_d_grass_sq
        comp.update(ce=ce_loss.detach(), geo_layers=geo_layers.detach(), mix=mix.detach())
        self.last_components = comp
        ops.trace("combine_queued")
        if selector_tail is not None:
            if self.sync_ranks:
                # The reference's timing: ranks read (and rank 0 raised) inside this call; called last so that
                # everything else of the step is queued before the host blocks.  While the host is about to idle
                # waiting for this step's ranks it first queues the PREVIOUS step's selector tail (~0.15 ms of
                # launches that would otherwise sit between the read-back and the caller's backward).
                sel.finish_pending()
                ranks = read_ranks()
                sel._pending_tail = lambda: queue_tail(ranks, gate_tail=True)
            else:
                # deferred: the PREVIOUS step's ranks are read now (its rank kernel finished long ago; if not,
                # this wait is the back-pressure that keeps the host at most one step ahead), this step's by the
                # next forward or by the first reader of ``subspace_ranks``
                previous, sel._pending_tail = sel._pending_tail, selector_tail
                if previous is not None:
                    previous()
        ops.trace("fwd_out")
        return total
