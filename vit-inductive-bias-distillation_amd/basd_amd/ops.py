"""Host-side orchestration of the HIP kernels (thin wrappers over the C ABI).

PyTorch is used here for device memory (the caching allocator hands out scratch
without synchronising), the current HIP stream, and autograd bookkeeping; all
arithmetic on the path runs in ``libbasd_hip.so``.
"""
from __future__ import annotations

from dataclasses import dataclass
from functools import lru_cache

import torch

from . import _lib

import os

F32, BF16 = 0, 1
MAX_SWEEPS = 20
TWO_PASS_SVD = True      # test hook: False sends cores of 129..196 tokens through the block solver instead
TRANSPOSED_MIX_GRAD = True   # test hook: False keeps the stacked cores (riding rows) whenever the mixing weights need a gradient;
                             # "force": the transposed route also for small batches (with basd_procrustes_tuning(2))
TWO_PASS_LOG_LIMIT = 16 << 30     # bytes of rotation log above which the block solver is used (cfg-4: 1.6 GB per call)
# Symmetric eigen-solver for the selector's D_s x D_s Grams when only eigenvalues / leading eigenvectors are
# needed: "tridiag" (Householder + bisection + inverse iteration) or "jacobi" (block one-sided Jacobi).
EIG_SOLVER = os.environ.get("BASD_EIG_SOLVER", "tridiag")
if os.environ.get("BASD_GEMM_SPLIT") in ("0", "1"):      # A/B hook: 0 = fp32 MFMA Gram launches (basd_gemm_tuning)
    _lib.call("basd_gemm_tuning", int(os.environ["BASD_GEMM_SPLIT"]))
if os.environ.get("BASD_JACOBI_ORDERING") in ("0", "1"):      # A/B hook: 0 = round-robin through LDS (basd_jacobi_ordering)
    _lib.call("basd_jacobi_ordering", int(os.environ["BASD_JACOBI_ORDERING"]))


def _stream() -> int:
    # the raw hipStream_t of the current stream: torch.cuda.current_stream() builds a Stream object through three layers of
    # Python (~10 us, ~20 calls a step); this is one C call
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def as_supported(t: torch.Tensor) -> torch.Tensor:
    """fp32 and bf16 are consumed in place; anything else is widened to fp32."""
    if t.dtype in (torch.float32, torch.bfloat16):
        return t
    return t.float()


def _require_cuda(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("basd_amd kernels need CUDA/HIP tensors (there is no CPU fallback)")


def _ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


# Host-side timeline of a step (tools/host_timeline.py): list of (label, perf_counter) when switched on.
HOST_TRACE: list | None = None
CHAIN_EVENTS: list = []          # (start, rank-ready) timing events of the teacher chain while HOST_TRACE is on


def trace(label: str) -> None:
    if HOST_TRACE is not None:
        import time
        HOST_TRACE.append((label, time.perf_counter()))


GPU_MARKS: list | None = None    # (label, timing event recorded on the current stream) while switched on (tools/step_clock.py)


def gpu_mark(label: str) -> None:
    if GPU_MARKS is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        GPU_MARKS.append((label, ev))


def _tok3(x: torch.Tensor) -> tuple[int, int, int, int, int, int]:
    """(ptr, dtype, sb, sn, sd, rows_per_batch) of a (B, N, D) or (M, D) view."""
    if x.dim() == 2:
        return x.data_ptr(), _dtype_code(x), 0, x.stride(0), x.stride(1), 1 << 30
    assert x.dim() == 3
    return x.data_ptr(), _dtype_code(x), x.stride(0), x.stride(1), x.stride(2), x.shape[1]


# --------------------------------------------------------------------------- #
# dense contractions
# --------------------------------------------------------------------------- #
def gemm_nt(a: torch.Tensor, b: torch.Tensor, *, scale: float = 1.0, out: torch.Tensor | None = None,
            a_batch_stride: int = 0, b_batch_stride: int = 0, batch: int = 1, rows: int | None = None,
            n_cols: int | None = None, bias: torch.Tensor | None = None, beta: float = 0.0,
            col_mean: bool = False, mean_out: torch.Tensor | None = None, col_sums: torch.Tensor | None = None):
    """C = beta * C + scale * A @ B.T - bias.  A: (B,N,K) or (M,K) view in fp32/bf16; B: (N,K) fp32 row-major.
    With batch > 1, element z uses a/b advanced by the given batch strides.  ``col_mean``: also return the column
    means of C (from the kernel's epilogue; no second pass over C) -> (C, means (batch?, N))."""
    _require_cuda(a, b)
    ptr, dt, sb, sn, sd, rpb = _tok3(a)
    M = rows if rows is not None else (a.shape[0] * a.shape[1] if a.dim() == 3 else a.shape[0])
    K = a.shape[-1]
    N = n_cols if n_cols is not None else b.shape[-2]
    assert b.dtype == torch.float32 and b.stride(-1) == 1 and b.shape[-1] == K
    ldb = b.stride(-2)
    if out is None:
        out = torch.empty((batch, M, N) if batch > 1 else (M, N), device=a.device, dtype=torch.float32)
    part = mean = None
    tiles = gemm_nt_row_tiles(M, N, K, batch, beta)
    if col_sums is not None:            # row-tile column sums only (folded later, e.g. inside centered_grams)
        assert col_sums.is_contiguous() and col_sums.numel() == batch * tiles * N
        part = col_sums
    if col_mean:
        part = torch.empty((batch, tiles, N), device=a.device, dtype=torch.float32)
        mean = mean_out if mean_out is not None else \
            torch.empty((batch, N) if batch > 1 else (N,), device=a.device, dtype=torch.float32)
        assert mean.dtype == torch.float32 and mean.is_contiguous() and mean.numel() == batch * N
    _lib.call("basd_gemm_nt", ptr, dt, sb, sn, sd, rpb, a_batch_stride, b.data_ptr(), ldb, b_batch_stride,
              M, N, K, batch, out.data_ptr(), N, M * N, scale, _ptr(bias), beta, _ptr(part), _ptr(mean), _stream())
    return (out, mean) if col_mean else out


def gemm_nt_row_tiles(M: int, N: int, K: int, batch: int = 1, beta: float = 0.0) -> int:
    """Number of row tiles ``gemm_nt`` leaves column sums for (128-row tiles)."""
    return (M + 127) // 128


def gemm_tn(a: torch.Tensor, b: torch.Tensor, *, mean_a: torch.Tensor | None = None,
            mean_b: torch.Tensor | None = None, scale: float = 1.0, batch: int = 1, a_batch_stride: int = 0,
            b_batch_stride: int = 0, krows: int | None = None, m_cols: int | None = None,
            n_cols: int | None = None, split: bool = True) -> torch.Tensor:
    """C = scale * (A - mean_a)^T (B - mean_b), contraction over rows.  A, B: (B,N,D) / (M,D) views."""
    _require_cuda(a, b)
    pa, dt, asb, asn, asd, rpb = _tok3(a)
    pb, dtb, bsb, bsn, bsd, rpb_b = _tok3(b)
    assert dt == dtb and rpb == rpb_b
    kr = krows if krows is not None else (a.shape[0] * a.shape[1] if a.dim() == 3 else a.shape[0])
    M = m_cols if m_cols is not None else a.shape[-1]
    N = n_cols if n_cols is not None else b.shape[-1]
    splits = _lib.query("basd_gemm_tn_splits", kr) if (split and batch == 1) else 1
    out = torch.empty((batch, M, N) if batch > 1 else (M, N), device=a.device, dtype=torch.float32)
    slabs = torch.empty((batch * splits, M, N), device=a.device, dtype=torch.float32) if splits > 1 else None
    _lib.call("basd_gemm_tn", pa, pb, dt, asb, asn, asd, a_batch_stride, bsb, bsn, bsd, b_batch_stride, rpb, kr, M, N,
              batch, _ptr(mean_a), _ptr(mean_b), splits, _ptr(slabs), out.data_ptr(), N, M * N, scale, _stream())
    return out


def colmean(x: torch.Tensor) -> torch.Tensor:
    """Column means of a (B,N,D) / (M,D) view -> (D,) fp32."""
    _require_cuda(x)
    ptr, dt, sb, sn, sd, rpb = _tok3(x)
    rows = x.shape[0] * x.shape[1] if x.dim() == 3 else x.shape[0]
    cols = x.shape[-1]
    parts = _lib.query("basd_colmean_parts", rows)
    partial = torch.empty((parts, cols), device=x.device, dtype=torch.float32)
    mean = torch.empty((cols,), device=x.device, dtype=torch.float32)
    _lib.call("basd_colmean", ptr, dt, sb, sn, sd, rpb, 0, rows, cols, 1, parts, partial.data_ptr(),
              mean.data_ptr(), _stream())
    return mean


def _gram_layout(xs: list[torch.Tensor]):
    _require_cuda(*xs)
    ptr, dt, sb, sn, sd, rpb = _tok3(xs[0])
    for x in xs[1:]:
        assert _tok3(x)[1:] == (dt, sb, sn, sd, rpb) and x.shape == xs[0].shape, "Gram operands must share a layout"
    rows = xs[0].shape[0] * xs[0].shape[1] if xs[0].dim() == 3 else xs[0].shape[0]
    vec_ok = int(all(x.data_ptr() % 16 == 0 for x in xs))
    return (dt, sb, sn, sd, rpb), rows, xs[0].shape[-1], vec_ok


_COLMEAN_SCRATCH: dict = {}


def column_means(xs: list[torch.Tensor], out: torch.Tensor | None = None) -> torch.Tensor:
    """Column means of same-layout (B,N,D) / (M,D) views in one launch (+ fold) -> (n, D) fp32.
    ``out``: a contiguous fp32 buffer of n * D elements to receive them (e.g. a slice of a gradient bucket)."""
    (dt, sb, sn, sd, rpb), rows, cols, vec_ok = _gram_layout(xs)
    n, dev = len(xs), xs[0].device
    table = _ptr_table(xs)
    if out is None:
        means = torch.empty((n, cols), device=dev, dtype=torch.float32)
    else:
        assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == n * cols and out.device == dev
        means = out
    parts = _lib.query("basd_colmean_parts", rows)
    # the per-slice partial sums only live between the two launches of this call: one buffer per (shape, stream)
    key = (n, parts, cols, dev, _stream())
    partial = _COLMEAN_SCRATCH.get(key)
    if partial is None:
        if len(_COLMEAN_SCRATCH) >= 16:
            _COLMEAN_SCRATCH.clear()
        partial = _COLMEAN_SCRATCH[key] = torch.empty((n, parts, cols), device=dev, dtype=torch.float32)
    _lib.call("basd_colmean_multi", table.data_ptr(), dt, sb, sn, sd, rpb, rows, cols, n, parts, partial.data_ptr(),
              means.data_ptr(), vec_ok, _stream())
    return means.view(n, cols)


def centered_grams(xs: list[torch.Tensor], *, centered: list[bool] | None = None,
                   scales: list[float] | None = None, out: torch.Tensor | None = None,
                   splits: int | None = None, means: torch.Tensor | None = None,
                   fold: tuple[torch.Tensor, int] | None = None) -> tuple[torch.Tensor, torch.Tensor]:
    """Gram matrices of same-layout (B,N,D) / (M,D) views in two launches (+ reductions):
    out[z] = scales[z] * (X_z - 1 mu_z^T)^T (X_z - 1 mu_z^T), mu_z the column means (0 where not centred).
    ``means``: column means already queued by ``column_means`` (lets the caller put other work between the two
    stages).  Returns (out (n, D, D), means (n, D))."""
    (dt, sb, sn, sd, rpb), rows, cols, vec_ok = _gram_layout(xs)
    n, dev = len(xs), xs[0].device
    table = _ptr_table(xs)
    if fold is not None:
        part, fold_from = fold
        assert part.is_contiguous() and part.dtype == torch.float32 and part.shape[0] == n - fold_from
        means_used = None
        centered = None
    elif means is None:
        means = column_means(xs)
    if fold is not None:
        pass
    elif centered is not None and not all(centered):
        keep = _device_consts(tuple(1.0 if c else 0.0 for c in centered), torch.float32, dev)
        means_used = means * keep.unsqueeze(1)
    else:
        means_used = means
    sc = None
    if scales is not None:
        sc = _device_consts(tuple(float(s) for s in scales), torch.float32, dev)
    if splits is None:
        splits = _lib.query("basd_syrk_splits", rows, cols, n)
    slabs = torch.empty((n * splits, cols, cols), device=dev, dtype=torch.float32)
    if out is None:
        out = torch.empty((n, cols, cols), device=dev, dtype=torch.float32)
    assert out.dtype == torch.float32 and out.shape == (n, cols, cols) and out[0].is_contiguous()
    _lib.call("basd_syrk_multi", table.data_ptr(), dt, sb, sn, sd, rpb, rows, cols, n, _ptr(means_used),
              _ptr(sc), splits, slabs.data_ptr(), out.data_ptr(), out.stride(0), vec_ok,
              _ptr(fold[0]) if fold is not None else None, fold[0].shape[1] if fold is not None else 0,
              fold[1] if fold is not None else 0, _stream())
    return out, means


# --------------------------------------------------------------------------- #
# Jacobi solver
# --------------------------------------------------------------------------- #
def jacobi_onesided(W: torch.Tensor, rows_dot: int, *, n_arr: torch.Tensor | None = None,
                    want_sweeps: bool = False, tol: float = 0.0):
    """In-place one-sided Jacobi on W: (batch, n, rows_tot) memory == column-major (rows_tot x n).
    Returns column norms (batch, n) [and the sweep counts]."""
    _require_cuda(W)
    assert W.dtype == torch.float32 and W.is_contiguous() and W.dim() == 3
    batch, n, rows_tot = W.shape
    colnorm = torch.empty((batch, n), device=W.device, dtype=torch.float32)
    flags = torch.empty((_lib.query("basd_jacobi_workspace_ints", batch, MAX_SWEEPS),), device=W.device,
                        dtype=torch.int32)
    sweeps = torch.zeros((batch,), device=W.device, dtype=torch.int32) if want_sweeps else None
    _lib.call("basd_jacobi_onesided", W.data_ptr(), n * rows_tot, rows_dot, rows_tot, n, batch, _ptr(n_arr),
              colnorm.data_ptr(), n, MAX_SWEEPS, tol, flags.data_ptr(), _ptr(sweeps), _stream())
    return (colnorm, sweeps) if want_sweeps else colnorm


def sym_eig(G: torch.Tensor, kmax: int = 0) -> tuple[torch.Tensor, torch.Tensor | None, torch.Tensor, torch.Tensor]:
    """Eigen-decomposition of symmetric PSD matrices G (batch, n, n); G is DESTROYED.
    Returns (vals_desc (batch, n), vecs (batch, kmax, n) rows | None, W, colnorm) -- the last two let the
    caller extract more vectors later without re-solving."""
    assert G.dim() == 3 and G.shape[1] == G.shape[2]
    colnorm = jacobi_onesided(G, G.shape[1])
    vals, vecs = sort_extract(G, colnorm, kmax)
    return vals, vecs, G, colnorm


@dataclass
class TridiagState:
    """Householder tridiagonalisation of a batch of symmetric matrices + all eigenvalues (descending)."""
    d: torch.Tensor
    e: torch.Tensor
    tau: torch.Tensor
    vh: torch.Tensor
    vals: torch.Tensor      # (batch, n) descending
    err: torch.Tensor | None = None   # (8,) int32; [0] non-zero if the workgroups sharing a matrix lost each other
                                      # ([1:6] then: step + 1, row, member | matrix << 8, tag seen, tag wanted)
    ranks: torch.Tensor | None = None   # Marchenko-Pastur ranks of the leading matrices (``tridiagonalise(mp_rank=)``)


def tridiag_slice(ts: TridiagState, first: int, count: int) -> TridiagState:
    """Factorisations [first, first + count) of a batch as a state of their own (views; the status words are the
    batch's: one launch factored them all)."""
    sl = slice(first, first + count)
    return TridiagState(ts.d[sl], ts.e[sl], ts.tau[sl], ts.vh[sl], ts.vals[sl], ts.err, None)


def tridiagonalise(G: torch.Tensor, mp_rank: tuple | None = None) -> TridiagState:
    """G (batch, n, n) symmetric, DESTROYED.  Queues the Householder tridiagonalisation; ``vals`` is allocated but
    not filled (``tridiag_spectrum``).  No host sync.
    ``mp_rank`` = (M, D, cap, count, host_mirror | None[, mid_event handle | None]): also the Marchenko-Pastur ranks of the first ``count``
    matrices (``ts.ranks``, int32 on device), from the kernel that finishes the factorisation; ``host_mirror``: pinned
    int32 host tensor of count + 8 elements filled with the ranks and the status words (read it after an event
    recorded behind this call)."""
    _require_cuda(G)
    assert G.dtype == torch.float32 and G.is_contiguous() and G.dim() == 3 and G.shape[1] == G.shape[2]
    batch, n, _ = G.shape
    f32 = dict(device=G.device, dtype=torch.float32)
    d, e, tau = (torch.empty((batch, n), **f32) for _ in range(3))
    vh = torch.empty((batch, n, n), **f32)
    vals = torch.empty((batch, n), **f32)
    work = torch.empty((_lib.query("basd_tridiag_workspace_bytes", n, batch),), device=G.device, dtype=torch.uint8)
    ranks = None
    if mp_rank is None:
        _lib.call("basd_tridiag", G.data_ptr(), n * n, n, batch, d.data_ptr(), e.data_ptr(), tau.data_ptr(),
                  vh.data_ptr(), work.data_ptr(), _stream())
    else:
        M, D, cap, count, mirror, mid_event = (tuple(mp_rank) + (None,))[:6]
        factor = (1 + (D / M) ** 0.5) ** 2          # float64 on the host, as reference layer_selector.py:11,18
        ranks = torch.empty((count,), device=G.device, dtype=torch.int32)
        if mirror is not None:
            assert mirror.is_pinned() and mirror.dtype == torch.int32 and mirror.numel() == count + 8
        _lib.call("basd_tridiag_ranked", G.data_ptr(), n * n, n, batch, d.data_ptr(), e.data_ptr(), tau.data_ptr(),
                  vh.data_ptr(), work.data_ptr(), count, factor, cap, ranks.data_ptr(), _ptr(mirror), mid_event,
                  _stream())
    return TridiagState(d, e, tau, vh, vals, work[-32:].view(torch.int32), ranks)


def new_event() -> int:
    """A raw hipEvent_t handle (basd_event_create) for ordering streams from inside library calls."""
    import ctypes
    h = ctypes.c_void_p()
    _lib.call("basd_event_create", ctypes.byref(h))
    return h.value


def stream_wait_event(stream: "torch.cuda.Stream", event: int) -> None:
    _lib.call("basd_stream_wait_event", stream.cuda_stream, event)


def tridiag_spectrum(ts: TridiagState, first: int = 0, count: int | None = None) -> torch.Tensor:
    """All eigenvalues (descending) of matrices [first, first + count) by Sturm bisection -> view of ``ts.vals``."""
    batch, n = ts.vals.shape
    count = batch - first if count is None else count
    _lib.call("basd_tridiag_eigenvalues", ts.d[first].data_ptr(), ts.e[first].data_ptr(), n, count,
              ts.vals[first].data_ptr(), _stream())
    return ts.vals[first:first + count]


def tridiag_mp_rank(ts: TridiagState, M: int, D: int, cap: int, first: int = 0, count: int | None = None,
                    host_mirror: torch.Tensor | None = None) -> torch.Tensor:
    """Marchenko-Pastur ranks (int32, device) of matrices [first, first + count) straight from the tridiagonals
    (median by multisection + one Sturm count at the threshold; the spectra are not computed).
    ``host_mirror``: pinned int32 host tensor of count + 8 elements that the kernel fills with the ranks and the
    factorisation's status words (read it after an event recorded behind this call)."""
    batch, n = ts.vals.shape
    count = batch - first if count is None else count
    factor = (1 + (D / M) ** 0.5) ** 2          # float64 on the host, as reference layer_selector.py:11,18
    out = torch.empty((count,), device=ts.d.device, dtype=torch.int32)
    if host_mirror is not None:
        assert host_mirror.is_pinned() and host_mirror.dtype == torch.int32 and host_mirror.numel() == count + 8
    _lib.call("basd_tridiag_mp_rank", ts.d[first].data_ptr(), ts.e[first].data_ptr(), n, count, factor, cap,
              out.data_ptr(), None, _ptr(ts.err) if host_mirror is not None else None, _ptr(host_mirror), _stream())
    return out


def tridiag_mp_rank_rank1(ts: TridiagState, w: torch.Tensor, M: int, D: int, cap: int,
                          host_mirror: torch.Tensor | None = None) -> torch.Tensor:
    """Marchenko-Pastur ranks (int32, device) of Q (T + M w w^T) Q^T for every factorisation of ``ts``: the uncentred
    Grams of projected tokens from the factorisations of their centred Grams, w = Q^T zbar (batch, n) -- see
    basd_tridiag_mp_rank_rank1.  ``host_mirror``: pinned int32 tensor of batch + 8 elements (ranks + status words)."""
    batch, n = ts.d.shape
    assert w.dtype == torch.float32 and w.is_contiguous() and w.numel() == batch * n
    factor = (1 + (D / M) ** 0.5) ** 2          # float64 on the host, as reference layer_selector.py:11,18
    out = torch.empty((batch,), device=ts.d.device, dtype=torch.int32)
    if host_mirror is not None:
        assert host_mirror.is_pinned() and host_mirror.dtype == torch.int32 and host_mirror.numel() == batch + 8
    _lib.call("basd_tridiag_mp_rank_rank1", ts.d.data_ptr(), ts.e.data_ptr(), w.data_ptr(), n, batch, float(M), factor,
              cap, out.data_ptr(), _ptr(ts.err) if host_mirror is not None else None, _ptr(host_mirror), _stream())
    return out


def tridiag_eigenvalues(G: torch.Tensor) -> TridiagState:
    """G (batch, n, n) symmetric, DESTROYED.  Queues tridiagonalisation + Sturm bisection; no host sync."""
    ts = tridiagonalise(G)
    tridiag_spectrum(ts)
    return ts


def tridiag_eigenvectors(ts: TridiagState, k: int, first: int = 0, count: int | None = None) -> torch.Tensor:
    """Leading k eigenvectors (rows) of matrices [first, first + count) of the batch -> (count, k, n)."""
    batch, n = ts.vals.shape
    count = batch - first if count is None else count
    f32 = dict(device=ts.vals.device, dtype=torch.float32)
    z = torch.empty((count, k, n), **f32)
    vecs = torch.empty((count, k, n), **f32)
    sl = slice(first, first + count)
    _lib.call("basd_tridiag_eigenvectors", ts.d[sl].data_ptr(), ts.e[sl].data_ptr(), ts.tau[sl].data_ptr(),
              ts.vh[sl].data_ptr(), ts.vals[sl].data_ptr(), n, k, count, z.data_ptr(), vecs.data_ptr(), k, _stream())
    return vecs


def tridiag_apply_q(ts: TridiagState, x: torch.Tensor, transpose: bool) -> torch.Tensor:
    """Rows of x (batch, k, n) times the reflector product of the factorisation: Q x (back to the original basis) or
    Q^T x (``transpose``: into the tridiagonal's basis)."""
    batch, k, n = x.shape
    assert x.is_contiguous() and x.dtype == torch.float32 and batch == ts.d.shape[0] and n == ts.d.shape[1]
    out = torch.empty_like(x)
    _lib.call("basd_tridiag_apply_q", ts.tau.data_ptr(), ts.vh.data_ptr(), n, k, batch, x.data_ptr(), out.data_ptr(), k,
              int(bool(transpose)), _stream())
    return out


def tridiag_shifted_solve(ts: TridiagState, shifts: torch.Tensor, rhs: torch.Tensor) -> torch.Tensor:
    """x[z][t] = (T_z - shifts[z][t] I)^{-1} rhs[z][t]  (rhs (batch, k, n) in the tridiagonal's basis; the shifts are
    eigenvalues: LU with partial pivoting, tiny pivots perturbed -- project the singular direction out afterwards)."""
    batch, k, n = rhs.shape
    assert rhs.is_contiguous() and shifts.dtype == torch.float32 and shifts.shape[0] == batch and shifts.stride(1) == 1
    out = torch.empty_like(rhs)
    _lib.call("basd_tridiag_shifted_solve", ts.d.data_ptr(), ts.e.data_ptr(), shifts.data_ptr(), shifts.stride(0), n, k,
              batch, rhs.data_ptr(), out.data_ptr(), _stream())
    return out


def selector_tail(t_ts: TridiagState, t_first: int, L: int, s_ts: TridiagState, kmax: int, ranks_dev: torch.Tensor,
                  proj_s_t: torch.Tensor) -> torch.Tensor:
    """d_grass_sq (E, L) from the two factorisations and the device-side ranks: everything behind the rank
    read-back in ONE library call (see basd_selector_tail); scratch comes from two allocations."""
    E, n = s_ts.vals.shape
    dev = s_ts.vals.device
    items = E * L
    kn, kk = kmax * n, kmax * kmax
    sizes = [E * kn, E * kn, L * kn, L * kn, L * kn, L * kmax, items * kk, items * kmax, items]
    buf = torch.empty((sum(sizes),), device=dev, dtype=torch.float32)
    z_s, v_s, z_t, u_t, u_rot, sw, cos, sigma, d_out = torch.split(buf, sizes)
    ints = torch.empty((items + _lib.query("basd_jacobi_workspace_ints", items, MAX_SWEEPS),), device=dev,
                       dtype=torch.int32)
    sw_index = _device_consts(tuple(range(L)) * E, torch.int32, dev)
    sl = slice(t_first, t_first + L)
    trace("tail_alloc")
    _lib.call("basd_selector_tail", t_ts.d[sl].data_ptr(), t_ts.e[sl].data_ptr(), t_ts.tau[sl].data_ptr(),
              t_ts.vh[sl].data_ptr(), t_ts.vals[sl].data_ptr(), s_ts.d.data_ptr(), s_ts.e.data_ptr(),
              s_ts.tau.data_ptr(), s_ts.vh.data_ptr(), s_ts.vals.data_ptr(), n, E, L, kmax, ranks_dev.data_ptr(),
              proj_s_t.data_ptr(), z_s.data_ptr(), v_s.data_ptr(), z_t.data_ptr(), u_t.data_ptr(), u_rot.data_ptr(),
              sw.data_ptr(), cos.data_ptr(), ints.data_ptr(), sw_index.data_ptr(), sigma.data_ptr(),
              ints[items:].data_ptr(), d_out.data_ptr(), _stream())
    return d_out.view(E, L)


def sort_extract(W: torch.Tensor, colnorm: torch.Tensor, kmax: int, rows: int | None = None):
    batch, n, rows_tot = W.shape
    rows = rows_tot if rows is None else rows
    vals = torch.empty((batch, n), device=W.device, dtype=torch.float32)
    vecs = torch.empty((batch, kmax, rows), device=W.device, dtype=torch.float32) if kmax > 0 else None
    _lib.call("basd_sort_extract", W.data_ptr(), n * rows_tot, rows, rows_tot, n, batch, colnorm.data_ptr(), n,
              vals.data_ptr(), _ptr(vecs), kmax, _stream())
    return vals, vecs


def mp_rank_device(vals_desc: torch.Tensor, M: int, D: int, cap: int) -> torch.Tensor:
    """Marchenko-Pastur ranks (int32, on device) from descending eigenvalues (batch, n)."""
    q = D / M
    factor = (1 + q ** 0.5) ** 2          # float64 on the host, as reference layer_selector.py:11,18
    batch, n = vals_desc.shape
    out = torch.empty((batch,), device=vals_desc.device, dtype=torch.int32)
    _lib.call("basd_mp_rank", vals_desc.data_ptr(), n, batch, factor, cap, out.data_ptr(), None, _stream())
    return out


def grassmann_distance(colnorm: torch.Tensor, k_arr: torch.Tensor, sw: torch.Tensor,
                       sw_index: torch.Tensor) -> torch.Tensor:
    items, stride = colnorm.shape
    d = torch.empty((items,), device=colnorm.device, dtype=torch.float32)
    _lib.call("basd_grassmann_distance", colnorm.data_ptr(), stride, k_arr.data_ptr(), sw.data_ptr(), sw.stride(0),
              sw_index.data_ptr(), items, d.data_ptr(), None, _stream())
    return d


def sqrt_clamp(x: torch.Tensor) -> torch.Tensor:
    x = x.contiguous()
    out = torch.empty_like(x)
    _lib.call("basd_sqrt_clamp", x.data_ptr(), out.data_ptr(), x.numel(), _stream())
    return out


# --------------------------------------------------------------------------- #
# interpolation taps (reference combined.py:9-14 / relational.py:29-32), host-side tables
# --------------------------------------------------------------------------- #
@dataclass(frozen=True)
class Taps:
    tap0: torch.Tensor    # (n_out,) int32
    tap1: torch.Tensor    # (n_out,) int32
    lam: torch.Tensor     # (n_out,) fp32
    range0: torch.Tensor  # (n_in,) int32: first output row touching input row j
    range1: torch.Tensor  # (n_in,) int32: one past the last


@lru_cache(maxsize=64)
def _taps_cpu(n_in: int, n_out: int):
    i = torch.arange(n_out, dtype=torch.float32)
    scale = torch.tensor(n_in, dtype=torch.float32) / torch.tensor(n_out, dtype=torch.float32)
    src = torch.clamp(scale * (i + 0.5) - 0.5, min=0.0)
    i0 = src.floor().to(torch.int64)
    i1 = torch.clamp(i0 + 1, max=n_in - 1)
    lam = src - i0.to(torch.float32)
    r0 = torch.full((n_in,), n_out, dtype=torch.int64)
    r1 = torch.zeros((n_in,), dtype=torch.int64)
    for s in range(n_out):
        for j in (int(i0[s]), int(i1[s])):
            r0[j] = min(int(r0[j]), s)
            r1[j] = max(int(r1[j]), s + 1)
    r0 = torch.minimum(r0, r1)   # input rows no output touches get an empty range
    return i0.to(torch.int32), i1.to(torch.int32), lam, r0.to(torch.int32), r1.to(torch.int32)


_taps_dev: dict[tuple[int, int, str], Taps] = {}


def taps(n_in: int, n_out: int, device: torch.device) -> Taps | None:
    """None when the grids coincide (identity)."""
    if n_in == n_out:
        return None
    key = (n_in, n_out, str(device))
    if key not in _taps_dev:
        _taps_dev[key] = Taps(*[t.to(device) for t in _taps_cpu(n_in, n_out)])
    return _taps_dev[key]


def _ptr_table(tensors: list[torch.Tensor]) -> torch.Tensor:
    """Device array of base pointers (K8 of SURVEY.md: a pointer table instead of torch.stack)."""
    return _device_consts(tuple(t.data_ptr() for t in tensors), torch.int64, tensors[0].device)


_CONSTS: dict = {}
_CONSTS_RETIRED: list = []       # older generations of the cache, kept alive (see below)
_CONSTS_LIMIT = 512


def _device_consts(values: tuple, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    """Small read-only device arrays (pointer tables, per-matrix scales), cached by value: the caching
    allocator hands the same addresses back every step, so the steady state uploads nothing.

    The tables are read by kernels on several streams (chains, tail, caller) without ``record_stream``, so their memory
    must not be handed out again while a queued reader may still run.  No device-wide wait for that (it would stall
    training every ~100 steps once input addresses churn): a full cache is RETIRED, not freed -- the two most recent
    retired generations stay alive, i.e. a table is released no sooner than 2 x 512 newer distinct tables later.  Every
    loss step makes the host wait for that step's rank kernel, so no stream is ever more than a few steps behind the
    host; a thousand tables later (hundreds of steps) every reader is long done."""
    key = (values, dtype, device)
    t = _CONSTS.get(key)
    if t is None:
        if len(_CONSTS) >= _CONSTS_LIMIT:
            _CONSTS_RETIRED.append(dict(_CONSTS))
            del _CONSTS_RETIRED[:-2]
            _CONSTS.clear()
        t = torch.tensor(values, dtype=dtype).to(device)
        _CONSTS[key] = t
    return t


# --------------------------------------------------------------------------- #
# Procrustes forward / backward over all extraction layers
# --------------------------------------------------------------------------- #
@dataclass
class ProcrustesContext:
    omega: torch.Tensor      # (G, B, n_s)   G = E, or 1 when the teacher side is shared
    mu_s: torch.Tensor       # (E, B, D_s)
    a_prime: torch.Tensor    # (E, B, n, D_s)
    k_prime: torch.Tensor    # (E, B, n, n)
    tr_s: torch.Tensor       # (E, B)
    tr_t: torch.Tensor       # (E, B)
    nuc: torch.Tensor        # (E, B)
    loss_b: torch.Tensor     # (E, B)
    sweeps: torch.Tensor | None
    mixgrad: dict | None = None   # extra state kept only when the mixing weights need a gradient
    dx: torch.Tensor | None = None   # (E, B, n_s, D_s) student gradients queued with the forward (``grad_layers``)
    uw: torch.Tensor | None = None   # (4 + 2 E,) UW-SO combination computed inside the call (``uwso_ce``)


def _check_common_layout(tensors: list[torch.Tensor], what: str) -> list[torch.Tensor]:
    ref = tensors[0]
    out = []
    for t in tensors:
        if t.shape != ref.shape:
            raise RuntimeError(f"all {what} must share one shape (reference stacks them): {t.shape} vs {ref.shape}")
        if t.dtype != ref.dtype or t.stride() != ref.stride():
            t = t.to(ref.dtype).contiguous()
        out.append(t)
    if any(o.stride() != out[0].stride() for o in out):
        out = [o.contiguous() for o in out]
    return out


class _ProcrustesPlan:
    """Persistent scratch + a pre-filled ``BasdProcrustesArgs`` for one (shapes, layouts, options, stream) key.

    Scratch (mixed teacher, fp64 Grams / factors, stacked cores, Jacobi flags, ...) is only alive inside the call's own
    launches on ONE stream, so it is allocated once and reused by every step; what outlives the call -- the tensors kept
    for the backward and the student gradients handed to autograd -- is allocated per call (two ``torch.empty``)."""

    def __init__(self, E, L, B, n_s, d_s, n_t, d_t, H, A, has_cls, need_bwd, need_mix_grad, want_sweeps, want_dx,
                 want_uw, dev):
        import ctypes
        self.E, self.L, self.B, self.n_s, self.d_s, self.n_t, self.d_t = E, L, B, n_s, d_s, n_t, d_t
        n_a = A - (1 if has_cls else 0)
        n = min(n_s, n_t)
        self.n, self.n_a = n, n_a
        self.tp = taps(n_t, n_s, dev) if n_t < n_s else None                 # student-side (transposed) taps
        self.gt = taps(n_t, n_s, dev) if n_t > n_s else None                 # teacher-side gather taps
        self.atp = taps(n_a, n_s, dev)
        G = self.G = 1 if L == 1 else E          # softmax over one layer is exactly 1: the teacher side is shared
        f32 = dict(device=dev, dtype=torch.float32)
        slabs = (d_s + 31) // 32
        EB, GB = E * B, G * B
        self.need_bwd, self.need_mix_grad, self.want_dx, self.want_uw = need_bwd, need_mix_grad, want_dx, want_uw
        # ---- kept sizes (fresh per call): omega, mu_s, a_prime, k_prime, terms (4, E, B), uw
        self.kept_sizes = [G * B * n_s, E * B * d_s, EB * n * d_s, EB * n * n if need_bwd else 0, 4 * EB,
                           (4 + 2 * E) if want_uw else 0]
        # every piece starts on a 256-byte boundary, as separate allocations would (the kernels read them as float4)
        self.kept_padded = [(c + 63) // 64 * 64 for c in self.kept_sizes]
        self.kept_total = sum(self.kept_padded)
        # ---- scratch
        g_splits = 1
        if GB < 1024 and d_t >= 512:           # few teacher Grams with a long feature axis: split the contraction
            g_splits = max(1, min(1024 // GB, d_t // 256, 16))
        self.g_splits = g_splits
        self.g_all = torch.empty((EB + GB + (g_splits * GB if g_splits > 1 else 0), n, n), device=dev, dtype=torch.float64)
        self.l_all = torch.empty((EB + GB, n, n), device=dev, dtype=torch.float64)
        self.W = torch.empty((EB, n, 2 * n), **f32)        # memory == column-major (2n x n)
        self.tc = torch.empty((G, B, n, d_t), **f32)
        sizes = [GB * n, GB * d_t, EB * slabs, EB * n]
        padded = [(c + 63) // 64 * 64 for c in sizes]
        self.omega_t, self.mu_t, self.tr_part, self.sigma = (
            piece[:c] for piece, c in zip(torch.split(torch.empty((sum(padded),), **f32), padded), sizes))
        self.ints = torch.empty((_lib.query("basd_jacobi_workspace_ints", EB, MAX_SWEEPS) + EB,), device=dev,
                                dtype=torch.int32)
        self.sweeps = self.ints[-EB:] if want_sweeps else None
        transposed_mix = bool(need_mix_grad and need_bwd and _lib.query("basd_jacobi_plain4_fits", n)
                              and (EB >= 128 or TRANSPOSED_MIX_GRAD == "force") and TRANSPOSED_MIX_GRAD)
        jac_bytes = _lib.query("basd_jacobi_twopass_workspace_bytes", n, EB, MAX_SWEEPS) \
            if TWO_PASS_SVD and not transposed_mix else 0
        if jac_bytes > TWO_PASS_LOG_LIMIT or (jac_bytes > (256 << 20)
                                              and jac_bytes > torch.cuda.mem_get_info(dev)[0] // 2):
            jac_bytes = 0       # very large batches / little free memory: the block solver needs no log
        self.jac_ws = torch.empty((jac_bytes // 8 + 1,), device=dev, dtype=torch.int64) if jac_bytes > 0 else None
        self.h = torch.empty((E, B, n, d_s), **f32) if want_dx else None
        # gradients through the mixing weights read U Sigma (the top half of the stacked cores): where the forward takes
        # the transposed route (cores the plain LDS solver holds), it is rebuilt into this buffer
        self.w_stack = None
        self.sigma_u = None
        if transposed_mix:
            self.w_stack = torch.empty((EB, n, 2 * n), **f32)
            self.sigma_u = torch.empty((EB, n), **f32)
        self.host_ptrs = (ctypes.c_void_p * E)()

        a = self.args = _lib.ProcrustesArgs()
        a.student_host_ptrs = ctypes.cast(self.host_ptrs, ctypes.c_void_p)
        (a.E, a.L, a.G, a.B, a.n_s, a.n_t, a.d_s, a.d_t, a.H, a.A, a.has_cls, a.n_a, a.n, a.max_sweeps) = \
            E, L, G, B, n_s, n_t, d_s, d_t, H, A, int(has_cls), n_a, n, MAX_SWEEPS
        if self.atp:
            a.atap0, a.atap1, a.alam = self.atp.tap0.data_ptr(), self.atp.tap1.data_ptr(), self.atp.lam.data_ptr()
        if self.tp:
            a.tap0, a.tap1, a.lam = self.tp.tap0.data_ptr(), self.tp.tap1.data_ptr(), self.tp.lam.data_ptr()
            a.range0, a.range1 = self.tp.range0.data_ptr(), self.tp.range1.data_ptr()
        if self.gt:
            a.g0, a.g1, a.glam = self.gt.tap0.data_ptr(), self.gt.tap1.data_ptr(), self.gt.lam.data_ptr()
        a.omega_t, a.mu_t, a.tc, a.tr_part = (self.omega_t.data_ptr(), self.mu_t.data_ptr(), self.tc.data_ptr(),
                                              self.tr_part.data_ptr())
        a.g_all, a.l_all, a.W = self.g_all.data_ptr(), self.l_all.data_ptr(), self.W.data_ptr()
        a.sigma, a.jflags, a.sweeps = self.sigma.data_ptr(), self.ints.data_ptr(), _ptr(self.sweeps)
        a.h = _ptr(self.h)
        a.g_slabs = self.g_all[EB + GB:].data_ptr() if g_splits > 1 else None
        a.g_splits = g_splits
        a.jac_ws = _ptr(self.jac_ws)
        a.w_stack = _ptr(self.w_stack)
        a.sigma_u = _ptr(self.sigma_u)


_PROCRUSTES_PLANS: dict = {}


def procrustes_forward(students: list[torch.Tensor], teachers: list[torch.Tensor], attns: list[torch.Tensor],
                       mix: torch.Tensor, has_cls: bool, *, need_backward: bool = True,
                       want_sweeps: bool = False, need_mix_grad: bool = False,
                       grad_layers: torch.Tensor | None = None,
                       uwso_ce: torch.Tensor | None = None) -> ProcrustesContext:
    """students: E tensors (B, N_s, D_s); teachers: L tensors (B, N_t, D_t); attns: L tensors (B, H, A, A);
    mix: (E, L) fp32 mixing weights on device.  Returns per-sample terms for every extraction layer.
    ``grad_layers`` ((E,) fp32 on device): also queue the student-token gradients for these upstream gradients
    (``ctx.dx``: (E, B, N_s, D_s) fp32).  ``uwso_ce`` (one fp32 on the device, the base loss): the UW-SO combination
    (combined.py:76-85) is computed inside the call as well -- ``ctx.uw`` = [w_ce, w_geo, total, geo, E x w_geo / E,
    E per-layer means] -- and ``ctx.dx`` are then the gradients of ``total`` for a unit upstream gradient.
    Everything is queued by ONE library call (basd_procrustes_forward_fused) into a persistent, shape-keyed workspace
    (``_ProcrustesPlan``): what the step's host time buys is two allocations and a handful of pointer updates."""
    import ctypes
    E, L = len(students), len(teachers)
    students = [as_supported(s) for s in students]
    students = _check_common_layout([s if s.stride(2) == 1 else s.contiguous() for s in students],
                                    "student token tensors")
    teachers = _check_common_layout([as_supported(t) for t in teachers], "teacher token tensors")
    attns = _check_common_layout([as_supported(a) for a in attns], "teacher attention tensors")
    _require_cuda(*students, *teachers, *attns, mix)
    s0, t0, a0 = students[0], teachers[0], attns[0]
    dev = s0.device
    B, n_s, d_s = s0.shape
    _, n_t, d_t = t0.shape
    H, A = a0.shape[1], a0.shape[2]
    need_bwd = need_backward or grad_layers is not None
    want_dx = grad_layers is not None or uwso_ce is not None
    stream = _stream()
    # mixing-weight gradients read the call's scratch in backward (W, sigma, tc, Grams): those calls get private scratch
    key = (E, L, B, n_s, d_s, n_t, d_t, H, A, bool(has_cls), need_bwd, need_mix_grad, want_sweeps, want_dx,
           uwso_ce is not None, dev, stream, TWO_PASS_SVD)
    plan = None if need_mix_grad else _PROCRUSTES_PLANS.get(key)
    if plan is None:
        plan = _ProcrustesPlan(E, L, B, n_s, d_s, n_t, d_t, H, A, has_cls, need_bwd, need_mix_grad, want_sweeps, want_dx,
                               uwso_ce is not None, dev)
        if not need_mix_grad:
            if len(_PROCRUSTES_PLANS) >= 8:
                torch.cuda.synchronize(dev)
                _PROCRUSTES_PLANS.clear()
            _PROCRUSTES_PLANS[key] = plan
    n, n_a, G = plan.n, plan.n_a, plan.G
    EB = E * B
    mix = mix.contiguous().float()
    kept = torch.empty((plan.kept_total,), device=dev, dtype=torch.float32)
    omega, mu_s, a_prime, k_prime, terms, uw = (piece[:c] for piece, c in
                                                zip(torch.split(kept, plan.kept_padded), plan.kept_sizes))
    omega, mu_s = omega.view(G, B, n_s), mu_s.view(E, B, d_s)
    a_prime = a_prime.view(E, B, n, d_s)
    k_prime = k_prime.view(E, B, n, n) if need_bwd else None
    terms = terms.view(4, E, B)
    uw = uw if uwso_ce is not None else None
    raw = torch.empty((G, B, n_a), device=dev, dtype=torch.float32) if need_mix_grad else None
    dx = torch.empty((E, B, n_s, d_s), device=dev, dtype=torch.float32) if want_dx else None
    if plan.sweeps is not None:
        plan.sweeps.zero_()
    if uwso_ce is not None:
        assert uwso_ce.dtype == torch.float32 and uwso_ce.numel() == 1 and uwso_ce.device == dev
    if grad_layers is not None:
        grad_layers = grad_layers.contiguous().float()

    args = plan.args
    for e, s in enumerate(students):
        plan.host_ptrs[e] = s.data_ptr()
    args.student_ptrs = _ptr_table(students).data_ptr()
    args.s_dtype, args.s_sb, args.s_sn = _dtype_code(s0), s0.stride(0), s0.stride(1)
    args.s_aligned = int(all(s.data_ptr() % 16 == 0 for s in students))
    args.tok_ptrs = _ptr_table(teachers).data_ptr()
    args.t_dtype = _dtype_code(t0)
    args.t_sb, args.t_sn, args.t_sd = t0.stride()
    args.attn_ptrs = _ptr_table(attns).data_ptr()
    args.a_dtype = _dtype_code(a0)
    args.a_sb, args.a_sh, args.a_sq, args.a_sk = a0.stride()
    args.mix = mix.data_ptr()
    args.omega, args.raw, args.mu_s = omega.data_ptr(), _ptr(raw), mu_s.data_ptr()
    args.tr_s, args.tr_t, args.nuc, args.loss_b = (terms[i].data_ptr() for i in range(4))
    args.a_prime, args.k_prime = a_prime.data_ptr(), _ptr(k_prime)
    args.dx, args.grad_layers = _ptr(dx), _ptr(grad_layers)
    args.uw_ce, args.uw_out = _ptr(uwso_ce), _ptr(uw)
    gpu_mark("procrustes_begin")
    _lib.call("basd_procrustes_forward_fused", ctypes.addressof(args), stream)
    gpu_mark("procrustes_end")

    mixgrad = None
    if need_mix_grad:
        l_a, g_b = plan.l_all[:EB], plan.g_all[EB:EB + G * B]
        mixgrad = dict(raw=raw, tc=plan.tc, l_a=l_a, g_b=g_b, W=plan.w_stack if plan.w_stack is not None else plan.W,
                       sigma=plan.sigma_u if plan.w_stack is not None else plan.sigma, omega_e=omega,
                       teachers=teachers, attns=attns, tok_tab=_ptr_table(teachers), att_tab=_ptr_table(attns),
                       has_cls=has_cls, n_a=n_a, n=n, n_s=n_s, gather=plan.gt, student_taps=plan.tp, attn_taps=plan.atp)
    sweeps = plan.sweeps.clone() if plan.sweeps is not None else None
    return ProcrustesContext(omega, mu_s, a_prime, k_prime, terms[0], terms[1], terms[2], terms[3], sweeps, mixgrad,
                             dx, uw)


def scale_unless_one(x: torch.Tensor, num: torch.Tensor, den: torch.Tensor | None = None) -> None:
    """x *= num / den in place unless the ratio is exactly 1 (the launch then returns at once)."""
    _require_cuda(x, num)
    assert x.dtype == torch.float32 and x.is_contiguous() and num.dtype == torch.float32 and num.numel() == 1
    _lib.call("basd_scale_unless_one", x.data_ptr(), x.numel(), num.data_ptr(), _ptr(den), _stream())


def procrustes_student_grads(students: list[torch.Tensor], ctx: ProcrustesContext, grad_layers: torch.Tensor,
                             tnorm2: torch.Tensor | None = None):
    """d(sum_e grad_layers[e] * mean_b loss_b[e]) / d students[e]   (fp32, contiguous).
    With ``tnorm2`` (E, B, n_s) also returns d loss_b / d omega (E, B, n_s), un-scaled."""
    E = len(students)
    _, B, n, d_s = ctx.a_prime.shape
    dev = ctx.a_prime.device
    n_s = students[0].shape[1]
    tp = taps(n, n_s, dev)
    t0, t1, lam = (tp.tap0.data_ptr(), tp.tap1.data_ptr(), tp.lam.data_ptr()) if tp else (None, None, None)
    st = _stream()
    grad_layers = grad_layers.contiguous().float()
    xs = [as_supported(x) for x in students]
    xs = _check_common_layout([x if x.stride(2) == 1 else x.contiguous() for x in xs], "student token tensors")
    dx = torch.empty((E, B, n_s, d_s), device=dev, dtype=torch.float32)
    shared = ctx.omega.shape[0] == 1
    if tnorm2 is None and n <= 64:
        # small cores: H = K' A' formed inside the gradient kernel (never stored)
        status = getattr(_lib.load(), "basd_student_grad_fused")(
            _ptr_table(xs).data_ptr(), _dtype_code(xs[0]), xs[0].stride(0), xs[0].stride(1), E, B, n_s, n, d_s,
            int(all(x.data_ptr() % 16 == 0 for x in xs)), ctx.omega.data_ptr(), 0 if shared else B * n_s,
            ctx.mu_s.data_ptr(), ctx.k_prime.data_ptr(), ctx.a_prime.data_ptr(), t0, t1, lam, grad_layers.data_ptr(),
            2.0 / B, dx.data_ptr(), st)
        if status == 0:
            return list(dx.unbind(0))
        if status != _lib.EUNSUPPORTED:
            raise RuntimeError(f"basd_student_grad_fused failed with status {status}")
    # H = K' A' per (layer, sample): K' is symmetric, so this is the TN contraction
    kp = ctx.k_prime.view(E * B, n, n)
    ap = ctx.a_prime.view(E * B, n, d_s)
    h = gemm_tn(kp[0], ap[0], batch=E * B, a_batch_stride=n * n, b_batch_stride=n * d_s, krows=n, m_cols=n,
                n_cols=d_s, split=False).view(E, B, n, d_s)
    gomega = torch.empty((E, B, n_s), device=dev, dtype=torch.float32) if tnorm2 is not None else None
    _lib.call("basd_student_grad_multi", _ptr_table(xs).data_ptr(), _dtype_code(xs[0]), xs[0].stride(0),
              xs[0].stride(1), E, B, n_s, n, d_s, ctx.omega.data_ptr(), 0 if shared else B * n_s, ctx.mu_s.data_ptr(),
              h.data_ptr(), t0, t1, lam, grad_layers.data_ptr(), 2.0 / B, dx.data_ptr(), _ptr(tnorm2), _ptr(gomega), st)
    grads = list(dx.unbind(0))
    return (grads, gomega) if tnorm2 is not None else grads


def procrustes_teacher_factor(ctx: ProcrustesContext) -> tuple[torch.Tensor, torch.Tensor]:
    """(Kt (E*B, n, n), |t_hat_c|^2 (E, B, n_s)) -- see basd_teacher_factor."""
    mg = ctx.mixgrad
    E, B = ctx.loss_b.shape
    n, n_s = mg["n"], mg["n_s"]
    dev = ctx.loss_b.device
    tp = mg["student_taps"]
    t0, t1, lam = (tp.tap0.data_ptr(), tp.tap1.data_ptr(), tp.lam.data_ptr()) if tp else (None, None, None)
    r0, r1 = (tp.range0.data_ptr(), tp.range1.data_ptr()) if tp else (None, None)
    kt = torch.empty((E * B, n, n), device=dev, dtype=torch.float32)
    tnorm2 = torch.empty((E, B, n_s), device=dev, dtype=torch.float32)
    common = (mg["W"].data_ptr(), 2 * n * n, mg["sigma"].data_ptr(), n, n_s, E * B, mg["l_a"].data_ptr(),
              mg["g_b"].data_ptr(), n * n, mg["omega_e"].data_ptr(), t0, t1, lam, r0, r1, kt.data_ptr(),
              tnorm2.data_ptr())
    # the tiled form (Z through scratch) for every order: 0.87 ms against 7.5 for the LDS-resident kernel at 512 cores of
    # 196 tokens, 0.04 against 0.19 at 64 tokens (tools/probe/teacher_factor_ab.py)
    scratch = torch.empty((E * B, n, n), device=dev, dtype=torch.float32)
    _lib.call("basd_teacher_factor_tiled", *common, scratch.data_ptr(), _stream())
    return kt, tnorm2


def procrustes_mix_grads(ctx: ProcrustesContext, kt: torch.Tensor, gomega: torch.Tensor,
                         grad_layers: torch.Tensor, want_inputs: bool = False):
    """d(sum_e grad_layers[e] * mean_b loss_b[e]) / d mix  -> (E, L).
    ``want_inputs``: also return (R, g_raw): R = Kt T_c (E*B, n, d_t) with d loss_b / d T_c = 2 R, and g_raw (E, B, n_a)
    = d loss_b / d (un-normalised attention-grid token weight) -- the gradients w.r.t. the teacher tokens / attention
    of the stand-alone ``geometric_relational_loss`` are assembled from them."""
    mg = ctx.mixgrad
    E, B = ctx.loss_b.shape
    n, n_s, n_a = mg["n"], mg["n_s"], mg["n_a"]
    teachers, attns = mg["teachers"], mg["attns"]
    L = len(teachers)
    d_t = teachers[0].shape[2]
    dev = ctx.loss_b.device
    st = _stream()
    tc = mg["tc"].view(E * B, n, d_t)
    # R = Kt T_c per (layer, sample); Kt is symmetric, so the TN contraction applies
    r = gemm_tn(kt[0], tc[0], batch=E * B, a_batch_stride=n * n, b_batch_stride=n * d_t, krows=n, m_cols=n,
                n_cols=d_t, split=False)
    gt = mg["gather"]
    g0, g1, glam = (gt.tap0.data_ptr(), gt.tap1.data_ptr(), gt.lam.data_ptr()) if gt else (None, None, None)
    part_tok = torch.empty((E, B, L), device=dev, dtype=torch.float32)
    tsb, tsn, tsd = teachers[0].stride()
    if E <= 4 and L > 1:
        # all extraction layers in one pass over the teacher layers
        scratch = torch.empty((_lib.query("basd_mix_grad_tokens_scratch_floats", E, B, L, n),), device=dev,
                              dtype=torch.float32)
        _lib.call("basd_mix_grad_tokens_onepass", r.data_ptr(), mg["tok_tab"].data_ptr(), _dtype_code(teachers[0]), L,
                  tsb, tsn, tsd, E, B, n, d_t, g0, g1, glam, part_tok.data_ptr(), scratch.data_ptr(), st)
    else:
        _lib.call("basd_mix_grad_tokens", r.data_ptr(), mg["tok_tab"].data_ptr(), _dtype_code(teachers[0]), L, tsb,
                  tsn, tsd, E, B, n, d_t, g0, g1, glam, part_tok.data_ptr(), st)
    atp = mg["attn_taps"]
    a0, a1, alam = (atp.tap0.data_ptr(), atp.tap1.data_ptr(), atp.lam.data_ptr()) if atp else (None, None, None)
    ar0, ar1 = (atp.range0.data_ptr(), atp.range1.data_ptr()) if atp else (None, None)
    sb, sh, sq, sk = attns[0].stride()
    H, A = attns[0].shape[1], attns[0].shape[2]
    part_att = torch.empty((E, B, L), device=dev, dtype=torch.float32)
    g_raw = torch.empty((E, B, n_a), device=dev, dtype=torch.float32) if want_inputs else None
    _lib.call("basd_token_weight_bwd", gomega.data_ptr(), mg["raw"].data_ptr(), E, B, n_a, n_s, a0, a1, alam, ar0,
              ar1, mg["att_tab"].data_ptr(), _dtype_code(attns[0]), L, sb, sh, sq, sk, H, A, int(mg["has_cls"]),
              part_att.data_ptr(), _ptr(g_raw), st)
    per_layer = (2.0 * part_tok + part_att).sum(dim=1) / B                 # (E, L): tiny torch glue
    g_mix = per_layer * grad_layers.float().view(E, 1)
    return (g_mix, r, g_raw) if want_inputs else g_mix


# --------------------------------------------------------------------------- #
# principal angles with a backward (multi-layer teachers)
# --------------------------------------------------------------------------- #
def angle_stack(cos: torch.Tensor, k_arr: torch.Tensor) -> torch.Tensor:
    """(items, kmax, kmax) cosine matrices -> (items, kmax, 2 kmax) [masked cos ; I] stacks (column-major)."""
    items, kmax, _ = cos.shape
    out = torch.empty((items, kmax, 2 * kmax), device=cos.device, dtype=torch.float32)
    _lib.call("basd_build_angle_stack", cos.data_ptr(), kmax, k_arr.data_ptr(), items, out.data_ptr(), _stream())
    return out


def grassmann_distance_bwd(stack: torch.Tensor, colnorm: torch.Tensor, k_arr: torch.Tensor, sw: torch.Tensor,
                           sw_index: torch.Tensor, gd: torch.Tensor) -> torch.Tensor:
    items, kmax, _ = stack.shape
    gwt = torch.empty((items, kmax, kmax), device=stack.device, dtype=torch.float32)
    _lib.call("basd_grassmann_distance_bwd", stack.data_ptr(), colnorm.data_ptr(), kmax, k_arr.data_ptr(),
              sw.data_ptr(), sw.stride(0), sw_index.data_ptr(), gd.contiguous().data_ptr(), items, gwt.data_ptr(),
              _stream())
    return gwt


def eigvec_k2(m: torch.Tensor, lam: torch.Tensor) -> torch.Tensor:
    batch, D, kmax = m.shape
    k2 = torch.empty((batch, D, D), device=m.device, dtype=torch.float32)
    _lib.call("basd_eigvec_k2", m.data_ptr(), lam.data_ptr(), D, kmax, batch, k2.data_ptr(), _stream())
    return k2
