"""CPU oracle for the BASD loss hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a from-scratch fp32 restatement (torch CPU ops as the arithmetic
substrate) of the algorithm in the reference's ``src/losses``.  It exists only
so that ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg can check / time the HIP path against something that
travels to the GPU box (the reference itself cannot).  Nothing under
``vit-inductive-bias-distillation_amd/`` may import it.

Where the arithmetic lives: the reference delegates every numerical step to
the third-party dependency ``torch==2.10.0`` (reference ``pyproject.toml:6``),
which is not vendored under ``/root/reference``.  The same torch build is
installed in this image, so the oracle calls the same LAPACK-backed kernels
(``torch.linalg.eigvalsh / svd / svdvals``) that the reference's call sites do.

Pinning: the reference ships no tests or golden vectors, so the
oracle is pinned against outputs of the reference itself, imported here on CPU
by ``tests/golden/make_goldens.py``; the resulting fixtures live in
``tests/golden/*.npz`` and ``tests/test_oracle_vs_golden.py`` checks this file
against every one of them.

Each function cites the reference file:line it follows (paths relative to
``/root/reference``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch

F32_EPS = float(torch.finfo(torch.float32).eps)


# --------------------------------------------------------------------------- #
# a10: extraction-layer placement                    (src/losses/combined.py:34-40)
# --------------------------------------------------------------------------- #
def extraction_layers(student_depth: int, num_points: int) -> list[int]:
    if num_points == 1:
        return [student_depth - 1]
    # Python round() == round-half-to-even, as in the reference.
    return [round(i * (student_depth - 1) / (num_points - 1)) for i in range(num_points)]


# --------------------------------------------------------------------------- #
# a1: Marchenko-Pastur rank                     (src/losses/layer_selector.py:8-20)
# --------------------------------------------------------------------------- #
def mp_threshold_factor(M: int, D: int) -> float:
    """(1 + sqrt(q))**2 with q = D / M, evaluated in Python float64 exactly as
    layer_selector.py:11,18 does (``q ** 0.5``, not ``math.sqrt``)."""
    q = D / M
    return (1 + q ** 0.5) ** 2


def mp_rank_from_eigenvalues(eigvals: torch.Tensor, M: int, D: int) -> int:
    """eigvals: fp32 eigenvalues (any order).  Lower median, strict '>' against
    the fp32-rounded threshold  (layer_selector.py:17-19)."""
    ev, _ = torch.sort(eigvals.float())
    n = ev.numel()
    sigma2 = float(ev[(n - 1) // 2])                # torch.median == lower median
    lam = sigma2 * mp_threshold_factor(M, D)        # float64 on the host
    lam32 = torch.tensor(lam, dtype=torch.float32)  # tensor-vs-scalar compare is done in fp32
    return int((ev > lam32).sum())


@torch.no_grad()
def mp_rank(features: torch.Tensor) -> int:
    M, D = features.shape
    x = features
    # Gram on the smaller side, NOT centred          (layer_selector.py:12-15)
    gram = (x.T @ x) / M if M >= D else (x @ x.T) / M
    return mp_rank_from_eigenvalues(torch.linalg.eigvalsh(gram), M, D)


# --------------------------------------------------------------------------- #
# a2: top-k PCA subspace                        (src/losses/layer_selector.py:23-37)
# --------------------------------------------------------------------------- #
def pca_subspace(z_flat: torch.Tensor, k: int) -> tuple[torch.Tensor, torch.Tensor]:
    z = z_flat.float()
    z = z - z.mean(dim=0, keepdim=True)
    _, S, Vh = torch.linalg.svd(z, full_matrices=False)
    return Vh[:k].T, S[:k]


# --------------------------------------------------------------------------- #
# a8: 1-D linear token-count interpolation           (src/losses/combined.py:9-14)
# --------------------------------------------------------------------------- #
def interp_taps(n_in: int, n_out: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Tap indices / weight of ``F.interpolate(mode='linear', align_corners=False)``
    along a length-``n_in`` axis, with every intermediate in fp32."""
    i = torch.arange(n_out, dtype=torch.float32)
    scale = torch.tensor(n_in, dtype=torch.float32) / torch.tensor(n_out, dtype=torch.float32)
    src = torch.clamp(scale * (i + 0.5) - 0.5, min=0.0)
    i0 = src.floor().to(torch.int64)
    i1 = torch.clamp(i0 + 1, max=n_in - 1)
    lam = src - i0.to(torch.float32)
    return i0, i1, lam


def resample_tokens(tokens: torch.Tensor, target_n: int) -> torch.Tensor:
    """(B, N, D) -> (B, target_n, D) along the flattened token axis."""
    if tokens.shape[1] == target_n:
        return tokens
    i0, i1, lam = interp_taps(tokens.shape[1], target_n)
    lam = lam.to(tokens.dtype).view(1, -1, 1)
    return (1 - lam) * tokens[:, i0, :] + lam * tokens[:, i1, :]


# --------------------------------------------------------------------------- #
# a9: attention-weighted Procrustes loss            (src/losses/relational.py:5-50)
# --------------------------------------------------------------------------- #
def token_weights(attn: torch.Tensor, has_cls: bool, n_s: int) -> torch.Tensor:
    """relational.py:22-34 -> (B, n_s) weights summing to 1 per sample."""
    if has_cls:
        w = attn[:, :, 0, 1:].mean(dim=1)           # CLS query row, heads averaged
    else:
        w = attn.mean(dim=(1, 2))                   # heads and queries averaged
    if w.shape[1] != n_s:
        w = resample_tokens(w.unsqueeze(-1), n_s).squeeze(-1)
    return w / w.sum(dim=-1, keepdim=True)


def procrustes_terms(
    s: torch.Tensor, t: torch.Tensor, w: torch.Tensor
) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Per-sample (tr_s, tr_t, nuclear norm) of relational.py:36-48."""
    s = s.float()
    t = t.float()
    w3 = w.unsqueeze(-1)
    s_c = s - (w3 * s).sum(dim=1, keepdim=True)
    t_c = t - (w3 * t).sum(dim=1, keepdim=True)
    rw = w3.sqrt()
    s_w = rw * s_c
    t_w = rw * t_c
    tr_s = s_w.square().sum(dim=(1, 2))
    tr_t = t_w.square().sum(dim=(1, 2))
    cross = torch.bmm(s_w.transpose(1, 2), t_w)
    nuc = torch.linalg.svdvals(cross).sum(dim=-1)
    return tr_s, tr_t, nuc


def geometric_relational_loss(
    student_tokens: torch.Tensor,
    teacher_tokens: torch.Tensor,
    teacher_attn: torch.Tensor,
    *,
    has_cls_token: bool,
) -> torch.Tensor:
    w = token_weights(teacher_attn, has_cls_token, student_tokens.shape[1])
    tr_s, tr_t, nuc = procrustes_terms(student_tokens, teacher_tokens, w)
    return (tr_s + tr_t - 2.0 * nuc).mean()


# --------------------------------------------------------------------------- #
# a3-a7: Grassmannian layer selector          (src/losses/layer_selector.py:40-152)
# --------------------------------------------------------------------------- #
@dataclass
class SelectorState:
    """The three tensors of the reference module's state_dict
    (layer_selector.py:51-63)."""
    proj_s: torch.Tensor            # (D_s, D_s) orthogonal
    proj_t: torch.Tensor            # (D_s, D_t) orthonormal rows
    log_temperatures: torch.Tensor  # (E,)
    ranks: dict[int, int] = field(default_factory=dict)

    @staticmethod
    def create(num_points: int, student_dim: int, teacher_dim: int) -> "SelectorState":
        # Same global-RNG consumption order as layer_selector.py:51-54.
        proj_s = torch.empty(student_dim, student_dim)
        proj_t = torch.empty(student_dim, teacher_dim)
        torch.nn.init.orthogonal_(proj_s)
        torch.nn.init.orthogonal_(proj_t)
        log_t = torch.full((num_points,), math.log(math.exp(1.0) - 1)).requires_grad_(True)
        return SelectorState(proj_s, proj_t, log_t)


@dataclass
class SelectorTrace:
    """Intermediates the parity tests compare (not part of the reference API)."""
    ranks: dict[int, int]
    d_grass_sq: dict[int, torch.Tensor]   # per student layer: (L,)
    mix_weights: dict[int, torch.Tensor]  # per student layer: (L,)


def selector_forward(
    state: SelectorState,
    student_tokens: dict[int, torch.Tensor],
    teacher_tokens: dict[int, torch.Tensor],
    teacher_attns: dict[int, torch.Tensor],
    extraction_indices: list[int],
) -> tuple[dict[int, torch.Tensor], dict[int, torch.Tensor], SelectorTrace]:
    t_keys = sorted(teacher_tokens.keys())                      # layer_selector.py:123
    D_s = state.proj_s.shape[0]
    D_t = teacher_tokens[t_keys[0]].shape[2]

    # ranks + teacher subspaces, no grad                        (:69-74, :131-138)
    bases: dict[int, torch.Tensor] = {}
    sweights: dict[int, torch.Tensor] = {}
    with torch.no_grad():
        for key in t_keys:
            z_t = teacher_tokens[key].reshape(-1, D_t) @ state.proj_t.T
            state.ranks[key] = min(mp_rank(z_t), D_s - 1)
            bases[key], sweights[key] = pca_subspace(z_t, state.ranks[key])

    tok_stack = torch.stack([teacher_tokens[k] for k in t_keys])     # (:128)
    att_stack = torch.stack([teacher_attns[k] for k in t_keys])      # (:129)
    tau_all = torch.nn.functional.softplus(state.log_temperatures)   # (:65-67)

    mixed_tok: dict[int, torch.Tensor] = {}
    mixed_att: dict[int, torch.Tensor] = {}
    trace = SelectorTrace(dict(state.ranks), {}, {})
    for i, s_layer in enumerate(extraction_indices):                 # (:143-150)
        x = student_tokens[s_layer]
        z_s = (x.reshape(-1, x.shape[2]) @ state.proj_s.T).float()   # (:86-88)
        z_s = z_s - z_s.mean(dim=0, keepdim=True)                    # (:90-91)
        Vh_s = torch.linalg.svd(z_s, full_matrices=False)[2]         # (:92)
        dists = []
        for key in t_keys:                                           # (:95-105)
            k = state.ranks[key]
            cosines = torch.linalg.svdvals(Vh_s[:k] @ bases[key])
            theta = torch.acos(cosines.clamp(max=1.0 - F32_EPS))
            sw = sweights[key]
            dists.append((sw * theta.square()).sum() / sw.sum())
        d = torch.stack(dists)
        wts = torch.softmax(-d / tau_all[i], dim=0)                  # (:107-108)
        trace.d_grass_sq[s_layer] = d.detach()
        trace.mix_weights[s_layer] = wts.detach()
        wts = wts.to(tok_stack.dtype)                                # (:110)
        mixed_tok[s_layer] = (wts.view(-1, 1, 1, 1) * tok_stack).sum(dim=0)      # (:111)
        mixed_att[s_layer] = (wts.view(-1, 1, 1, 1, 1) * att_stack).sum(dim=0)   # (:112)
    return mixed_tok, mixed_att, trace


# --------------------------------------------------------------------------- #
# a11: the combined loss                           (src/losses/combined.py:48-85)
# --------------------------------------------------------------------------- #
def uwso_combine(losses: list[torch.Tensor]) -> torch.Tensor:
    """combined.py:78-85.  w_i = (1/L_i)/sum_j(1/L_j) with detached L."""
    eps = torch.finfo(losses[0].dtype).eps
    inv = torch.stack([1.0 / v.detach().clamp(min=eps) for v in losses])
    w = inv / inv.sum()
    total = w[0] * losses[0]
    for i in range(1, len(losses)):
        total = total + w[i] * losses[i]
    return total


@dataclass
class BASDTrace:
    ce: torch.Tensor
    geo_per_layer: list[torch.Tensor]
    selector: SelectorTrace


def basd_forward(
    state: SelectorState,
    base_criterion,
    layers: list[int],
    num_student_tokens: int,
    teacher_has_cls: bool,
    logits: torch.Tensor,
    targets: torch.Tensor,
    student_tokens: dict[int, torch.Tensor],
    teacher_tokens: dict[int, torch.Tensor],
    teacher_attns: dict[int, torch.Tensor],
) -> tuple[torch.Tensor, BASDTrace]:
    ce = base_criterion(logits, targets)                                        # (:56)
    mixed_tok, mixed_att, sel = selector_forward(
        state, student_tokens, teacher_tokens, teacher_attns, layers)            # (:58-61)
    geo = []
    for layer in layers:                                                        # (:63-75)
        aligned = resample_tokens(mixed_tok[layer], num_student_tokens)
        geo.append(geometric_relational_loss(
            student_tokens[layer], aligned, mixed_att[layer], has_cls_token=teacher_has_cls))
    geo_mean = torch.stack(geo).mean()                                          # (:76)
    total = uwso_combine([ce, geo_mean])                                        # (:78-85)
    return total, BASDTrace(ce.detach(), [g.detach() for g in geo], sel)
