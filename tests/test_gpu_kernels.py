"""Per-kernel numerics on a real MI355X, each against an fp64 torch reference of the same op."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


def test_gemm_nt_layouts(dev):
    from basd_amd import ops
    g = torch.Generator().manual_seed(1)
    B, N, K, Nout = 5, 49, 260, 96
    x = torch.randn(B, N, K, generator=g).to(dev)
    w = torch.randn(Nout, K, generator=g).to(dev)
    ref = x.double().reshape(-1, K) @ w.double().T
    assert _rel(ops.gemm_nt(x, w), ref) < 2e-6
    full = torch.zeros(B, N + 1, K, device=dev)
    full[:, 1:] = x
    assert _rel(ops.gemm_nt(full[:, 1:], w), ref) < 2e-6              # CLS-sliced view
    chan = x.transpose(1, 2).contiguous().transpose(1, 2)             # channel-major view
    assert not chan.is_contiguous()
    assert _rel(ops.gemm_nt(chan, w), ref) < 2e-6
    assert _rel(ops.gemm_nt(x.bfloat16(), w), x.bfloat16().double().reshape(-1, K) @ w.double().T) < 2e-6
    big = torch.randn(700, 1024, generator=g).to(dev)
    wb = torch.randn(384, 1024, generator=g).to(dev)
    assert _rel(ops.gemm_nt(big, wb, scale=0.5), 0.5 * big.double() @ wb.double().T) < 2e-6


@pytest.mark.parametrize("M,K,N,cm", [(1000, 96, 200, False), (12544 // 4, 2048 // 4, 384, True), (77, 40, 64, False),
                                      # many row tiles: the XCD-aware 1-D grid, full and ragged
                                      (12544, 2048, 384, True), (8192 + 37, 300, 384, False), (9000, 260, 200, False)])
def test_gemm_nt_column_mean_epilogue(dev, M, K, N, cm):
    """Column means of C from the projection kernel's epilogue (ragged last row tile, channel-major A, bias)."""
    from basd_amd import ops
    g = torch.Generator().manual_seed(M + N)
    if cm:    # channel-major teacher view (B, K, n) -> (B, n, K)
        a = (torch.randn(M // 49, K, 49, generator=g) + 0.3).to(dev).transpose(1, 2)
    else:
        a = (torch.randn(M, K, generator=g) + 0.3).to(dev)
    b = torch.randn(N, K, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    out, mean = ops.gemm_nt(a, b, scale=0.5, bias=bias, col_mean=True)
    ref = 0.5 * a.double().reshape(-1, K) @ b.double().T - bias.double()
    assert _rel(out, ref) < 2e-6
    assert _rel(mean, ref.mean(0)) < 2e-6
    if M >= 8192:
        ab = a.bfloat16()
        outb = ops.gemm_nt(ab, b, scale=0.5, bias=bias)
        assert _rel(outb, 0.5 * ab.double().reshape(-1, K) @ b.double().T - bias.double()) < 2e-6
    slot = torch.zeros(3, N, device=dev)
    ops.gemm_nt(a, b, col_mean=True, mean_out=slot[1])
    assert _rel(slot[1], (a.double().reshape(-1, K) @ b.double().T).mean(0)) < 2e-6
    assert float(slot[0].abs().max()) == 0.0 and float(slot[2].abs().max()) == 0.0


def test_gemm_tn_and_colmean(dev):
    from basd_amd import ops
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(7, 197, 200, generator=g) + 1.5).to(dev)[:, 1:, :]   # strided rows
    flat = x.double().reshape(-1, 200)
    mean = ops.colmean(x)
    assert _rel(mean, flat.mean(0)) < 1e-6
    assert _rel(ops.gemm_tn(x, x, scale=1.0 / flat.shape[0]), flat.T @ flat / flat.shape[0]) < 2e-6
    c = flat - flat.mean(0)
    assert _rel(ops.gemm_tn(x, x, mean_a=mean, mean_b=mean), c.T @ c) < 5e-6
    # batched, no split:  C[z] = A[z]^T B[z]
    a = torch.randn(6, 49, 49, generator=g).to(dev)
    b = torch.randn(6, 49, 130, generator=g).to(dev)
    out = ops.gemm_tn(a[0], b[0], batch=6, a_batch_stride=49 * 49, b_batch_stride=49 * 130, krows=49, m_cols=49,
                      n_cols=130, split=False)
    assert _rel(out, a.double().transpose(1, 2) @ b.double()) < 2e-6


@pytest.mark.parametrize("B,N,D,dtype", [(7, 197, 200, torch.float32), (9, 50, 384, torch.float32),
                                         (33, 17, 130, torch.bfloat16), (3, 40, 64, torch.float32)])
def test_centered_grams_multi(dev, B, N, D, dtype):
    """basd_syrk_multi + basd_colmean_multi: symmetric tile pairs, pointer table, per-matrix mean / scale."""
    from basd_amd import ops
    g = torch.Generator().manual_seed(B * N + D)
    xs = [((torch.randn(B, N, D, generator=g) * (1 + 0.3 * i) + 0.7 * i).to(dtype).to(dev))[:, 1:, :] for i in range(3)]
    M = B * (N - 1)
    out, means = ops.centered_grams(xs, centered=[True, False, True], scales=[1.0, 1.0 / M, 0.5])
    tol = 5e-6 if dtype == torch.float32 else 2e-5
    for i, x in enumerate(xs):
        flat = x.double().reshape(-1, D)
        mu = flat.mean(0)
        assert _rel(means[i], mu) < 1e-6
        c = flat - (mu.float().double() if i != 1 else 0.0)
        ref = c.T @ c * (1.0, 1.0 / M, 0.5)[i]
        assert _rel(out[i], ref) < tol, (i, float(_rel(out[i], ref)))
        assert torch.equal(out[i], out[i].T), "mirrored tiles must be bit-identical"
    # default: every matrix centred, unit scale, 2-D operands
    flat = [x.reshape(-1, D).contiguous() for x in xs[:2]]
    out2, _ = ops.centered_grams(flat)
    for i in range(2):
        d = flat[i].double()
        c = d - d.mean(0).float().double()
        assert _rel(out2[i], c.T @ c) < tol


@pytest.mark.parametrize("n,rows_dot,rows_tot,batch", [
    (49, 49, 98, 5), (50, 50, 100, 5), (7, 7, 7, 5), (96, 96, 192, 5), (130, 130, 130, 5),
    # batch < 0: the 4-lanes-per-pair kernel shape, batch > 1000: the 8-lanes one (basd_jacobi_tuning)
    (49, 49, 98, -30), (31, 31, 62, -26), (64, 64, 128, -8), (9, 9, 18, -6), (24, 30, 61, -7),
    (49, 49, 98, 1030), (31, 31, 62, 1026), (64, 64, 128, 1008), (9, 9, 18, 1006), (24, 30, 61, 1007),
    # matrices that do not fit LDS: the block path (panels of 16 / 32 / 64 lanes per pair, as wide as LDS allows)
    (144, 144, 288, 3), (196, 196, 392, 2), (200, 200, 200, 2), (384, 384, 384, 2), (150, 160, 470, 2), (512, 512, 1024, 1)])
def test_jacobi_lds_invariants(dev, n, rows_dot, rows_tot, batch):
    from basd_amd import ops, _lib
    g = torch.Generator().manual_seed(n)
    _lib.call("basd_jacobi_tuning", 4 if batch < 0 else (8 if batch > 1000 else 0))
    batch = abs(batch) % 1000
    w0 = torch.randn(batch, n, rows_tot, generator=g)
    w0[:, :, :rows_dot] *= torch.logspace(0, -3, n).view(1, n, 1)       # spread the spectrum
    W = w0.clone().to(dev)
    sigma, sweeps = ops.jacobi_onesided(W, rows_dot, want_sweeps=True)
    assert int(sweeps.max()) < ops.MAX_SWEEPS, "did not converge"
    top0 = w0[:, :, :rows_dot].double()            # column c = w0[b, c, :]
    top = W[:, :, :rows_dot].double().cpu()
    sv = torch.linalg.svdvals(top0)
    got = sigma.double().cpu().sort(dim=1, descending=True).values
    assert ((got - sv).abs().max(dim=1).values / sv[:, 0]).max() < 3e-6
    gram = top @ top.transpose(1, 2)                                    # columns orthogonal -> diagonal
    off = gram - torch.diag_embed(torch.diagonal(gram, dim1=1, dim2=2))
    nrm = torch.diagonal(gram, dim1=1, dim2=2).sqrt()
    _lib.call("basd_jacobi_tuning", 0)
    # columns below 4 eps sigma_max are numerically null: the solver leaves them alone (they are round-off)
    live = (nrm > 1e-6 * nrm.amax(dim=1, keepdim=True)).double()
    cos = off.abs() / (nrm.unsqueeze(2) * nrm.unsqueeze(1)).clamp_min(1e-30) * live.unsqueeze(2) * live.unsqueeze(1)
    assert cos.max().item() < 5e-6, cos.max().item()
    if rows_tot > rows_dot:                                             # right-orthogonal invariants
        bot0 = w0[:, :, rows_dot:].double()
        bot = W[:, :, rows_dot:].double().cpu()
        assert _rel(bot.transpose(1, 2) @ bot, bot0.transpose(1, 2) @ bot0) < 1e-5
        assert _rel(top.transpose(1, 2) @ bot, top0.transpose(1, 2) @ bot0) < 1e-5


@pytest.mark.parametrize("ordering", [1, 0])
@pytest.mark.parametrize("n,rows,batch", [(49, 49, 130), (50, 50, 128), (144, 144, 130), (196, 196, 132), (195, 196, 128),
                                          (41, 64, 129), (96, 90, 128), (196, 196, 3)])
def test_jacobi_plain_batched_orderings(dev, n, rows, batch, ordering):
    """The plain batched solver (no riding rows: the transposed Procrustes cores) with both orderings -- odd-even with the
    columns in registers (jacobi_oe_kernel, the default) and round-robin through LDS: singular values against LAPACK
    (fp64), orthogonality of the live columns, X^T X unchanged (only right rotations were applied), exactly
    rank-deficient inputs (the centred cores are), odd orders (a zero column pads the line)."""
    from basd_amd import ops, _lib
    g = torch.Generator().manual_seed(7 * n + rows)
    _lib.call("basd_jacobi_ordering", ordering)
    _lib.call("basd_jacobi_tuning", 4 if batch < 128 else 0)
    try:
        w0 = torch.randn(batch, n, rows, generator=g)
        w0 *= torch.logspace(0, -3, n).view(1, n, 1)
        # exactly rank-deficient members: rows confined to a subspace of dimension r - 1 / 5
        for b, r in ((1, min(n, rows) - 1), (2, 5)):
            basis = torch.randn(r, rows, generator=g)
            w0[b] = torch.randn(n, r, generator=g) @ basis
        W = w0.clone().to(dev)
        sigma, sweeps = ops.jacobi_onesided(W, rows, want_sweeps=True)
        assert int(sweeps.max()) < ops.MAX_SWEEPS, "did not converge"
        x0 = w0.double()                                  # column c of the matrix = x0[b, c, :]
        x = W.double().cpu()
        sv = torch.linalg.svdvals(x0)
        k = sv.shape[1]
        got = sigma.double().cpu().sort(dim=1, descending=True).values[:, :k]
        assert ((got - sv).abs().max(dim=1).values / sv[:, 0]).max() < 3e-6
        rest = sigma.double().cpu().sort(dim=1, descending=True).values[:, k:]
        assert rest.numel() == 0 or float(rest.abs().max()) < 1e-5 * float(sv.max())
        assert torch.allclose(sigma.double().cpu(), x.norm(dim=2), rtol=2e-6, atol=1e-6 * float(sv.max()))
        gram = x @ x.transpose(1, 2)
        nrm = torch.diagonal(gram, dim1=1, dim2=2).sqrt()
        off = gram - torch.diag_embed(nrm ** 2)
        live = (nrm > 1e-6 * nrm.amax(dim=1, keepdim=True)).double()
        cos = off.abs() / (nrm.unsqueeze(2) * nrm.unsqueeze(1)).clamp_min(1e-30) * live.unsqueeze(2) * live.unsqueeze(1)
        assert cos.max().item() < 5e-6, cos.max().item()
        assert _rel(x.transpose(1, 2) @ x, x0.transpose(1, 2) @ x0) < 1e-5      # X V V^T X^T = X X^T
    finally:
        _lib.call("basd_jacobi_ordering", 1)
        _lib.call("basd_jacobi_tuning", 0)


def test_jacobi_per_matrix_order(dev):
    from basd_amd import ops
    g = torch.Generator().manual_seed(3)
    kmax = 40
    ks = [40, 17, 1, 0, 33]
    mats = torch.randn(len(ks), kmax, kmax, generator=g)
    W = mats.clone().to(dev)
    sigma = ops.jacobi_onesided(W, kmax, n_arr=torch.tensor(ks, dtype=torch.int32, device=dev)).cpu()
    for i, k in enumerate(ks):
        if k == 0:
            continue
        ref = torch.linalg.svdvals(mats[i, :k, :k].double())
        got = sigma[i, :k].double().sort(descending=True).values
        assert (got - ref).abs().max() / ref[0] < 3e-6


@pytest.mark.parametrize("n", [384, 200])
def test_sym_eig_block_path(dev, n):
    from basd_amd import ops
    g = torch.Generator().manual_seed(n)
    x = torch.randn(3, 4 * n, n, generator=g)
    x[:, :, :20] *= 6.0
    G0 = (x.transpose(1, 2) @ x).to(dev)
    vals, vecs, _, _ = ops.sym_eig(G0.clone(), kmax=24)
    ref = torch.linalg.eigvalsh(G0.double().cpu()).flip(1)
    assert ((vals.double().cpu() - ref).abs().max(dim=1).values / ref[:, 0]).max() < 3e-6
    v = vecs.double().cpu()                                             # rows are eigenvectors
    resid = G0.double().cpu() @ v.transpose(1, 2) - v.transpose(1, 2) * vals[:, :24].double().cpu().unsqueeze(1)
    assert resid.norm(dim=1).max() / ref[:, 0].max() < 2e-5
    eye = v @ v.transpose(1, 2)
    assert (eye - torch.eye(24)).abs().max() < 2e-5


def test_gram_chol_f64(dev):
    from basd_amd import _lib, ops
    g = torch.Generator().manual_seed(4)
    # 49 / 20 / 64 / 7: one wave per matrix; 196 / 65: one workgroup per matrix; 260: panels, factor in global memory
    for n, D in [(49, 384), (196, 96), (20, 36), (64, 80), (7, 16), (65, 80), (260, 272)]:
        p = torch.randn(3, n, D, generator=g).to(dev)
        p[2] = p[2] - p[2].mean(0, keepdim=True)      # exactly singular Gram
        G = torch.empty(3, n, n, device=dev, dtype=torch.float64)
        _lib.call("basd_gram_f64", p.data_ptr(), n * D, n, D, 3, G.data_ptr(), n * n, ops._stream())
        ref = p.double() @ p.double().transpose(1, 2)
        assert _rel(G, ref) < 1e-14
        L = torch.empty_like(G)
        _lib.call("basd_chol_f64", G.data_ptr(), n * n, n, 3, L.data_ptr(), n * n, ops._stream())
        assert torch.equal(L.triu(1), torch.zeros_like(L))
        if n <= D:
            assert _rel(L @ L.transpose(1, 2), ref) < 1e-12
        else:
            assert _rel(L @ L.transpose(1, 2), ref) < 1e-9   # rank-deficient: zero pivots


def test_resample_matches_interpolate(dev):
    from basd_amd.losses import _align_token_count
    g = torch.Generator().manual_seed(5)
    for n_in, n_out in [(49, 196), (256, 196), (144, 576), (1, 64)]:
        x = torch.randn(2, n_in, 24, generator=g).to(dev).requires_grad_(True)
        y = _align_token_count(x, n_out)
        ref_in = x.detach().clone().requires_grad_(True)
        ref = torch.nn.functional.interpolate(ref_in.transpose(1, 2), size=n_out, mode="linear",
                                              align_corners=False).transpose(1, 2)
        assert (y - ref).abs().max() < 5e-7
        gy = torch.randn(y.shape, generator=g).to(dev)
        y.backward(gy)
        ref.backward(gy)
        assert (x.grad - ref_in.grad).abs().max() < 1e-6 * ref_in.grad.abs().max()


@pytest.fixture
def tuning():
    """basd_tridiag_tuning(members, pad, lag, threads, tail): test hook of the library; restored afterwards."""
    from basd_amd import _lib

    def set_(members=-1, pad=-1, lag=-1, threads=-1, tail=-1, reset=0):
        _lib.call("basd_tridiag_tuning", members, pad, lag, threads, tail, reset)
    yield set_
    _lib.call("basd_tridiag_tuning", -1, -1, -1, -1, -1, 1)


@pytest.mark.parametrize("n", [384, 768, 100, 45, 33])
def test_tridiag_member_count_is_invisible(dev, n, tuning):
    """The workgroups sharing a matrix exchange rows of identical arithmetic: d, e, tau and the reflectors are
    bit-identical whatever the member count (vector+full, vector+ragged and scalar kernels)."""
    from basd_amd import ops
    g = torch.Generator().manual_seed(7 * n)
    x = torch.randn(3, 4 * n, n, generator=g)
    G0 = (x.transpose(1, 2) @ x).to(dev)
    outs = []
    for members in ("1", "2", "5", "16", "16x"):
        # tail=3: the shared stage + tail2 for every order (orders 257..384 otherwise never enter the shared stage)
        tuning(members=int(members.rstrip("x")), tail=3)
        if members.endswith("x"):       # ids of one matrix no longer agree mod 8: its members sit on different XCDs
            tuning(pad=3)
        ts = ops.tridiag_eigenvalues(G0.clone())
        assert int(ts.err[0].item()) == 0
        outs.append(ts)
    for ts in outs[1:]:
        for name in ("d", "e", "tau", "vh", "vals"):
            assert torch.equal(getattr(ts, name), getattr(outs[0], name)), (name, n)
    # and it is a tridiagonalisation: eigenvalues of T == eigenvalues of G
    ref = torch.linalg.eigvalsh(G0.double().cpu()).flip(1)
    assert ((outs[-1].vals.double().cpu() - ref).abs().max(dim=1).values / ref[:, 0]).max() < 3e-6


@pytest.mark.parametrize("tail", [1, 2, 3])
@pytest.mark.parametrize("n", [384, 256, 257, 300, 320, 383, 192, 768, 64, 7, 2])
def test_tridiag_tail_stage_reconstructs_the_matrix(dev, n, tail, tuning):
    """The register-resident stages -- tail=1: orders 257..384 whole in tridiag_packed_kernel (upper triangle in one CU's
    registers), other orders tail2 (whole factorisation for n <= 256, last 256 steps above); 3: shared stage + tail2
    for every order; 2: the four-barrier tail kernel: Q T Q^T gives the matrix back, Q is orthogonal, and T has the
    spectrum of the all-shared-stage factorisation."""
    from basd_amd import ops
    g = torch.Generator().manual_seed(31 * n)
    x = torch.randn(3, 3 * n + 5, n, generator=g)
    G0 = (x.transpose(1, 2) @ x).to(dev)
    tuning(tail=tail)
    ts = ops.tridiag_eigenvalues(G0.clone())
    assert int(ts.err[0].item()) == 0
    eye = torch.eye(n, device=dev).repeat(3, 1, 1).contiguous()
    Qt = ops.tridiag_apply_q(ts, eye, transpose=False).double()        # row i = Q e_i
    T = torch.diag_embed(ts.d.double()) + torch.diag_embed(ts.e[:, :n - 1].double(), 1) + \
        torch.diag_embed(ts.e[:, :n - 1].double(), -1)
    rec = Qt.transpose(1, 2) @ T @ Qt
    scale = G0.double().abs().amax(dim=(1, 2), keepdim=True)
    assert ((rec - G0.double()).abs() / scale).max() < 2e-5
    assert (Qt @ Qt.transpose(1, 2) - torch.eye(n, device=dev, dtype=torch.float64)).abs().max() < 2e-5
    tuning(tail=0)
    ref = ops.tridiag_eigenvalues(G0.clone())
    assert ((ts.vals - ref.vals).abs().amax(dim=1) / ref.vals[:, 0]).max() < 3e-6


@pytest.mark.parametrize("lag", ["0", "3", "11"])
def test_tridiag_lagging_member(dev, tuning, lag):
    """A member that is slower than the others every step (test hook) -- early on, in the middle, or the one that
    lives to the end -- must neither be lost (it only ever needs granules of members that wait for it) nor
    change a bit of the result.  (Regression: members whose rows were all reduced used to stay in the loop,
    nobody waited for them, and once they fell two steps behind their granules were gone.)"""
    from basd_amd import ops
    g = torch.Generator().manual_seed(5)
    n = 384
    x = torch.randn(2, 4 * n, n, generator=g)
    G0 = (x.transpose(1, 2) @ x).to(dev)
    tuning(members=12, tail=0)                          # the shared stage for the WHOLE factorisation
    ref = ops.tridiag_eigenvalues(G0.clone())           # same configuration, nobody lagging
    tuning(lag=int(lag))
    ts = ops.tridiag_eigenvalues(G0.clone())
    assert ts.err.tolist()[0] == 0, ts.err.tolist()
    for name in ("d", "e", "tau", "vh"):
        assert torch.equal(getattr(ts, name), getattr(ref, name)), name


@pytest.mark.parametrize("tail,n", [(3, 384), (1, 384), (1, 768)])
def test_tridiag_members_under_uneven_load(dev, tuning, tail, n):
    """Hand-off stress: two shared-matrix factorisations on two streams while a third stream keeps the chip busy
    with long MFMA workgroups (the situation of a training step).  Every launch must reproduce the idle-chip
    result bit for bit and leave its status word at zero.  tail=3: the shared stage (members polling each other);
    tail=1: the packed kernel at 384 (nothing to hand off: must be bit-stable under load all the same), the shared stage
    as production runs it at 768.  (Round 3: this test caught member workgroups of fewer than 1024 threads going wrong
    beside the split-operand Gram launch -- see tridiag_impl.)"""
    from basd_amd import ops
    tuning(tail=tail)
    g = torch.Generator().manual_seed(11)
    xa = torch.randn(2, 4 * n, n, generator=g)
    xb = torch.randn(4, 4 * n, n, generator=g)
    Ga, Gb = (xa.transpose(1, 2) @ xa).to(dev), (xb.transpose(1, 2) @ xb).to(dev)
    ra, rb = ops.tridiag_eigenvalues(Ga.clone()), ops.tridiag_eigenvalues(Gb.clone())    # idle chip
    big = [torch.randn(256 * 384 // n, 197, n, generator=g).to(dev)[:, 1:, :] for _ in range(4)]
    s1, s2, s3 = (torch.cuda.Stream() for _ in range(3))
    torch.cuda.synchronize()
    for it in range(25):
        with torch.cuda.stream(s3):
            ops.centered_grams(big)
        with torch.cuda.stream(s1):
            ta = ops.tridiag_eigenvalues(Ga.clone())
        with torch.cuda.stream(s2):
            tb = ops.tridiag_eigenvalues(Gb.clone())
        torch.cuda.synchronize()
        assert int(ta.err[0].item()) == 0 and int(tb.err[0].item()) == 0, it
        for got, ref in ((ta, ra), (tb, rb)):
            for name in ("d", "e", "tau", "vh"):
                assert torch.equal(getattr(got, name), getattr(ref, name)), (it, name)


@pytest.mark.parametrize("tail", [3, 1])
@pytest.mark.parametrize("n", [384, 768, 200])
def test_tridiag_does_not_read_stale_lds(dev, tuning, tail, n):
    """Whatever the previous tenant of a CU left in LDS (zeros, NaNs, another kernel's tiles) must not reach the
    factorisation: bit-identical d, e, tau, reflectors and spectra behind ``basd_debug_fill_lds`` with two patterns."""
    from basd_amd import ops, _lib
    tuning(tail=tail)
    g = torch.Generator().manual_seed(n)
    x = torch.randn(3, 4 * n, n, generator=g)
    G = (x.transpose(1, 2) @ x).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for pattern in (0x00000000, 0x7FC00000, 0x3F800000):
        _lib.call("basd_debug_fill_lds", pattern, st)
        ts = ops.tridiag_eigenvalues(G.clone())
        torch.cuda.synchronize()
        assert int(ts.err[0].item()) == 0
        out.append(ts)
    for other in out[1:]:
        for name in ("d", "e", "tau", "vh", "vals"):
            assert torch.equal(getattr(out[0], name), getattr(other, name)), name


@pytest.mark.parametrize("n,k", [(384, 48), (192, 20), (768, 80), (100, 100), (45, 10)])
def test_tridiag_eigensolver(dev, n, k):
    """Householder tridiagonalisation + Sturm bisection + inverse iteration against fp64 eigh."""
    from basd_amd import ops
    g = torch.Generator().manual_seed(n + k)
    x = torch.randn(2, 6 * n, n, generator=g)
    x[:, :, :24] *= torch.linspace(8.0, 3.0, 24)          # separated leading spectrum on top of a noise bulk
    G0 = (x.transpose(1, 2) @ x).to(dev)
    ts = ops.tridiag_eigenvalues(G0.clone())
    ref_vals, ref_vecs = torch.linalg.eigh(G0.double().cpu())
    ref_vals, ref_vecs = ref_vals.flip(1), ref_vecs.flip(2)
    assert ((ts.vals.double().cpu() - ref_vals).abs().max(dim=1).values / ref_vals[:, 0]).max() < 3e-6
    vecs = ops.tridiag_eigenvectors(ts, k).double().cpu()               # (2, k, n) rows
    eye = vecs @ vecs.transpose(1, 2)
    assert (eye - torch.eye(k, dtype=torch.float64)).abs().max() < 1e-4
    resid = G0.double().cpu() @ vecs.transpose(1, 2) - vecs.transpose(1, 2) * ts.vals[:, :k].double().cpu().unsqueeze(1)
    assert resid.norm(dim=1).max() / ref_vals[:, 0].max() < 2e-5
    # well separated leading subspace: projector agrees with fp64
    kk = min(k, 24)
    p = vecs[:, :kk].transpose(1, 2) @ vecs[:, :kk]
    pr = ref_vecs[:, :, :kk] @ ref_vecs[:, :, :kk].transpose(1, 2)
    assert (p - pr).abs().max() < 1e-4


@pytest.mark.parametrize("n,M", [(384, 12544), (192, 2048), (100, 300), (33, 64)])
def test_tridiag_mp_rank_matches_full_spectrum(dev, n, M):
    """basd_tridiag_mp_rank (median by multisection + one Sturm count) against the rank computed from the complete
    bisection spectrum, and against fp64 eigvalsh."""
    from basd_amd import ops
    g = torch.Generator().manual_seed(n)
    out = []
    mats = []
    for r in (3, 17, n // 3):
        x = torch.randn(M, r, generator=g) @ torch.randn(r, n, generator=g) * (4.0 / r ** 0.5) + torch.randn(M, n, generator=g)
        mats.append(x.T @ x / M)
    G0 = torch.stack(mats).to(dev)
    ts = ops.tridiag_eigenvalues(G0.clone())
    fast = ops.tridiag_mp_rank(ts, M, n, cap=n - 1).cpu()
    full = ops.mp_rank_device(ts.vals, M, n, cap=n - 1).cpu()
    assert torch.equal(fast, full), (fast, full)
    ev = torch.linalg.eigvalsh(G0.double().cpu())
    lam = ev[:, (n - 1) // 2] * (1 + (n / M) ** 0.5) ** 2
    ref = (ev > lam.unsqueeze(1)).sum(1).clamp(max=n - 1).to(torch.int32)
    assert torch.equal(fast, ref), (fast, ref)
    assert int(fast.min()) >= 1
    # the ranks of the leading matrices straight out of the factorisation's last kernel (basd_tridiag_ranked), with
    # the pinned host mirror the training step reads
    pin = torch.empty((2 + 8,), dtype=torch.int32, pin_memory=True)
    ts2 = ops.tridiagonalise(G0.clone(), mp_rank=(M, n, n - 1, 2, pin))
    torch.cuda.synchronize()
    assert torch.equal(ts2.ranks.cpu(), ref[:2]), (ts2.ranks, ref)
    assert pin.tolist() == ref[:2].tolist() + [0] * 8


@pytest.mark.parametrize("n,M", [(768, 25088), (384, 12544), (100, 300)])
def test_mp_rank_from_the_centred_factorisation(dev, n, M):
    """basd_tridiag_mp_rank_rank1: the Marchenko-Pastur rank of the UNCENTRED Gram z^T z / M counted in the basis that
    tridiagonalises the CENTRED one (z^T z = Q (T + M w w^T) Q^T, w = Q^T zbar; eigenvalues of a rank-one modification
    counted by one fp64 Sturm recurrence) -- against the rank from the uncentred Gram's own factorisation and against
    fp64 eigvalsh, for data with a LARGE mean (the rank-one term dominates the spectrum) and with none."""
    from basd_amd import ops
    g = torch.Generator().manual_seed(3 * n)
    grams_c, grams_u, zbars = [], [], []
    for r, shift in ((5, 3.0), (n // 8, 0.2), (17, 0.0)):
        z = (torch.randn(M, r, generator=g, dtype=torch.float64) @ torch.randn(r, n, generator=g, dtype=torch.float64)
             * (4.0 / r ** 0.5) + torch.randn(M, n, generator=g, dtype=torch.float64) + shift)
        zb = z.mean(0)
        zc = z - zb
        grams_c.append((zc.T @ zc).float())
        grams_u.append((z.T @ z / M).float())
        zbars.append(zb.float())
    Gc, Gu, zbar = torch.stack(grams_c).to(dev), torch.stack(grams_u).to(dev), torch.stack(zbars).to(dev)
    ts = ops.tridiagonalise(Gc.clone())
    w = ops.tridiag_apply_q(ts, zbar.view(3, 1, n).contiguous(), transpose=True).view(3, n)
    pin = torch.empty((3 + 8,), dtype=torch.int32, pin_memory=True)
    got = ops.tridiag_mp_rank_rank1(ts, w, M, n, n - 1, pin)
    direct = ops.tridiag_mp_rank(ops.tridiag_eigenvalues(Gu.clone()), M, n, cap=n - 1)
    ev = torch.linalg.eigvalsh(Gu.double().cpu())
    lam = ev[:, (n - 1) // 2] * (1 + (n / M) ** 0.5) ** 2
    ref = (ev > lam.unsqueeze(1)).sum(1).clamp(max=n - 1).to(torch.int32)
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), ref), (got, ref)
    assert torch.equal(direct.cpu(), ref), (direct, ref)
    assert pin.tolist()[:3] == ref.tolist() and not any(pin.tolist()[3:])


def test_rank_certificate_is_sound(dev):
    """basd_rank_certificate: flag = 1 must IMPLY that every Marchenko-Pastur rank (fp64 spectrum, lower median, strict
    '>', layer_selector.py:16-19) is >= 1 -- over flat spectra, barely-separated spikes, heavy tails, one matrix of a
    batch failing, NaNs; and it must actually fire on spectra like the ones teachers produce (a few dominant directions)."""
    from basd_amd import _lib
    g = torch.Generator().manual_seed(11)
    n, M = 96, 3000
    factor = (1 + (n / M) ** 0.5) ** 2
    q, _ = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))

    def gram(ev):
        return ((q * ev) @ q.T).float()

    def mp_rank(G):
        ev = torch.linalg.eigvalsh(G.double())
        lam = ev[(n - 1) // 2] * factor
        return int((ev > lam).sum())

    spectra = {
        "flat": torch.ones(n, dtype=torch.float64),
        "noise": torch.distributions.Chi2(M).sample((n,)).double() / M,
        "spike_below": torch.cat([torch.tensor([factor * 0.999]), torch.ones(n - 1)]).double(),
        "spike_just_above": torch.cat([torch.tensor([factor * 1.01]), torch.ones(n - 1)]).double(),
        "spike_x3": torch.cat([torch.tensor([3.0]), torch.ones(n - 1)]).double(),
        "spike_x40": torch.cat([torch.tensor([40.0]), torch.ones(n - 1)]).double(),
        "signal_12": torch.cat([torch.full((12,), 50.0), torch.ones(n - 12)]).double(),
        "power_law": (torch.arange(1, n + 1, dtype=torch.float64)) ** -1.5,
        "upper_half_big": torch.cat([torch.full((n // 2 + 2,), 10.0), torch.ones(n - n // 2 - 2)]).double(),
        "zero": torch.zeros(n, dtype=torch.float64),
    }
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch = torch.zeros(_lib.query("basd_rank_certificate_scratch_bytes", 2), dtype=torch.uint8, device=dev)
    fired = {}
    for name, ev in spectra.items():
        G = gram(ev).to(dev).contiguous()
        _lib.call("basd_rank_certificate", G.data_ptr(), None, n, 1, factor, scratch.data_ptr(), flag.data_ptr(), torch.cuda.current_stream().cuda_stream)
        fired[name] = int(flag.item())
        if fired[name]:
            assert mp_rank(G.cpu()) >= 1, name
    assert not fired["flat"] and not fired["noise"] and not fired["spike_below"] and not fired["zero"]
    assert not fired["upper_half_big"]                       # rank 0 by construction (the median is one of the big ones)
    assert fired["spike_x40"] and fired["signal_12"] and fired["power_law"], fired
    # random spectra: soundness only
    for trial in range(40):
        k = int(torch.randint(0, 8, (1,), generator=g))
        ev = torch.rand(n, generator=g, dtype=torch.float64) + 0.5
        ev[:k] *= 10.0 ** float(torch.rand(1, generator=g) * 2)
        G = gram(ev).to(dev).contiguous()
        _lib.call("basd_rank_certificate", G.data_ptr(), None, n, 1, factor, scratch.data_ptr(), flag.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if int(flag.item()):
            assert mp_rank(G.cpu()) >= 1, (trial, k)
    # a flat spectrum plus a large mean: only the mean-based bounds can see that (uncentred Grams of features with a mean)
    zb = torch.randn(n, generator=g, dtype=torch.float64)
    zb *= 3.0 / zb.norm()
    for scale, expect in ((1.0, 1), (0.05, 0)):
        z = (zb * scale).float()
        G = (gram(torch.ones(n, dtype=torch.float64)).double() + torch.outer(z.double(), z.double())).float().to(dev).contiguous()
        zd = z.to(dev).contiguous()
        _lib.call("basd_rank_certificate", G.data_ptr(), zd.data_ptr(), n, 1, factor, scratch.data_ptr(), flag.data_ptr(),
                  torch.cuda.current_stream().cuda_stream)
        assert int(flag.item()) == expect
        if expect:
            assert mp_rank(G.cpu()) >= 1
    # a batch is certified only if EVERY matrix is; NaN never certifies
    both = torch.stack([gram(spectra["signal_12"]), gram(spectra["flat"])]).to(dev).contiguous()
    _lib.call("basd_rank_certificate", both.data_ptr(), None, n, 2, factor, scratch.data_ptr(), flag.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert int(flag.item()) == 0
    both[1] = both[0]
    _lib.call("basd_rank_certificate", both.data_ptr(), None, n, 2, factor, scratch.data_ptr(), flag.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert int(flag.item()) == 1
    assert not scratch.any(), "every launch leaves its scratch zeroed"
    both[1, 3, 5] = float("nan")
    _lib.call("basd_rank_certificate", both.data_ptr(), None, n, 2, factor, scratch.data_ptr(), flag.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert int(flag.item()) == 0


def _ulp_step(x: np.float32, steps: int) -> np.float32:
    x = np.float32(x)
    for _ in range(abs(steps)):
        x = np.nextafter(x, np.float32(np.inf if steps > 0 else -np.inf), dtype=np.float32)
    return x


@pytest.mark.parametrize("n,M", [(384, 12544), (100, 300)])
def test_mp_rank_near_the_threshold(dev, n, M):
    """Tie policy of the Marchenko-Pastur rank (layer_selector.py:16-19: lower median, float64 factor, threshold rounded
    to fp32, STRICT ``>``), DESIGN.md section 2.  Diagonal Grams make every eigenvalue exact in every route, so the
    comparison is decidable to the last bit: one eigenvalue is planted 0, +-1, +-4, +-64 ulps from
    fp32(median * factor).
      * from a given spectrum (``basd_mp_rank``, what ``marchenko_pastur_rank`` / ``_estimate_ranks`` end in): exact at
        every offset, equality is NOT counted;
      * straight from the tridiagonal (``basd_tridiag_ranked`` / ``basd_tridiag_mp_rank``, the training step): the
        median comes out of a bisection that stops at a 2-ulp bracket, so the threshold itself is only known to
        +-2 ulps: exact from 4 ulps on, within the two neighbouring answers below that."""
    from basd_amd import ops
    rng = np.random.default_rng(n)
    factor = (1 + (n / M) ** 0.5) ** 2
    k_sig = 7
    # the bulk stays below the threshold (median ~0.8, factor >= 1.38): only the planted eigenvalue sits next to it
    bulk = rng.uniform(0.6, 1.0, n - k_sig - 1).astype(np.float32)
    signal = rng.uniform(20.0, 60.0, k_sig).astype(np.float32)
    for off in (0, 1, -1, 4, -4, 64, -64):
        base = np.concatenate([bulk, signal])
        # the planted value replaces nothing below the median: it is far above it for every offset
        med = np.sort(np.concatenate([base, [np.float32(10.0)]]))[(n - 1) // 2]
        thr = np.float32(np.float64(med) * factor)
        planted = _ulp_step(thr, off)
        lam = np.concatenate([base, [planted]]).astype(np.float32)
        assert np.sort(lam)[(n - 1) // 2] == med
        expect = int((lam > thr).sum())
        with_planted, without = k_sig + 1, k_sig
        assert expect == (with_planted if off > 0 else without)        # equality (off == 0) is NOT counted
        perm = rng.permutation(n)
        diag = torch.from_numpy(lam[perm])
        # (1) from the spectrum
        vals = torch.from_numpy(np.sort(lam)[::-1].copy()).to(dev).unsqueeze(0)
        assert int(ops.mp_rank_device(vals, M, n, cap=n - 1)[0]) == expect, off
        # (2) straight from the tridiagonal, both entry points
        G = torch.diag(diag).to(dev).unsqueeze(0).contiguous()
        pin = torch.empty((1 + 8,), dtype=torch.int32, pin_memory=True)
        ts = ops.tridiagonalise(G.clone(), mp_rank=(M, n, n - 1, 1, pin))
        ops.tridiag_spectrum(ts)
        alone = ops.tridiag_mp_rank(ts, M, n, cap=n - 1)
        torch.cuda.synchronize()
        for got in (int(ts.ranks[0]), int(alone[0]), int(pin[0])):
            if abs(off) >= 4:
                assert got == expect, (off, got, expect)
            else:
                assert got in (with_planted, without), (off, got)
        # the bisection spectrum of a diagonal matrix is the diagonal itself, to the 2-ulp bracket it stops at
        np.testing.assert_allclose(ts.vals[0].cpu().numpy(), np.sort(lam)[::-1], rtol=3e-7)


def test_mp_rank_dense_with_a_planted_gap(dev):
    """The same question for dense Grams, where it is only decidable up to the accuracy of ANY fp32 symmetric
    eigen-solver (the reference's LAPACK route included): ~n eps lambda_max.  An eigenvalue planted 2e-3 (relative)
    above / below the threshold must be counted / not counted, as the fp64 spectrum of the same fp32 matrix says."""
    from basd_amd import ops
    n, M = 384, 12544
    g = torch.Generator().manual_seed(5)
    q, _ = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))
    factor = (1 + (n / M) ** 0.5) ** 2
    rng = np.random.default_rng(5)
    for rel in (2e-3, -2e-3):
        lam = np.concatenate([rng.uniform(0.6, 1.0, n - 9), rng.uniform(8.0, 14.0, 8)])
        med = np.sort(np.concatenate([lam, [10.0]]))[(n - 1) // 2]
        lam = np.concatenate([lam, [med * factor * (1 + rel)]])
        G = (q @ torch.diag(torch.from_numpy(lam)) @ q.T)
        G = ((G + G.T) / 2).float()
        ev = torch.linalg.eigvalsh(G.double())
        thr = float(np.float32(float(ev[(n - 1) // 2]) * factor))
        ref = int((ev > thr).sum())
        assert ref == (9 if rel > 0 else 8)
        pin = torch.empty((1 + 8,), dtype=torch.int32, pin_memory=True)
        ts = ops.tridiagonalise(G.to(dev).unsqueeze(0).contiguous(), mp_rank=(M, n, n - 1, 1, pin))
        torch.cuda.synchronize()
        assert int(ts.ranks[0]) == ref and int(pin[0]) == ref, (rel, int(ts.ranks[0]), ref)


def test_align_token_count_golden(dev, golden):
    """``_align_token_count`` (combined.py:9-14) against the reference's own outputs (tests/golden/align.npz, written by
    the imported reference): 49 -> 196, 256 -> 196, 144 -> 576, 1 -> 64, identity, and a down-sampling case."""
    from basd_amd.losses import _align_token_count
    g = golden("align.npz")
    for tag in ("49_196", "256_196", "144_576", "1_64", "64_64", "7_3"):
        x = torch.from_numpy(g[f"{tag}_x"]).to(dev)
        want = g[f"{tag}_y"]
        y = _align_token_count(x, want.shape[1])
        np.testing.assert_allclose(y.cpu().numpy(), want, rtol=2e-6, atol=1e-7, err_msg=tag)
        # strided input (a CLS-sliced view), as the trainer hands it over
        full = torch.zeros(x.shape[0], x.shape[1] + 1, x.shape[2], device=dev)
        full[:, 1:] = x
        np.testing.assert_allclose(_align_token_count(full[:, 1:], want.shape[1]).cpu().numpy(), want, rtol=2e-6,
                                   atol=1e-7, err_msg=tag + " strided")


@pytest.mark.parametrize("n", [384, 100, 45])
def test_tridiag_apply_q_and_shifted_solve(dev, n):
    """The reflector product in both directions and the shifted tridiagonal solve behind the multi-layer backward."""
    from basd_amd import ops
    g = torch.Generator().manual_seed(3 * n)
    x = torch.randn(2, 4 * n, n, generator=g)
    x[:, :, :8] *= torch.linspace(6.0, 2.0, 8)
    G0 = (x.transpose(1, 2) @ x).to(dev)
    ts = ops.tridiag_eigenvalues(G0.clone())
    eye = torch.eye(n, device=dev).repeat(2, 1, 1).contiguous()
    q_rows = ops.tridiag_apply_q(ts, eye, transpose=False)          # row i = Q e_i  ->  q_rows = Q^T
    Q = q_rows.transpose(1, 2).double()
    assert (Q.transpose(1, 2) @ Q - torch.eye(n, device=dev, dtype=torch.float64)).abs().max() < 5e-5
    T = torch.diag_embed(ts.d.double()) + torch.diag_embed(ts.e.double()[:, :n - 1], 1) + torch.diag_embed(ts.e.double()[:, :n - 1], -1)
    assert ((Q @ T @ Q.transpose(1, 2) - G0.double()).abs().max() / G0.abs().max()) < 5e-6
    v = torch.randn(2, 5, n, generator=g).to(dev)
    back = ops.tridiag_apply_q(ts, ops.tridiag_apply_q(ts, v, transpose=True), transpose=False)
    assert _rel(back, v.double()) < 5e-6
    assert _rel(ops.tridiag_apply_q(ts, v, transpose=True), v.double() @ Q) < 5e-6      # rows: (Q^T v)^T = v^T Q
    # a regular shift (between two eigenvalues): plain solve
    vals = ts.vals.double()
    shifts = (0.5 * (vals[:, 2:7] + vals[:, 3:8])).float().contiguous()                  # (2, 5)
    sol = ops.tridiag_shifted_solve(ts, shifts, v)
    for z in range(2):
        for t in range(5):
            ref = torch.linalg.solve(T[z] - shifts[z, t].double() * torch.eye(n, device=dev, dtype=torch.float64), v[z, t].double())
            assert _rel(sol[z, t], ref) < 2e-3, (z, t, float(_rel(sol[z, t], ref)))
    # an eigenvalue as the shift: finite, and exact on the orthogonal complement of its eigenvector
    lam = ts.vals[:, :5].contiguous()
    w_t = torch.linalg.eigh(T)[1].flip(2)[:, :, :5]                                       # eigenvectors of T (fp64)
    rhs = v.double() - (v.double() @ w_t) @ w_t.transpose(1, 2)                            # rhs orthogonal to the leading 5
    sol = ops.tridiag_shifted_solve(ts, lam, rhs.float().contiguous()).double()
    assert torch.isfinite(sol).all()
    sol = sol - (sol @ w_t) @ w_t.transpose(1, 2)
    ev = torch.linalg.eigvalsh(T).flip(1)
    for z in range(2):
        for t in range(5):
            pinv = torch.linalg.pinv(T[z] - ev[z, t] * torch.eye(n, device=dev, dtype=torch.float64), rtol=1e-9)
            ref = pinv @ rhs[z, t]
            ref = ref - w_t[z] @ (w_t[z].T @ ref)
            assert _rel(sol[z, t], ref) < 5e-3, (z, t, float(_rel(sol[z, t], ref)))


def test_centered_grams_fold_means_from_projection_epilogue(dev):
    """Teacher chain as the selector runs it: projection with row-tile column sums, then [uncentred / M, centred]
    Grams whose means are folded inside the Gram kernel."""
    from basd_amd import ops
    g = torch.Generator().manual_seed(17)
    B, n_t, K, N = 37, 49, 256, 384
    t = (torch.randn(B, K, n_t, generator=g) + 0.4).to(dev).transpose(1, 2)       # channel-major teacher
    p = torch.randn(N, K, generator=g).to(dev) / 16
    M = B * n_t
    sums = torch.empty((1, (M + 127) // 128, N), device=dev)
    z = ops.gemm_nt(t, p, col_sums=sums[0])
    zd = t.double().reshape(-1, K) @ p.double().T
    assert _rel(z, zd) < 2e-6
    assert _rel(sums.sum(dim=(0, 1)), zd.sum(0)) < 2e-6
    out, _ = ops.centered_grams([z, z], scales=[1.0 / M, 1.0], fold=(sums, 1))
    c = zd - zd.mean(0)
    assert _rel(out[0], zd.T @ zd / M) < 5e-6
    assert _rel(out[1], c.T @ c) < 5e-6


@pytest.mark.parametrize("soft", [False, True])
@pytest.mark.parametrize("eps", [0.0, 0.1])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_cross_entropy_matches_torch(dev, soft, eps, dtype):
    """The one-launch cross entropy (loss + logits gradient) against torch.nn.CrossEntropyLoss: hard labels (with an
    ignored row), soft (mixup-style) targets, label smoothing, fp32 / bf16 logits."""
    from basd_amd import losses
    g = torch.Generator().manual_seed(3)
    B, C = 37, 1000
    logits = (torch.randn(B, C, generator=g) * 3).to(dev).to(dtype)
    if soft:
        targets = torch.softmax(torch.randn(B, C, generator=g) * 2, dim=1).to(dev)
    else:
        targets = torch.randint(0, C, (B,), generator=g).to(dev)
        targets[5] = -100
    crit = torch.nn.CrossEntropyLoss(label_smoothing=eps)
    a = logits.clone().requires_grad_(True)
    b = logits.clone().float().requires_grad_(True)
    la = losses._base_loss(crit, a, targets)
    lb = crit(b, targets)
    assert type(la.grad_fn).__name__.startswith("_FusedCrossEntropy")
    tol = 1e-5 if dtype == torch.float32 else 4e-3
    assert abs(la.item() - lb.item()) <= tol * abs(lb.item())
    (la * 1.7).backward()
    (lb * 1.7).backward()
    ref = b.grad
    assert ((a.grad.float() - ref).abs().max() / ref.abs().max()).item() < (1e-5 if dtype == torch.float32 else 1e-2)


def test_fused_cross_entropy_edge_cases(dev):
    """What nn.CrossEntropyLoss does at the edges: soft targets that do not sum to 1 (no normalisation is assumed),
    every row ignored (0 / 0 = NaN), and a class index outside [0, C) (torch device-asserts; here the loss is NaN and
    nothing is read out of bounds)."""
    from basd_amd import losses
    g = torch.Generator().manual_seed(4)
    B, C = 9, 50
    logits = torch.randn(B, C, generator=g).to(dev)
    for eps in (0.0, 0.1):
        crit = torch.nn.CrossEntropyLoss(label_smoothing=eps)
        soft = (torch.rand(B, C, generator=g) * 0.1).to(dev)            # row sums ~2.5, all different
        a = logits.clone().requires_grad_(True)
        b = logits.clone().requires_grad_(True)
        la, lb = losses._base_loss(crit, a, soft), crit(b, soft)
        assert type(la.grad_fn).__name__.startswith("_FusedCrossEntropy")
        assert abs(la.item() - lb.item()) <= 1e-5 * abs(lb.item())
        la.backward()
        lb.backward()
        assert ((a.grad - b.grad).abs().max() / b.grad.abs().max()).item() < 1e-5
        ignored = torch.full((B,), -100, dtype=torch.int64, device=dev)
        assert math.isnan(losses._base_loss(crit, logits, ignored).item()) and math.isnan(crit(logits, ignored).item())
        bad = torch.randint(0, C, (B,), generator=g).to(dev)
        bad[3] = C + 5
        a = logits.clone().requires_grad_(True)
        out = losses._base_loss(crit, a, bad)
        assert math.isnan(out.item())
        out.backward()
        assert not a.grad[3].any()


@pytest.mark.parametrize("side_s, side_t, d_s, E, B, dtype", [
    (14, 7, 384, 2, 3, torch.float32),      # cfg-2's core: 49 tokens, 13 rows of H per wave
    (14, 7, 200, 1, 2, torch.bfloat16),     # last slab partly outside D; bf16 rows
    (4, 4, 64, 1, 2, torch.float32),        # no interpolation, 16 tokens (4 rows per wave)
    (8, 8, 104, 2, 2, torch.float32),       # 64 tokens: the largest core the kernel takes
    (6, 3, 64, 1, 2, torch.float32),        # 9 tokens: waves 2 and 3 recompute rows of their neighbours
    (10, 5, 48, 3, 2, torch.float32),       # 25 tokens
])
def test_student_grad_fused_forms_h_in_the_kernel(dev, side_s, side_t, d_s, E, B, dtype):
    """basd_student_grad_fused (H = K' A' formed per workgroup, never stored) against an fp64 evaluation of
    dX = g_e (2/B) w_s ((x_s - mu) - interp(K' A')_s) from the forward's own K', A', mu, omega (autograd of
    relational.py:36-50), and against the two-launch form."""
    from basd_amd import ops
    g = torch.Generator().manual_seed(side_s * 1000 + side_t * 10 + E)
    n_s, n_t, d_t, H = side_s * side_s, side_t * side_t, 96, 2
    students = [torch.randn(B, n_s, d_s, generator=g).to(dev).to(dtype) for _ in range(E)]
    teacher = torch.randn(B, n_t, d_t, generator=g).to(dev)
    attn = torch.rand(B, H, n_t, n_t, generator=g).to(dev)
    ctx = ops.procrustes_forward(students, [teacher], [attn], torch.ones(E, 1, device=dev), False)
    grad_layers = torch.rand(E, generator=g).to(dev) + 0.5
    grads = ops.procrustes_student_grads(students, ctx, grad_layers)
    n = min(n_s, n_t)
    assert n <= 64
    tp = ops.taps(n, n_s, dev)
    Hm = torch.matmul(ctx.k_prime.double(), ctx.a_prime.double())                   # (E, B, n, d)
    if tp is not None:
        lam = tp.lam.double()[None, None, :, None]
        tgt = (1 - lam) * Hm[:, :, tp.tap0.long()] + lam * Hm[:, :, tp.tap1.long()]
    else:
        tgt = Hm
    x = torch.stack([s.double() for s in students])
    om = ctx.omega.double().expand(E, B, n_s)
    ref = grad_layers.double()[:, None, None, None] * (2.0 / B) * om[..., None] * ((x - ctx.mu_s.double()[:, :, None]) - tgt)
    for e in range(E):
        assert _rel(grads[e], ref[e]) < 1e-5, (e, _rel(grads[e], ref[e]))
    # the two-launch form (GEMM + gradient kernel) gives the same tensor up to fp32 rounding of H
    h = torch.matmul(ctx.k_prime, ctx.a_prime).contiguous()
    dx2 = torch.empty((E, B, n_s, d_s), device=dev, dtype=torch.float32)
    xs = [s.contiguous() for s in students]
    t0, t1, lm = (tp.tap0.data_ptr(), tp.tap1.data_ptr(), tp.lam.data_ptr()) if tp else (None, None, None)
    ops._lib.call("basd_student_grad_multi", ops._ptr_table(xs).data_ptr(), ops._dtype_code(xs[0]), xs[0].stride(0),
                  xs[0].stride(1), E, B, n_s, n, d_s, ctx.omega.data_ptr(), 0, ctx.mu_s.data_ptr(), h.data_ptr(), t0, t1,
                  lm, grad_layers.data_ptr(), 2.0 / B, dx2.data_ptr(), None, None, ops._stream())
    torch.cuda.synchronize()
    assert _rel(torch.stack(grads), dx2) < 1e-5


@pytest.mark.parametrize("n_t, n, D, L, G, layout, dtype", [
    (196, 196, 100, 5, 4, "row", torch.float32),        # ViT teacher: CLS-sliced rows, last slab partly outside D
    (49, 49, 64, 3, 2, "channel", torch.float32),       # channel-major (CNN feature maps)
    (256, 64, 48, 4, 3, "row", torch.bfloat16),         # finer teacher grid gathered to 64 tokens, bf16
])
def test_teacher_center_multi_matches_per_group(dev, n_t, n, D, L, G, layout, dtype):
    """basd_teacher_center_multi (all groups of mixing weights in one pass over the teacher layers) against
    basd_teacher_center called once per group: same mixed, centred tokens and weighted means
    (layer_selector.py:110-111 + relational.py:37,39)."""
    from basd_amd import ops
    g = torch.Generator().manual_seed(n_t + D)
    B = 3
    if layout == "row":
        full = [torch.randn(B, n_t + 1, D, generator=g).to(dev).to(dtype) for _ in range(L)]
        toks = [f[:, 1:, :] for f in full]
    else:
        full = [torch.randn(B, D, n_t, generator=g).to(dev).to(dtype) for _ in range(L)]
        toks = [f.transpose(1, 2) for f in full]
    mix = torch.softmax(torch.randn(G, L, generator=g), dim=-1).to(dev)
    omega_t = torch.softmax(torch.randn(G, B, n, generator=g), dim=-1).to(dev)
    gt = ops.taps(n_t, n, dev) if n_t != n else None
    g0, g1, gl = (gt.tap0.data_ptr(), gt.tap1.data_ptr(), gt.lam.data_ptr()) if gt else (None, None, None)
    tab = ops._ptr_table(toks)
    sb, sn, sd = toks[0].stride()
    mu1, tc1 = torch.empty(G, B, D, device=dev), torch.empty(G, B, n, D, device=dev)
    mu2, tc2 = torch.empty_like(mu1), torch.empty_like(tc1)
    for k in range(G):
        ops._lib.call("basd_teacher_center", tab.data_ptr(), ops._dtype_code(toks[0]), mix[k].data_ptr(), L, sb, sn, sd,
                      B, n, D, g0, g1, gl, omega_t[k].data_ptr(), mu1[k].data_ptr(), tc1[k].data_ptr(), ops._stream())
    ops._lib.call("basd_teacher_center_multi", tab.data_ptr(), ops._dtype_code(toks[0]), mix.data_ptr(), L, G, sb, sn,
                  sd, B, n, D, g0, g1, gl, omega_t.data_ptr(), mu2.data_ptr(), tc2.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    assert _rel(mu2, mu1) < 1e-5 and _rel(tc2, tc1) < 1e-5, (_rel(mu2, mu1), _rel(tc2, tc1))
    if layout == "row":      # the streaming form (no LDS tile, per-chunk column sums folded in a second pass)
        mu3, tc3 = torch.empty_like(mu1), torch.empty_like(tc1)
        scratch = torch.empty((ops._lib.query("basd_teacher_center_stream_scratch_floats", G, B, n, D),), device=dev)
        ops._lib.call("basd_teacher_center_stream", tab.data_ptr(), ops._dtype_code(toks[0]), mix.data_ptr(), L, G, sb,
                      sn, sd, B, n, D, g0, g1, gl, omega_t.data_ptr(), mu3.data_ptr(), tc3.data_ptr(),
                      scratch.data_ptr(), ops._stream())
        torch.cuda.synchronize()
        assert _rel(mu3, mu1) < 1e-5 and _rel(tc3, tc1) < 1e-5, (_rel(mu3, mu1), _rel(tc3, tc1))


@pytest.mark.parametrize("n, batch", [(144, 3), (196, 2), (130, 2), (177, 2), (190, 2)])
def test_jacobi_two_pass_invariants(dev, n, batch):
    """basd_jacobi_stacked_twopass (top n x n solved in LDS with a rotation log, the log replayed on the riding rows):
    singular values, orthogonality of the top half, and the right-orthogonal invariants that tie the riding rows to
    the same rotations -- the checks of test_jacobi_lds_invariants."""
    from basd_amd import ops, _lib
    g = torch.Generator().manual_seed(n)
    w0 = torch.randn(batch, n, 2 * n, generator=g)
    w0[:, :, :n] *= torch.logspace(0, -3, n).view(1, n, 1)
    W = w0.clone().to(dev)
    nbytes = _lib.query("basd_jacobi_twopass_workspace_bytes", n, batch, ops.MAX_SWEEPS)
    assert nbytes > 0
    ws = torch.empty((nbytes // 8 + 1,), device=dev, dtype=torch.int64)
    sigma = torch.empty((batch, n), device=dev)
    sweeps = torch.zeros((batch,), device=dev, dtype=torch.int32)
    _lib.call("basd_jacobi_stacked_twopass", W.data_ptr(), 2 * n * n, n, batch, sigma.data_ptr(), n, ops.MAX_SWEEPS, 0.0,
              ws.data_ptr(), sweeps.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    assert 0 < int(sweeps.max()) < ops.MAX_SWEEPS, "did not converge"
    top0, bot0 = w0[:, :, :n].double(), w0[:, :, n:].double()
    top, bot = W[:, :, :n].double().cpu(), W[:, :, n:].double().cpu()
    sv = torch.linalg.svdvals(top0)
    got = sigma.double().cpu().sort(dim=1, descending=True).values
    assert ((got - sv).abs().max(dim=1).values / sv[:, 0]).max() < 3e-6
    gram = top @ top.transpose(1, 2)
    off = gram - torch.diag_embed(torch.diagonal(gram, dim1=1, dim2=2))
    nrm = torch.diagonal(gram, dim1=1, dim2=2).sqrt()
    live = (nrm > 1e-6 * nrm.amax(dim=1, keepdim=True)).double()
    cos = off.abs() / (nrm.unsqueeze(2) * nrm.unsqueeze(1)).clamp_min(1e-30) * live.unsqueeze(2) * live.unsqueeze(1)
    assert cos.max().item() < 5e-6, cos.max().item()
    assert _rel(bot.transpose(1, 2) @ bot, bot0.transpose(1, 2) @ bot0) < 1e-5
    assert _rel(top.transpose(1, 2) @ bot, top0.transpose(1, 2) @ bot0) < 1e-5
    assert _lib.query("basd_jacobi_twopass_workspace_bytes", 49, 1024, ops.MAX_SWEEPS) == 0     # fits the one-kernel solver
    assert _lib.query("basd_jacobi_twopass_workspace_bytes", 576, 4, ops.MAX_SWEEPS) == 0      # past LDS: block solver


@pytest.mark.parametrize("n_s, n_t", [(576, 144), (196, 196)])
def test_procrustes_two_pass_svd_matches_block_solver(dev, n_s, n_t, monkeypatch):
    """The Procrustes forward with the two-pass SVD against the block solver on the same cores (cfg-5's 144 tokens,
    cfg-4's 196): per-sample terms and K'."""
    from basd_amd import ops, synth
    g = torch.Generator().manual_seed(n_s + n_t)
    B, d_s, d_t = 3, 256, 320
    s = synth.structured(g, B, n_s, d_s, 16).to(dev)
    t = synth.structured(g, B, n_t, d_t, 12).to(dev)
    attn = torch.softmax(torch.randn(B, 2, n_t, n_t, generator=g), dim=-1).to(dev)
    mix = torch.ones(1, 1, device=dev)
    a = ops.procrustes_forward([s], [t], [attn], mix, False, want_sweeps=True)
    monkeypatch.setattr(ops, "TWO_PASS_SVD", False)
    b = ops.procrustes_forward([s], [t], [attn], mix, False, want_sweeps=True)
    torch.cuda.synchronize()
    assert int(a.sweeps.max()) < ops.MAX_SWEEPS
    assert _rel(a.nuc, b.nuc) < 2e-6 and _rel(a.loss_b, b.loss_b) < 1e-5
    assert _rel(a.k_prime, b.k_prime) < 2e-5, _rel(a.k_prime, b.k_prime)


@pytest.mark.parametrize("n_t, n, D, L, E, dtype", [
    (196, 196, 96, 24, 4, torch.float32),      # cfg-4's layer count: three passes of 8 layers
    (100, 100, 40, 5, 3, torch.float32),       # last chunk of 4 rows, partial layer group
    (256, 64, 48, 3, 2, torch.bfloat16),       # finer teacher grid gathered to 64 tokens, bf16 rows
])
def test_mix_grad_tokens_onepass_matches_per_layer(dev, n_t, n, D, L, E, dtype):
    """basd_mix_grad_tokens_onepass (all extraction layers, one pass over the teacher layers, per-chunk sums folded in
    a fixed order) against basd_mix_grad_tokens and an fp64 evaluation of <R[e][b], That_l[b]>
    (layer_selector.py:111)."""
    from basd_amd import ops, _lib
    g = torch.Generator().manual_seed(n_t * 3 + L)
    B = 3
    full = [torch.randn(B, n_t + 1, D, generator=g).to(dev).to(dtype) for _ in range(L)]
    toks = [f[:, 1:, :] for f in full]
    R = torch.randn(E, B, n, D, generator=g).to(dev)
    gt = ops.taps(n_t, n, dev) if n_t != n else None
    g0, g1, gl = (gt.tap0.data_ptr(), gt.tap1.data_ptr(), gt.lam.data_ptr()) if gt else (None, None, None)
    tab = ops._ptr_table(toks)
    sb, sn, sd = toks[0].stride()
    p1 = torch.empty(E, B, L, device=dev)
    p2 = torch.empty_like(p1)
    _lib.call("basd_mix_grad_tokens", R.data_ptr(), tab.data_ptr(), ops._dtype_code(toks[0]), L, sb, sn, sd, E, B, n, D,
              g0, g1, gl, p1.data_ptr(), ops._stream())
    scratch = torch.empty((_lib.query("basd_mix_grad_tokens_scratch_floats", E, B, L, n),), device=dev)
    _lib.call("basd_mix_grad_tokens_onepass", R.data_ptr(), tab.data_ptr(), ops._dtype_code(toks[0]), L, sb, sn, sd, E, B,
              n, D, g0, g1, gl, p2.data_ptr(), scratch.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    T = torch.stack([t.double() for t in toks])                                   # (L, B, n_t, D)
    if gt is not None:
        lam = gt.lam.double()[None, None, :, None]
        T = (1 - lam) * T[:, :, gt.tap0.long()] + lam * T[:, :, gt.tap1.long()]
    ref = torch.einsum("ebjd,lbjd->ebl", R.double(), T)
    assert _rel(p2, ref) < 1e-5 and _rel(p1, ref) < 1e-5, (_rel(p2, ref), _rel(p1, ref))


@pytest.mark.parametrize("n_s, n_t, d_s, d_t, E, B", [
    (196, 49, 384, 512, 2, 3),       # cfg-2's cores
    (576, 144, 256, 320, 1, 3),      # cfg-5's: one plain LDS solve instead of the two-pass SVD
    (64, 64, 192, 256, 2, 2),        # no interpolation
    (100, 100, 48, 64, 1, 2),        # rank-deficient cores (47 < 100): truncated singular values
])
def test_transposed_cores_match_the_stacked_route(dev, n_s, n_t, d_s, d_t, E, B):
    """The route without riding rows (Jacobi on M^T leaves V Sigma; Y = L_b V formed with K') against the stacked cores
    [M; L_b] on the same inputs: per-sample terms, K' and the student gradients (relational.py:36-50 and autograd)."""
    from basd_amd import ops, synth, _lib
    g = torch.Generator().manual_seed(n_s + 7 * n_t)
    students = [synth.structured(g, B, n_s, d_s, 16).to(dev) for _ in range(E)]
    t = synth.structured(g, B, n_t, d_t, 12).to(dev)
    attn = torch.softmax(torch.randn(B, 2, n_t, n_t, generator=g), dim=-1).to(dev)
    mix = torch.ones(E, 1, device=dev)
    gl = torch.rand(E, generator=g).to(dev) + 0.5
    out = {}
    try:
        for mode in (0, 2):
            _lib.call("basd_procrustes_tuning", mode)
            pc = ops.procrustes_forward(students, [t], [attn], mix, False, want_sweeps=True, grad_layers=gl)
            torch.cuda.synchronize()
            out[mode] = pc
    finally:
        _lib.call("basd_procrustes_tuning", 1)
    a, b = out[2], out[0]
    assert 0 < int(a.sweeps.max()) < ops.MAX_SWEEPS
    assert _rel(a.nuc, b.nuc) < 2e-6 and _rel(a.loss_b, b.loss_b) < 1e-5
    # K' carries 1 / sigma: both routes have fp32 round-off there (1e-5; 1e-4 with singular values near the truncation
    # threshold), so each is held to the fp64 gradient rather than to the other
    assert _rel(a.k_prime, b.k_prime) < 3e-4, _rel(a.k_prime, b.k_prime)
    from oracle import basd_oracle as O
    w = O.token_weights(attn.cpu(), False, n_s).double()
    t_al = O.resample_tokens(t.cpu(), n_s).double()
    w3 = w.unsqueeze(-1)
    t_w = w3.sqrt() * (t_al - (w3 * t_al).sum(1, keepdim=True))
    for e in range(E):
        s64 = students[e].double().cpu().requires_grad_(True)
        s_w = w3.sqrt() * (s64 - (w3 * s64).sum(1, keepdim=True))
        ref = (s_w.square().sum((1, 2)) + t_w.square().sum((1, 2))
               - 2 * torch.linalg.svdvals(torch.bmm(s_w.transpose(1, 2), t_w)).sum(-1)).mean() * gl[e].item()
        ref.backward()
        for pc in (a, b):
            err = ((pc.dx[e].double().cpu() - s64.grad).norm() / s64.grad.norm()).item()
            assert err < 1e-4, (e, err)
