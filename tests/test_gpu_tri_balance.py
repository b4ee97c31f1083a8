"""GPU: balanced triangular products on few output tiles (csrc/gemm_tn_f64_kg.h, gemm_tn_f64_kg_tri_kernel).

The solves of the inducing-point basis (gpytorch.solve at inducing_point.py:89-93, :130-137) run here as triangular
products with L_c^-1, and the noise colouring / un-whitening as products with L_c.  On the narrow particle shard of an
8-GPU run every 64 x 64 output tile has a workgroup and a CU of its own, so an unbalanced launch lasts as long as the
heaviest tile row; the balanced kernel pairs tile rows and lets two workgroups share each pair, the heavy tile's two
partial sums meeting through a scratch slot.  Held here: the product against torch's fp64 `@` on the host for every
branch (even / odd tile-row counts, ragged last tiles, a middle row that pairs with itself, both triangle kinds), repeated
launches on one scratch (the flag words come back zero), equality with the unbalanced kernel to rounding, untouched guard
bands, and the inducing-point step through it against the CPU oracle.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pls_oracle as O
from test_gpu_parity import cu, make_problem, relerr


@pytest.fixture(scope="module")
def P():
    import projected_langevin_sampling_amd as pkg
    from projected_langevin_sampling_amd import basis, costs, link_functions

    assert torch.cuda.is_available(), "these tests need the MI355X"
    pkg._lib.load()

    class NS:
        pass

    ns = NS()
    ns.pkg, ns.basis, ns.costs, ns.links = pkg, basis, costs, link_functions
    return ns


@pytest.fixture(autouse=True)
def _f64_default():
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(prev)


class balance:
    """with balance(P, 0 / 1): PLS_OPT_TRI_BALANCE for the block"""

    def __init__(self, P, value):
        self.L, self.lib, self.value = P.pkg._lib, P.pkg._lib.load(), value

    def __enter__(self):
        self.prev = self.lib.pls_get_option(self.L.OPT_TRI_BALANCE)
        self.L.check(self.lib.pls_set_option(self.L.OPT_TRI_BALANCE, self.value))

    def __exit__(self, *exc):
        self.L.check(self.lib.pls_set_option(self.L.OPT_TRI_BALANCE, self.prev))
        return False


def _factor_like(P, m, seed):
    """A CholDesc whose `inverse factor` is an arbitrary lower-triangular T (the product kernels do not care that the real
    one is L_c^-1): pls_chol_forward_solve computes T U (operand T^T, k <= row), pls_chol_solve_ws T^T (T U) (operand T,
    k >= row)."""
    from projected_langevin_sampling_amd.basis.base import alloc_matrix

    L, lib = P.pkg._lib, P.pkg._lib.load()
    g = torch.Generator().manual_seed(seed)
    t = torch.tril(torch.randn(m, m, generator=g))
    linv, linvt = alloc_matrix(m, m, "cuda"), alloc_matrix(m, m, "cuda")
    linv.copy_(t)
    linvt.copy_(t.T)
    nbytes = int(lib.pls_tri_scratch_bytes(m, 4096))
    scratch = torch.zeros((nbytes + 7) // 8, dtype=torch.float64, device="cuda")
    d = L.CholDesc()
    d.m = m
    d.Lc, d.ldlc, d.LcT, d.ldlct = linv.data_ptr(), L.ld(linv), linvt.data_ptr(), L.ld(linvt)  # (unused by the products)
    d.Linv, d.ldlinv, d.LinvT, d.ldlinvt = linv.data_ptr(), L.ld(linv), linvt.data_ptr(), L.ld(linvt)
    d.tri_scratch, d.tri_scratch_bytes = scratch.data_ptr(), scratch.numel() * 8
    return t, d, scratch, (linv, linvt)


def _aligned(host: torch.Tensor) -> torch.Tensor:
    """device copy with the library's padded leading dimension (the few-tiles kernels need 16-byte aligned operand rows)"""
    from projected_langevin_sampling_amd.basis.base import alloc_matrix

    out = alloc_matrix(host.shape[0], host.shape[1], "cuda")
    out.copy_(host)
    return out


def _products(P, d, u, guard=3):
    """(T U, T^T T U) through the C ABI, outputs embedded in NaN guard bands that must come back untouched."""
    L, lib = P.pkg._lib, P.pkg._lib.load()
    m, j = u.shape
    ldo = j + guard
    y = torch.full((m + 2 * guard, ldo), float("nan"), device="cuda")
    v = torch.full((m + 2 * guard, ldo), float("nan"), device="cuda")
    ws = torch.empty(m * j, device="cuda")
    yv, vv = y[guard:guard + m], v[guard:guard + m]
    L.check(lib.pls_chol_forward_solve(d, u.data_ptr(), L.ld(u), j, yv.data_ptr(), ldo, L.stream_ptr()), "forward")
    L.check(lib.pls_chol_solve_ws(d, u.data_ptr(), L.ld(u), j, vv.data_ptr(), ldo, ws.data_ptr(), ws.numel() * 8, L.stream_ptr()),
            "solve_ws")
    torch.cuda.synchronize()
    for out in (y, v):
        assert torch.isnan(out[:guard]).all() and torch.isnan(out[guard + m:]).all() and torch.isnan(out[guard:guard + m, j:]).all()
    return yv[:, :j].clone(), vv[:, :j].clone()


# (M, J): two tile rows .. sixteen; odd counts (a middle row that pairs with itself); ragged last tile rows and columns;
# a single column; M a little above one tile (the second row holds one line)
SHAPES = [(128, 64), (65, 100), (130, 1), (192, 192), (200, 130), (320, 64), (511, 77), (512, 256), (1000, 200),
          (1024, 1024), (1024, 1000), (1088, 320), (2048, 1024), (4096, 512)]


@pytest.mark.parametrize("m,j", SHAPES)
def test_balanced_products_against_the_host(P, m, j):
    t, d, scratch, keep = _factor_like(P, m, seed=m + j)
    g = torch.Generator().manual_seed(7 * m + j)
    u_host = torch.randn(m, j, generator=g)
    u = _aligned(u_host)
    want_y = t @ u_host
    want_v = t.T @ want_y
    flags = scratch.view(torch.int32)[:4096]
    with balance(P, 1):
        for rep in range(3):  # the same scratch again and again: the finisher leaves the flag words zero
            y, v = _products(P, d, u)
            assert relerr(y, want_y) < 1e-13 and relerr(v, want_v) < 1e-13, (m, j, rep)
            assert int(flags.abs().sum()) == 0, "a flag word was left set"
    used = bool((scratch.view(torch.int64)[2048:] != 0).any())
    assert used, "the balanced kernel did not run (no partial sum was ever written)"
    with balance(P, 0):
        y0, v0 = _products(P, d, u)
    assert relerr(y, y0) < 1e-14 and relerr(v, v0) < 1e-14


def test_balanced_products_are_reproducible_and_survive_a_dirty_slot(P):
    """Two launches give the same bits whoever finishes a tile (a + b == b + a), and partial-sum slots full of NaN from
    an earlier life of the scratch are never read before they are written."""
    m, j = 1024, 1024
    t, d, scratch, keep = _factor_like(P, m, seed=5)
    u = _aligned(torch.randn(m, j, generator=torch.Generator().manual_seed(6)))
    with balance(P, 1):
        a = _products(P, d, u)
        scratch.view(torch.int64)[2048:] = -1  # all-ones = NaN in every partial slot; the flag words stay zero
        b = _products(P, d, u)
        c = _products(P, d, u)
    assert all(torch.equal(x, y) for x, y in zip(a, b)) and all(torch.equal(x, y) for x, y in zip(a, c))


def test_scratch_too_small_or_missing_falls_back(P):
    m, j = 512, 512
    t, d, scratch, keep = _factor_like(P, m, seed=9)
    u_host = torch.randn(m, j, generator=torch.Generator().manual_seed(10))
    want = t @ u_host
    d.tri_scratch_bytes = 20000  # flags + not even one slot
    y, _ = _products(P, d, _aligned(u_host))
    assert relerr(y, want) < 1e-13 and not bool((scratch.view(torch.int64)[2048:] != 0).any())
    d.tri_scratch, d.tri_scratch_bytes = None, 0
    y, _ = _products(P, d, _aligned(u_host))
    assert relerr(y, want) < 1e-13


@pytest.mark.parametrize("j", [1024, 2048, 700])
def test_inducing_point_step_on_a_narrow_shard_against_the_oracle(P, j):
    """The per-call step of the drop-in loop (`particles += pls.calculate_particle_update(...)`, trainers.py:153-157) at the
    shard size of an 8- / 4-GPU run, M = 1024 (the problem of test_mid_size_ipb_step_against_the_oracle): forward solve and
    product with L_c are balanced triangular products.  Against the CPU oracle with injected noise, and against the
    unbalanced kernels to rounding."""
    n, m, d = 8000, 1024, 8
    pr = make_problem(n, m, j, d, seed=777)
    ok, gk = O.RBFARDKernel(pr["ls"], 1.3), P.pkg.ARDKernel(pr["ls"], 1.3)
    yz = pr["y"][:m]
    oi = O.InducingPointBasis(ok, pr["z"], yz, pr["x"])
    gi = P.basis.InducingPointBasis(P.pkg.PLSKernel(gk, pr["z"]), pr["z"], yz, pr["x"])
    oc = O.GaussianCost(0.3, pr["y"], O.IdentityLink())
    gc = P.costs.GaussianCost(0.3, pr["y"], P.links.IdentityLinkFunction())
    e = torch.randn(m, j, generator=pr["gen"])
    want = O.PLS(oi, oc).calculate_particle_update(pr["u"].clone(), 1e-4, noise=e)
    L, lib = P.pkg._lib, P.pkg._lib.load()
    results = {}
    for operator in (1, 0):  # 1: dS = -eta (P U - ct) + noise straight from U, then Lc dS; 0: forward solve, Q S, Lc dS
        L.check(lib.pls_set_option(L.OPT_IPB_STEP_OPERATOR, operator))
        try:
            for bal in (1, 0):
                with balance(P, bal):
                    results[operator, bal] = P.pkg.PLS(gi, gc).calculate_particle_update(cu(pr["u"]), 1e-4, noise=cu(e))
        finally:
            L.check(lib.pls_set_option(L.OPT_IPB_STEP_OPERATOR, 1))
    for key, got in results.items():
        assert relerr(got, want) < 1e-8, (key, relerr(got, want))
    assert relerr(results[1, 1], results[1, 0]) < 1e-12 and relerr(results[0, 1], results[0, 0]) < 1e-12
    assert relerr(results[1, 1], results[0, 1]) < 1e-10  # the folded operator against solve-then-multiply
    sc = gi._chol.tri_scratch()
    assert int(sc.view(torch.int32)[:4096].abs().sum()) == 0 and bool((sc.view(torch.int64)[2048:] != 0).any())


def test_stale_flag_words_are_repaired_and_streams_do_not_share_a_scratch(P):
    """The balanced products meet through flag words that must be zero on entry.  A factor keeps one scratch per stream (an
    eager call beside a graph replay on another stream must not share flags); poisoned flags -- what a launch that died half
    way leaves -- make a later product wrong, and the repair the error path of every solve applies (reset_tri_scratch)
    puts it right."""
    from projected_langevin_sampling_amd._chol import cholesky_factor

    g = torch.Generator().manual_seed(31)
    m, j = 512, 192
    a = torch.randn(m, m, generator=g)
    k = cu(a @ a.T / m + torch.eye(m))
    f = cholesky_factor(k).build_inverse()
    u = _aligned(torch.randn(m, j, generator=g))
    with balance(P, 1):
        want = f.solve(u)
        assert relerr(k @ want, u) < 1e-10
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            other = f.solve(u)
        torch.cuda.synchronize()
        assert torch.equal(other, want) and len(f._tri_scratch) == 2
        assert len({sc.data_ptr() for sc in f._tri_scratch.values()}) == 2
        mine = f.tri_scratch()
        flags = mine.view(torch.int32)[: f.TRI_FLAG_BYTES // 4]
        assert int(flags.abs().sum()) == 0
        flags.fill_(1)  # every pair already "has a partial sum waiting": the first arriver adds garbage instead of publishing
        bad = f.solve(u)
        torch.cuda.synchronize()
        assert not torch.equal(bad, want), "poisoned flags are expected to corrupt the product (else this test tests nothing)"
        f.reset_tri_scratch()
        assert int(flags.abs().sum()) == 0
        assert torch.equal(f.solve(u), want)
        # the error path: a refused call zeroes the flags before it re-raises
        flags.fill_(1)
        with pytest.raises(P.pkg._lib.PlsHipError):
            f._check(1, "a failing launch")
        assert int(flags.abs().sum()) == 0
