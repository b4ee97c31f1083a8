"""GPU: the k-split 64 x 64 contraction kernel (csrc/gemm_tn_f64_kg.h) that narrow particle shards take.

A rank of an 8-GPU run owns J / 8 particle columns (the columns are independent: orthonormal.py:151-158,
inducing_point.py:143-149), so the M_k x M_k x J products of the Gaussian step have too few 128 x 128 tiles to fill
256 CUs.  These tests hold that kernel to the fp64 product computed by torch on the host (the oracle's own `@`), to
the round-2 kernels it replaces, and to the CPU oracle's step; and they extend the J-shard invariance check to
J / G = 1024 and 2048 at M_k = 1024.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pls_oracle as O
from test_gpu_parity import FUZZ_SEED, TOL, build_onb, cu, make_problem, relerr


@pytest.fixture(scope="module")
def P():
    import projected_langevin_sampling_amd as pkg
    from projected_langevin_sampling_amd import basis, costs, distributed, link_functions

    assert torch.cuda.is_available(), "these tests need the MI355X"
    pkg._lib.load()

    class NS:
        pass

    ns = NS()
    ns.pkg, ns.basis, ns.costs, ns.links, ns.dist = pkg, basis, costs, link_functions, distributed
    return ns


@pytest.fixture(autouse=True)
def _f64_default():
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(prev)


class ksplit:
    """with ksplit(P, mode[, max_tiles]): ...  -- PLS_OPT_KSPLIT_MODE / _MAX_TILES for the block, restored after"""

    def __init__(self, P, mode, max_tiles=256):
        self.L, self.lib = P.pkg._lib, P.pkg._lib.load()
        self.mode, self.max_tiles = mode, max_tiles

    def __enter__(self):
        L, lib = self.L, self.lib
        self.prev = (lib.pls_get_option(L.OPT_KSPLIT_MODE), lib.pls_get_option(L.OPT_KSPLIT_MAX_TILES))
        L.check(lib.pls_set_option(L.OPT_KSPLIT_MODE, self.mode))
        L.check(lib.pls_set_option(L.OPT_KSPLIT_MAX_TILES, self.max_tiles))

    def __exit__(self, *exc):
        L, lib = self.L, self.lib
        L.check(lib.pls_set_option(L.OPT_KSPLIT_MODE, self.prev[0]))
        L.check(lib.pls_set_option(L.OPT_KSPLIT_MAX_TILES, self.prev[1]))
        return False


def gemm_tn(P, l, r):
    from projected_langevin_sampling_amd import _ops

    return _ops.gemm_tn(l, r)


# every branch of the k-loop: K below one super-step (tail only), exact multiples, one / two DMA steps + tail, odd
# remainders; I and J on and off the 64-tile grid, odd sizes (a pair that straddles the edge), a single column
SHAPES = [
    (64, 64, 32), (64, 64, 16), (64, 64, 7), (64, 64, 1), (64, 64, 33), (64, 64, 48), (64, 64, 64), (64, 64, 65),
    (64, 64, 96), (64, 64, 100), (128, 192, 257), (100, 70, 130), (33, 129, 95), (1, 64, 40), (65, 1, 40),
    (200, 1000, 200), (1024, 1024, 1024), (1000, 1536, 1000),
]


@pytest.mark.parametrize("mode", [2, 3])
def test_ksplit_contraction_matches_the_host_product(P, mode):
    g = torch.Generator().manual_seed(7 + FUZZ_SEED)
    for (i, j, k) in SHAPES:
        l = torch.randn(k, i, generator=g)
        r = torch.randn(k, j, generator=g)
        want = l.T @ r
        with ksplit(P, mode):
            got = gemm_tn(P, cu(l), cu(r))
        with ksplit(P, 0):
            old = gemm_tn(P, cu(l), cu(r))
        scale = (l.abs().T @ r.abs()).max().item()
        err = (got.cpu() - want).abs().max().item() / scale
        assert err < 4e-16 * max(4, k) ** 0.5, f"mode {mode} shape {(i, j, k)}: {err:.2e}"
        assert relerr(got, old) < 1e-13, f"mode {mode} shape {(i, j, k)} vs the round-2 kernel"


@pytest.mark.parametrize("mode", [2, 3])
def test_ksplit_triangular_product(P, mode):
    """e = Lc xi with the transposed factor as the k-major operand (pls_tri_multiply): only k <= row is contracted"""
    from projected_langevin_sampling_amd import _lib as L

    lib = L.load()
    g = torch.Generator().manual_seed(11 + FUZZ_SEED)
    for m, j in ((64, 64), (100, 96), (257, 130), (1024, 512)):
        lc = torch.tril(torch.randn(m, m, generator=g))
        x = torch.randn(m, j, generator=g)
        lct = cu(lc.T.contiguous())
        xg, out = cu(x), torch.empty(m, j, device="cuda")
        with ksplit(P, mode):
            L.check(lib.pls_tri_multiply(lct.data_ptr(), L.ld(lct), m, xg.data_ptr(), L.ld(xg), j, out.data_ptr(), L.ld(out),
                                         L.stream_ptr()))
        assert relerr(out, lc @ x) < 1e-13, (m, j)


@pytest.mark.parametrize("mode", [2, 3])
def test_ksplit_fast_path_step_and_energy_against_the_oracle(P, mode):
    """Gaussian/identity fast path (B U with the Langevin update in the epilogue) through the k-split kernel: step,
    in-place form, energy by-product and the stand-alone energy against the CPU oracle with injected noise."""
    for (n, m, j, seed) in ((400, 24, 48, 3), (900, 70, 130, 5), (1500, 200, 333, 9)):
        pr = make_problem(n, m, j, 3, seed=seed + FUZZ_SEED)
        ob, gb = build_onb(P, pr)
        mk = ob.approximation_dimension
        oc = O.GaussianCost(0.3, pr["y"], O.IdentityLink())
        gc = P.costs.GaussianCost(0.3, pr["y"], P.links.IdentityLinkFunction())
        u = pr["u"][:mk].contiguous()
        xi = torch.randn(mk, j, generator=pr["gen"])
        want = O.PLS(ob, oc).calculate_particle_update(u.clone(), 1e-3, noise=xi)
        e_want = O.PLS(ob, oc).calculate_energy_potential(u)  # the mean over the particles (orthonormal.py:110-126)
        with ksplit(P, mode):
            e_in = torch.empty(j, device="cuda")
            got = gb.fused_step(gc, cu(u), 1e-3, noise=P.basis.NoiseSpec(injected=cu(xi)), input_energy=e_in)
            new = gb.fused_step(gc, cu(u), 1e-3, noise=P.basis.NoiseSpec(injected=cu(xi)), new_state=True)
            e_sep = gb.fused_particle_energy(gc, cu(u))
        with ksplit(P, 0):
            e_old = gb.fused_particle_energy(gc, cu(u))
        assert relerr(got, want) < TOL, (n, m, j)
        assert relerr(new, u + want) < TOL
        assert relerr(e_in, e_sep) < 1e-11 and relerr(e_sep, e_old) < 1e-11
        assert abs(e_sep.mean().item() - e_want) < TOL * abs(e_want)


def _big_gaussian_basis(P, mk=1024, n=3000, seed=0):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(mk, n, generator=g) / mk ** 0.5
    lam = torch.rand(mk, generator=g) + 0.5
    basis = P.basis.OrthonormalBasis.from_projection(cu(a), cu(lam))
    y = torch.randn(n, generator=g)
    cost = P.costs.GaussianCost(0.5, y, P.links.IdentityLinkFunction())
    return basis, cost, g


def test_narrow_shards_reproduce_the_full_run_at_rank_1024(P):
    """J-shard invariance at the sizes an 8- and a 4-GPU run of configs[1] hand to a rank (M_k = 1024, J / G = 1024 and
    2048): the shard's step -- k-split kernel -- is the same numbers as its columns of the full J = 8192 launch
    (128 x 128 tiles), Philox noise included (global column counters), and so are the energies."""
    basis, cost, g = _big_gaussian_basis(P)
    j = 8192
    u = cu(torch.randn(1024, j, generator=g))
    e_full = torch.empty(j, device="cuda")
    full = basis.fused_step(cost, u, 1e-3, noise=P.basis.NoiseSpec(seed=5, step=2), input_energy=e_full)
    for world in (8, 4):
        w = j // world
        for rank in (0, world - 1, 3):
            j0 = rank * w
            e_sh = torch.empty(w, device="cuda")
            shard = basis.fused_step(cost, u[:, j0:j0 + w].contiguous(), 1e-3,
                                     noise=P.basis.NoiseSpec(seed=5, step=2, j_offset=j0), input_energy=e_sh)
            assert relerr(shard, full[:, j0:j0 + w]) < 1e-13, (world, rank)
            assert relerr(e_sh, e_full[j0:j0 + w]) < 1e-12, (world, rank)
    # and the two kernels agree on the same shard
    sh = u[:, :1024].contiguous()
    with ksplit(P, 0):
        old = basis.fused_step(cost, sh, 1e-3, noise=P.basis.NoiseSpec(seed=5, step=2))
    with ksplit(P, 1):
        new = basis.fused_step(cost, sh, 1e-3, noise=P.basis.NoiseSpec(seed=5, step=2))
    assert relerr(new, old) < 1e-13


def test_ksplit_kernel_on_the_full_width_launch(P):
    """MAX_TILES above the tile count sends the full J = 8192 launch through the k-split kernel too (the A/B of DESIGN
    section 8): same step."""
    basis, cost, g = _big_gaussian_basis(P, seed=1)
    u = cu(torch.randn(1024, 4096, generator=g))
    with ksplit(P, 0):
        old = basis.fused_step(cost, u, 1e-3, noise=P.basis.NoiseSpec(seed=9, step=1))
    with ksplit(P, 1, max_tiles=1 << 30):
        new = basis.fused_step(cost, u, 1e-3, noise=P.basis.NoiseSpec(seed=9, step=1))
    assert relerr(new, old) < 1e-13


def test_noise_drawn_in_front_of_the_k_loop_is_the_same_noise(P):
    """PLS_OPT_KG_NOISE_PREGEN: the k-split kernel draws the Philox pairs of its output block while its first operand rows
    travel.  Bit for bit the step of the in-epilogue generator: tiles on and off the 64 grid, a k range below one super-step,
    shard offsets, per-block step sizes with a Philox column restart, the energy by-product, in-place and delta forms."""
    from projected_langevin_sampling_amd import _lib as L
    from projected_langevin_sampling_amd.basis.base import BlockSpec

    lib = L.load()
    g = torch.Generator().manual_seed(31 + FUZZ_SEED)
    for (mk, j, n) in ((64, 64, 300), (24, 48, 200), (100, 130, 400), (200, 333, 500), (257, 65, 600), (1024, 1024, 1500)):
        a = cu(torch.randn(mk, n, generator=g) / mk ** 0.5)
        lam = cu(torch.rand(mk, generator=g) + 0.5)
        basis = P.basis.OrthonormalBasis.from_projection(a, lam)
        cost = P.costs.GaussianCost(0.5, torch.randn(n, generator=g), P.links.IdentityLinkFunction())
        u = cu(torch.randn(mk, j, generator=g))
        bc = max(1, j // 3)
        eta = cu(torch.tensor([1e-3, 0.0, 2e-3, 5e-4][: (j + bc - 1) // bc]))

        def run(pregen):
            L.check(lib.pls_set_option(L.OPT_KG_NOISE_PREGEN, pregen))
            try:
                with ksplit(P, 2):
                    ns = P.basis.NoiseSpec(seed=77, step=5, j_offset=4096)
                    e = torch.empty(j, device="cuda")
                    d = basis.fused_step(cost, u, 1e-3, noise=ns, input_energy=e)
                    new = basis.fused_step(cost, u, 1e-3, noise=ns, new_state=True)
                    blk = basis.fused_step(cost, u, 0.0, noise=P.basis.NoiseSpec(seed=78, step=1), new_state=True,
                                           blocks=BlockSpec(bc, eta))
                return d, new, blk, e
            finally:
                L.check(lib.pls_set_option(L.OPT_KG_NOISE_PREGEN, 1))

        on, off = run(1), run(0)
        for x, y in zip(on, off):
            assert torch.equal(x, y), (mk, j)
        assert (on[0] - (on[1] - u)).abs().max().item() < 1e-12


def test_the_noise_of_a_particle_does_not_depend_on_the_tiling(P):
    """The Philox / Box-Muller code is inlined into every tiling's epilogue (128 x 128 direct, 64 x 64 through LDS, k-split with
    the noise drawn in front of the k-loop); its floating-point contractions are written out (csrc/fmath.h, philox.h), so a
    particle's noise is the same BITS whichever kernel its shard takes.  With U = 0 the contraction contributes exact zeros
    and the step is -eta (0 - c) / sigma^2 + sqrt(2 eta) xi: full-width launch against the shards of an 8-GPU run."""
    basis, cost, g = _big_gaussian_basis(P, seed=3)
    j = 8192
    u = torch.zeros(1024, j, device="cuda")
    full = basis.fused_step(cost, u, 1e-3, noise=P.basis.NoiseSpec(seed=11, step=4))
    assert full.abs().max().item() > 0.05  # (noise of standard deviation sqrt(2e-3) is in there)
    for rank in (0, 5, 7):
        j0 = rank * 1024
        sh = u[:, j0:j0 + 1024].contiguous()
        for mode in (1, 0):  # k-split kernel; the 64 x 64 tiles of round 2
            with ksplit(P, mode):
                got = basis.fused_step(cost, sh, 1e-3, noise=P.basis.NoiseSpec(seed=11, step=4, j_offset=j0))
            assert torch.equal(got, full[:, j0:j0 + 1024]), (rank, mode)
