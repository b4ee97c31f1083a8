"""GPU: the back-projection D = A G for a rank that is not a multiple of the 128-row tile (csrc/gemm_tn_f64_rows.h).

Every eigenvalue threshold leaves such a rank (orthonormal.py:51-60); the launch cuts the rank into equal-height tiles
and deals their 16-row blocks to the wave rows, so the MFMA count follows cdiv(M_k, 16) and G is read once.  The step
(orthonormal.py:128-159) is held to plain torch fp64 on the host and to the round-2 launch sequence (128-row tiles plus
remainder pieces) it replaces; padding rows of the projection are poisoned, so a store or a contraction that strays
over the tile's last row shows.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from test_gpu_ksplit import P, _f64_default  # noqa: F401  (fixtures)
from test_gpu_parity import cu, relerr


class row_blocks:
    """with row_blocks(P, 0 | 1): PLS_OPT_ROW_BLOCKS for the block, restored after"""

    def __init__(self, P, mode):
        self.L, self.lib, self.mode = P.pkg._lib, P.pkg._lib.load(), mode

    def __enter__(self):
        self.prev = self.lib.pls_get_option(self.L.OPT_ROW_BLOCKS)
        self.L.check(self.lib.pls_set_option(self.L.OPT_ROW_BLOCKS, self.mode))

    def __exit__(self, *exc):
        self.L.check(self.lib.pls_set_option(self.L.OPT_ROW_BLOCKS, self.prev))
        return False


class two_gemm_path:
    """with two_gemm_path(P): (ranks 129 .. 256 always take the two-GEMM path since the wave-pair fused kernel of round 3
    left the build; kept as a no-op so that the test bodies read as before)"""

    def __init__(self, P):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _problem(mk, n, j, seed):
    gen = torch.Generator().manual_seed(seed)
    a = torch.randn(mk, n, generator=gen) / mk ** 0.5
    lam = torch.rand(mk, generator=gen) + 0.5
    u = torch.randn(mk, j, generator=gen)
    xi = torch.randn(mk, j, generator=gen)
    y = torch.randn(n, generator=gen)
    return a, lam, u, xi, y


# 129: 80 + 49 rows (blocks 3|2 and 2|2); 144, 160: equal tiles; 150: a partly filled last block; 255 / 257 / 300 / 383:
# either side of two and three tiles; 1000: the rank a threshold leaves of 1024 inducing points (8 tiles, last one 7 blocks)
@pytest.mark.parametrize("mk", [129, 144, 150, 160, 176, 192, 200, 224, 241, 255, 257, 300, 383, 1000])
def test_step_through_the_row_block_launch(P, mk):
    n, j, eta, s2 = (40000, 2048, 1e-3, 0.4) if mk < 400 else (12000, 1024, 1e-3, 0.4)
    a, lam, u, xi, y = _problem(mk, n, j, 900 + mk)
    basis = P.basis.OrthonormalBasis.from_projection(cu(a), cu(lam), poison_padding=True)
    gc = P.costs.GaussianCost(s2, y, P.links.IdentityLinkFunction())
    e_in = torch.empty(j, device="cuda")
    with two_gemm_path(P):
        with row_blocks(P, 1):
            got = basis.fused_step(gc, cu(u), eta, noise=P.basis.NoiseSpec(injected=cu(xi)), force_generic=True, input_energy=e_in)
        with row_blocks(P, 0):
            old = basis.fused_step(gc, cu(u), eta, noise=P.basis.NoiseSpec(injected=cu(xi)), force_generic=True)
    f = a.T @ u
    want = -eta * (a @ ((f - y[:, None]) / s2)) - eta * u / lam[:, None] + math.sqrt(2 * eta) * xi
    assert torch.isfinite(got).all()
    assert relerr(got, want) < 1e-11
    assert relerr(got, old) < 1e-12
    e_want = ((f - y[:, None]) ** 2).sum(0) / (2 * s2) + 0.5 * (u * u / lam[:, None]).sum(0)
    assert relerr(e_in, e_want) < 1e-11


@pytest.mark.parametrize("mk,j", [(129, 2200), (176, 1999), (208, 2049), (250, 1153), (1000, 1100)])
def test_row_block_launch_with_a_ragged_last_column_tile(P, mk, j):
    """J off the 128-column grid (odd J: the last pair of a k-row straddles the edge): lanes beyond J store nothing"""
    n, eta, s2 = (40000, 1e-3, 0.4) if mk < 400 else (12000, 1e-3, 0.4)
    a, lam, u, xi, y = _problem(mk, n, j, 1300 + mk)
    basis = P.basis.OrthonormalBasis.from_projection(cu(a), cu(lam), poison_padding=True)
    gc = P.costs.GaussianCost(s2, y, P.links.IdentityLinkFunction())
    e_in = torch.empty(j, device="cuda")
    with two_gemm_path(P):
        with row_blocks(P, 1):
            got = basis.fused_step(gc, cu(u), eta, noise=P.basis.NoiseSpec(injected=cu(xi)), force_generic=True, input_energy=e_in)
        with row_blocks(P, 0):
            old = basis.fused_step(gc, cu(u), eta, noise=P.basis.NoiseSpec(injected=cu(xi)), force_generic=True)
    f = a.T @ u
    want = -eta * (a @ ((f - y[:, None]) / s2)) - eta * u / lam[:, None] + math.sqrt(2 * eta) * xi
    assert relerr(got, want) < 1e-11
    assert relerr(got, old) < 1e-12
    e_want = ((f - y[:, None]) ** 2).sum(0) / (2 * s2) + 0.5 * (u * u / lam[:, None]).sum(0)
    assert relerr(e_in, e_want) < 1e-11


@pytest.mark.parametrize("mk", [129, 200, 300])
def test_row_block_launch_with_the_rows_streamed_in_chunks(P, mk):
    """a workspace for a third of the rows: the launch accumulates onto the slabs of the previous chunk (beta = 1)"""
    n, j, eta = 30000, 1024, 2e-3
    a, lam, u, xi, y = _problem(mk, n, j, 1700 + mk)
    basis = P.basis.OrthonormalBasis.from_projection(cu(a), cu(lam), poison_padding=True)
    lib = P.pkg._lib.load()
    basis.workspace_bytes = lib.pls_onb_step_workspace_bytes(basis._desc(), j, n // 3)
    fstar = (a.T @ u)[:, 0]
    gc = P.costs.BernoulliCost((fstar > 0).double(), P.links.SigmoidLinkFunction())
    with two_gemm_path(P):
        with row_blocks(P, 1):
            got = basis.fused_step(gc, cu(u), eta, noise=P.basis.NoiseSpec(injected=cu(xi)), force_generic=True)
        with row_blocks(P, 0):
            old = basis.fused_step(gc, cu(u), eta, noise=P.basis.NoiseSpec(injected=cu(xi)), force_generic=True)
    f = a.T @ u
    p = torch.sigmoid(f).clamp(1e-10, 1 - 1e-10)
    yy = (fstar > 0).double()[:, None]
    g = -yy * (1 - p) + (1 - yy) * p
    want = -eta * (a @ g) - eta * u / lam[:, None] + math.sqrt(2 * eta) * xi
    assert relerr(got, want) < 1e-10
    assert relerr(got, old) < 1e-12


def test_row_block_contraction_fuzz_against_the_host_product_with_guard_bands(P):
    """pls_gemm_tn takes the row-block launch for many tiles and I off the 128-row grid.  60 seeded draws of (I, J, K,
    alpha, beta, leading dimensions): the result against torch's fp64 product on the host, and the memory AROUND the
    output -- the row padding beyond J inside ldc, guard rows above and below -- must be untouched: the launch drops rows
    of the next tile or beyond I through the range of a buffer descriptor and columns >= J through a lane offset, and
    this is the check that neither leaks."""
    import numpy as np

    from projected_langevin_sampling_amd import _lib as L

    lib = L.load()
    rng = np.random.default_rng(20240 + 7)
    guard = 7.25e300
    for draw in range(60):
        i = int(rng.integers(129, 256)) if draw % 2 == 0 else int(rng.choice([300, 383, 497, 500, 620, 761, 1000, 1023, 1090]))
        j = int(rng.choice([128, 256, 1000, 1024, 1153, 2048, 2049, 3000]))
        k = int(rng.choice([1, 3, 4, 15, 16, 17, 33, 64, 100, 257]))
        while -(-i // 128) * -(-j // 128) < 256:  # (fewer tiles take the 64 x 64 kernels: not this test's subject)
            j += 1024
        alpha = float(rng.choice([1.0, -0.5, 2.25]))
        beta = float(rng.choice([0.0, 0.0, 1.0, -0.75]))
        ldl, ldr = i + (i & 1) + 2 * int(rng.integers(0, 5)), j + (j & 1) + 2 * int(rng.integers(0, 5))  # even: the DMA path
        ldc = max(j + int(rng.integers(0, 9)), 128)
        g = torch.Generator().manual_seed(5000 + draw)
        lm = torch.randn(k, ldl, generator=g)
        rm = torch.randn(k, ldr, generator=g)
        c0 = torch.randn(i, j, generator=g)
        lm[:, i:] = float("nan")  # row padding of the operands must never reach the result
        rm[:, j:] = float("nan")
        buf = torch.full((i + 6, ldc), guard)
        buf[3:3 + i, :j] = c0
        bg, lg, rg = cu(buf), cu(lm), cu(rm)
        out = bg[3:3 + i]
        L.check(lib.pls_gemm_tn(lg.data_ptr(), ldl, rg.data_ptr(), ldr, out.data_ptr(), ldc, i, j, k, alpha, beta, L.stream_ptr()))
        got = bg.cpu()
        want = alpha * (lm[:, :i].T @ rm[:, :j]) + beta * c0
        tag = f"draw {draw}: I={i} J={j} K={k} alpha={alpha} beta={beta} ldl={ldl} ldr={ldr} ldc={ldc}"
        scale = (lm[:, :i].abs().T @ rm[:, :j].abs()).max().item() + c0.abs().max().item()
        assert (got[3:3 + i, :j] - want).abs().max().item() < 1e-14 * scale * max(4, k) ** 0.5, tag
        assert (got[:3] == guard).all() and (got[3 + i:] == guard).all(), tag + ": guard rows written"
        assert (got[3:3 + i, j:] == guard).all(), tag + ": row padding written"
        with row_blocks(P, 0):
            bg2 = cu(buf)
            L.check(lib.pls_gemm_tn(lg.data_ptr(), ldl, rg.data_ptr(), ldr, bg2[3:3 + i].data_ptr(), ldc, i, j, k, alpha, beta,
                                    L.stream_ptr()))
        assert relerr(bg2[3:3 + i, :j], got[3:3 + i, :j]) < 1e-13, tag + ": vs the 128-row tiles"
