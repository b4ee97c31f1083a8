"""Device Cholesky factor of an SPD matrix and the solves with it (pls_chol_factor / pls_chol_solve).

Replaces ``gpytorch.solve(lhs=K, input=K, rhs=U)`` of the reference (basis/inducing_point.py:89-93, :104-106, :130-132,
:235-239).  gpytorch factorises through ``psd_safe_cholesky``: plain Cholesky first, then -- if a pivot fails -- with
jitter 1e-8, 1e-7, 1e-6 added to the diagonal (float64 defaults: ``cholesky_jitter`` 1e-8, ``cholesky_max_tries`` 3),
a warning per attempt and ``NotPSDError`` after the last.  Same schedule here; the factorisation itself (blocked
right-looking, MFMA trailing updates) and the triangular solves run in libplship."""
from __future__ import annotations

import warnings

import torch

from . import _lib as L

#: gpytorch.settings.cholesky_jitter (float64) and cholesky_max_tries defaults
CHOLESKY_JITTER = 1e-8
CHOLESKY_MAX_TRIES = 3


class NotPSDError(RuntimeError):
    """The matrix is not numerically positive definite even with the largest jitter (gpytorch's exception name)."""


class CholeskyFactor:
    """K (+ jitter I) = Lc Lc^T on the device: ``Lc`` (lower), ``LcT`` (upper) and the substitution operators."""

    def __init__(self, lc: torch.Tensor, lct: torch.Tensor, sf: torch.Tensor, sb: torch.Tensor, jitter: float):
        self.Lc, self.LcT, self.Sf, self.Sb, self.jitter = lc, lct, sf, sb, jitter
        self.m = lc.shape[0]
        self.Linv = self.LinvT = None
        self._tri_scratch: dict = {}

    #: bytes at the head of a scratch that hold the flag words of the balanced products (csrc/gemm_tn_f64_kg.h, kKgTriFlagBytes)
    TRI_FLAG_BYTES = 16384

    def tri_scratch(self) -> torch.Tensor | None:
        """Scratch of the balanced triangular products (pls_chol_desc.tri_scratch): allocated once PER STREAM, zeroed, at the
        size the largest product that takes the few-tiles kernel needs (fewer than 256 tiles of 128 x 128: <= 32 MB whatever
        M is), so that its address never changes under a captured graph -- launches on one stream are ordered, an eager call
        beside a graph replay on another stream must not share flag words --; None when M has a single 64-row tile row
        (nothing to balance).  The library leaves the flag words zero after every call; after a FAILED call they are zeroed
        again (reset_tri_scratch), because stale flags would make every later product silently wrong."""
        if self.m <= 64:
            return None
        key = L.stream_ptr() if torch.cuda.is_available() else 0
        sc = self._tri_scratch.get(key)
        if sc is None:
            rows128 = (self.m + 127) // 128
            j_max = 128 * max(1, -(-256 // rows128))
            nbytes = int(L.load().pls_tri_scratch_bytes(self.m, j_max))
            sc = self._tri_scratch[key] = torch.zeros((nbytes + 7) // 8, dtype=torch.float64, device=self.Lc.device)
        return sc

    def reset_tri_scratch(self) -> None:
        """Zero the flag words of every scratch (after a failed or aborted launch)."""
        for sc in self._tri_scratch.values():
            sc[: self.TRI_FLAG_BYTES // 8].zero_()

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            self.reset_tri_scratch()
            L.check(rc, what)

    def build_inverse(self) -> "CholeskyFactor":
        """Linv = Lc^-1 and its transpose (pls_chol_build_inverse: the identity through the block forward substitution).
        With them a solve is one or two triangular PRODUCTS on the MFMA contraction -- what fills the chip when the
        right-hand side is a narrow J-shard -- instead of a substitution that is serial over the block rows."""
        if self.Linv is None:
            from .basis.base import alloc_matrix

            linv, linvt = alloc_matrix(self.m, self.m, self.Lc.device), alloc_matrix(self.m, self.m, self.Lc.device)
            L.check(L.load().pls_chol_build_inverse(self.desc(), linv.data_ptr(), L.ld(linv), linvt.data_ptr(), L.ld(linvt),
                                                    L.stream_ptr()), "pls_chol_build_inverse")
            self.Linv, self.LinvT = linv, linvt
        return self

    def desc(self) -> L.CholDesc:
        d = L.CholDesc()
        d.m = self.m
        d.Lc, d.ldlc = self.Lc.data_ptr(), L.ld(self.Lc)
        d.LcT, d.ldlct = self.LcT.data_ptr(), L.ld(self.LcT)
        d.Sf, d.ldsf = self.Sf.data_ptr(), L.ld(self.Sf)
        d.Sb, d.ldsb = self.Sb.data_ptr(), L.ld(self.Sb)
        if self.Linv is not None:
            d.Linv, d.ldlinv = self.Linv.data_ptr(), L.ld(self.Linv)
            d.LinvT, d.ldlinvt = self.LinvT.data_ptr(), L.ld(self.LinvT)
            sc = self.tri_scratch()
            if sc is not None:
                d.tri_scratch, d.tri_scratch_bytes = sc.data_ptr(), sc.numel() * 8
        return d

    def solve(self, rhs: torch.Tensor) -> torch.Tensor:
        """K^-1 rhs for a (M, J) device matrix: two triangular products with the inverse factor if build_inverse() has
        run (and PLS_OPT_SOLVE_MODE is 1), block forward + backward substitution in one launch otherwise."""
        u = L.require_gpu_tensor(rhs, "rhs")
        u = u if u.dim() == 2 and u.stride(1) == 1 else u.reshape(self.m, -1).contiguous()
        j = u.shape[1]
        v = torch.empty((self.m, j), dtype=torch.float64, device=u.device)
        if j:
            ws = torch.empty((self.m, j), dtype=torch.float64, device=u.device) if self.Linv is not None else None
            self._check(L.load().pls_chol_solve_ws(self.desc(), u.data_ptr(), L.ld(u), j, v.data_ptr(), max(j, 1), L.ptr(ws),
                                                   0 if ws is None else ws.numel() * 8, L.stream_ptr()), "pls_chol_solve_ws")
        return v

    def forward_solve(self, rhs: torch.Tensor) -> torch.Tensor:
        """Lc^-1 rhs (pls_chol_forward_solve)."""
        u = L.require_gpu_tensor(rhs, "rhs")
        u = u if u.dim() == 2 and u.stride(1) == 1 else u.reshape(self.m, -1).contiguous()
        j = u.shape[1]
        y = torch.empty((self.m, j), dtype=torch.float64, device=u.device)
        if j:
            self._check(L.load().pls_chol_forward_solve(self.desc(), u.data_ptr(), L.ld(u), j, y.data_ptr(), max(j, 1),
                                                        L.stream_ptr()), "pls_chol_forward_solve")
        return y

    def colour(self, xi: torch.Tensor) -> torch.Tensor:
        """Lc xi: standard normals (M, J) -> N(0, K) samples (triangular product)."""
        x = L.require_gpu_tensor(xi, "xi").contiguous()
        j = x.shape[1]
        out = torch.empty_like(x)
        if j:
            L.check(L.load().pls_tri_multiply(self.LcT.data_ptr(), L.ld(self.LcT), self.m, x.data_ptr(), L.ld(x), j,
                                              out.data_ptr(), L.ld(out), L.stream_ptr()), "pls_tri_multiply")
        return out


def _alloc(m: int, device):
    from .basis.base import alloc_matrix  # (imported here: basis/ imports this module)

    return [alloc_matrix(m, m, device) for _ in range(4)]


def cholesky_factor(k: torch.Tensor, jitter: float | None = None, max_tries: int | None = None) -> CholeskyFactor:
    """psd_safe_cholesky on the device (schedule in the module docstring).  ``k``: (M, M) device float64, only read."""
    k = L.require_gpu_tensor(k, "matrix")
    assert k.dim() == 2 and k.shape[0] == k.shape[1], "square matrix expected"
    k = k if k.stride(1) == 1 else k.contiguous()
    m = k.shape[0]
    if torch.isnan(k).any().item():
        raise NotPSDError("cholesky: the matrix contains NaN")
    jitter = CHOLESKY_JITTER if jitter is None else jitter
    max_tries = CHOLESKY_MAX_TRIES if max_tries is None else max_tries
    lc, lct, sf, sb = _alloc(m, k.device)
    info = torch.zeros(1, dtype=torch.int32, device=k.device)
    lib = L.load()
    attempts = [0.0] + [jitter * 10**i for i in range(max_tries)]
    for jit in attempts:
        if jit > 0.0:
            warnings.warn(f"A not p.d., added jitter of {jit:.1e} to the diagonal", RuntimeWarning, stacklevel=2)
        L.check(
            lib.pls_chol_factor(k.data_ptr(), L.ld(k), m, float(jit), lc.data_ptr(), L.ld(lc), lct.data_ptr(), L.ld(lct),
                                sf.data_ptr(), L.ld(sf), sb.data_ptr(), L.ld(sb), info.data_ptr(), L.stream_ptr()),
            "pls_chol_factor",
        )
        if int(info.item()) == 0:  # (one host sync per attempt; setup only)
            return CholeskyFactor(lc, lct, sf, sb, jit)
    raise NotPSDError(f"Matrix not positive definite after repeatedly adding jitter up to {attempts[-1]:.1e}.")


def factor_from_host(chol_lower: torch.Tensor) -> CholeskyFactor:
    """A factor computed elsewhere (e.g. the oracle's LAPACK Cholesky in a parity run): upload Lc and Lc^T, build the
    substitution operators on the device (pls_chol_build_operators)."""
    lh = torch.tril(chol_lower.detach().to(device="cpu", dtype=torch.float64))
    m = lh.shape[0]
    dev = torch.device("cuda", torch.cuda.current_device())
    lc, lct, sf, sb = _alloc(m, dev)
    lc.copy_(lh)
    lct.copy_(lh.T)
    sf.zero_()
    sb.zero_()
    L.check(
        L.load().pls_chol_build_operators(lc.data_ptr(), L.ld(lc), lct.data_ptr(), L.ld(lct), m, sf.data_ptr(), L.ld(sf),
                                          sb.data_ptr(), L.ld(sb), L.stream_ptr()),
        "pls_chol_build_operators",
    )
    return CholeskyFactor(lc, lct, sf, sb, 0.0)
