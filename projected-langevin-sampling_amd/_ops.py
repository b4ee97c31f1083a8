"""Small wrappers over libplship's primitive entry points, shared by the prediction-side code."""
from __future__ import annotations

import torch

from . import _lib as L


def gemm_tn(lm: torch.Tensor, rm: torch.Tensor, alpha: float = 1.0, beta: float = 0.0, out: torch.Tensor | None = None) -> torch.Tensor:
    """out (I x J) = alpha * lm^T rm + beta * out with lm (K x I), rm (K x J) row-major device float64 (pls_gemm_tn)."""
    L.require_gpu_tensor(lm, "lm")
    L.require_gpu_tensor(rm, "rm")
    lm = lm if lm.stride(1) == 1 else lm.contiguous()
    rm = rm if rm.stride(1) == 1 else rm.contiguous()
    k, i = lm.shape
    k2, j = rm.shape
    assert k == k2, f"contraction mismatch {lm.shape} vs {rm.shape}"
    if out is None:
        assert beta == 0.0
        out = torch.empty((i, j), dtype=torch.float64, device=lm.device)
    assert out.shape == (i, j) and (out.stride(1) == 1 or j <= 1)
    L.check(
        L.load().pls_gemm_tn(lm.data_ptr(), L.ld(lm), rm.data_ptr(), L.ld(rm), out.data_ptr(), L.ld(out), i, j, k, float(alpha),
                             float(beta), L.stream_ptr()),
        "pls_gemm_tn",
    )
    return out


def row_power_sums(s: torch.Tensor, power: int, shift: torch.Tensor | None = None) -> torch.Tensor:
    """(rows,) vector of sum_j (s[r, j] - shift[r])^power (pls_row_power_sums; fixed summation order)."""
    L.require_gpu_tensor(s, "samples")
    s = s if s.stride(1) == 1 else s.contiguous()
    rows, cols = s.shape
    out = torch.empty(rows, dtype=torch.float64, device=s.device)
    sh = None if shift is None else L.require_gpu_tensor(shift, "shift").contiguous()
    L.check(
        L.load().pls_row_power_sums(s.data_ptr(), L.ld(s), rows, cols, L.ptr(sh), int(power), out.data_ptr(), L.stream_ptr()),
        "pls_row_power_sums",
    )
    return out


def row_quantiles(s: torch.Tensor, qs) -> torch.Tensor:
    """(rows, len(qs)) quantiles over dim 1 with torch.quantile's linear interpolation (pls_row_quantiles: one LDS sort per
    row up to 16384 samples, radix selection of the two order statistics beyond -- a calibration split above 16384 points,
    the gathered samples of a J-sharded run)."""
    L.require_gpu_tensor(s, "samples")
    s = s if s.stride(1) == 1 else s.contiguous()
    rows, cols = s.shape
    q = torch.as_tensor(list(qs), dtype=torch.float64).to(s.device)
    out = torch.empty((rows, q.numel()), dtype=torch.float64, device=s.device)
    L.check(
        L.load().pls_row_quantiles(s.data_ptr(), L.ld(s), rows, cols, q.data_ptr(), q.numel(), out.data_ptr(), q.numel(),
                                   L.stream_ptr()),
        "pls_row_quantiles",
    )
    return out


def block_means(e: torch.Tensor, block_cols: int | None = None, out: torch.Tensor | None = None, out_ptr: int | None = None) -> torch.Tensor | None:
    """Means of consecutive blocks of ``block_cols`` entries of the per-particle vector ``e`` (pls_block_means; the whole
    vector when block_cols is None): the ``.mean()`` of orthonormal.py:126 / inducing_point.py:115 as a fixed-order
    libplship reduction.  ``out``: device vector to fill; ``out_ptr``: raw address instead (pinned host memory mapped into
    the device: the value is read on the host after an event, with no copy kernel)."""
    L.require_gpu_tensor(e, "energies")
    e = e.contiguous()
    j = e.numel()
    bc = j if block_cols is None else int(block_cols)
    nb = (j + bc - 1) // bc if j else 0
    if out_ptr is None:
        if out is None:
            out = torch.empty(nb, dtype=torch.float64, device=e.device)
        assert out.numel() >= nb and out.is_contiguous()
        out_ptr = out.data_ptr()
    if j:
        L.check(L.load().pls_block_means(e.data_ptr(), j, max(bc, 1), out_ptr, L.stream_ptr()), "pls_block_means")
    return out
