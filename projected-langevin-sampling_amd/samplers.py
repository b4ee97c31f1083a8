"""Samplers (drop-in for src/samplers.py:6-44).

``sample_multivariate_normal`` keeps the reference's stream: torch.normal on the CPU generator, coloured by
eigh(cov).  It is setup / prediction code (the per-step Langevin noise uses the in-kernel Philox stream
instead, see basis/); the eigh is torch.linalg.eigh on the device the covariance lives on, exactly as
`torch.linalg.eigh(cov)` of the reference (samplers.py:27) resolves, and the product is done by libplship on
the device."""
from __future__ import annotations

from typing import Tuple

import torch

from . import _lib as L
from .kernel import _dev


#: where the eigendecompositions of setup and prediction run (k(Z,Z)/M of the orthonormal basis, the covariances of the
#: sampler): "auto" = on the device the matrix lives on, which is what the reference's `torch.linalg.eigh(matrix)` does;
#: "cpu" = host LAPACK (the reference's CPU path: its eigenvector gauge, so coordinates and draws compare one to one with a
#: CPU run -- the parity tests pin this); "cuda" = the GPU.  An (M_k + N*)-sized eigh of the predictive sampler takes 8 s on
#: the host share of a GPU box for 2 000 test points and 0.05 s on the device; k(Z,Z)/M at M = 4096 takes 21 s against
#: 0.15 s (tools/eigh_probe.py, tools/predict_probe.py).  Every choice gives a valid factor Q sqrt(Lambda) of the same
#: matrix; the SAMPLE differs (another eigenvector gauge).
DEFAULT_EIGH_DEVICE = "auto"


def resolve_eigh_device(requested: str | None, matrix: torch.Tensor) -> str:
    """'cpu' or 'cuda' for an eigh of `matrix`: the explicit request, else DEFAULT_EIGH_DEVICE, 'auto' = where it lives"""
    where = requested or DEFAULT_EIGH_DEVICE
    assert where in ("auto", "cpu", "cuda"), "eigh_device must be 'auto', 'cpu' or 'cuda'"
    if where == "auto":
        where = "cuda" if matrix.is_cuda else "cpu"
    return where


def sample_multivariate_normal(
    mean: torch.Tensor,
    cov: torch.Tensor,
    size: Tuple[int] | None = None,
    seed: int | None = None,
    eigh_device: str | None = None,
) -> torch.Tensor:
    """samplers.py:6-44.  Returns a (size..., n) float64 device tensor."""
    generator = torch.Generator().manual_seed(seed) if seed is not None else None
    size = (1,) if not size else size
    where = resolve_eigh_device(eigh_device, cov)
    c64 = cov.detach().to(torch.float64)
    eigenvalues, eigenvectors = torch.linalg.eigh(c64.cpu() if where == "cpu" else _dev(c64))  # samplers.py:27
    eigenvalues = torch.clip(eigenvalues, 0, None)
    n = eigenvalues.shape[0]
    normal_sample = torch.normal(mean=0.0, std=1.0, size=(n, *size), generator=generator)  # samplers.py:30-35
    j = int(normal_sample.numel() // n)
    xi = _dev(normal_sample.reshape(n, j))
    # (Q sqrt(Lambda))^T stored k-major: L[k][i] = Q[i][k] * sqrt(lam_k)
    lt = _dev((eigenvectors * torch.sqrt(eigenvalues)[None, :]).T)
    out = torch.empty((n, j), dtype=torch.float64, device=xi.device)
    L.check(
        L.load().pls_gemm_tn(lt.data_ptr(), n, xi.data_ptr(), j, out.data_ptr(), j, n, j, n, 1.0, 0.0, L.stream_ptr()),
        "pls_gemm_tn",
    )
    out = out + _dev(mean)[:, None]
    return out.reshape(n, *size).movedim(0, -1) if len(size) > 1 else out.T
