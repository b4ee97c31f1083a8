"""Samplers (drop-in for src/samplers.py:6-44).

``sample_multivariate_normal`` colours standard normals by the spectral factor Q sqrt(max(Lambda, 0)) of the covariance,
as the reference does.  It is setup / prediction code (the per-step Langevin noise is generated inside the step kernels,
see basis/); the eigh is torch.linalg.eigh on the device the covariance lives on, exactly as `torch.linalg.eigh(cov)` of
the reference (samplers.py:27) resolves; the normals come from libplship's generator on the device (or, on request, from
the reference's host stream), and the product is libplship's contraction."""
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

#: ... except for SMALL matrices: under "auto" an n x n matrix with n <= this goes to host LAPACK wherever it lives.  The
#: reference's own benchmark builds bases of 10 .. 100 inducing points (experiments/profiler/config.yaml:6-10), where the
#: device call is all latency: rocSOLVER 0.6 ms at n = 10 and 2.0 ms at n = 60 .. 100 against 0.02 / 0.2 .. 0.6 ms on one
#: host thread plus a 0.05 ms round trip (profiles/r05_profiler_grid.txt).  Also the reference's CPU gauge, for free.
EIGH_HOST_BELOW = 128


def resolve_eigh_device(requested: str | None, matrix: torch.Tensor) -> str:
    """'cpu' or 'cuda' for an eigh of `matrix`: the explicit request, else DEFAULT_EIGH_DEVICE; 'auto' = where it lives,
    small matrices (EIGH_HOST_BELOW) on the host"""
    where = requested or DEFAULT_EIGH_DEVICE
    assert where in ("auto", "cpu", "cuda"), "eigh_device must be 'auto', 'cpu' or 'cuda'"
    if where == "auto":
        where = "cuda" if (matrix.is_cuda and matrix.shape[-1] > EIGH_HOST_BELOW) else "cpu"
    return where


class one_host_thread:
    """Context: torch's intra-op pool narrowed to one thread.  A small LAPACK call (an eigh of a 50 x 50 matrix: 0.2 ms) is
    pure overhead for a thread pool -- 130 ms measured with 8 threads on an 8-core container, 98 ms with the 128 threads
    torch assumes on a GPU box whose cgroup grants 16 cores."""

    def __enter__(self):
        self.prev = torch.get_num_threads()
        if self.prev != 1:
            torch.set_num_threads(1)
        return self

    def __exit__(self, *exc):
        if self.prev != 1:
            torch.set_num_threads(self.prev)
        return False


def host_eigh(matrix: torch.Tensor):
    """torch.linalg.eigh on the host; small matrices (n <= 256) on one thread"""
    m = matrix.cpu()
    if m.shape[-1] <= 256:
        with one_host_thread():
            return torch.linalg.eigh(m)
    return torch.linalg.eigh(m)


#: where the standard normals that sample_multivariate_normal colours come from.
#:   "auto" (default): "reference" for a run that is not J-sharded, "device" as soon as it is (``j_offset`` != 0, or
#:       torch.distributed initialised with more than one rank).  The drop-in contract comes first: after ``set_seed(0)``
#:       ``sample_multivariate_normal(zeros(2), eye(2), size=(2,), seed=0)`` returns the reference's pinned draws
#:       (tests/test_samplers.py:19-26 of the reference), and a sharded run -- which the reference does not have -- gets
#:       the stream that makes its result independent of the GPU count.
#:   "reference": torch.normal on the host generator and a host -> device copy, the reference's stream, sample for sample
#:       (samplers.py:30-40); 0.3 s of host time for 2 000 test points x 8 192 particles.
#:   "device": libplship's counter-based generator on the GPU (pls_normal_fill: Philox4x32-10 + Box-Muller), keyed
#:       by ``seed`` -- or, with ``seed=None``, by ONE 63-bit draw from torch's global CPU generator, so that the
#:       reference's reproducibility contract (set_seed before a run) holds -- and by the GLOBAL particle column, so the
#:       ranks of a J-sharded prediction draw different columns of one matrix and the result does not depend on the GPU
#:       count.  (With the host stream every rank seeded alike would draw the SAME normals for different particles.)
#:       Set ``samplers.DEFAULT_NORMAL_STREAM = "device"`` for single-GPU runs that predict on thousands of points.
#: Same law either way: N(mean, Q max(Lambda, 0) Q^T).
DEFAULT_NORMAL_STREAM = "auto"


def resolve_normal_stream(requested: str | None, j_offset: int = 0) -> str:
    """'device' or 'reference': the explicit request, else DEFAULT_NORMAL_STREAM; 'auto' = 'device' only for a J-sharded run"""
    stream = requested or DEFAULT_NORMAL_STREAM
    assert stream in ("auto", "device", "reference"), "normal_stream must be 'auto', 'device' or 'reference'"
    if stream != "auto":
        return stream
    if j_offset:
        return "device"
    import torch.distributed as dist

    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    return "device" if sharded else "reference"


def spectral_factor(cov: torch.Tensor, eigh_device: str | None = None) -> torch.Tensor:
    """(Q sqrt(max(Lambda, 0)))^T of the symmetric ``cov`` as an (n, n) device matrix: the k-major operand that colours
    standard normals (samplers.py:27-28, :37-44).  The eigendecomposition is not optional: the reference's predictive
    covariances are INDEFINITE (orthonormal.py:186-204 mixes r over Z u x with the spectrum of k(Z,Z)/M: a hundred negative
    eigenvalues at M = 128, N* = 300, tests/test_oracle_goldens.py), its law is the one of the clipped spectrum, and no
    Cholesky factor -- with whatever jitter -- has that law."""
    where = resolve_eigh_device(eigh_device, cov)
    c64 = cov.detach().to(torch.float64)
    eigenvalues, eigenvectors = host_eigh(c64) if where == "cpu" else torch.linalg.eigh(_dev(c64))  # samplers.py:27
    eigenvalues = torch.clip(eigenvalues, 0, None)
    # (Q sqrt(Lambda))^T stored k-major: L[k][i] = Q[i][k] * sqrt(lam_k)
    return _dev((eigenvectors * torch.sqrt(eigenvalues)[None, :]).T)


def standard_normals(n: int, size: Tuple[int], seed: int | None = None, normal_stream: str | None = None,
                     j_offset: int = 0) -> torch.Tensor:
    """(n, prod(size)) standard normals on the device from the chosen stream (DEFAULT_NORMAL_STREAM)."""
    stream = resolve_normal_stream(normal_stream, j_offset)
    j = 1
    for v in size:
        j *= int(v)
    if stream == "reference":
        generator = torch.Generator().manual_seed(seed) if seed is not None else None
        normal_sample = torch.normal(mean=0.0, std=1.0, size=(n, *size), generator=generator)  # samplers.py:30-35
        return _dev(normal_sample.reshape(n, j))
    key = int(seed) if seed is not None else int(torch.randint(0, 2**63 - 1, (1,)).item())
    xi = torch.empty((n, j), dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
    if n and j:
        L.check(L.load().pls_normal_fill(xi.data_ptr(), j, n, j, key & (2**64 - 1), 0, int(j_offset), L.stream_ptr()),
                "pls_normal_fill")
    return xi


def colour(factor_t: torch.Tensor, xi: torch.Tensor) -> torch.Tensor:
    """(Q sqrt(Lambda)) xi on the device (pls_gemm_tn)."""
    n, j = xi.shape
    out = torch.empty((n, j), dtype=torch.float64, device=xi.device)
    if n and j:
        L.check(
            L.load().pls_gemm_tn(factor_t.data_ptr(), L.ld(factor_t), xi.data_ptr(), L.ld(xi), out.data_ptr(), j, n, j, n, 1.0, 0.0,
                                 L.stream_ptr()),
            "pls_gemm_tn",
        )
    return out


def sample_multivariate_normal(
    mean: torch.Tensor,
    cov: torch.Tensor,
    size: Tuple[int] | None = None,
    seed: int | None = None,
    eigh_device: str | None = None,
    normal_stream: str | None = None,
    j_offset: int = 0,
    factor: torch.Tensor | None = None,
) -> torch.Tensor:
    """samplers.py:6-44.  Returns a (size..., n) float64 device tensor.  ``factor`` (extension): spectral_factor(cov) computed
    earlier -- repeated predictions at the same test points pay the eigh once; ``j_offset``: global index of the first
    sample (device stream)."""
    size = (1,) if not size else size
    lt = spectral_factor(cov, eigh_device) if factor is None else factor
    n = lt.shape[0]
    xi = standard_normals(n, size, seed=seed, normal_stream=normal_stream, j_offset=j_offset)
    out = colour(lt, xi)
    out = out + _dev(mean)[:, None]
    return out.reshape(n, *size).movedim(0, -1) if len(size) > 1 else out.T
