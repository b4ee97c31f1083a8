"""ONE spectrum per job for the orthonormal basis (host logic around orthonormal.py:46-68 of the reference).

The reference is one process with one `torch.linalg.eigh(k(Z,Z) / M)` call; particles are coordinates in THAT
eigenvector gauge.  A J-sharded run has one process per GPU, and an eigendecomposition is only defined up to the sign of
every eigenvector (and a rotation inside a cluster of equal eigenvalues): ranks that factorise on their own may keep
different counts M_k at a threshold that cuts through rounding-level eigenvalues, or hold coordinates in different gauges
that `gather_particles` / `save_pls` would then mix silently.  So:

* ``shared_spectrum``: OPT-IN (``group=True`` for the default process group, or a ProcessGroup).  Rank ``src`` runs the
  eigh and broadcasts (lambda, V) -- M (M + 1) doubles, 8 MB at M = 1024 -- and every rank keeps the SAME bits, whatever
  its own Gram matrix rounded to.  Before a rank adopts another rank's spectrum the group checks that all ranks hold the
  same matrix up to rounding (M and two weighted sums of k(Z,Z)/M, one fixed-size all-reduce): ranks that were handed
  different inducing points or hyper-parameters get a RuntimeError instead of the eigenvectors of somebody else's Gram
  matrix.  ``group=None`` (the default) is the reference's behaviour: a local eigh and no collective at all, so a basis
  built by ONE rank of a running job (a rank-0 evaluation, a per-rank sweep) never waits for peers that do not come;
* ``canonicalise_signs``: the largest-magnitude component of every eigenvector is made positive (ties: the first), so
  that two solvers which agree on the eigenvectors up to sign give the same matrix.  Applied to the device (rocSOLVER)
  route; the host LAPACK route keeps LAPACK's raw signs, the gauge the reference's CPU path and its goldens have;
* ``spectrum_fingerprint`` / ``compare_fingerprints``: what a checkpoint records about the gauge its particles are
  coordinates in, and the check a resume makes against the basis it was handed (experiments/loaders.py:10-28 restores the
  particles into whatever basis the caller rebuilt)."""
from __future__ import annotations

import hashlib
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def canonicalise_signs(eigenvectors: torch.Tensor) -> torch.Tensor:
    """Columns scaled by +-1 so that each one's largest-|.| component (first one on ties) is positive."""
    if eigenvectors.numel() == 0:
        return eigenvectors
    pivot = eigenvectors.abs().argmax(dim=0)  # first maximal index per column
    s = torch.sign(eigenvectors.gather(0, pivot[None, :]))[0]
    s = torch.where(s == 0, torch.ones_like(s), s)
    return eigenvectors * s[None, :]


def _resolve_group(group):
    """(active, process group or None for the default one).  None / False: no collective (the caller's own eigh); True: the
    default process group; anything else: that ProcessGroup."""
    if group is None or group is False:
        return False, None
    pg = None if group is True else group
    active = dist.is_available() and dist.is_initialized() and dist.get_world_size(pg) > 1
    return active, pg


def _dist_active(group) -> bool:
    return _resolve_group(group)[0]


def assert_same_matrix(gram_scaled: torch.Tensor, group, rtol: float = 1e-9) -> None:
    """Collective: every rank of ``group`` holds the same k(Z,Z) / M up to rounding.  One all-reduce of a fixed-size vector
    (so ranks with different M meet in it instead of hanging in a broadcast of mismatched shapes): M, the plain sum and a
    sum weighted by an incommensurate probe.  Raises on EVERY rank when any two disagree."""
    active, pg = _resolve_group(group)
    if not active:
        return
    g = gram_scaled.detach().to(torch.float64)
    m = g.shape[0]
    w = _probe_vector(m).to(g.device)
    sig = torch.stack([torch.tensor(float(m), dtype=torch.float64, device=g.device), g.sum(), (w @ g @ w),
                       g.abs().sum()]).cpu()
    dev = _collective_device(pg, gram_scaled)
    both = torch.cat([sig, -sig]).to(dev)
    dist.all_reduce(both, op=dist.ReduceOp.MAX, group=pg)
    both = both.cpu()
    hi, lo = both[:4], -both[4:]
    if float(hi[0]) != float(lo[0]):
        raise RuntimeError(f"orthonormal basis: the ranks of the group hold different numbers of inducing points "
                           f"({int(lo[0])} .. {int(hi[0])}, this rank {m}); a shared spectrum needs ONE k(Z,Z) -- build with "
                           f"group=None for per-rank bases")
    scale = float(hi[3]) if float(hi[3]) > 0 else 1.0
    if float((hi[1:3] - lo[1:3]).abs().max()) > rtol * scale:
        raise RuntimeError("orthonormal basis: the ranks of the group hold different k(Z,Z) / M (other inducing points or "
                           "kernel hyper-parameters); a shared spectrum needs ONE matrix -- build with group=None for "
                           "per-rank bases")


def _collective_device(group, like: torch.Tensor) -> torch.device:
    """Tensors of a collective live on the GPU under RCCL ("nccl") and on the host under gloo."""
    backend = str(dist.get_backend(group)).lower()
    if "nccl" in backend:
        return like.device if like.is_cuda else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def shared_spectrum(gram_scaled: torch.Tensor, eigh_where: str, group=None, src: int = 0,
                    canonical_signs: Optional[bool] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(eigenvalues, eigenvectors) of ``gram_scaled`` = k(Z,Z) / M as float64 CPU tensors (ascending, like torch.linalg.eigh).
    ``group`` None: this process's own eigh, no collective.  ``group`` True (default process group) or a ProcessGroup:
    identical on every rank of the group, after a check that the ranks hold the same matrix (assert_same_matrix).
    ``eigh_where``: "cpu" (host LAPACK) or "cuda" (the device ``gram_scaled`` lives on).  ``src`` is a rank of ``group``.
    ``canonical_signs`` None = True for the device route, False for the host route."""
    assert eigh_where in ("cpu", "cuda")
    m = gram_scaled.shape[0]
    active, group = _resolve_group(group)
    if active:
        assert_same_matrix(gram_scaled, group if group is not None else True)
    rank = dist.get_rank(group) if active else 0
    if canonical_signs is None:
        canonical_signs = eigh_where == "cuda"
    packed = None
    if not active or rank == src:
        g = gram_scaled.detach().to(torch.float64)
        from ..samplers import host_eigh

        lam, vec = host_eigh(g) if eigh_where == "cpu" else torch.linalg.eigh(g)  # orthonormal.py:46-48
        lam, vec = lam.cpu(), vec.cpu()
        if canonical_signs:
            vec = canonicalise_signs(vec)
        packed = torch.cat([lam.reshape(1, m), vec], dim=0).contiguous()  # (M + 1, M)
    if active:
        dev = _collective_device(group, gram_scaled)
        buf = packed.to(dev) if packed is not None else torch.empty((m + 1, m), dtype=torch.float64, device=dev)
        dist.broadcast(buf, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
        packed = buf.cpu()
    return packed[0].clone(), packed[1:].clone()


def assert_same_count(mk: int, group=None) -> None:
    """Every rank of a J-sharded run must keep the same number of eigen-directions (orthonormal.py:52-60): the particle
    matrices are concatenated along J (gather_particles) and reduced row by row (predictive_moments)."""
    active, group = _resolve_group(group)
    if not active:
        return
    dev = _collective_device(group, torch.empty(0))
    lo = torch.tensor([mk, -mk], dtype=torch.int64, device=dev)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    lo = lo.cpu()
    if int(lo[0]) != mk or int(-lo[1]) != mk:
        raise RuntimeError(f"orthonormal basis: ranks disagree on the number of kept eigenvalues ({int(lo[0])} .. {int(-lo[1])}, "
                           f"this rank {mk}); pass one spectrum= to every rank")


def _probe_vector(m: int) -> torch.Tensor:
    i = torch.arange(m, dtype=torch.float64)
    return torch.sin(1.0 + i * 2.0**0.5) + 0.25 * torch.cos(0.3 + i * 3.0**0.5)


def spectrum_fingerprint(eigenvalues: torch.Tensor, eigenvectors: torch.Tensor) -> dict:
    """What identifies the gauge of an orthonormal basis: the kept eigenvalues, the projections V^T p of one fixed probe
    vector p (they flip with an eigenvector's sign and move with a rotation inside a cluster) and a hash of the exact bits.
    Small (2 M_k doubles), plain data, unknown to the reference's loader, which ignores extra keys."""
    lam = eigenvalues.detach().cpu().to(torch.float64).contiguous()
    vec = eigenvectors.detach().cpu().to(torch.float64).contiguous()
    h = hashlib.sha256()
    h.update(lam.numpy().tobytes())
    h.update(vec.numpy().tobytes())
    return {
        "m": int(vec.shape[0]),
        "mk": int(lam.shape[0]),
        "eigenvalues": lam.clone(),
        "probe": (vec.T @ _probe_vector(vec.shape[0])).contiguous(),
        "sha256": h.hexdigest(),
    }


def compare_fingerprints(saved: dict, current: dict, rtol: float = 1e-8) -> Optional[str]:
    """None if particles saved under ``saved`` are coordinates in the basis ``current`` describes (same bits, or the same
    eigenvectors to rounding); otherwise a sentence saying what differs."""
    if saved.get("sha256") is not None and saved.get("sha256") == current.get("sha256"):
        return None
    if int(saved["m"]) != int(current["m"]) or int(saved["mk"]) != int(current["mk"]):
        return (f"the checkpoint was written with {saved['mk']} of {saved['m']} eigen-directions, the basis has "
                f"{current['mk']} of {current['m']}")
    ls, lc = saved["eigenvalues"].double(), current["eigenvalues"].double()
    scale = float(lc.abs().max().clamp_min(1e-300)) if lc.numel() else 1.0
    if lc.numel() and float((ls - lc).abs().max()) > rtol * scale:
        return "the eigenvalues differ (another kernel, inducing set or threshold)"
    ps, pc = saved["probe"].double(), current["probe"].double()
    pscale = float(pc.abs().max().clamp_min(1e-300)) if pc.numel() else 1.0
    bad = (ps - pc).abs() > 1e-6 * pscale
    if bool(bad.any()):
        flipped = bool(((ps + pc).abs() <= 1e-6 * pscale)[bad].all())
        n = int(bad.sum())
        return (f"{n} eigenvector(s) have the opposite sign" if flipped else
                f"{n} eigenvector(s) differ by more than a sign (a rotation inside a cluster of equal eigenvalues)") + \
            ": the particles are coordinates in another eigenvector gauge"
    return None
