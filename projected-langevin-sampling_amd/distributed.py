"""J-sharding across the GPUs of one node.

Every particle column evolves independently given the replicated basis, so a rank owns a contiguous block of
columns and the Langevin step needs NO collective.  RCCL (torch.distributed backend "nccl") is used only where
the reference reduces over J: the mean energy (orthonormal.py:126), predictive moments (gaussian.py:49-52) and the
per-test-point quantiles of the conformal wrapper (conformalise/pls.py:36-45).
The noise counters use GLOBAL column indices (basis.j_offset), so 1/2/4/8-GPU runs give identical particles."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_bounds(number_of_particles: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Columns [j0, j1) owned by ``rank``: contiguous, sizes differ by at most one, earlier ranks get the extras."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of size {world_size}")
    base, extra = divmod(number_of_particles, world_size)
    j0 = rank * base + min(rank, extra)
    return j0, j0 + base + (1 if rank < extra else 0)


def shard_particles(particles: torch.Tensor, rank: int, world_size: int) -> torch.Tensor:
    """This rank's columns of a full (M, J) particle matrix (contiguous copy)."""
    j0, j1 = shard_bounds(particles.shape[1], rank, world_size)
    return particles[:, j0:j1].contiguous()


def attach_shard(basis, number_of_particles: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Tell ``basis`` which global columns its particles are, so its noise stream is GPU-count invariant."""
    j0, j1 = shard_bounds(number_of_particles, rank, world_size)
    basis.j_offset = j0
    return j0, j1


def mean_over_particles(local_values: torch.Tensor, number_of_particles: int, group=None) -> float:
    """Global mean over J of a per-particle vector: all-reduce(sum) of one fp64 (collective C1 of SURVEY.md 2.2)."""
    s = local_values.sum().reshape(1)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    return (s / number_of_particles).item()


class EnergyMean:
    """``train_pls(..., energy_reduce=EnergyMean(J))`` for a J-sharded run: the mean energy over ALL particles, which the stop
    rule of the reference's loop looks at after every step (experiments/trainers.py:158-161) -- the one exchange between the
    ranks that a training iteration needs (collective C1 of SURVEY.md 2.2).

    Called with a tensor it is ``mean_over_particles`` (an all-reduce the caller waits for).  The pipelined loop of
    trainers.py instead hands over the rank's LOCAL energy sum as a host float -- it arrives in pinned host memory from the
    step launch itself, while further steps are already queued -- and ``reduce_local_sum`` exchanges those floats between the
    HOST processes without touching the GPU queues:

      * ranks of one node (what bench.py --gpus N and a torchrun --nnodes=1 job are): a board in POSIX shared memory,
        /dev/shm/pls_energy_<id> -- ring of 16 iterations x world (sequence number, value).  A rank stores its value, then
        the iteration's sequence number (x86-64: stores become visible in program order; aligned 8-byte stores are single
        copies), polls until every rank's number has arrived, and adds the values in RANK order: every rank gets the same
        bits, so every rank takes the same stop decision.  About a microsecond per iteration, no system call, no kernel;
      * otherwise (several hosts, or board=False): a blocking all-reduce of one double per iteration on the group.

    Slot t % 16 is reused at iteration t + 16, which a rank only reaches after every rank has published -- hence finished
    reading -- iteration t + 15 > t.  A rank that leaves its loop through an exception says so on the board (``abort``: one
    word per rank behind the ring; trainers.py calls it on the way out), and its peers raise within a fraction of a second
    instead of polling for a value that will never come; ``timeout_s`` bounds the wait for a rank that died without a word.
    The board relies on x86-64's store ordering (value, then sequence number, two plain stores): on other hosts it is not
    used and the exchange is the all-reduce."""

    RING = 16

    def __init__(self, number_of_particles: int, group=None, board: bool | None = None, timeout_s: float = 600.0):
        self.number_of_particles = int(number_of_particles)
        self.group = group
        self.timeout_s = float(timeout_s)
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self._t = 0
        self._seq = self._val = self._map = self._abort = None
        self._dead = None  # why this exchange can no longer be used (an abort seen or sent)
        if self.world > 1 and board is not False:
            import platform

            if platform.machine().lower() in ("x86_64", "amd64"):
                self._open_board(required=board is True)
            elif board is True:
                raise RuntimeError("EnergyMean(board=True): the shared-memory board needs x86-64's store ordering")

    def _on_device(self) -> bool:
        """Tensors of a collective live on the GPU under RCCL ("nccl", also as part of a composite backend string)"""
        return "nccl" in str(dist.get_backend(self.group)).lower()

    def _open_board(self, required: bool) -> None:
        """Collective over the group.  Every step that can fail locally (creating, mapping the file) is followed by an
        agreement between the ranks, so that either all of them use the board or none does -- never a mixture that would wait
        for each other in different places."""
        import os
        import socket
        import uuid

        import numpy as np

        def all_agree(ok: bool) -> bool:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            if self._on_device():
                flag = flag.cuda()
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return bool(flag.item())

        hosts = [None] * self.world
        dist.all_gather_object(hosts, (socket.gethostname(), os.path.isdir("/dev/shm")), group=self.group)
        if not (all(h == hosts[0] for h in hosts) and hosts[0][1]):
            if required:
                raise RuntimeError("EnergyMean(board=True): the ranks of the group do not share one host with /dev/shm")
            return
        name = [None]
        nwords = 2 * self.RING * self.world + self.world  # sequence numbers, values, one abort word per rank
        nbytes = nwords * 8
        if self.rank == 0:
            try:
                path = f"/dev/shm/pls_energy_{os.getpid()}_{uuid.uuid4().hex[:12]}"
                with open(path, "wb") as f:
                    f.write(b"\0" * nbytes)
                name[0] = path
            except OSError:
                name[0] = None
        dist.broadcast_object_list(name, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        mapped = None
        try:
            if name[0] is not None:
                try:
                    mapped = np.memmap(name[0], dtype=np.int64, mode="r+", shape=(nwords,))
                except (OSError, ValueError):
                    mapped = None
            ok = all_agree(mapped is not None)  # (also the point after which the file's name can go: everybody has mapped it, or given up)
        finally:  # whatever happened above -- a failed collective included -- the name does not stay behind in /dev/shm
            if self.rank == 0 and name[0] is not None:
                try:
                    os.unlink(name[0])
                except OSError:
                    pass
        if not ok:
            if required:
                raise RuntimeError("EnergyMean(board=True): the board could not be created or mapped on every rank")
            return
        self._map = mapped
        ring = self.RING * self.world
        self._seq = mapped[:ring].reshape(self.RING, self.world)
        self._val = mapped[ring:2 * ring].view(np.float64).reshape(self.RING, self.world)
        self._abort = mapped[2 * ring:]

    @property
    def uses_board(self) -> bool:
        return self._seq is not None

    def __call__(self, local_values: torch.Tensor) -> float:
        return mean_over_particles(local_values, self.number_of_particles, self.group)

    def reduce_local_sum(self, local_sum: float) -> float:
        """The global MEAN from this rank's local SUM of per-particle values; every rank of the group must call it once per
        iteration, in the same order."""
        if self.world == 1:
            return float(local_sum) / self.number_of_particles
        if self._dead is not None:
            raise RuntimeError(f"EnergyMean: {self._dead}")
        if self._seq is None:
            s = torch.tensor([float(local_sum)], dtype=torch.float64)
            if self._on_device():
                s = s.cuda()
            dist.all_reduce(s, op=dist.ReduceOp.SUM, group=self.group)
            return s.item() / self.number_of_particles
        import time

        t = self._t
        slot, tag = t % self.RING, t + 1
        self._val[slot, self.rank] = float(local_sum)
        self._seq[slot, self.rank] = tag  # (after the value)
        row = self._seq[slot]
        spins, t0 = 0, None
        while not bool((row == tag).all()):
            spins += 1
            if spins % 64 == 0 and bool(self._abort.any()):
                gone = [r for r in range(self.world) if self._abort[r]]
                self._dead = f"rank(s) {gone} left the training loop through an exception at or before iteration {t}"
                raise RuntimeError(f"EnergyMean: {self._dead}")
            if spins % 4096 == 0:
                t0 = t0 or time.monotonic()
                if time.monotonic() - t0 > self.timeout_s:
                    raise RuntimeError(f"EnergyMean: rank(s) {[r for r in range(self.world) if row[r] != tag]} did not publish "
                                       f"iteration {t} within {self.timeout_s:.0f} s")
                time.sleep(0)
        total = 0.0
        for r in range(self.world):  # rank order: the same bits on every rank
            total += float(self._val[slot, r])
        self._t = t + 1
        return total / self.number_of_particles


def _energy_mean_abort(self, reason: str = "") -> None:
    """This rank leaves its loop through an exception: tell the board, so that the peers polling for its next value raise at
    once (idempotent; a no-op without a board -- a blocking all-reduce fails through the process group's own time-out)."""
    if self._abort is not None:
        self._abort[self.rank] = 1
    self._dead = self._dead or f"this rank aborted the exchange{': ' + reason if reason else ''}"


EnergyMean.abort = _energy_mean_abort


def predictive_moments(local_samples: torch.Tensor, number_of_particles: int, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Mean and unbiased variance over all J particles of (N*, J_local) samples, two passes (sum, then squared deviations
    about the global mean), each followed by one all-reduce of N* doubles (collective C2; replaces
    prediction_samples.mean(dim=1) / .var(axis=1) at gaussian.py:49-52).  Device tensors go through libplship's
    fixed-order row reduction; CPU tensors (tests of the bookkeeping) through torch."""
    on_gpu = local_samples.device.type == "cuda"
    if on_gpu:
        from . import _ops

        s1 = _ops.row_power_sums(local_samples, 1)
    else:
        s1 = local_samples.sum(dim=1)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(s1, op=dist.ReduceOp.SUM, group=group)
    mean = s1 / number_of_particles
    if on_gpu:
        s2 = _ops.row_power_sums(local_samples, 2, shift=mean)
    else:
        s2 = ((local_samples - mean[:, None]) ** 2).sum(dim=1)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(s2, op=dist.ReduceOp.SUM, group=group)
    return mean, s2 / (number_of_particles - 1)


def gather_particles(local_particles: torch.Tensor, number_of_particles: int, group=None) -> torch.Tensor:
    """All-gather the column shards into the full (M, J) matrix on every rank (collective C3: conformal quantiles
    need all samples of a test point, conformalise/pls.py:36-45).  Shards may differ by one column."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_particles
    world = dist.get_world_size(group)
    m = local_particles.shape[0]
    widths = [shard_bounds(number_of_particles, r, world) for r in range(world)]
    wmax = max(b - a for a, b in widths)
    padded = torch.zeros((wmax, m), dtype=local_particles.dtype, device=local_particles.device)
    padded[: local_particles.shape[1]] = local_particles.T
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return torch.cat([o[: b - a].T for o, (a, b) in zip(out, widths)], dim=1)


def sharded_row_quantiles(local_samples: torch.Tensor, q, group=None) -> torch.Tensor:
    """Quantiles over ALL J particles of every row of (N*, J_local) prediction samples -> (N*, len(q)) on every rank
    (collective C3 of SURVEY.md 8e; replaces torch.quantile(samples, q, dim=1) at conformalise/pls.py:36-45, :57-62).

    An order statistic needs every sample of its row, but not every row on every rank: the rows are dealt out over the
    ranks (all-to-all of N*/G x J_local blocks), each rank sorts its N*/G rows of all J samples, and the N* x len(q)
    quantiles are all-gathered -- 1/G of the sorting work and of the receive volume of gathering all samples everywhere
    (round 2 did that, and round 1 gathered the (M, J) particles and repeated the prediction on every rank).  The
    result does not depend on the world size: a quantile is a function of the multiset of a row's samples.
    Device tensors go through libplship's row quantiles; CPU tensors (gloo tests of the bookkeeping) through torch."""
    qs = [float(v) for v in q]

    def quantiles(block: torch.Tensor) -> torch.Tensor:
        if block.device.type == "cuda":
            from . import _ops

            return _ops.row_quantiles(block, qs)
        return torch.quantile(block, torch.tensor(qs, dtype=block.dtype), dim=1).T.contiguous()

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return quantiles(local_samples.contiguous())
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = local_samples.shape[0]
    widths = [torch.zeros(1, dtype=torch.int64, device=local_samples.device) for _ in range(world)]
    dist.all_gather(widths, torch.tensor([local_samples.shape[1]], dtype=torch.int64, device=local_samples.device), group=group)
    widths = [int(w.item()) for w in widths]
    rows = [shard_bounds(n, r, world) for r in range(world)]
    r0, r1 = rows[rank]
    send = [local_samples[a:b, :].contiguous() for a, b in rows]
    recv = [torch.empty((r1 - r0, w), dtype=local_samples.dtype, device=local_samples.device) for w in widths]
    # the all-to-all as one batch of point-to-point operations (RCCL groups them into one exchange over the xGMI links;
    # gloo, used by the CPU tests, has no all_to_all)
    recv[rank].copy_(send[rank])
    ops = []
    for peer in range(world):
        if peer != rank:
            # P2POp addresses its peer by GLOBAL rank; inside a sub-group (ConformalisePLS forwards the caller's group) rank r
            # of the group is some other process of the job
            to = dist.get_global_rank(group, peer) if group is not None else peer
            ops.append(dist.P2POp(dist.isend, send[peer], to, group))
            ops.append(dist.P2POp(dist.irecv, recv[peer], to, group))
    for req in dist.batch_isend_irecv(ops) if ops else []:
        req.wait()
    mine = quantiles(torch.cat(recv, dim=1).contiguous()) if r1 > r0 else torch.empty((0, len(qs)), dtype=local_samples.dtype,
                                                                                     device=local_samples.device)
    hmax = max(b - a for a, b in rows)
    padded = torch.zeros((hmax, len(qs)), dtype=local_samples.dtype, device=local_samples.device)
    padded[: r1 - r0] = mine
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[: b - a] for p, (a, b) in zip(parts, rows)], dim=0).contiguous()
