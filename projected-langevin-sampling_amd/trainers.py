"""Caller of the hot path (drop-in for experiments/trainers.py:139-162 and experiments/early_stopper.py:4-24)."""
from __future__ import annotations

import time
from typing import List, Tuple

import numpy as np
import torch

from . import _lib as L
from . import _ops
from .projected_langevin_sampling import PLS


class EarlyStopper:
    """Stops when the simulated time without an improving loss reaches ``patience`` (early_stopper.py:4-24)."""

    def __init__(self, patience: float = 1e-4):
        self.patience = patience
        self.simulation_time = 0
        self.min_loss = float("inf")

    def should_stop(self, loss: float, step_size: float) -> bool:
        if not np.isfinite(loss):
            return True
        elif loss >= self.min_loss:
            self.simulation_time += step_size
            return self.simulation_time >= self.patience
        else:
            self.min_loss = loss
            self.simulation_time = 0
            return False


def mean_from_chunk_sums(sums, count: int) -> float:
    """Mean energy from the 256-column chunk sums the step's finishing launch leaves (pls_block_desc.energy_sums: what
    pls_block_means computes, bit for bit) or from the 16-column sums of the one-launch small-rank step
    (pls_block_desc.energy_sums16): the entries added in ascending order, divided by the particle count."""
    if isinstance(sums, np.ndarray) and sums.shape[0] > 16:
        return float(np.cumsum(sums)[-1]) / count  # (a cumulative sum is strictly sequential: the loop below, vectorised)
    total = 0.0
    for v in sums:
        total += float(v)
    return total / count


def _supports_energy_sums(pls: PLS) -> bool:
    return bool(getattr(pls.basis, "supports_energy_sums", lambda c: False)(pls.cost))


class _LoopSpace:
    """The coordinates a training loop keeps its particles in between steps.  Identity for every basis but one: the
    inducing-point basis under the Gaussian cost with the identity link, whose step is ONE contraction per iteration in
    whitened coordinates S = Lc^-1 U (InducingPointBasis.whitened_step: 2 M^2 J flop) against forward solve + contraction
    + Lc dS per call (4 M^2 J) -- the loop whitens once, steps S, and maps back once (same chain, same noise counters,
    same energies to rounding: tests/test_gpu_whitened.py).  Injected noise matrices are coloured (N(0, k(Z,Z)) samples
    the reference's sampler would have drawn), so a loop that injects them stays in the original coordinates."""

    def __init__(self, pls: PLS, noises, j: int = 0):
        basis, cost = pls.basis, pls.cost
        self.pls = pls
        can = bool(noises is None and getattr(basis, "whitened", False) and hasattr(basis, "whitened_step")
                   and getattr(cost, "is_native", lambda: False)())
        self.whitened = bool(can and basis._is_gaussian(cost, False))
        # ... and, round 5, the same basis under ANY native cost on at most 128 inducing points while the problem is launch-bound:
        # in whitened coordinates the prior is M more rows of the forward operand and the noise is white, so an iteration is
        # ONE launch (InducingPointBasis.whitened_generic_applies) instead of solve + coloured noise + step -- same draws
        self.whitened_generic = bool(can and not self.whitened and j > 0
                                     and getattr(basis, "whitened_generic_applies", lambda c, n: False)(cost, j))
        self.whitened = self.whitened or self.whitened_generic

    def enter(self, particles: torch.Tensor) -> torch.Tensor:
        return self.pls.basis.whiten(particles) if self.whitened else particles

    def step(self, state, step_size, out, noise, input_energy, blocks=None):
        basis, cost = self.pls.basis, self.pls.cost
        fn = basis.whitened_step if self.whitened else basis.fused_step
        return fn(cost, state, float(step_size), out=out, new_state=True, noise=noise, input_energy=input_energy, blocks=blocks)

    def general_launcher(self, state, step_size):
        """step + energy by-product + mean as a pre-bound call (basis.step_launcher), or None"""
        make = None if self.whitened else getattr(self.pls.basis, "step_launcher", None)
        return None if make is None else make(self.pls.cost, state, step_size)

    def sums_launcher(self, state, eta_dev):
        """step + energies + their chunk sums as a pre-bound call (basis.sums_step_launcher: the one-launch small-rank step), or None"""
        if self.whitened_generic:
            return self.pls.basis.whitened_generic_sums_step_launcher(self.pls.cost, state, eta_dev)
        make = None if self.whitened else getattr(self.pls.basis, "sums_step_launcher", None)
        return None if make is None else make(self.pls.cost, state, eta_dev)

    def lagged_launcher(self, state, eta_dev):
        """the lagged Gaussian step as a pre-bound call (basis.lagged_step_launcher), or None"""
        make = getattr(self.pls.basis, "lagged_step_launcher", None)
        return None if make is None else make(self.pls.cost, state, eta_dev)

    def flush(self, state, blocks) -> None:
        """finish the partial rows of the last step launch (lagged energies: BlockSpec.energy_flush)"""
        self.pls.basis.flush_energies(self.pls.cost, state, blocks)

    def energy(self, state) -> torch.Tensor:
        if self.whitened:
            return self.pls.basis.whitened_particle_energy(self.pls.cost, state)
        return self.pls.particle_energy_potential(state)

    def leave(self, state: torch.Tensor, particles: torch.Tensor) -> None:
        """final state -> the caller's particle tensor"""
        if self.whitened:
            self.pls.basis.unwhiten(state, out=particles)
        elif state.data_ptr() != particles.data_ptr():
            particles.copy_(state)


def _mean_energy(e: torch.Tensor) -> float:
    """Mean over the particles of the per-particle energies (orthonormal.py:126's .mean().item()): libplship's fixed-order
    reduction for device vectors, so that every loop variant (plain, pipelined, captured) reports identical values."""
    return _ops.block_means(e).item() if e.is_cuda else e.mean().item()


def train_pls(
    pls: PLS,
    particles: torch.Tensor,
    number_of_epochs: int,
    step_size: float,
    early_stopper_patience: float,
    tqdm_desc: str | None = None,
    noises: List[torch.Tensor] | None = None,
    energy_reduce=None,
) -> Tuple[torch.Tensor, List[float]]:
    """trainers.py:139-162: update, in-place add, energy, early stop.

    ``noises`` (extension) injects the noise of each step; ``energy_reduce`` (extension) maps the local
    per-particle energy vector to the global mean (distributed.mean_over_particles for J-sharded runs).

    When the step kernel can emit the energy of its input particles as a by-product (Gaussian/identity on the
    orthonormal basis), the loop is software-pipelined: the launch of step t+1 also produces the energy the reference
    evaluates after step t, so every iteration is ONE kernel.  Results (particles, energy list, stop index, torch RNG
    state) are those of the plain loop: a step launched speculatively past the stop is discarded."""
    if particles.is_cuda and particles.dtype in L.PROMOTED_DTYPES:
        # float32 particles: the run is carried in float64 (one rounding at the end instead of one per step) and the caller's
        # tensor receives the final state, which is also what is returned -- trainers.py:157 mutates its argument
        state, energy_potentials = train_pls(pls, particles.double(), number_of_epochs, step_size, early_stopper_patience,
                                             tqdm_desc, noises, energy_reduce)
        particles.copy_(state)
        return particles, energy_potentials
    reduce = energy_reduce if energy_reduce is not None else _mean_energy
    early_stopper = EarlyStopper(patience=early_stopper_patience)
    energy_potentials: List[float] = []
    pipelined = (
        number_of_epochs > 0 and pls._fused() and getattr(pls.basis, "supports_input_energy", lambda c: False)(pls.cost)
    )
    if not pipelined:
        for t in range(number_of_epochs):
            pls.step_(particles, step_size, noise=None if noises is None else noises[t])
            energy_potential = reduce(pls.particle_energy_potential(particles))
            if early_stopper.should_stop(loss=energy_potential, step_size=step_size):
                break
            energy_potentials.append(energy_potential)
        return particles, energy_potentials

    from .basis.base import NoiseSpec

    if particles.is_cuda and (energy_reduce is None or hasattr(energy_reduce, "reduce_local_sum")):
        # (a J-sharded run hands in distributed.EnergyMean: the ranks' local sums meet on the host, the GPU queues stay full)
        return _train_pls_in_flight(pls, particles, number_of_epochs, step_size, early_stopper, noises, mean=energy_reduce)

    space = _LoopSpace(pls, noises, particles.shape[1])
    cur = space.enter(particles)
    nxt = torch.empty_like(particles, memory_format=torch.contiguous_format)
    e_in = torch.empty(particles.shape[1], dtype=torch.float64, device=particles.device)
    stopped = False
    for t in range(number_of_epochs):
        rng_state = torch.get_rng_state()  # the speculative launch below may have to be un-drawn
        spec = NoiseSpec(injected=noises[t]) if noises is not None else None
        space.step(cur, step_size, nxt, spec, e_in)
        if t >= 1:  # e_in = energy of `cur`, i.e. of the particles after update t-1 (trainers.py:158)
            energy_potential = reduce(e_in)
            if early_stopper.should_stop(loss=energy_potential, step_size=step_size):
                torch.set_rng_state(rng_state)
                stopped = True
                break
            energy_potentials.append(energy_potential)
        cur, nxt = nxt, cur
    if not stopped:  # energy after the last update
        energy_potential = reduce(space.energy(cur))
        if not early_stopper.should_stop(loss=energy_potential, step_size=step_size):
            energy_potentials.append(energy_potential)
    space.leave(cur, particles)
    return particles, energy_potentials


#: step launches the pipelined loop keeps queued ahead of the energy it is waiting for (>= 2).  Two keep the GPU busy over
#: the host's ordinary round trip (~30 us); on a shared host the Python thread is now and then descheduled for milliseconds
#: (measured on the GPU boxes: the same loop 0.275 or 0.31 ms per iteration from one repetition to the next, the host's
#: share of an iteration 29 or 63 us, tools/train_loop_probe.py), and a queue of eight 0.27 ms launches rides that out.
#: Costs depth + 1 particle buffers and up to `depth` speculative launches past the stop (discarded).
IN_FLIGHT_DEPTH = 8
#: Gaussian fast paths: launch k + 1 finishes the energies of launch k at its start (pls_block_desc.energy_partials ...), so
#: that no launch carries the reduction's serial tail; False: every launch finishes its own (energy_sync)
LAGGED_ENERGIES = True
#: ... as long as the rotating particle buffers stay below this many bytes (depth is reduced, never below 2)
IN_FLIGHT_BUFFER_BYTES = 4 << 30


def _train_pls_in_flight(pls: PLS, particles: torch.Tensor, number_of_epochs: int, step_size: float,
                         early_stopper: EarlyStopper, noises, depth: int | None = None, mean=None) -> Tuple[torch.Tensor, List[float]]:
    """The pipelined loop with `depth` step launches queued: launch k computes U_{k+1} from U_k and, as a by-product, the
    energy of U_k.  The mean energy travels to pinned host memory by the launch that finishes the by-product (the host
    polls that slot; costs without fused chunk sums: a mean launch followed by an event), and launches k+1 .. k+depth-1
    are already queued behind it while the host waits -- so the GPU never idles over the host's round trip (early-stop
    logic + next launch, ~30 us against a 270 us step at configs[1]) nor over a descheduled host thread.  depth + 1 particle buffers rotate, so the launches made speculatively
    past the stop never touch the returned state, and the torch RNG state is rewound to what the plain loop would have
    consumed.  Same particles, energies and stop index as the plain loop (tests/test_gpu_parity.py)."""
    from .basis.base import NoiseSpec

    from .basis.base import BlockSpec

    T = number_of_epochs
    j = particles.shape[1]
    if depth is None:
        depth = IN_FLIGHT_DEPTH
        while depth > 2 and (depth + 1) * particles.numel() * 8 > IN_FLIGHT_BUFFER_BYTES:
            depth -= 1
    depth = max(2, min(int(depth), max(T, 2)))
    NB = depth + 1  # rotating slots: particle buffers, energy vectors, host sums, events
    space = _LoopSpace(pls, noises, j)
    bufs = [space.enter(particles)] + [torch.empty_like(particles, memory_format=torch.contiguous_format) for _ in range(NB - 1)]
    e_dev = [torch.empty(j, dtype=torch.float64, device=particles.device) for _ in range(NB)]
    # Gaussian/identity fast paths: the launch that finishes the energy by-product also leaves the 256-column chunk sums of
    # the energies -- straight in pinned host memory -- so an iteration is the step kernel and ONE small launch (round 2: a
    # finishing launch plus a mean launch, 15 us of a 280 us iteration); other costs keep the separate mean launch
    fused_sums = _supports_energy_sums(pls)
    # (costs without the Gaussian algebra on a small basis: the one-launch step leaves the sums of 16 columns each -- for free,
    # where the 256-column chunk sums cost it a second hand-over between workgroups, csrc/small_rank_step.h)
    sums16 = bool(fused_sums and getattr(pls.basis, "uses_sums16", lambda c: False)(pls.cost))
    nchunk = ((j + 15) // 16 if sums16 else (j + 255) // 256) if fused_sums else 1
    host = torch.empty(NB * nchunk, dtype=torch.float64).pin_memory()
    host_ptr = host.data_ptr()  # (hipHostMalloc'ed by torch: host and device addresses coincide)
    # Fused chunk sums: no event per launch.  An event record puts a barrier packet between the finishing launch and the
    # next step (5.8 us of idle GPU per iteration in rocprofv3's trace, 2 % of a 0.27 ms iteration); instead the host
    # fills a slot with a NaN of a payload no computation produces before it queues the launch, and reads the slot once
    # every entry has been overwritten (each chunk sum is ONE 8-byte store by the finishing launch into coherent pinned
    # memory; a diverged run's NaN / inf energies are ordinary values here)
    host_bits = host.view(torch.int64)
    from .basis.base import UNWRITTEN_ENERGY_BITS as UNWRITTEN
    eta_dev = torch.full((1,), float(step_size), dtype=torch.float64, device=particles.device) if fused_sums else None
    # ... and, with one zeroed counter per chunk (pls_block_desc.energy_sync), the step launch finishes the energies itself:
    # an iteration is ONE launch (the finishing launch was 5-6 us of a 47 us iteration on the shard of an 8-GPU run)
    sync = torch.zeros(nchunk, dtype=torch.int32, device=particles.device) if fused_sums else None
    # ... or, better, launch k + 1 finishes the energies of launch k at its START, under the landing of its first operand rows
    # (pls_block_desc.energy_partials / _prev): the reduction over the tile rows -- 4.4-5 us of serial tail behind the last MFMA
    # when a launch finishes its own -- leaves the critical path altogether; E(U_k) then arrives with launch k + 1, and the
    # last launch's partial rows are finished by a small launch of their own (flush)
    lagged = bool(fused_sums and LAGGED_ENERGIES and getattr(pls.basis, "supports_lagged_energies", lambda c: False)(pls.cost)
                  and (space.whitened or getattr(pls.basis, "fused_step_takes_lagged_energies", False)))
    fast = None
    if lagged:
        pbytes = pls.basis.energy_partial_rows_bytes(j)
        parts = [torch.empty((pbytes + 7) // 8, dtype=torch.float64, device=particles.device) for _ in range(2)]
        if noises is None and all(b.dim() == 2 and b.stride(1) == 1 for b in bufs):
            # The host's share of an iteration is what bounds the loop at the reference's own problem sizes (a 10 us kernel
            # against 20 us of Python): the step call is bound once (descriptors, stream), the buffers' addresses are looked up
            # once, the pinned slots are polled through numpy views, and the per-step keys are drawn from torch's global
            # generator in batches -- the same stream of draws as one per step, and the generator is left exactly where the
            # plain loop leaves it (below)
            fast = space.lagged_launcher(bufs[0], eta_dev)
    sums = None
    if fused_sums and fast is None and not lagged and noises is None and pls.cost.is_native() \
            and all(b.dim() == 2 and b.stride(1) == 1 for b in bufs):
        # costs without the Gaussian algebra on a small basis: ONE launch per iteration leaves the new state, the energies of its
        # input and their chunk sums in the pinned slot (csrc/small_rank_step.h), bound once like the lagged Gaussian step
        sums = space.sums_launcher(bufs[0], eta_dev)
    general = None
    if not fused_sums and noises is None and pls.cost.is_native() and all(b.dim() == 2 and b.stride(1) == 1 for b in bufs):
        # (other costs: the same host-side trim around pls_onb_step + pls_block_means; the mean's pinned slot is polled like
        # the chunk sums instead of waited for through an event)
        general = space.general_launcher(bufs[0], step_size)
    if fast is not None or general is not None or sums is not None:
        buf_ptr, buf_ld = [b.data_ptr() for b in bufs], [L.ld(b) for b in bufs]
        e_ptr = [e.data_ptr() for e in e_dev]
        part_ptr = [p_.data_ptr() for p_ in parts] if fast is not None else None
        keys: List[int] = []
        rng_start = torch.get_rng_state()
    host_np = host.numpy()  # (shares the pinned pages)
    host_np_bits = host_np.view(np.int64)
    flushed = [False]
    events = [torch.cuda.Event() for _ in range(NB)]
    rng_states = {}
    launched = 0

    def launch():
        nonlocal launched
        k = launched
        if fast is not None:  # launch k leaves the partial rows of E(U_k) and finishes those of E(U_{k-1}) into slot k - 1
            if k >= len(keys):
                keys.extend(torch.randint(0, 2**62, (256,), dtype=torch.int64).tolist())
            a, b, prev = k % NB, (k + 1) % NB, (k - 1) % NB
            if k > 0:
                host_np_bits[prev * nchunk:(prev + 1) * nchunk] = UNWRITTEN
                fast(buf_ptr[a], buf_ld[a], buf_ptr[b], buf_ld[b], keys[k], part_ptr[k % 2], part_ptr[(k - 1) % 2], e_ptr[prev],
                     host_ptr + 8 * nchunk * prev)
            else:
                fast(buf_ptr[a], buf_ld[a], buf_ptr[b], buf_ld[b], keys[k], part_ptr[0], None, None, None)
            launched += 1
            return
        if sums is not None:  # launch k: U_{k+1} from U_k, E(U_k) and its chunk sums (slot k) from the same launch
            if k >= len(keys):
                keys.extend(torch.randint(0, 2**62, (256,), dtype=torch.int64).tolist())
            a, b = k % NB, (k + 1) % NB
            host_np_bits[a * nchunk:(a + 1) * nchunk] = UNWRITTEN
            sums(buf_ptr[a], buf_ld[a], buf_ptr[b], buf_ld[b], keys[k], e_ptr[a], host_ptr + 8 * nchunk * a)
            launched += 1
            return
        if general is not None:  # launch k: U_{k+1} from U_k, E(U_k) as a by-product, its mean into slot k
            if k >= len(keys):
                keys.extend(torch.randint(0, 2**62, (256,), dtype=torch.int64).tolist())
            a, b = k % NB, (k + 1) % NB
            host_np_bits[a] = UNWRITTEN
            general(buf_ptr[a], buf_ld[a], buf_ptr[b], buf_ld[b], keys[k], e_ptr[a], host_ptr + 8 * a)
            launched += 1
            return
        rng_states[k] = torch.get_rng_state()  # (a speculative launch may have to be un-drawn)
        spec = NoiseSpec(injected=noises[k]) if noises is not None else None
        if lagged:  # launch k leaves the partial rows of E(U_k) and finishes those of E(U_{k-1}) into slot k - 1
            prev = (k - 1) % NB
            if k > 0:
                host_bits[prev * nchunk:(prev + 1) * nchunk] = UNWRITTEN
            blocks = BlockSpec(j, eta_dev, energy_partials=parts[k % 2], energy_partials_prev=parts[(k - 1) % 2] if k > 0 else None,
                               energy_prev=e_dev[prev] if k > 0 else None,
                               energy_sums_prev=host_ptr + 8 * nchunk * prev if k > 0 else None)
            space.step(bufs[k % NB], step_size, bufs[(k + 1) % NB], spec, None, blocks=blocks)
        elif fused_sums:  # one column block = all particles, its step size from a device word, chunk sums to the host slot
            host_bits[(k % NB) * nchunk:(k % NB + 1) * nchunk] = UNWRITTEN
            slot = host_ptr + 8 * nchunk * (k % NB)
            blocks = BlockSpec(j, eta_dev, energy_sums16=slot) if sums16 else BlockSpec(j, eta_dev, energy_sums=slot, energy_sync=sync)
            space.step(bufs[k % NB], step_size, bufs[(k + 1) % NB], spec, e_dev[k % NB], blocks=blocks)
        else:
            space.step(bufs[k % NB], step_size, bufs[(k + 1) % NB], spec, e_dev[k % NB])
            # E(U_k): the reduction kernel stores the mean straight into pinned host memory (mapped into the device's
            # address space); the host reads it after the event -- no torch reduce kernel, no copy kernel per iteration
            _ops.block_means(e_dev[k % NB], out_ptr=host_ptr + 8 * (k % NB))
            events[k % NB].record()
        rng_states.pop(k - NB - 1, None)
        launched += 1

    def read_energy(slot: int) -> float:
        if mean is not None:  # J-sharded run (distributed.EnergyMean): the local SUM goes to the ranks' host-side exchange
            local = mean_from_chunk_sums(host_np[slot * nchunk:(slot + 1) * nchunk], 1) if fused_sums else float(host_np[slot]) * j
            return mean.reduce_local_sum(local)
        if fused_sums:
            return mean_from_chunk_sums(host_np[slot * nchunk:(slot + 1) * nchunk], j)
        return float(host_np[slot])

    def wait_for(slot: int) -> None:
        if not fused_sums and general is None:
            events[slot].synchronize()
            return
        bits = host_np_bits[slot * nchunk:(slot + 1) * nchunk]
        spins = 0
        while bool((bits == UNWRITTEN).any()):
            spins += 1
            if spins > 1024:  # far beyond a fast-path step (tens to hundreds of microseconds): stop burning the core
                time.sleep(1e-4)
            if spins % 4096 == 0 and torch.cuda.current_stream().query():  # the queue has drained and the slot is still
                if bool((bits == UNWRITTEN).any()):                       # unwritten: a failed launch, not a slow one
                    raise RuntimeError("train_pls: the step launch did not deliver its energy sums")

    energy_potentials: List[float] = []
    final = None
    keys_used = T  # (the plain loop draws one key per step it executes: T, or t + 1 when iteration t stops it)
    try:
        for t in range(T):  # iteration t of the plain loop: update t done (U_{t+1}), its energy E(U_{t+1}) wanted
            # launch t+1 carries E(U_{t+1}); t+2 .. t+depth keep the queue non-empty.  Launch k writes slot (k+1) % NB and
            # the stop below may return slot (t+1) % NB: k + 1 - (t + 1) <= depth < NB, so that slot is never overwritten
            while launched < T and launched <= t + depth:
                launch()
            if lagged and launched == T and not flushed[0]:
                # every step is queued: E(U_{T-1}) has no following launch to ride on -- a small finishing launch of its own
                k = T - 1
                host_bits[(k % NB) * nchunk:(k % NB + 1) * nchunk] = UNWRITTEN
                space.flush(bufs[k % NB], BlockSpec(j, eta_dev, energy_partials_prev=parts[k % 2], energy_prev=e_dev[k % NB],
                                                    energy_sums_prev=host_ptr + 8 * nchunk * (k % NB), energy_flush=True))
                flushed[0] = True
            if t + 1 < T:
                wait_for((t + 1) % NB)
                energy_potential = read_energy((t + 1) % NB)
            else:  # the energy after the last update has no following launch to ride on
                last = space.energy(bufs[T % NB])
                energy_potential = _mean_energy(last) if mean is None else mean.reduce_local_sum(last.sum().item())
            if early_stopper.should_stop(loss=energy_potential, step_size=step_size):
                if fast is not None or general is not None or sums is not None:
                    keys_used = t + 1
                elif launched > t + 1:
                    torch.set_rng_state(rng_states[t + 1])
                final = bufs[(t + 1) % NB]
                break
            energy_potentials.append(energy_potential)
    except BaseException as exc:
        # a J-sharded run: the other ranks are polling for this rank's next energy sum -- tell them it will not come
        if mean is not None and hasattr(mean, "abort"):
            mean.abort(repr(exc)[:200])
        raise
    finally:
        # up to `depth` launches are still queued: they write the pinned slots and the rotating buffers, which must not go
        # back to torch's allocators (on ANY exit: a raising early stopper, a failed launch) before they have drained
        torch.cuda.current_stream().synchronize()
    if fast is not None or general is not None or sums is not None:  # leave torch's generator where one draw per executed step leaves it
        torch.set_rng_state(rng_start)
        if keys_used > 0:
            torch.randint(0, 2**62, (keys_used,), dtype=torch.int64)
    if final is None:
        final = bufs[T % NB]
    space.leave(final, particles)
    return particles, energy_potentials


def train_pls_captured(pls: PLS, particles: torch.Tensor, number_of_epochs: int, step_size: float,
                       early_stopper_patience: float, steps_per_replay: int = 16, seed: int | None = None
                       ) -> Tuple[torch.Tensor, List[float]]:
    """train_pls for launch-bound problems (extension): K steps and their energies per hipGraph replay
    (graph.CapturedTraining).  Same loop semantics as experiments/trainers.py:139-162 -- update, energy of the updated
    particles, early stop, the failing update is kept -- over the library's counter-based noise stream: ``seed`` (one
    draw from torch's global generator if None) keys it, step t uses counter t.  The result equals the eager loop over
    that stream exactly (an overshooting replay is rolled back and re-run up to the stop index)."""
    from .graph import CapturedTraining

    if seed is None:
        seed = int(torch.randint(0, 2**63 - 1, (1,)).item())
    early_stopper = EarlyStopper(patience=early_stopper_patience)
    energies: List[float] = []
    T, K = number_of_epochs, max(1, min(steps_per_replay, number_of_epochs))
    if T == 0:
        return particles, energies
    cap = CapturedTraining(pls, particles, step_size, K, seed)
    t = 0  # next plain-loop iteration to judge: needs E(U_{t+1})
    while t < T:
        k0 = cap.steps_done
        if k0 + K <= T:  # a full replay: launches k0 .. k0+K-1 report E(U_k0) .. E(U_{k0+K-1})
            known = cap.replay().tolist()
        else:  # fewer than K steps left: finish eagerly, one energy per step
            known = None
        if known is not None:
            stop_at = None
            for s, e in enumerate(known):
                it = k0 + s - 1  # E(U_{k0+s}) closes plain-loop iteration k0+s-1
                if it < 0:
                    continue  # E(U_0): the plain loop never looks at it
                if early_stopper.should_stop(loss=e, step_size=step_size):
                    stop_at = it
                    break
                energies.append(e)
                t = it + 1
            if stop_at is not None:  # keep U_{stop_at+1}: rewind the replay, roll forward eagerly
                cap.roll_back()
                cap.eager_steps(stop_at + 1 - cap.steps_done)
                return particles, energies
        else:
            while t < T:
                if cap.steps_done < t + 1:
                    cap.eager_steps(t + 1 - cap.steps_done)
                e = _mean_energy(pls.particle_energy_potential(particles))
                if early_stopper.should_stop(loss=e, step_size=step_size):
                    return particles, energies
                energies.append(e)
                t += 1
            return particles, energies
        if cap.steps_done == T:  # all updates done; E(U_T) has no following launch to ride on
            e = _mean_energy(pls.particle_energy_potential(particles))
            if not early_stopper.should_stop(loss=e, step_size=step_size):
                energies.append(e)
            return particles, energies
    return particles, energies
