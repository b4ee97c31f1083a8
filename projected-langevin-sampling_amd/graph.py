"""hipGraph-captured Langevin steps (extension; no reference counterpart).

The reference's loop body is `particles += pls.calculate_particle_update(particles, eta)` (experiments/profiler/
main.py:77-82).  When a rank's particle shard is small the step is launch-bound, so K consecutive fused steps are
captured once into a hipGraph (libplship's launch functions neither allocate nor synchronise) and replayed.  The noise
stream is (seed, device step counter): launch arguments are frozen by the capture, the counter is read by the kernels
at run time and advanced by K inside the graph, so every replay draws fresh noise and a replayed run equals the eager
run with the same (seed, step) sequence bit for bit."""
from __future__ import annotations

import torch

from . import _lib as L
from . import _ops
from .basis.base import UNWRITTEN_ENERGY_BITS as UNWRITTEN
from .basis.base import NoiseSpec
from .projected_langevin_sampling import PLS


def _own_workspace(basis, cost, particles: torch.Tensor, with_energy: bool, force_generic: bool = False) -> torch.Tensor | None:
    """A workspace buffer that belongs to ONE capture (never shared with, nor freed by, the basis)."""
    cd = cost.desc()
    if cd.cost == L.COST_GAUSSIAN and cd.link == L.LINK_IDENTITY and not force_generic:
        # (allocates B, c and the whitened operator: must not happen inside the capture)
        if hasattr(basis, "_prepare_for"):
            basis._prepare_for(cost)
        else:
            basis.prepare_gaussian(cost.y_device())
    nbytes = basis.step_workspace_bytes(cost, particles.shape[1], with_energy, force_generic)
    return torch.empty(max(nbytes // 8 + 1, 1), dtype=torch.float64, device=particles.device)


class CapturedSteps:
    """K fused steps per replay on a fixed particle buffer.  ``particles`` is updated in place by every replay()."""

    def __init__(self, pls: PLS, particles: torch.Tensor, step_size: float, steps_per_replay: int, seed: int,
                 force_generic: bool = False):
        if not pls._fused():
            raise L.PlsHipError("graph capture needs a native basis and cost (the fused step)")
        assert steps_per_replay >= 1
        L.require_gpu_tensor(particles, "particles")
        assert particles.is_contiguous()
        self.pls, self.particles, self.k = pls, particles, steps_per_replay
        self._pong = torch.empty_like(particles)
        self.counter = torch.zeros(1, dtype=torch.int64, device=particles.device)
        basis, cost = pls.basis, pls.cost
        # the capture freezes the workspace ADDRESS: it owns the buffer (the basis' own scratch is reallocated whenever a
        # later eager call asks for more bytes, e.g. an energy evaluation between two replays)
        self._ws = _own_workspace(basis, cost, particles, with_energy=False, force_generic=force_generic)
        # ... and, for the same reason, the step-size word and the arrival counters of the one-launch small-rank step
        # (pls_block_desc.step_sync): the basis keeps such counters per stream for eager calls, a capture brings its own
        from .basis.base import BlockSpec

        self._eta = torch.full((1,), float(step_size), dtype=torch.float64, device=particles.device)
        self._step_sync = torch.zeros(max(int(L.load().pls_step_sync_words(particles.shape[1])), 1), dtype=torch.int32,
                                      device=particles.device)
        blocks = BlockSpec(max(int(particles.shape[1]), 1), self._eta, step_sync=self._step_sync)

        def body():
            cur, nxt = self.particles, self._pong
            for s in range(self.k):
                spec = NoiseSpec(seed=seed, step=s, j_offset=basis.j_offset, step_base=self.counter)
                basis.fused_step(cost, cur, float(step_size), out=nxt, new_state=True, noise=spec, force_generic=force_generic,
                                 workspace=self._ws, blocks=blocks)
                cur, nxt = nxt, cur
            if cur is not self.particles:
                self.particles.copy_(cur)
            L.check(L.load().pls_counter_add(self.counter.data_ptr(), self.k, L.stream_ptr()), "pls_counter_add")

        # warm-up outside the capture (workspaces, Gaussian constants, kernel attributes), then rewind the state it touched
        saved = particles.clone()
        body()
        torch.cuda.synchronize()
        self.particles.copy_(saved)
        self.counter.zero_()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(self.graph, stream=side):
                body()
        torch.cuda.current_stream().wait_stream(side)
        self.particles.copy_(saved)
        self.counter.zero_()

    def replay(self, times: int = 1) -> torch.Tensor:
        for _ in range(times):
            self.graph.replay()
        return self.particles

    @property
    def steps_done(self) -> int:
        return int(self.counter.item())


class CapturedTraining:
    """The train loop (step + energy + early stop, experiments/trainers.py:139-162) with K steps per hipGraph replay.

    Each captured step launch also emits the energy of its input particles (pls_onb_step / pls_ipb_step ``energy_in``);
    their means land in a (K,) device vector that is read back once per replay, so a launch-bound problem (the
    reference's own experiment sizes: N ~ 1e2..1e4, M ~ 1e1..1e2) pays one graph launch and one host sync per K
    iterations instead of ~5 launches and a sync per iteration.  The early-stop rule is evaluated on the host after
    every replay; a replay that overshoots the stop is rolled back to its saved starting particles and re-run eagerly
    up to the stop index with the same (seed, step) noise counters -- the kernels are deterministic, so the returned
    particles, energy list and stop index are exactly those of the eager loop over the same noise stream
    (tests/test_gpu_parity.py::test_captured_training_matches_the_eager_loop)."""

    def __init__(self, pls: PLS, particles: torch.Tensor, step_size: float, steps_per_replay: int, seed: int):
        if not pls._fused() or not getattr(pls.basis, "supports_input_energy", lambda c: False)(pls.cost):
            raise L.PlsHipError("captured training needs a native basis and cost (fused step with the energy by-product)")
        assert steps_per_replay >= 1
        L.require_gpu_tensor(particles, "particles")
        assert particles.is_contiguous()
        self.pls, self.particles, self.k, self.seed, self.step_size = pls, particles, steps_per_replay, int(seed), float(step_size)
        self._pong = torch.empty_like(particles)
        self._start = torch.empty_like(particles)  # particles at the start of the last replay (roll-back point)
        self._e = torch.empty(particles.shape[1], dtype=torch.float64, device=particles.device)
        self.means = torch.zeros(steps_per_replay, dtype=torch.float64, device=particles.device)
        # Gaussian/identity fast paths: the finishing launch of the energy by-product leaves the 256-column chunk sums; the
        # host adds them after the replay (no mean launch inside the graph)
        j = particles.shape[1]
        self._fused_sums = bool(getattr(pls.basis, "supports_energy_sums", lambda c: False)(pls.cost))
        self._nchunk = (j + 255) // 256
        self._sums = torch.zeros(steps_per_replay, self._nchunk, dtype=torch.float64, device=particles.device)
        self._sync = torch.zeros(self._nchunk, dtype=torch.int32, device=particles.device)  # (pls_block_desc.energy_sync)
        # (pls_block_desc.step_sync: the one-launch small-rank step's counters, owned by the capture like its workspace)
        self._step_sync = torch.zeros(max(int(L.load().pls_step_sync_words(j)), 1), dtype=torch.int32, device=particles.device)
        # lagged energies (trainers.LAGGED_ENERGIES): launch s + 1 of the replay finishes the energies of launch s at its start,
        # one small finishing launch closes the replay
        from . import trainers as _tr

        self._lagged = bool(self._fused_sums and _tr.LAGGED_ENERGIES and getattr(pls.basis, "fused_step_takes_lagged_energies", False)
                            and getattr(pls.basis, "supports_lagged_energies", lambda c: False)(pls.cost))
        if self._lagged:
            pbytes = pls.basis.energy_partial_rows_bytes(j)
            self._parts = [torch.empty((pbytes + 7) // 8, dtype=torch.float64, device=particles.device) for _ in range(2)]
        self._eta = torch.full((1,), float(step_size), dtype=torch.float64, device=particles.device)
        self.counter = torch.zeros(1, dtype=torch.int64, device=particles.device)
        basis, cost = pls.basis, pls.cost
        self._ws = _own_workspace(basis, cost, particles, with_energy=True)  # owned by the capture (see CapturedSteps)

        def body():
            self._start.copy_(self.particles)
            if self._fused_sums:
                # one fill per replay: a step route that does not deliver its chunk sums leaves the sentinel behind and
                # replay() raises instead of handing zeros to the early stop (trainers.UNWRITTEN: a NaN payload no
                # computation produces; a diverged run's NaN / inf energies are ordinary values)
                self._sums.view(torch.int64).fill_(UNWRITTEN)
            cur, nxt = self.particles, self._pong
            for s in range(self.k):
                spec = NoiseSpec(seed=self.seed, step=s, j_offset=basis.j_offset, step_base=self.counter)
                if self._lagged:
                    from .basis.base import BlockSpec

                    blocks = BlockSpec(cur.shape[1], self._eta, energy_partials=self._parts[s % 2],
                                       energy_partials_prev=self._parts[(s - 1) % 2] if s > 0 else None,
                                       energy_prev=self._e if s > 0 else None,
                                       energy_sums_prev=self._sums[s - 1].data_ptr() if s > 0 else None)
                    basis.fused_step(cost, cur, self.step_size, out=nxt, new_state=True, noise=spec, blocks=blocks)
                elif self._fused_sums:
                    from .basis.base import BlockSpec

                    blocks = BlockSpec(cur.shape[1], self._eta, energy_sums=self._sums[s].data_ptr(), energy_sync=self._sync,
                                       step_sync=self._step_sync)
                    basis.fused_step(cost, cur, self.step_size, out=nxt, new_state=True, noise=spec, input_energy=self._e,
                                     workspace=self._ws, blocks=blocks)
                else:
                    basis.fused_step(cost, cur, self.step_size, out=nxt, new_state=True, noise=spec, input_energy=self._e,
                                     workspace=self._ws)
                    _ops.block_means(self._e, out=self.means[s: s + 1])  # E(U_{k0 + s}), the INPUT of launch k0 + s
                cur, nxt = nxt, cur
            if self._lagged:  # the last step's partial rows: a finishing launch of their own
                from .basis.base import BlockSpec

                basis.flush_energies(cost, cur, BlockSpec(cur.shape[1], self._eta, energy_partials_prev=self._parts[(self.k - 1) % 2],
                                                          energy_prev=self._e, energy_sums_prev=self._sums[self.k - 1].data_ptr(),
                                                          energy_flush=True))
            if cur is not self.particles:
                self.particles.copy_(cur)
            L.check(L.load().pls_counter_add(self.counter.data_ptr(), self.k, L.stream_ptr()), "pls_counter_add")

        saved = particles.clone()
        body()  # warm-up outside the capture (workspaces, Gaussian constants, kernel attributes)
        torch.cuda.synchronize()
        self.particles.copy_(saved)
        self.counter.zero_()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(self.graph, stream=side):
                body()
        torch.cuda.current_stream().wait_stream(side)
        self.particles.copy_(saved)
        self.counter.zero_()
        self.steps_done = 0

    def eager_steps(self, count: int) -> None:
        """``count`` steps without the graph, continuing the same noise stream (tail of a run / roll-forward)."""
        basis, cost = self.pls.basis, self.pls.cost
        cur, nxt = self.particles, self._pong
        for _ in range(count):
            spec = NoiseSpec(seed=self.seed, step=self.steps_done, j_offset=basis.j_offset)
            basis.fused_step(cost, cur, self.step_size, out=nxt, new_state=True, noise=spec)
            cur, nxt = nxt, cur
            self.steps_done += 1
        if cur is not self.particles:
            self.particles.copy_(cur)
        self.counter.fill_(self.steps_done)

    def replay(self) -> torch.Tensor:
        """K more steps; returns the (K,) host vector [E(U_k0), ..., E(U_{k0+K-1})], k0 = steps done before the call."""
        self.graph.replay()
        self.steps_done += self.k
        if self._fused_sums:
            from .trainers import mean_from_chunk_sums

            host = self._sums.cpu()
            if bool((host.view(torch.int64) == UNWRITTEN).any()):
                raise RuntimeError("CapturedTraining: a captured step did not deliver its energy sums")
            rows = host.tolist()
            return torch.tensor([mean_from_chunk_sums(r, self.particles.shape[1]) for r in rows], dtype=torch.float64)
        return self.means.cpu()

    def roll_back(self) -> None:
        """Undo the last replay (particles and step counter)."""
        self.particles.copy_(self._start)
        self.steps_done -= self.k
        self.counter.fill_(self.steps_done)
