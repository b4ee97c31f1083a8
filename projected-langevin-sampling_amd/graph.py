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
from .basis.base import NoiseSpec
from .projected_langevin_sampling import PLS


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

        def body():
            cur, nxt = self.particles, self._pong
            for s in range(self.k):
                spec = NoiseSpec(seed=seed, step=s, j_offset=basis.j_offset, step_base=self.counter)
                basis.fused_step(cost, cur, float(step_size), out=nxt, new_state=True, noise=spec, force_generic=force_generic)
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
