"""Batched step-size search (the caller of the training loop; replaces experiments/runners.py:331-446).

The reference tries ``number_of_step_searches`` log-spaced step sizes one after the other; every candidate re-runs
``train_pls`` from the same initial particles after ``set_seed(seed)`` for ``int(simulation_duration / step_size)`` epochs.
The candidates are independent samplers, and because each is re-seeded they even consume the SAME noise keys -- so here
they run side by side as column blocks of ONE particle matrix:

* block layout: the candidate with the most epochs first, so the blocks still running always form a prefix of the
  columns; a launch covers that prefix only, and the total number of column-steps equals the sequential search's;
* one fused launch per epoch for all running candidates: a per-block step size (``pls_block_desc``), the per-block
  Philox column index (every block draws the stream a stand-alone run of J particles would), and the per-particle
  energies of the launch's INPUT particles as a by-product, reduced to one mean per block (``pls_block_means``);
* a block's own EarlyStopper sees the same energies in the same order as in a stand-alone run; a block that stops (or
  completes its epochs) hands over its particles and is frozen with step size 0;
* the search's rules are applied in the reference's candidate order as results become available: a run counts if it
  produced energies and finite particles (runners.py:373), the best one by ``metric_to_optimise`` is kept (:411-422), and
  the whole search ends -- abandoning the candidates still running -- as soon as two consecutive accepted runs agree on
  their final energy to ``minimum_change_in_energy_potential`` (:423-433).

A PLS with a user-defined Python cost or basis has no fused step; its candidates are trained one at a time."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _ops
from .basis.base import BlockSpec
from .metrics import calculate_mae, calculate_mse, calculate_nll
from .projected_langevin_sampling import PLS
from .trainers import EarlyStopper, train_pls
from .utils import set_seed

_LOWER_IS_BETTER = ("nll", "mse", "mae", "loss")
_HIGHER_IS_BETTER = ("acc", "auc", "f1")


def candidate_step_sizes(step_size_upper: float, simulation_duration: float, maximum_number_of_steps: int,
                         number_of_step_searches: int) -> np.ndarray:
    """runners.py:356-360: log-spaced from the upper bound down to duration / maximum steps."""
    return np.logspace(np.log10(step_size_upper), np.log10(simulation_duration / maximum_number_of_steps),
                       number_of_step_searches)


class _CandidateRun:
    """What one candidate's training produced: its final particles and the energies its loop accepted."""

    def __init__(self, step_size: float, number_of_epochs: int):
        self.step_size, self.number_of_epochs = step_size, number_of_epochs
        self.particles: Optional[torch.Tensor] = None
        self.energies: List[float] = []
        self.rng_state: Optional[torch.Tensor] = None  # torch CPU generator state right after its training
        self.finished = False


class _SearchLedger:
    """The search's bookkeeping, applied to the candidates in the reference's order (index 0 = largest step size)."""

    def __init__(self, pls: PLS, metric: str, x_train: torch.Tensor, y_train: torch.Tensor, tolerance: float,
                 fallback_particles: torch.Tensor):
        if metric in _LOWER_IS_BETTER:
            self.best_value = float("inf")
        elif metric in _HIGHER_IS_BETTER:
            self.best_value = 0
        else:
            raise NotImplementedError(f"Unknown metric to optimise {metric}.")
        self.pls, self.metric, self.x_train, self.y_train, self.tolerance = pls, metric, x_train, y_train, tolerance
        self.best_step_size: Optional[float] = None
        self.best_particles = fallback_particles
        self.final_energy: Dict[int, float] = {}  # accepted candidates only
        self.accepted_epochs: Dict[float, int] = {}
        self.closed = False
        self.cursor = 0  # next candidate index to judge

    def _metric_value(self, run: _CandidateRun) -> float:
        if self.metric == "loss":
            # (the reference also calls pls.predict here, :374-377, and discards the result; every candidate re-seeds, so
            # the draws it would consume never reach another candidate)
            return run.energies[-1]
        keep = torch.get_rng_state()
        if run.rng_state is not None:
            torch.set_rng_state(run.rng_state)  # predict draws its noise exactly where the sequential search would
        try:
            prediction = self.pls.predict(x=self.x_train, particles=run.particles)
        finally:
            torch.set_rng_state(keep)
        if self.metric == "nll":
            return calculate_nll(prediction=prediction, y=self.y_train)
        if self.metric == "mse":
            return calculate_mse(prediction=prediction, y=self.y_train)
        if self.metric == "mae":
            return calculate_mae(prediction=prediction, y=self.y_train)
        import sklearn.metrics  # acc / auc / f1 go through sklearn like the reference (:392-406)

        yt = self.y_train.cpu().detach().numpy()
        probs = prediction.probs.cpu().detach().numpy()
        if self.metric == "acc":
            return sklearn.metrics.accuracy_score(y_true=yt, y_pred=probs.round())
        if self.metric == "auc":
            return sklearn.metrics.roc_auc_score(y_true=yt, y_score=probs)
        return sklearn.metrics.f1_score(y_true=yt, y_pred=probs.round())

    def judge_ready(self, runs: List[_CandidateRun]) -> None:
        """Judge, in index order, every candidate whose run has finished; stop at the first unfinished one."""
        while not self.closed and self.cursor < len(runs) and runs[self.cursor].finished:
            i, run = self.cursor, runs[self.cursor]
            self.cursor += 1
            if not run.energies or not bool(torch.isfinite(run.particles).all()):
                continue  # diverged or stopped at once: the candidate does not count
            self.final_energy[i] = run.energies[-1]
            self.accepted_epochs[run.step_size] = len(run.energies)
            value = self._metric_value(run)
            improved = value < self.best_value if self.metric in _LOWER_IS_BETTER else value > self.best_value
            if improved:
                self.best_value, self.best_step_size = value, run.step_size
                self.best_particles = run.particles.detach().clone()
            previous = self.final_energy.get(i - 1)
            if previous is not None and abs(previous - run.energies[-1]) / previous < self.tolerance:
                self.closed = True  # two consecutive accepted runs agree: smaller steps will not change the answer
        if self.cursor >= len(runs):
            self.closed = True

    def result(self) -> Tuple[torch.Tensor, float, int]:
        return self.best_particles, self.best_step_size, self.accepted_epochs[self.best_step_size]


def _run_blocks(pls: PLS, particles: torch.Tensor, runs: List[_CandidateRun], patience: float, seed: int,
                ledger: _SearchLedger) -> None:
    """All candidates as column blocks of one particle matrix (module docstring)."""
    basis, cost = pls.basis, pls.cost
    j = particles.shape[1]
    # block b <- candidate order[b]; most epochs first (stable), so the running blocks are a prefix
    order = sorted(range(len(runs)), key=lambda i: -runs[i].number_of_epochs)
    nblocks = len(order)
    epochs = [runs[i].number_of_epochs for i in order]
    stoppers = [EarlyStopper(patience=patience) for _ in order]
    live = [True] * nblocks  # still taking updates
    for b, i in enumerate(order):
        if epochs[b] == 0:  # int(duration / step) == 0: train_pls returns the initial particles and no energy
            runs[i].particles, runs[i].finished, live[b] = particles.detach().clone(), True, False
    cur = particles.detach().repeat(1, nblocks).contiguous()
    nxt = torch.empty_like(cur)
    eta_host = torch.tensor([runs[i].step_size for i in order], dtype=torch.float64)
    eta_dev = eta_host.to(cur.device)
    e_dev = torch.empty(nblocks * j, dtype=torch.float64, device=cur.device)
    means = torch.zeros(nblocks, dtype=torch.float64).pin_memory()
    done_event = torch.cuda.Event()
    set_seed(seed)  # runners.py:364 -- once: every candidate would re-seed to this same state and draw the same keys
    ledger.judge_ready(runs)
    k = 0  # launch k maps U_k -> U_{k+1} and reports E(U_k)
    while not ledger.closed:
        width = sum(1 for b in range(nblocks) if epochs[b] >= k)  # blocks with an update or a last energy outstanding
        if width == 0:
            break
        state_before = torch.get_rng_state()
        for b in range(width):  # a block past its last update (or stopped) is frozen: it only reports its energy
            want = runs[order[b]].step_size if (live[b] and k < epochs[b]) else 0.0
            if eta_host[b] != want:
                eta_host[b] = want
                eta_dev[b: b + 1].fill_(want)
        cols = width * j
        spec = basis._draw_noise_spec(None)  # one key from torch's generator per epoch, as in a stand-alone run
        basis.fused_step(cost, cur[:, :cols], 0.0, out=nxt[:, :cols], new_state=True, noise=spec,
                         input_energy=e_dev[:cols], blocks=BlockSpec(j, eta_dev))
        _ops.block_means(e_dev[:cols], block_cols=j, out_ptr=means.data_ptr())
        done_event.record()
        done_event.synchronize()
        if k >= 1:  # means[b] = E(U_k) of block b: closes iteration k - 1 of its stand-alone loop (trainers.py:158-161)
            for b in range(width):
                if not live[b]:
                    continue
                run = runs[order[b]]
                energy = float(means[b])
                stopped = stoppers[b].should_stop(loss=energy, step_size=run.step_size)
                if not stopped:
                    run.energies.append(energy)
                if stopped or k == epochs[b]:  # U_k (this launch's input) is the block's final state
                    run.particles = cur[:, b * j: (b + 1) * j].clone()
                    run.rng_state = state_before  # k keys drawn: what its own train_pls would have left behind
                    run.finished, live[b] = True, False
            ledger.judge_ready(runs)
        cur, nxt = nxt, cur
        k += 1


def _run_one_by_one(pls: PLS, particles: torch.Tensor, runs: List[_CandidateRun], patience: float, seed: int,
                    ledger: _SearchLedger, train_fn: Callable) -> None:
    """No fused step (user-defined cost or basis), or an explicit ``train_fn``: the candidates train in turn."""
    for run in runs:
        if ledger.closed:
            break
        set_seed(seed)
        run.particles, run.energies = train_fn(pls=pls, particles=particles.detach().clone(),
                                               number_of_epochs=run.number_of_epochs, step_size=run.step_size,
                                               early_stopper_patience=patience)
        run.rng_state, run.finished = torch.get_rng_state(), True
        ledger.judge_ready(runs)


#: A step of J particles shorter than this is launch-bound: the S candidates then run as column blocks of one launch per
#: epoch.  Above it the block launch buys nothing -- its column-steps equal the sequential search's -- and pays a host
#: round trip per epoch, while train_pls keeps eight launches queued.  Measured at N = 1e5, M_k = 1024, Gaussian, 8
#: candidates (tools/runner_probe.py): J = 256 (25 us per step) 0.18 s in blocks vs 0.25 s one by one; J = 1024 (40 us)
#: 0.30 s vs 0.26 s.
LAUNCH_BOUND_STEP_SECONDS = 30e-6


def _step_is_launch_bound(pls: PLS, j: int) -> bool:
    """estimated duration of one step of j particles (at half the fp64 MFMA peak) below LAUNCH_BOUND_STEP_SECONDS"""
    basis = pls.basis
    mk = int(basis.approximation_dimension)
    if getattr(basis, "supports_lagged_energies", lambda c: False)(pls.cost):
        flop = 2.0 * mk * mk * j  # Gaussian/identity fast path: one M_k x M_k x J contraction
    else:
        n = int(getattr(pls.cost, "y_train").shape[0])
        flop = 4.0 * n * mk * j  # F = A^T U and A G
    return flop / 39.3e12 < LAUNCH_BOUND_STEP_SECONDS


def train_pls_runner(
    pls: PLS,
    particle_name: str,
    x_train: torch.Tensor,
    y_train: torch.Tensor,
    simulation_duration: float,
    maximum_number_of_steps: int,
    early_stopper_patience: float,
    number_of_step_searches: int,
    step_size_upper: float,
    minimum_change_in_energy_potential: float,
    seed: int,
    particles: torch.Tensor,
    metric_to_optimise: str = "nll",
    train_fn: Optional[Callable] = None,
    batched: Optional[bool] = None,
) -> Tuple[torch.Tensor, float, int]:
    """Same arguments and return value as the reference's runner -- (best particles, best step size, number of energies
    of the best run), runners.py:446 -- with ``experiment_data`` replaced by the two tensors it is read for
    (experiment_data.train.x / .y, :374-391); plotting is out of scope.  ``train_fn`` (extension): train the candidates
    one at a time with this loop (e.g. ``train_pls_captured``) instead of the batched launch.  ``batched`` (extension):
    force the column-block launch (True) or the one-at-a-time loop (False); None chooses by the step's size."""
    step_sizes = candidate_step_sizes(step_size_upper, simulation_duration, maximum_number_of_steps, number_of_step_searches)
    runs = [_CandidateRun(float(s), int(simulation_duration / s)) for s in step_sizes]  # :363
    ledger = _SearchLedger(pls, metric_to_optimise, x_train, y_train, minimum_change_in_energy_potential,
                           fallback_particles=particles.detach().clone())
    can_batch = (train_fn is None and particles.is_cuda and pls._fused()
                 and getattr(pls.basis, "supports_input_energy", lambda c: False)(pls.cost))
    if batched is None:
        batched = can_batch and _step_is_launch_bound(pls, particles.shape[1])
    else:
        assert not batched or can_batch, "batched=True needs a fused step with the energy by-product on the GPU"
    if batched:
        _run_blocks(pls, particles, runs, early_stopper_patience, seed, ledger)
    else:
        _run_one_by_one(pls, particles, runs, early_stopper_patience, seed, ledger, train_fn or train_pls)
    return ledger.result()
