#!/usr/bin/env python3
"""Langevin steps/sec of the MI355X hot path on BASELINE.json's headline configuration
(configs[1]: synthetic UCI-style regression, N=1e5, M=1024, J=8192, RBF/ARD kernel, Gaussian cost, fp64).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = particles <- particles + PLS.calculate_particle_update(particles, eta) for ALL J particles
(the loop body of the reference's experiments/profiler/main.py:77-82).  The particle axis is sharded over the
ranks (strong scaling: J is fixed); the step has no collective.

Rank 0 prints ONE JSON line.  `value` is the like-for-like path: the same 4*N*M*J flop per step the reference
performs (F = A^T U, d cost/d f, A G), i.e. the kernel that also serves the Poisson / Bernoulli / Student-t costs.
The Gaussian/identity algebraic shortcut (B = A A^T precomputed, 2*M^2*J flop per step; README.md:9's O(M^3 + J M^2))
is measured in the same run and reported under "gaussian_fast_path" -- never mixed into `value`.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix (= vector) peak, AMD datasheet; MI355X_MICROARCH.md has no fp64 row
HBM_PEAK_GBS = 8000.0

CONFIGS = {
    # name: (N, M, J, D, cost)
    "c2": dict(n=100_000, m=1024, j=8192, d=8, cost="gaussian", eta=1e-5, obs=0.01,
               workload="configs[1]: synthetic regression N=1e5 M=1024 J=8192 D=8, RBF/ARD, ONB + Gaussian/identity, fp64"),
    "c3": dict(n=50_000, m=512, j=16384, d=1, cost="poisson", eta=1e-6, obs=None, threshold=1e-7,
               workload="configs[2]: Poisson (f^2 link) regression N=5e4 M=512 J=16384 D=1, ONB, fp64"),
    "c4": dict(n=200_000, m=2048, j=8192, d=8, cost="bernoulli", eta=1e-6, obs=None,
               workload="configs[3], one GPU's shard (J = 32768 / 4): binary classification N=2e5 M=2048 D=8, ONB + Bernoulli/sigmoid, fp64"),
    "c5": dict(n=1_000_000, m=4096, j=8192, d=8, cost="gaussian", eta=1e-5, obs=0.01,
               workload="configs[4], one GPU's shard (J = 65536 / 8): regression N=1e6 M=4096 D=8, ONB + Gaussian/identity, fp64"),
    "c1": dict(n=100, m=10, j=64, d=1, cost="gaussian", eta=1e-3, obs=0.25,
               workload="configs[0]: 1D regression N=100 M=10 J=64, ONB + Gaussian/identity, fp64 (launch-bound)"),
    "small": dict(n=4096, m=128, j=512, d=4, cost="gaussian", eta=1e-4, obs=0.01,
                  workload="smoke-sized regression N=4096 M=128 J=512"),
}


T0 = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def make_data(cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    n, m, d = cfg["n"], cfg["m"], cfg["d"]
    if d == 1:
        x = torch.linspace(-3, 3, n, dtype=torch.float64).reshape(-1, 1)
    else:
        x = torch.rand(n, d, generator=g, dtype=torch.float64) * 2 - 1
    z = x[torch.randperm(n, generator=g)[:m]].clone()
    w = torch.randn(d, generator=g, dtype=torch.float64)
    fstar = torch.sin(2.0 * (x @ w))
    if cfg["cost"] == "poisson":
        y = torch.poisson((2.0 * fstar) ** 2 + 0.1, generator=g)
    elif cfg["cost"] == "bernoulli":  # labels ~ Bernoulli(sigmoid(f*)) as curves/curves.py:31-38
        y = (torch.rand(n, generator=g, dtype=torch.float64) < torch.sigmoid(2.0 * fstar)).double()
    else:
        y = fstar + 0.1 * torch.randn(n, generator=g, dtype=torch.float64)
    ls = 0.5 + torch.rand(d, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    if d == 1:
        ls = ls * 0.2
    return x, z, y, ls


def host_cores() -> int:
    """CPU threads this process may really use: scheduler affinity capped by the cgroup CPU quota
    (os.cpu_count() reports the whole host and oversubscribes a containerised box)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or platform.machine()


def cpu_baseline(cfg, x, z, y, ls, lam_all, vec_all):
    """The oracle ("port": faithful torch-CPU restatement of the reference's step, oracle/pls_oracle.py) timed on this
    box's host cores.  The step is MEASURED at the configuration's full (N, J) whenever 1 warm-up + 2 timed steps fit the
    ~30 s budget (configs[1], configs[2]); a larger configuration is timed on a stated (N, J) sub-sample and scaled by
    the flop ratio of the step, and the sample string says so."""
    import resource

    from oracle import pls_oracle as O

    cores = host_cores()
    torch.set_num_threads(cores)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        m, mk = z.shape[0], lam_all.shape[0]
        n_full, j_full = x.shape[0], cfg["j"]

        def step_flop(n, j):  # reference association: (K_XZ V~) U, V~^T K_ZX then @ G, + the dense diag(1/lam) @ U
            return 4.0 * n * m * mk + 4.0 * n * mk * j + 2.0 * mk * mk * j

        budget = 1.8e13  # 1 warm-up + 3 timed steps in ~30 s at the ~0.5 TFLOP/s these 16 host cores sustain in fp64
        n_s, j_s = n_full, j_full
        while 4 * step_flop(n_s, j_s) > budget and j_s > max(256, j_full // 8):
            j_s //= 2
        while 4 * step_flop(n_s, j_s) > budget and n_s > 20_000:
            n_s //= 2
        xs, ys = x[:n_s], y[:n_s]
        ob = O.OrthonormalBasis.__new__(O.OrthonormalBasis)  # reuse the spectrum already computed for the GPU basis
        kern = O.RBFARDKernel(ls, 1.0)
        ob.base_kernel, ob.x_induce = kern, z
        ob.base_gram_induce = None
        ob.base_gram_induce_train = torch.empty(m, n_s)
        for r0 in range(0, n_s, 8192):  # chunked k(Z,X): the broadcasted build would need N*M*D doubles
            ob.base_gram_induce_train[:, r0:r0 + 8192] = kern(z, xs[r0:r0 + 8192])
            if (r0 // 8192) % 8 == 0:
                log(f"cpu baseline: host k(Z,X) rows {r0}/{n_s}")
        ob.eigenvalues, ob.eigenvectors = lam_all, vec_all
        ob.scaled_eigenvectors = vec_all / torch.sqrt(mk * lam_all)[None, :]
        if cfg["cost"] == "poisson":
            oc = O.PoissonCost(ys, O.SquareLink())
        elif cfg["cost"] == "bernoulli":
            oc = O.BernoulliCost(ys, O.SigmoidLink())
        else:
            oc = O.GaussianCost(cfg["obs"], ys, O.IdentityLink())
        pls = O.PLS(ob, oc)
        log(f"cpu baseline: k(Z,X) built on the host, {cores} threads; timing the oracle step at N={n_s}, J={j_s}")
        u = torch.randn(mk, j_s, generator=torch.Generator().manual_seed(3))
        t0 = time.perf_counter()
        u += pls.calculate_particle_update(u, 1e-12)  # warm-up (thread pool, page faults of the N x J temporaries)
        log(f"cpu baseline: warm-up step {time.perf_counter() - t0:.1f} s")
        reps = 3  # (at least three timed steps after one warm-up)
        t0 = time.perf_counter()
        for _ in range(reps):
            u += pls.calculate_particle_update(u, 1e-12)  # faithful: eigh(I) noise, dense diag @ U, full F and G
            log(f"cpu baseline: step done ({time.perf_counter() - t0:.1f} s since start of timing)")
        t_s = (time.perf_counter() - t0) / reps
        scale = step_flop(n_full, j_full) / step_flop(n_s, j_s)
        t_full = t_s * scale
        rss_gb = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0
        full = (n_s, j_s) == (n_full, j_full)
        # the *fair* CPU variant SURVEY 8(d) asks for next to the faithful one: A = V~^T k(Z,X) precomputed once, IID noise
        # (no eigh(I), no dense diag @ U) -- the same 4*N*Mk*J flop per step as the GPU path, so the ratio is not inflated
        # by the reference's avoidable work
        a_host = ob.scaled_eigenvectors.T @ ob.base_gram_induce_train  # (Mk, N), setup
        inv_lam = torch.reciprocal(ob.eigenvalues)[:, None]

        def fair_step():
            g = oc.calculate_cost_derivative(a_host.T @ u)
            u.add_(-1e-12 * (a_host @ g) - 1e-12 * inv_lam * u + math.sqrt(2e-12) * torch.randn(mk, j_s))

        fair_step()
        t0 = time.perf_counter()
        for _ in range(reps):
            fair_step()
        t_fair = (time.perf_counter() - t0) / reps
        fair_scale = (4.0 * n_full * mk * j_full) / (4.0 * n_s * mk * j_s)
        log(f"cpu baseline: fair variant {t_fair:.2f} s/step at N={n_s}, J={j_s}")
        del a_host
        on_device = None
        if torch.cuda.is_available() and 40.0 * n_full * j_full < 150e9:
            on_device = reference_style_on_device(O, ob, oc, x, y, z, ls, mk, j_full)
        how = ("measured at the full configuration, no extrapolation" if full else
               f"sub-sample N={n_s} of {n_full}, J={j_s} of {j_full}; scaled by the step's flop ratio x{scale:.2f} to "
               f"{t_full:.1f} s/step (the full step would take minutes of host time)")
        return {
            "value": 1.0 / t_full,
            "unit": "steps/s",
            "cores": cores,
            "cpu_model": cpu_model(),
            "kind": "port",
            "extrapolated": not full,
            "fair_variant": {"value": 1.0 / (t_fair * fair_scale), "unit": "steps/s", "s_per_step": t_fair * fair_scale,
                             "note": "A precomputed, IID torch.randn noise, row-scale prior drift: the GPU path's own 4*N*Mk*J "
                                     "flop per step on the host cores" + ("" if full else f" (sub-sample, scaled x{fair_scale:.2f})")},
            "reference_style_torch_rocm": on_device,
            "sample": f"oracle step (reference op order incl. per-step eigh(I) noise, full N x J F and G) at N={n_s}, J={j_s}: "
                      f"1 warm-up + {reps} timed steps, {t_s:.2f} s/step, peak RSS {rss_gb:.1f} GB; {how}",
        }
    finally:
        torch.set_default_dtype(prev)


def reference_style_on_device(O, ob, oc, x, y, z, ls, mk, j, dev=None):
    """What the reference's own ``.cuda()`` branches would do on this GPU (SURVEY 8d "reference-style torch-on-ROCm"): the
    oracle's step with its operands resident on the device, i.e. torch's rocBLAS matmuls in the reference's association,
    the full N x J F and G, the dense diag(1/lam) @ U -- and the noise exactly as samplers.py:27-44 makes it: eigh(I) and
    torch.normal on the HOST, then copied over.  Reported next to the CPU baseline; never part of ``value``."""
    dev = torch.device("cuda", torch.cuda.current_device()) if dev is None else dev
    n = x.shape[0]
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    try:
        kern = O.RBFARDKernel(ls.to(dev), 1.0)
        ob.base_gram_induce_train = torch.empty(z.shape[0], n, device=dev)
        zd, xd = z.to(dev), x.to(dev)
        for r0 in range(0, n, 16384):
            ob.base_gram_induce_train[:, r0:r0 + 16384] = kern(zd, xd[r0:r0 + 16384])
        ob.eigenvalues, ob.eigenvectors = ob.eigenvalues.to(dev), ob.eigenvectors.to(dev)
        ob.scaled_eigenvectors = ob.scaled_eigenvectors.to(dev)
        oc.y_train = y[:n].to(dev)
        pls = O.PLS(ob, oc)
        u = torch.randn(mk, j, generator=torch.Generator().manual_seed(3)).to(dev)

        def noise():  # samplers.py:27-44 under torch.cuda.is_available(): host eigh + host normals, products on the device
            lam, q = torch.linalg.eigh(torch.eye(mk))
            xi = torch.normal(mean=0.0, std=1.0, size=(mk, j))
            lam, q, xi = torch.clip(lam, 0, None).to(dev), q.to(dev), xi.to(dev)
            return torch.real(q @ torch.diag(torch.sqrt(lam)) @ xi)

        def one(with_noise: bool):
            e = noise() if with_noise else torch.zeros(mk, j, device=dev)
            u.add_(pls.calculate_particle_update(u, 1e-12, noise=e))

        one(True)
        sync()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            one(True)
        sync()
        t_step = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            one(False)
        sync()
        t_dev = (time.perf_counter() - t0) / reps
        log(f"reference-style torch-on-ROCm step: {t_step * 1e3:.1f} ms ({t_dev * 1e3:.1f} ms without the host-side noise)")
        return {"value": 1.0 / t_step, "unit": "steps/s", "ms_per_step": t_step * 1e3,
                "ms_per_step_device_part": t_dev * 1e3,
                "note": "oracle step with operands on the GPU (torch matmuls = rocBLAS, reference association, full N x J F "
                        "and G) + the reference's sampler (host eigh(I), host torch.normal, H2D copy); "
                        f"{reps} timed steps after 1 warm-up"}
    except Exception as exc:  # a comparator only: never fail the bench line over it
        log(f"reference-style torch-on-ROCm variant failed: {exc!r}")
        return {"value": None, "error": repr(exc)}
    finally:
        if dev.type == "cuda":
            torch.cuda.empty_cache()


def self_launch(n_ranks: int) -> int:
    """Run this script as ``n_ranks`` ranks under torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1) in a
    child process and return its exit code.  stdout/stderr are inherited, so rank 0's JSON line is this process's too."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    log(f"--gpus {n_ranks} without a launcher: starting {n_ranks} ranks under torch.distributed.run (port {port})")
    return subprocess.run(cmd, env=env).returncode


def profiler_grid_line(args):
    """`--config profiler-grid`: the reference's only timing harness (experiments/profiler/main.py:50-82, :141-169; grid of
    profiler/config.yaml:1-22) -- one timed block per grid point = basis construction + cost + PLS + particle initialisation +
    T x `particles += pls.calculate_particle_update(particles, 1e-10)` through the drop-in API -- with the CPU oracle timed in
    the same run.  Not BASELINE.json's metric (that is the default --config c2): printed as its own JSON line."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import profiler_grid

    sys.argv = [sys.argv[0], "--repeats", "5"] + (["--out", os.environ["PLS_PROFILER_GRID_OUT"]] if os.environ.get("PLS_PROFILER_GRID_OUT") else [])
    summary = profiler_grid.main()
    rows = summary["rows"]
    default = [r for r in rows if (r["n"], r["m"], r["t"], r["j"]) == (100, 10, 10, 100)]
    line = {
        "metric": "construct + T steps of the reference's profiler protocol (experiments/profiler/main.py:50-82), seconds per block",
        "config": {"workload": "experiments/profiler/config.yaml grid: N, J 100..1000, M, T 10..100 around N=100, M=10, T=10, J=100; "
                               "Gaussian/identity and Bernoulli/sigmoid", "dtype": "f64"},
        "n_gpus": 1, "grid_points": len(rows), "cpu_threads": summary["cpu_threads"], "cpu_model": summary["cpu_model"],
        "default_point": {r["cost"]: {"gpu": r["gpu"], "cpu_oracle": r["cpu"], "gpu_over_cpu_total": r["gpu_total_over_cpu_total"]}
                          for r in default},
        "gpu_over_cpu_total": {"min": min(r["gpu_total_over_cpu_total"] for r in rows),
                               "max": max(r["gpu_total_over_cpu_total"] for r in rows)},
        "points_where_the_gpu_block_is_slower_than_the_cpu_oracle": summary["points_where_the_gpu_block_is_slower_than_the_cpu_oracle"],
        "higher_is_better": False, "data": "synthetic (the reference's Curve1 regression data)",
    }
    print(json.dumps(line), flush=True)


def reference_scale_block(pkg, iterations: int):
    """`train_pls` (experiments/trainers.py:139-162) at the sizes the reference's own experiments and profiler run at
    (experiments/curves/*/config.yaml, experiments/profiler/config.yaml): microseconds per iteration for both bases and a cost
    with / without the Gaussian algebra -- the launch-bound regime of csrc/small_rank_step.h and csrc/ipb_prep.h.  Median of
    three runs of `iterations` iterations each; extra to BASELINE.json's metric."""
    import statistics

    from projected_langevin_sampling_amd.basis import InducingPointBasis, OrthonormalBasis
    from projected_langevin_sampling_amd.costs import BernoulliCost, GaussianCost
    from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction, SigmoidLinkFunction
    from projected_langevin_sampling_amd.trainers import train_pls

    rows = []
    for (n, m, j, d) in ((100, 10, 64, 1), (1000, 32, 100, 1), (4096, 128, 512, 4)):
        g = torch.Generator().manual_seed(0)
        x = torch.rand(n, d, generator=g, dtype=torch.float64) * 2 - 1
        z = x[torch.randperm(n, generator=g)[:m]].clone()
        y = torch.sin(2.0 * x.sum(dim=1)) + 0.1 * torch.randn(n, generator=g, dtype=torch.float64)
        kern = pkg.PLSKernel(pkg.ARDKernel(torch.full((d,), 0.5, dtype=torch.float64), 1.0), z)
        row = {"n": n, "m": m, "j": j}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # (the Cholesky jitter note of a random Z)
            bases = (("orthonormal", OrthonormalBasis(kern, z, x, 1e-8, verbose=False)), ("inducing_point", InducingPointBasis(kern, z, y[:m], x)))
        for bname, basis in bases:
            for cname, cost in (("gaussian_identity", GaussianCost(0.1, y, IdentityLinkFunction())),
                                ("bernoulli_sigmoid", BernoulliCost((y > 0).double(), SigmoidLinkFunction()))):
                pls = pkg.PLS(basis, cost)
                u = (1.0 + 0.1 * torch.randn(basis.approximation_dimension, j, generator=g, dtype=torch.float64)).cuda()
                eta_t = 1e-13  # (timing only: inside the stability bound of a random Z's stiffest prior mode, so no run stops early)
                train_pls(pls, u.clone(), 30, eta_t, 1e9)
                runs = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    _, e = train_pls(pls, u.clone(), iterations, eta_t, 1e9)
                    torch.cuda.synchronize()
                    assert len(e) == iterations, f"reference_scale: the run stopped after {len(e)} of {iterations} iterations"
                    runs.append((time.perf_counter() - t0) / iterations * 1e6)
                row[f"{bname}/{cname}"] = round(statistics.median(runs), 2)
        rows.append(row)
    return {"unit": "us per train_pls iteration (step + energy + early-stop test)", "iterations": iterations, "rows": rows,
            "protocol": "experiments/trainers.py:139-162 through the drop-in train_pls; sizes of experiments/curves/*/config.yaml and "
                        "experiments/profiler/config.yaml; median of 3 runs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS) + ["profiler-grid"],
                    help="c2 = configs[1] of BASELINE.json (the headline); profiler-grid = the reference's own timing protocol "
                         "(experiments/profiler/config.yaml) through tools/profiler_grid.py, GPU beside the CPU oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workspace-gb", type=float, default=8.0, help="cap of the per-step G-chunk workspace")
    ap.add_argument("--converge-steps", type=int, default=4000,
                    help="step cap of the wall-clock-to-converged-energy run (0 = skip it)")
    ap.add_argument("--ipb-steps", type=int, default=3, help="timed steps of the inducing-point-basis extra (0 = skip)")
    ap.add_argument("--select-inducing", action="store_true", help="also time the greedy inducing-point selection (setup)")
    ap.add_argument("--eigh-device", default="cuda", choices=["cpu", "cuda"],
                    help="where the one-time eigh of k(Z,Z)/M runs (cuda = torch.linalg.eigh on the device the matrix lives on, "
                         "like the reference's .cuda() branch; cpu = the reference's host LAPACK call: 0.9 s at M = 1024 and 21 s "
                         "at M = 4096 on the GPU box's host share, against 0.03 / 0.15 s)")
    ap.add_argument("--sustained-steps", type=int, default=300,
                    help="like-for-like steps of the `sustained` block (>= 10 s at configs[1]; 0 = skip)")
    ap.add_argument("--profiler-steps", type=int, default=100,
                    help="T of the reference's profiler protocol: construction + T steps timed as one block (0 = skip)")
    ap.add_argument("--reference-scale-iterations", type=int, default=1000,
                    help="iterations per run of the `reference_scale` block: train_pls at the reference's experiment sizes (0 = skip)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="development aid: run rank 0's particle shard of an N-GPU job on ONE GPU (no collectives); "
                         "the JSON line is marked emulated and is not a scaling result")
    args = ap.parse_args()
    if args.config == "profiler-grid":
        return profiler_grid_line(args)
    cfg = CONFIGS[args.config]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a CHILD process, before this
        # process has touched the GPU (nothing above initialises HIP), and relay rank 0's JSON line and the exit code
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # one rank per GPU; (the modulo only matters for the development rehearsal of several ranks on a one-GPU box)
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    import torch.distributed as dist

    if world > 1 or "RANK" in os.environ:
        # RCCL ("nccl") in production; PLS_BENCH_BACKEND=gloo lets a one-GPU box rehearse the multi-rank control flow
        backend = os.environ.get("PLS_BENCH_BACKEND", "nccl")
        extra = {"device_id": torch.device("cuda", device_index)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **extra)

    import projected_langevin_sampling_amd as pkg
    from projected_langevin_sampling_amd import distributed as D
    from projected_langevin_sampling_amd.basis import NoiseSpec, OrthonormalBasis
    from projected_langevin_sampling_amd.costs import GaussianCost, PoissonCost
    from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction, SquareLinkFunction

    L = pkg._lib
    x, z, y, ls = make_data(cfg)
    log("synthetic data ready")
    setup = {}
    t0 = time.perf_counter()
    torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    setup["hip_context_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    _w = torch.ones(64, 64, dtype=torch.float64, device="cuda")
    pkg._ops.gemm_tn(_w, _w)  # the first libplship launch loads the library's code object onto the device
    torch.cuda.synchronize()
    setup["code_object_load_first_launch_s"] = time.perf_counter() - t0
    t_setup = time.perf_counter()
    kernel = pkg.PLSKernel(pkg.ARDKernel(ls, 1.0), z)
    basis = OrthonormalBasis(kernel, z, x, eigenvalue_threshold=cfg.get("threshold", 0.0), verbose=False, keep_gram=False,
                             eigh_device=args.eigh_device, setup_times=setup, group=True if world > 1 else None)
    basis.workspace_bytes = int(args.workspace_gb * (1 << 30))
    setup["eigh_device"] = args.eigh_device  # (cuda: the first call of the process also loads the solver library, ~0.2 s)
    mk = basis.approximation_dimension
    if cfg["cost"] == "poisson":
        cost = PoissonCost(y, SquareLinkFunction())
    elif cfg["cost"] == "bernoulli":
        from projected_langevin_sampling_amd.costs import BernoulliCost
        from projected_langevin_sampling_amd.link_functions import SigmoidLinkFunction
        cost = BernoulliCost(y, SigmoidLinkFunction())
    else:
        cost = GaussianCost(cfg["obs"], y, IdentityLinkFunction())
    j_total = cfg["j"]
    shard_world = args.emulate_world if (args.emulate_world > 1 and world == 1) else world
    j0, j1 = D.attach_shard(basis, j_total, rank, shard_world)
    j_loc = j1 - j0
    eta = min(cfg["eta"], 0.5 * basis.eigenvalues.min().item())  # eta / lambda_min < 2 keeps the prior drift stable
    # identical initial particles for any GPU count: one seeded (Mk, J) draw, every rank keeps its columns
    u_full = torch.normal(0.0, 1.0, size=(mk, j_total), generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    ping = u_full[:, j0:j1].contiguous().cuda()
    pong = torch.empty_like(ping)
    del u_full
    rho_max = None
    if cfg["cost"] == "gaussian":
        t0 = time.perf_counter()
        basis.prepare_gaussian(cost.y_device())
        torch.cuda.synchronize()
        setup["gaussian_constants_B_c_s"] = time.perf_counter() - t0
        # Euler-Maruyama is stable for eta * rho < 2, rho = largest eigenvalue of the drift Jacobian B/sigma2 + Lambda^-1;
        # the reference finds a usable step by search (experiments/runners.py:356-433), here it is read off the spectrum:
        # block power iteration on the device (the contraction is pls_gemm_tn), Rayleigh quotient at the end.  (Round 2
        # took a full host eigvalsh of the 1024 x 1024 Jacobian to read off this one number: 0.4 s of the setup.)
        t0 = time.perf_counter()
        inv_lam, inv_obs = (1.0 / basis.eigenvalues)[:, None], 1.0 / cfg["obs"]
        v = torch.randn(mk, 4, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
        jv = None
        for it in range(60):
            v, _ = torch.linalg.qr(v)
            jv = pkg._ops.gemm_tn(basis._B, v.contiguous()) * inv_obs + inv_lam * v
            if it < 59:
                v = jv
        rho_max = float(torch.linalg.eigvalsh(v.T @ jv).max().item())  # (4 x 4 Ritz values)
        setup["step_size_power_iteration_s"] = time.perf_counter() - t0
        eta = min(eta, 1.0 / rho_max)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    setup = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in setup.items()}
    log(f"basis ready (M_k = {mk}), setup {t_setup:.2f} s: {setup}")

    # ---- the kernel build k(Z, X) on the roofline (north_star: "achieved HBM GB/s on the kernel build"): the launch of the
    # setup again, five times, each bracketed by HIP events on the launch stream (the library timeline) ----
    gram_roofline = None
    if rank == 0:
        gx, gz = x.cuda(), z.cuda()
        kernel.base_kernel(x1=gz, x2=gx)  # warm-up (allocation of the 8 N M bytes)
        torch.cuda.synchronize()
        with L.Timeline(capacity=64) as tlg:
            for _ in range(5):
                kzx = kernel.base_kernel(x1=gz, x2=gx)
        gsum = tlg.summary().get("kernel_gram")
        if gsum and gsum["launches"]:
            gbytes = 8.0 * cfg["m"] * cfg["n"]
            gms = gsum["total_ms"] / gsum["launches"]
            gram_roofline = {
                "kernel": "kernel_gram_kernel<RBF-ARD, D> (k(Z,X): M x N float64 written once, inputs 8 (N + M) D bytes)",
                "bound": "hbm", "achieved": gbytes / (gms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": gbytes / (gms * 1e-3) / 1e9 / 8000.0, "bytes_per_launch": gbytes, "avg_launch_ms": gms,
                "launches": gsum["launches"], "D": int(x.shape[1]),
                "valu_busy_frac": 0.64, "store_stream_frac": gbytes / (gms * 1e-3) / 1e9 / 5350.0,
                "bound_note": "co-limited: SQ_ACTIVE_INST_VALU / cycles = 0.64 (40 fp64 vector instructions per entry, "
                              "profiles/r05_gram_pmc.txt) next to 0.72-0.79 of the rate a pure store stream of this geometry "
                              "reaches (5.1-5.6 TB/s, store_stream_frac is against their mean); `bound` names the larger",
                "note": "algorithmic bytes (8 N M written) / launch duration from HIP events around each launch; a pure store "
                        "stream of this geometry reaches 5.1-5.6 TB/s on the chip (profiles/r02_gram_store_sweep.txt)",
            }
        del kzx, gx, gz
        torch.cuda.empty_cache()

    def settle(seconds: float = 1.0):
        torch.cuda.synchronize()
        time.sleep(seconds)

    def barrier():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    region_ms = [0.0]  # device time of the last run's timed region (HIP events on the launch stream)
    rank_dt = [0.0]    # wall time of the last run's timed region on every rank (rank order)

    def run(force_generic: bool, steps: int, warmup: int, timeline: bool):
        nonlocal ping, pong
        step_id = [0]

        def one():
            nonlocal ping, pong
            basis.fused_step(cost, ping, eta, out=pong, new_state=True, force_generic=force_generic,
                             noise=NoiseSpec(seed=1234, step=step_id[0], j_offset=j0))
            step_id[0] += 1
            ping, pong = pong, ping

        for _ in range(warmup):
            one()
        barrier()
        tl = L.Timeline(capacity=max(64, steps * 64)) if timeline else None
        if tl:
            tl.__enter__()
        # HIP events on the launch stream (libplship launches on torch's current stream) around the whole timed region
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            one()
        ev1.record()
        barrier()
        dt = time.perf_counter() - t0
        region_ms[0] = ev0.elapsed_time(ev1)
        if tl:
            tl.__exit__(None, None, None)
        rank_dt[:] = [dt]
        if dist.is_initialized():
            every = [None] * world
            dist.all_gather_object(every, dt)  # (after the barrier: not in the timed region)
            rank_dt[:] = every
            dt = max(every)
        assert torch.isfinite(ping).all().item(), "particles diverged"
        return dt, (tl.summary() if tl else {})

    n, m = cfg["n"], mk
    # ---- like-for-like path (headline) ----
    dt, tl = run(force_generic=True, steps=args.steps, warmup=args.warmup, timeline=True)
    ms_per_step = dt / args.steps * 1e3
    value = args.steps / dt
    headline_rank_ms = [d / args.steps * 1e3 for d in rank_dt]
    log(f"like-for-like path: {ms_per_step:.2f} ms/step")
    # dominant kernels of the step: the two GEMM launches, or (rank <= 128) the one fused small-rank launch
    dom = ("small_rank_drift",) if "small_rank_drift" in tl else ("gemm_cost_deriv", "gemm_store")
    gemm_ms = sum(tl[k]["total_ms"] for k in dom if k in tl)
    gemm_launches = sum(tl[k]["launches"] for k in dom if k in tl)
    flop_per_step_rank = 4.0 * n * m * j_loc
    achieved = flop_per_step_rank * args.steps / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    # HBM bytes per launch of the same two kernels on the same workload come from separate rocprofv3 --pmc passes
    # (FETCH_SIZE doubled, WRITE_SIZE as is: MI355X_MICROARCH.md "HBM"; tools/profile_bench.sh + tools/summarize_prof.py).
    # A counter pass cannot run inside this process, so the committed summary is used ONLY when it was collected from
    # exactly the kernel sources of this tree (tools/source_hash.py stamp); otherwise traffic stays null.
    traffic, traffic_src = None, None
    if args.config == "c2" and world == 1 and shard_world == 1:
        import glob

        from tools.source_hash import kernel_source_hash

        here = kernel_source_hash()
        traffic_src = "no PMC summary under profiles/ was collected from the current kernel sources"
        # (rNN_pmc_summary.json only: that name is the configs[1] workload on one GPU; other configs and shard runs carry a tag)
        import re

        for prof in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
            if not re.fullmatch(r"r\d+_pmc_summary\.json", os.path.basename(prof)):
                continue
            pm = json.load(open(prof))
            if pm.get("_kernel_source_hash") != here:
                continue
            ks = [pm.get("gemm_tn_f64<128x128>::EpiGaussDeriv", pm.get("gemm_tn_f64<128x128>::EpiCostDeriv", {})),
                  pm.get("gemm_tn_f64<128x128>::EpiStore", {})]
            if all("hbm_bytes_per_launch" in k for k in ks):
                traffic = sum(k["hbm_bytes_per_launch"] for k in ks) / len(ks)
                traffic_src = f"profiles/{os.path.basename(prof)} (kernel sources {here[:12]})"
                break
    step_tflops = flop_per_step_rank / (ms_per_step * 1e-3) / 1e12  # whole step: every launch and every gap of the timed region
    roofline = {
        "kernel": ("small_rank_kernel<KB,drift> (F, d cost/d f and back-projection fused; N x J intermediates never written)"
                   if dom[0] == "small_rank_drift" else
                   "gemm_tn_f64_kernel<128,128,64,64,16,*> (cost-derivative + back-projection launches of the step)"),
        "bound": "mfma",
        "achieved": achieved,
        "peak": FP64_MFMA_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
        "frac_note": "algorithmic flop of the step's dominant launches / their summed durations (HIP events around every launch)",
        "step_achieved": step_tflops,
        "step_frac": step_tflops / FP64_MFMA_PEAK_TFLOPS,
        "step_frac_note": "the same flop / ms_per_step: update kernel, launch gaps and host overhead of the timed region included",
        "algorithmic_bytes_per_step": 8.0 * (n * m + 2 * m * j_loc),
        "algorithmic_bytes_note": "SURVEY 8(d): A once, U in, U out; the two-GEMM path below also writes and re-reads G",
        "traffic": traffic,
        "traffic_unit": "bytes per launch (fabric-side L2 misses incl. Infinity-Cache hits)",
        "traffic_source": traffic_src,
        "algorithmic_bytes_per_launch": (8.0 * (n * m + 2 * m * j_loc) if dom[0] == "small_rank_drift"
                                         else 8.0 * (n * m + m * j_loc + 2 * n * j_loc) / 2.0),
        "flop_per_launch": flop_per_step_rank / max(gemm_launches / args.steps, 1),
        "avg_launch_ms": gemm_ms / max(gemm_launches, 1),
        "launches_per_step": gemm_launches / args.steps,
        "per_kernel_ms": {k: round(v["avg_ms"], 4) for k, v in tl.items()},
    }
    out = {
        "metric": "Langevin steps/sec",
        "value": value,
        "unit": "steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        **({"emulated_world": shard_world} if shard_world != world else {}),
        "config": {"workload": cfg["workload"], "N": n, "M": cfg["m"], "M_k": mk, "J": j_total, "J_per_gpu": j_loc,
                   "path": "like-for-like: F=A^T U -> d cost/d f -> A G, 4*N*M*J flop/step", "step_size": eta,
                   "parallelism": f"J-sharded x{world}, no per-step collective", "setup_s": round(t_setup, 2),
                   "setup_breakdown": setup},
        "roofline": roofline,
    }
    if dist.is_initialized():
        # who ran: the process group as the ranks see it, and every rank's own clock over the headline region (`ms_per_step` is
        # the slowest rank's)
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "local_rank": local_rank, "device_index": device_index,
                                      "device": torch.cuda.get_device_name(device_index),
                                      "uuid": str(getattr(torch.cuda.get_device_properties(device_index), "uuid", ""))})
        out["backend"] = str(dist.get_backend())
        out["ranks_seen"] = seen
        out["per_rank_ms_per_step"] = {"min": min(headline_rank_ms), "max": max(headline_rank_ms),
                                       "slowest_rank": int(max(range(world), key=lambda r: headline_rank_ms[r])),
                                       "all": [round(v, 4) for v in headline_rank_ms]}
    if gram_roofline is not None:
        out["roofline_gram"] = gram_roofline
    # ---- sustained: the same like-for-like step for >= 10 s, in windows of 50 steps (does the clock hold?) ----
    if args.sustained_steps >= 50:
        # 50-step windows at configs[1] (300 steps = 13.4 s); a configuration whose step takes seconds (configs[4]: 1.8 s) gets
        # the same six windows over ~20 s instead of 300 steps = 9 minutes
        nwin = max(args.sustained_steps // 50, 1)
        win = max(1, min(50, int(round(20.0 / nwin / (ms_per_step * 1e-3)))))
        run(force_generic=True, steps=2, warmup=0, timeline=False)
        windows = []
        t_all = time.perf_counter()
        for w in range(nwin):
            dtw, _ = run(force_generic=True, steps=win, warmup=0, timeline=False)
            windows.append(dtw / win * 1e3)
            log(f"sustained window {w + 1}/{nwin}: {windows[-1]:.2f} ms/step")
        t_all = time.perf_counter() - t_all
        out["sustained"] = {
            "steps": nwin * win, "wall_s": t_all, "ms_per_step": sum(windows) / len(windows),
            "steps_per_s": nwin * win / (sum(windows) * win * 1e-3),
            "window_ms_per_step": [round(w, 3) for w in windows], "window_steps": win,
            "frac": flop_per_step_rank / (sum(windows) / len(windows) * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            "note": "like-for-like steps back to back; every window is its own timed region (barrier + sync on both sides)",
        }
        log(f"sustained: {nwin * win} steps, {out['sustained']['ms_per_step']:.2f} ms/step, windows {windows[0]:.2f} .. {windows[-1]:.2f}")
    # ---- Gaussian algebraic fast path, same run ----
    if cfg["cost"] == "gaussian":
        fsteps = max(args.steps * 10, 50)
        # every section starts from a settled chip, like the headline run does after the setup: after the 45 ms-per-step
        # phase above (0.6 s of sustained full-rate MFMA) the power controller holds the clock lower for SECONDS -- these
        # same launches measured 0.282 ms one second after it and 0.261-0.267 ms from a rested chip
        # (tools/fastpath_probe.py; the train_pls section further down, later in the same run: 0.267 ms)
        settle(4.0)
        # two passes: per-launch HIP events (two hipEventRecord per launch: ~6 us of bubbles per step, nothing at 45 ms per
        # step, a tenth of a 50 us step on an 8-GPU shard) for the kernel's own duration, then the timed region without them
        _, tlf = run(force_generic=False, steps=fsteps, warmup=max(args.warmup, 5), timeline=True)
        dtf, _ = run(force_generic=False, steps=fsteps, warmup=2, timeline=False)
        log(f"gaussian fast path: {dtf / fsteps * 1e3:.3f} ms/step")
        k = tlf.get("gemm_langevin_gaussian", {"total_ms": 0.0, "launches": 0, "avg_ms": 0.0})
        fl = 2.0 * m * m * j_loc
        # per launch: the un-instrumented timed region (launch gaps included) -- the per-launch event pairs of the
        # timeline pass put bubbles between 0.26 ms launches and read 8 % long (rocprofv3's kernel duration agrees
        # with the region, profiles/)
        ach = fl / (region_ms[0] / fsteps * 1e-3) / 1e12
        out["gaussian_fast_path"] = {
            "value": fsteps / dtf, "unit": "steps/s", "steps": fsteps, "ms_per_step": dtf / fsteps * 1e3,
            "note": "B = A A^T, c = A y precomputed once (setup); per step 2*Mk^2*J flop in ONE fused kernel "
                    "(contraction + prior drift + Philox noise + axpy)",
            "roofline": {"kernel": "gemm_tn_f64_kernel<...,EpiLangevinGaussian>", "bound": "mfma", "achieved": ach,
                         "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                         "traffic": None, "region_ms_per_launch": region_ms[0] / fsteps,
                         "frac_note": "2*Mk^2*J flop / (HIP-event time of the whole timed region / launches): gaps between "
                                      "launches included",
                         "event_pair_avg_launch_ms": k["avg_ms"],
                         "event_pair_note": "a separate pass with a HIP event pair around every launch (the library "
                                            "timeline): reads long at this launch length, kept for reference"},
        }
        # sustained: >= 10 s of the same fused steps back to back, NO pause in front (the figure above is what a rested
        # chip does in a 50-100 launch burst; this is what a long training run sees)
        if args.sustained_steps >= 50:
            per = out["gaussian_fast_path"]["ms_per_step"] * 1e-3
            wsteps = max(200, int(1.0 / per))  # ~1 s per window
            wins = []
            t_all = time.perf_counter()
            for w in range(10):
                dtw, _ = run(force_generic=False, steps=wsteps, warmup=0, timeline=False)
                wins.append((dtw / wsteps * 1e3, region_ms[0] / wsteps))
            t_all = time.perf_counter() - t_all
            mean_ms = sum(w[0] for w in wins) / len(wins)
            mean_region = sum(w[1] for w in wins) / len(wins)
            out["gaussian_fast_path"]["sustained"] = {
                "steps": 10 * wsteps, "wall_s": t_all, "ms_per_step": mean_ms, "region_ms_per_launch": mean_region,
                "window_ms_per_step": [round(w[0], 5) for w in wins], "window_steps": wsteps,
                "frac": fl / (mean_region * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                "frac_wall": fl / (mean_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                "note": "ten windows of ~1 s of fused steps back to back with no settle; frac from the HIP-event time of each "
                        "window's region, frac_wall from its wall-clock (barrier + sync on both sides)",
            }
            log(f"gaussian fast path sustained: {mean_ms:.4f} ms/step over {10 * wsteps} steps "
                f"(windows {wins[0][0]:.4f} .. {wins[-1][0]:.4f})")
        # the same K steps as a captured hipGraph (10 steps per replay): what the launch overhead costs on small shards
        from projected_langevin_sampling_amd.graph import CapturedSteps

        # (a replay costs 10-16 us of launch: narrow shards, whose step is tens of microseconds, get more steps per replay)
        gsteps = 10 if j_loc >= 4096 else 50
        run_g = CapturedSteps(pkg.PLS(basis, cost), ping.clone(), eta, steps_per_replay=gsteps, seed=4321)
        run_g.replay(2)
        barrier()
        t0 = time.perf_counter()
        reps = max(fsteps // gsteps, 20)
        run_g.replay(reps)
        barrier()
        dtg = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dtg], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dtg = tt.item()
        out["gaussian_fast_path"]["hipgraph_steps_per_replay"] = gsteps
        out["gaussian_fast_path"]["hipgraph_ms_per_step"] = dtg / (reps * gsteps) * 1e3
        out["gaussian_fast_path"]["hipgraph_steps_per_s"] = reps * gsteps / dtg
        del run_g
    # ---- wall-clock to converged energy: the reference's train_pls loop (step, energy, early stop) ----
    if cfg["cost"] == "gaussian" and args.converge_steps > 0:
        from projected_langevin_sampling_amd.trainers import train_pls

        eta_c = 1.0 / rho_max
        patience = 100 * eta_c
        u0 = torch.normal(0.0, 1.0, size=(mk, j_total), generator=torch.Generator().manual_seed(0), dtype=torch.float64)
        particles = u0[:, j0:j1].contiguous().cuda()
        del u0
        pls = pkg.PLS(basis, cost)
        torch.manual_seed(0)
        # J-sharded run: the stop rule needs the mean energy over ALL ranks' particles after every step.  EnergyMean hands the
        # ranks' local sums (which arrive in pinned host memory from the step launches) across on the host -- a shared-memory
        # board between the processes of this node -- so the pipelined loop keeps every GPU's queue full
        reduce_fn = D.EnergyMean(j_total) if world > 1 else None
        # warm-up, untimed like the headline's: the loop's one-time allocations (second particle buffer, pinned energy
        # sums, workspace) cost tens of milliseconds against 0.3 s of iterations and moved the per-iteration figure by
        # 0.04 ms from one box to the next
        train_pls(pls, particles.clone(), 3, eta_c, patience, energy_reduce=reduce_fn)
        torch.manual_seed(0)
        barrier()
        t0 = time.perf_counter()
        particles, energies = train_pls(pls, particles, args.converge_steps, eta_c, patience, energy_reduce=reduce_fn)
        barrier()
        wall = time.perf_counter() - t0
        out["converged_energy"] = {
            "wall_s": wall, "steps": len(energies), "step_cap": args.converge_steps, "stopped_early": len(energies) < args.converge_steps,
            "step_size": eta_c, "patience_simulated_time": patience, "first_energy": energies[0] if energies else None,
            "final_energy": energies[-1] if energies else None, "ms_per_step_with_energy": wall / max(len(energies), 1) * 1e3,
            "loop": "train_pls (experiments/trainers.py:139-162): fused step + energy (.item() sync) + EarlyStopper every step; "
                    "3 untimed warm-up iterations first",
            "relaxation_rate_lower_bound": 1.0 / basis.eigenvalues.max().item(), "stiffness_max": rho_max,
            "energy_exchange": None if reduce_fn is None else ("shared-memory board between the ranks' host processes (distributed.EnergyMean)"
                                                              if reduce_fn.uses_board else "blocking all-reduce of one double per iteration"),
        }
        log(f"train_pls: {len(energies)} steps in {wall:.2f} s, energy {energies[0]:.4g} -> {energies[-1]:.4g}")
        if world == 1:
            # same loop with 32 steps + energies per hipGraph replay (extension; pays off when the step is launch-bound)
            from projected_langevin_sampling_amd.trainers import train_pls_captured
            u0 = torch.normal(0.0, 1.0, size=(mk, j_total), generator=torch.Generator().manual_seed(0), dtype=torch.float64)
            particles = u0[:, j0:j1].contiguous().cuda()
            del u0
            barrier()
            t0 = time.perf_counter()
            _, energies_c = train_pls_captured(pls, particles, args.converge_steps, eta_c, patience, steps_per_replay=32, seed=1)
            barrier()
            wall_c = time.perf_counter() - t0
            out["converged_energy"]["captured"] = {
                "wall_s": wall_c, "steps": len(energies_c), "ms_per_step_with_energy": wall_c / max(len(energies_c), 1) * 1e3,
                "note": "train_pls_captured: 32 steps + energies per hipGraph replay (wall includes the capture itself)",
            }
            log(f"train_pls_captured: {len(energies_c)} steps in {wall_c:.3f} s")
    # ---- train_pls iteration cost for the other costs: step + energy, pipelined (energy as a by-product of the step's
    # own F) against the plain loop (step, then a separate energy pass as the reference does) ----
    if cfg["cost"] != "gaussian" and args.converge_steps > 0:
        from projected_langevin_sampling_amd.trainers import train_pls
        iters = min(args.converge_steps, 20)
        res = {}
        for mode in ("pipelined", "plain"):
            u0 = torch.normal(0.0, 1.0, size=(mk, j_total), generator=torch.Generator().manual_seed(0), dtype=torch.float64)
            particles = u0[:, j0:j1].contiguous().cuda()
            del u0
            pls = pkg.PLS(basis, cost)
            if mode == "plain":
                basis.supports_input_energy = lambda c: False
            # (sharded runs: the pipelined loop with the ranks' host-side exchange, the plain one with a blocking all-reduce)
            reduce_fn = None if world == 1 else (D.EnergyMean(j_total) if mode == "pipelined" else (lambda e: D.mean_over_particles(e, j_total)))
            try:
                train_pls(pls, particles.clone(), 2, eta, 1e30, energy_reduce=reduce_fn)  # warm-up
                barrier()
                t0 = time.perf_counter()
                _, energies = train_pls(pls, particles, iters, eta, 1e30, energy_reduce=reduce_fn)
                barrier()
                res[mode] = ((time.perf_counter() - t0) / max(len(energies), 1) * 1e3, len(energies))
            finally:
                if mode == "plain":
                    del basis.supports_input_energy
        out["train_pls_iteration"] = {
            "ms_pipelined": res["pipelined"][0], "ms_plain": res["plain"][0], "iterations": res["pipelined"][1],
            "loop": "train_pls (experiments/trainers.py:139-162): step + energy (.item() sync) + EarlyStopper every step",
            "note": "pipelined: the step launch also emits the energy of its input particles (same F); plain: separate energy pass",
        }
        log(f"train_pls iteration: pipelined {res['pipelined'][0]:.3f} ms, plain {res['plain'][0]:.3f} ms")
    # ---- setup extra: greedy conditional-variance inducing-point selection at this N, M ----
    if rank == 0 and args.select_inducing:
        import numpy as np
        from projected_langevin_sampling_amd.inducing_point_selectors import ConditionalVarianceInducingPointSelector

        np.random.seed(0)
        sel = ConditionalVarianceInducingPointSelector()
        sel(x[:4096], 16, kernel.base_kernel)  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, idx_sel = sel(x, cfg["m"], kernel.base_kernel)
        torch.cuda.synchronize()
        t_sel = time.perf_counter() - t0
        alg_bytes = 8.0 * n * cfg["m"] * (cfg["m"] - 1) / 2 * 1.0  # the c[:i, n] reads of the recurrence dominate
        out["inducing_point_selection"] = {
            "seconds": t_sel, "selected": int(idx_sel.shape[0]), "algorithmic_GB": alg_bytes / 1e9,
            "achieved_GBps": alg_bytes / t_sel / 1e9, "bound": "hbm",
            "note": "partial pivoted Cholesky of k(X,X): M iterations of an N-long recurrence, 3 launches each, no host sync",
        }
        log(f"inducing-point selection: {t_sel:.3f} s")
    # ---- inducing-point basis: same data, same cost, one rank's shard ----
    if args.ipb_steps > 0:
        from projected_langevin_sampling_amd.basis import InducingPointBasis

        t_ipb = time.perf_counter()
        ipb = InducingPointBasis(kernel, z, y[: z.shape[0]], x)
        ipb.workspace_bytes = basis.workspace_bytes
        D.attach_shard(ipb, j_total, rank, shard_world)
        torch.cuda.synchronize()
        t_ipb = time.perf_counter() - t_ipb
        a = torch.normal(0.0, 1.0, size=(cfg["m"], j_loc), generator=torch.Generator().manual_seed(1), dtype=torch.float64).cuda()
        b = torch.empty_like(a)
        eta_i = 1e-9  # timing only: k(Z,Z)^-1 is stiff, the dynamics are not the point here
        for w in range(1 + args.ipb_steps):
            if w == 1:
                barrier()
                t0 = time.perf_counter()
            ipb.fused_step(cost, a, eta_i, out=b, new_state=True, noise=NoiseSpec(seed=7, step=w, j_offset=j0), force_generic=True)
            a, b = b, a
        barrier()
        dti = (time.perf_counter() - t0) / args.ipb_steps
        dti_fast = dti_fast3 = dti_fast_e = None
        if cfg["cost"] == "gaussian":  # the M x M x J algebraic path of the inducing-point basis (B = Kzx Kxz)
            reps = 50 if j_loc <= 2048 else 20

            def per_call(step0, **kw):
                nonlocal a, b
                for w in range(2 + reps):
                    if w == 2:
                        barrier()
                        t0 = time.perf_counter()
                    ipb.fused_step(cost, a, eta_i, out=b, new_state=True, noise=NoiseSpec(seed=7, step=step0 + w, j_offset=j0), **kw)
                    a, b = b, a
                barrier()
                return (time.perf_counter() - t0) / reps

            dti_fast = per_call(100)
            L.check(L.load().pls_set_option(L.OPT_IPB_STEP_OPERATOR, 0))
            try:
                dti_fast3 = per_call(300)
            finally:
                L.check(L.load().pls_set_option(L.OPT_IPB_STEP_OPERATOR, 1))
            e_in = torch.empty(j_loc, dtype=torch.float64, device="cuda")
            dti_fast_e = per_call(500, input_energy=e_in)
        dti_white = None
        if cfg["cost"] == "gaussian":  # a loop that keeps S = Lc^-1 U between the steps: one contraction per step
            sw = ipb.whiten(a)
            sb = torch.empty_like(sw)
            reps = 50
            for w in range(2 + reps):
                if w == 2:
                    barrier()
                    t0 = time.perf_counter()
                ipb.whitened_step(cost, sw, eta_i, out=sb, new_state=True, noise=NoiseSpec(seed=7, step=200 + w, j_offset=j0))
                sw, sb = sb, sw
            barrier()
            dti_white = (time.perf_counter() - t0) / reps
            del sw, sb
            # the reference's train loop on this basis (trainers.py:139-162): whitened between the steps, energy every step
            from projected_langevin_sampling_amd.trainers import train_pls as _train

            torch.manual_seed(1)
            _train(pkg.PLS(ipb, cost), a.clone(), 10, eta_i, 1e30)
            barrier()
            t0 = time.perf_counter()
            _, e_ipb = _train(pkg.PLS(ipb, cost), a, 200, eta_i, 1e30)
            barrier()
            dti_train = (time.perf_counter() - t0) / max(len(e_ipb), 1)
        # what the timed matrix is: the parity suite holds the inducing-point step to 1e-9 up to cond(K_ZZ) ~ 1e8; these
        # TIMINGS do not depend on it, but a reader should see which k(Z,Z) they were taken on
        kzz_eig = torch.linalg.eigvalsh(ipb.base_gram_induce.double())
        cond_kzz = float((kzz_eig.max() / kzz_eig.min().clamp_min(1e-300)).item())
        out["inducing_point_basis"] = {
            "ms_per_step": dti * 1e3, "steps": args.ipb_steps, "setup_s": round(t_ipb, 2),
            "cond_K_ZZ": cond_kzz, "cholesky_jitter": float(getattr(ipb._chol, "jitter", 0.0)),
            "cond_note": "2-norm condition number of k(Z,Z) (torch.linalg.eigvalsh on the device) and the jitter its Cholesky factor "
                         "was computed with; the parity tests of this basis run up to cond ~ 1e8 (tests/test_gpu_configs.py), the "
                         "figures of this block are timings of the same kernels on this matrix, not accuracy claims",
            "step": "V = K_ZZ^-1 U (MFMA) -> F = K_XZ V -> d cost/d f -> K_ZX G -> e = L_c xi (Philox + MFMA) -> update",
            "flop_per_step": 4.0 * n * cfg["m"] * j_loc + 4.0 * cfg["m"] ** 2 * j_loc,
            "tflops": (4.0 * n * cfg["m"] * j_loc + 4.0 * cfg["m"] ** 2 * j_loc) / dti / 1e12,
            "gaussian_fast_path_ms_per_step": None if dti_fast is None else dti_fast * 1e3,
            "gaussian_fast_path_note": "U -> U per call, two launches: dS = -eta (P U - c~) + sqrt(2 eta) xi with P = Q Lc^-1 (the "
                                       "whitened operator with the forward solve folded in: one fused kernel, 2 M^2 J flop), then "
                                       "dU = Lc dS (balanced triangular product, M^2 J flop)",
            "gaussian_fast_path_three_launch_ms_per_step": None if dti_fast3 is None else dti_fast3 * 1e3,
            "gaussian_fast_path_three_launch_note": "PLS_OPT_IPB_STEP_OPERATOR 0 (round 3's route): S = Lc^-1 U, dS from Q S, dU = Lc dS: "
                                                    "4 M^2 J flop",
            "gaussian_fast_path_with_energy_ms_per_step": None if dti_fast_e is None else dti_fast_e * 1e3,
            "gaussian_whitened_loop_ms_per_step": None if dti_white is None else dti_white * 1e3,
            "gaussian_whitened_loop_note": "a loop that keeps S between steps (pls_ipb_whitened_step): 2 M^2 J flop, one kernel per step",
            "gaussian_train_pls_ms_per_iteration": None if dti_white is None else dti_train * 1e3,
        }
        log(f"inducing-point basis: {dti * 1e3:.2f} ms/step")
        del ipb, a, b
    # ---- a multi-rank line proves its own shards (distributed.py:1-7: "1/2/4/8-GPU runs give identical particles").  After the
    # timed sections every rank advances ITS columns of one seeded particle matrix by K fast-path steps and K like-for-like
    # steps (library noise keyed by the GLOBAL column); the shards are gathered and rank 0 recomputes every shard on its own
    # GPU -- the same column block, the same column offset, hence the same kernels: bit for bit or the line says so -- and the
    # whole matrix unsharded (other tile shapes for a wider matrix: equal to rounding) ----
    if dist.is_initialized() or shard_world > 1:
        t_chk = time.perf_counter()
        seed_chk = 97531
        k_fast = 20 if cfg["cost"] == "gaussian" else 0
        k_like = max(2, min(20, int(10.0 / max(ms_per_step * 1e-3 * shard_world, 1e-6))))
        u_chk = torch.normal(0.0, 1.0, size=(mk, j_total), generator=torch.Generator().manual_seed(1), dtype=torch.float64)

        def advance(u0, j_off):
            prev = basis.j_offset
            basis.j_offset = j_off
            try:
                cur, nxt = u0.contiguous().cuda(), None
                nxt = torch.empty_like(cur)
                for t in range(k_fast + k_like):
                    basis.fused_step(cost, cur, eta, out=nxt, new_state=True, force_generic=t >= k_fast,
                                     noise=NoiseSpec(seed=seed_chk, step=t, j_offset=j_off))
                    cur, nxt = nxt, cur
                e = basis.fused_particle_energy(cost, cur)
                torch.cuda.synchronize()
                return cur, e
            finally:
                basis.j_offset = prev

        mine, e_mine = advance(u_chk[:, j0:j1], j0)
        collective_ok = None
        if dist.is_initialized():  # one real collective on device memory whatever the world size (RCCL under "nccl")
            on_dev = "nccl" in str(dist.get_backend()).lower()
            ones = torch.ones(4, dtype=torch.float64, device="cuda" if on_dev else "cpu")
            dist.all_reduce(ones, op=dist.ReduceOp.SUM)
            collective_ok = bool((ones == float(world)).all().item())
        if dist.is_initialized() and world > 1:
            on_host = "nccl" not in str(dist.get_backend()).lower()  # (gloo gathers host tensors)
            got_u = D.gather_particles(mine.cpu() if on_host else mine, j_total).cuda()
            got_e = D.gather_particles((e_mine.cpu() if on_host else e_mine)[None, :], j_total).cuda()[0]
            mean_e = D.mean_over_particles(e_mine.cpu() if on_host else e_mine, j_total)
        else:
            got_u, got_e, mean_e = mine, e_mine, None
        if rank == 0:
            max_abs, e_equal, shards = 0.0, True, []
            for r in range(shard_world):
                a, b = D.shard_bounds(j_total, r, shard_world)
                if not dist.is_initialized() and r != 0:
                    continue  # (an emulated shard run holds rank 0's columns only)
                ref_u, ref_e = advance(u_chk[:, a:b], a)
                lo = a if got_u.shape[1] == j_total else 0
                d = float((got_u[:, lo:lo + (b - a)] - ref_u).abs().max().item())
                max_abs = max(max_abs, d)
                e_equal = e_equal and bool(torch.equal(got_e[lo:lo + (b - a)], ref_e))
                shards.append({"rank": r, "columns": [a, b], "max_abs_diff": d})
            whole_u, whole_e = advance(u_chk, 0)
            cols = slice(0, j_total) if got_u.shape[1] == j_total else slice(j0, j1)
            rel_unsharded = float(((got_u - whole_u[:, cols]).abs().max() / whole_u.abs().max()).item())
            out["shard_check"] = {
                "steps": {"gaussian_fast_path": k_fast, "like_for_like": k_like},
                "max_abs_diff": max_abs, "energies_equal": e_equal, "shards": shards,
                "max_rel_diff_vs_one_unsharded_matrix": rel_unsharded,
                "mean_energy_all_reduce": mean_e, "all_reduce_of_ones_equals_world": collective_ok,
                "mean_energy_rank0_recomputed": float(whole_e.double().sum().item() / j_total),
                "seconds": time.perf_counter() - t_chk,
                "note": "every rank advances its columns of one seeded particle matrix; gathered over the process group; rank 0 "
                        "recomputes each shard (same columns, same global column offset: same kernels, bit for bit) and the "
                        "unsharded matrix (wider tiles: equal to rounding)",
            }
            log(f"shard check: {k_fast} + {k_like} steps, max |diff| {max_abs:.1e}, energies equal {e_equal}, vs unsharded {rel_unsharded:.1e}")
            assert max_abs == 0.0 and e_equal, "the ranks' shards are not the particles rank 0 computes for the same columns"
        del u_chk, mine, e_mine, got_u, got_e
        barrier()
    # ---- the reference's profiler protocol (experiments/profiler/main.py:41-82, :141-169): construction of kernel, basis,
    # cost and PLS, particle initialisation and T steps `particles += pls.calculate_particle_update(particles, eta)` timed
    # as ONE block ----
    if rank == 0 and world == 1 and args.profiler_steps > 0 and shard_world == 1:
        del ping, pong
        torch.cuda.empty_cache()

        def profiler_block(eigh_device):
            barrier()
            t0 = time.perf_counter()
            k2 = pkg.PLSKernel(pkg.ARDKernel(ls, 1.0), z)
            b2 = OrthonormalBasis(k2, z, x, eigenvalue_threshold=cfg.get("threshold", 0.0), verbose=False, keep_gram=False,
                                  eigh_device=eigh_device, group=True if world > 1 else None)
            b2.workspace_bytes = basis.workspace_bytes
            if cfg["cost"] == "poisson":
                c2 = PoissonCost(y, SquareLinkFunction())
            elif cfg["cost"] == "bernoulli":
                c2 = type(cost)(y, cost.link_function)
            else:
                c2 = GaussianCost(cfg["obs"], y, IdentityLinkFunction())
            pls2 = pkg.PLS(b2, c2)
            torch.manual_seed(0)
            p2 = pls2.initialise_particles(number_of_particles=j_total, noise_only=True)
            torch.cuda.synchronize()
            t_c = time.perf_counter() - t0
            for _ in range(args.profiler_steps):
                p2 += pls2.calculate_particle_update(particles=p2, step_size=eta)
            torch.cuda.synchronize()
            return t_c, time.perf_counter() - t0

        t_construct, t_block = profiler_block(args.eigh_device)
        other = "cpu" if args.eigh_device == "cuda" else "cuda"
        t_construct_o, t_block_o = profiler_block(other)
        out["construct_plus_T_steps"] = {
            "T": args.profiler_steps, "seconds": t_block, "construction_s": t_construct, "steps_s": t_block - t_construct,
            "eigh_device": args.eigh_device,
            f"seconds_eigh_on_{other}": t_block_o, f"construction_s_eigh_on_{other}": t_construct_o,
            "protocol": "experiments/profiler/main.py:41-82 inside one timed block: PLSKernel + OrthonormalBasis (torch.linalg.eigh "
                        "of k(Z,Z)/M on the device the matrix lives on, as the reference's .cuda() branch does; the host-LAPACK "
                        "variant of the reference's CPU path beside it) + cost + PLS + initialise_particles (host generator, like "
                        "the reference) + T x `particles += pls.calculate_particle_update(particles, step_size)` through the "
                        "drop-in API",
        }
        log(f"profiler protocol: construction {t_construct:.2f} s + {args.profiler_steps} steps = {t_block:.2f} s")
    if rank == 0 and world == 1 and shard_world == 1 and args.reference_scale_iterations > 0:
        out["reference_scale"] = reference_scale_block(pkg, args.reference_scale_iterations)
        log("reference scale: " + "; ".join(f"N={r['n']} M={r['m']} J={r['j']}: onb {r['orthonormal/gaussian_identity']} / {r['orthonormal/bernoulli_sigmoid']}, "
                                            f"ipb {r['inducing_point/gaussian_identity']} / {r['inducing_point/bernoulli_sigmoid']} us" for r in out["reference_scale"]["rows"]))
    # ---- CPU baseline (rank 0, N=1 only) ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        lam_all, vec_all = basis.eigenvalues.cpu(), basis.eigenvectors.cpu()
        out["cpu_baseline"] = cpu_baseline(cfg, x, z, y, ls, lam_all, vec_all)
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        out["gpu_over_cpu_fair_variant"] = value / out["cpu_baseline"]["fair_variant"]["value"]
        ref_dev = out["cpu_baseline"].get("reference_style_torch_rocm")
        if ref_dev and ref_dev.get("value"):
            out["gpu_over_reference_style_torch_rocm"] = value / ref_dev["value"]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
