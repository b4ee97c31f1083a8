"""The reference's own timing protocol (experiments/profiler/main.py:50-82, :141-169, profiler/config.yaml:1-22), the GPU
drop-in beside the CPU oracle in the same run.

One measurement = ONE timed block, exactly what `profile_pls` wraps in `record_function("model_training")`:
    OrthonormalBasis(kernel, x_induce, x_train)  ->  cost  ->  PLS  ->  initialise_particles(J, noise_only=True)
    ->  T x { update = pls.calculate_particle_update(particles, step_size = 1e-10);  particles += update }
on the reference's data (x = linspace(-3, 3, N), Curve1 standardised + 0.2 N(0, 1), experiments/curves/curves.py:41-47;
ScaleKernel(RBFKernel) at gpytorch's initial hyper-parameters softplus(0) = ln 2; observation noise 0.01; inducing points by
greedy conditional variance, selected OUTSIDE the block like the reference does).  The grid is the reference's: one
parameter at a time around the default point N = 100, M = 10, T = 10, J = 100 -- N and J 100 .. 1000, M and T 10 .. 100.

Reported per grid point and cost (the reference's Gaussian/identity, and Bernoulli/sigmoid for the costs without the Gaussian
algebra): construction and the T steps separately (median of `--repeats` blocks after one warm-up block), for
  gpu       the drop-in API on the MI355X (eigh where the library's default puts it),
  gpu_host_eigh / gpu_device_eigh   the same with the eigh forced to host LAPACK / rocSOLVER,
  cpu       the CPU oracle (oracle/pls_oracle.py: the reference's op sequence in torch fp64) on this box's host cores: with
            as many torch threads as the process may really use (cgroup quota / affinity, bench.host_cores; torch's own
            default is the whole host's count, which oversubscribes a containerised box 8-fold) AND with one thread (at
            these sizes the thread pool costs more than it brings); the better of the two is what the GPU block is held to.
usage: python tools/profiler_grid.py [--repeats 5] [--quick] [--out profiles/r05_profiler_grid.json]"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

STEP_SIZE = 1e-10  # profile_pls(step_size=1e-10), main.py:150
OBS_NOISE = 0.01   # run_experiment(observation_noise=0.01), main.py:207
LN2 = 0.6931471805599453  # gpytorch's softplus(raw = 0): lengthscale and outputscale of a fresh ScaleKernel(RBFKernel)


def make_data(n: int, seed: int = 0):
    x = torch.linspace(-3, 3, n, dtype=torch.float64).reshape(-1, 1)
    curve = 2 * torch.sin((x**2) * 0.35 * torch.pi)
    curve = (curve - curve.mean()) / curve.std()
    y = (curve + 0.2 * torch.normal(mean=0.0, std=1.0, generator=torch.Generator().manual_seed(seed), size=x.shape)).reshape(-1)
    return x, y


def grid_points(quick: bool):
    d = dict(n=100, m=10, t=10, j=100)
    pts = [tuple(d.values())]
    rng = {"n": range(100, 1001, 100), "m": range(10, 101, 10), "t": range(10, 101, 10), "j": range(100, 1001, 100)}
    for key, values in rng.items():
        for v in values:
            p = dict(d)
            p[key] = v
            if quick and v not in (values[0], values[len(values) // 2], values[-1]):
                continue
            if tuple(p.values()) not in pts:
                pts.append(tuple(p.values()))
    return pts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    torch.set_default_dtype(torch.float64)  # main.py:505

    import projected_langevin_sampling_amd as P
    from projected_langevin_sampling_amd import samplers
    from projected_langevin_sampling_amd.basis import OrthonormalBasis
    from projected_langevin_sampling_amd.costs import BernoulliCost, GaussianCost
    from projected_langevin_sampling_amd.inducing_point_selectors import ConditionalVarianceInducingPointSelector
    from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction, SigmoidLinkFunction
    from oracle import pls_oracle as O

    import bench

    threads = bench.host_cores()
    torch.set_num_threads(threads)

    def gpu_block(x, y, z, t, j, cost_name, eigh_device):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kernel = P.PLSKernel(P.ARDKernel(torch.full((1,), LN2), LN2), z)
        basis = OrthonormalBasis(kernel=kernel, x_induce=z, x_train=x, verbose=False, eigh_device=eigh_device)
        cost = (GaussianCost(OBS_NOISE, y, IdentityLinkFunction()) if cost_name == "gaussian"
                else BernoulliCost((y > 0).double(), SigmoidLinkFunction()))
        pls = P.PLS(basis=basis, cost=cost)
        particles = pls.initialise_particles(number_of_particles=j, noise_only=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(t):
            particle_update = pls.calculate_particle_update(particles=particles, step_size=STEP_SIZE)
            particles += particle_update
        torch.cuda.synchronize()
        return t1 - t0, time.perf_counter() - t1

    def cpu_block(x, y, z, t, j, cost_name):
        t0 = time.perf_counter()
        basis = O.OrthonormalBasis(O.RBFARDKernel(torch.full((1,), LN2), LN2), z, x)
        cost = (O.GaussianCost(OBS_NOISE, y, O.IdentityLink()) if cost_name == "gaussian"
                else O.BernoulliCost((y > 0).double(), O.SigmoidLink()))
        pls = O.PLS(basis, cost)
        particles = basis.initialise_particles(j, noise_only=True)
        t1 = time.perf_counter()
        for _ in range(t):
            particle_update = pls.calculate_particle_update(particles, STEP_SIZE)
            particles += particle_update
        return t1 - t0, time.perf_counter() - t1

    def median_block(fn, reps):
        fn()  # warm-up (allocator pools, code objects, LAPACK / rocSOLVER handles)
        runs = [fn() for _ in range(reps)]
        return statistics.median(r[0] for r in runs), statistics.median(r[1] for r in runs)

    rows = []
    selector = ConditionalVarianceInducingPointSelector()
    for (n, m, t, j) in grid_points(args.quick):
        x, y = make_data(n)
        torch.manual_seed(0)
        import numpy as np
        np.random.seed(0)
        z, _ = selector(x=x, m=m, kernel=P.ARDKernel(torch.full((1,), LN2), LN2))
        z = z.detach().cpu().double()
        for cost_name in ("gaussian", "bernoulli"):
            row = {"n": n, "m": m, "t": t, "j": j, "cost": cost_name}
            for label, dev in (("gpu", None), ("gpu_host_eigh", "cpu"), ("gpu_device_eigh", "cuda")):
                c, s = median_block(lambda: gpu_block(x, y, z, t, j, cost_name, dev), args.repeats)
                row[label] = {"construction_ms": c * 1e3, "steps_ms": s * 1e3, "us_per_step": s / t * 1e6, "total_ms": (c + s) * 1e3}
            if not args.no_cpu:
                best = None
                for label, nthreads in ((f"cpu_{threads}_threads", threads), ("cpu_1_thread", 1)):
                    torch.set_num_threads(nthreads)
                    c, s = median_block(lambda: cpu_block(x, y, z, t, j, cost_name), args.repeats)
                    row[label] = {"construction_ms": c * 1e3, "steps_ms": s * 1e3, "us_per_step": s / t * 1e6, "total_ms": (c + s) * 1e3,
                                  "threads": nthreads}
                    if best is None or row[label]["total_ms"] < best["total_ms"]:
                        best = row[label]
                torch.set_num_threads(threads)
                row["cpu"] = best
                row["gpu_total_over_cpu_total"] = row["gpu"]["total_ms"] / row["cpu"]["total_ms"]
            rows.append(row)
            g, c = row["gpu"], row.get("cpu")
            print(f"N={n:5d} M={m:4d} T={t:4d} J={j:5d} {cost_name:9s} | gpu: construct {g['construction_ms']:7.3f} ms + steps {g['steps_ms']:7.3f} ms "
                  f"({g['us_per_step']:6.1f} us/step) [host eigh {row['gpu_host_eigh']['construction_ms']:6.3f} / device eigh "
                  f"{row['gpu_device_eigh']['construction_ms']:6.3f} ms]"
                  + (f" | cpu ({c['threads']} of {threads} threads): construct {c['construction_ms']:7.3f} ms + steps {c['steps_ms']:7.3f} ms ({c['us_per_step']:7.1f} us/step)"
                     f" | gpu/cpu total {row['gpu_total_over_cpu_total']:.2f}" if c else ""), flush=True)
    summary = {
        "protocol": "experiments/profiler/main.py:50-82 in one timed block per grid point (construction + T x particles += "
                    "calculate_particle_update, step 1e-10), grid of experiments/profiler/config.yaml:1-22; median of "
                    f"{args.repeats} blocks after one warm-up block",
        "cpu_threads": threads, "cpu_model": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
        "default_eigh_device": samplers.DEFAULT_EIGH_DEVICE,
        "rows": rows,
    }
    if not args.no_cpu:
        losing = [(r["n"], r["m"], r["t"], r["j"], r["cost"], round(r["gpu_total_over_cpu_total"], 2)) for r in rows if r["gpu_total_over_cpu_total"] > 1.0]
        summary["points_where_the_gpu_block_is_slower_than_the_cpu_oracle"] = losing
        print(f"grid points where construct + T steps on the GPU take longer than on the CPU oracle: {len(losing)} of {len(rows)}: {losing}")
    if args.out:
        with open(args.out, "w") as f:
            json.dump(summary, f, indent=1)
    return summary


if __name__ == "__main__":
    main()
