"""Inducing-point basis, Gaussian cost: the solve with k(Z,Z) (block substitution vs triangular products with the inverse
factor), the per-call step (round-2 route vs whitened route) and the whitened step, at M = 1024 (and 4096) for a full
particle matrix and for the shards of a 4- and an 8-GPU run.  us per call, back-to-back launches in one timed region."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import _lib as L
from projected_langevin_sampling_amd.basis import InducingPointBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction

lib = L.load()
torch.manual_seed(0)
torch.set_default_dtype(torch.float64)


def region(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


ms = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1024]
for m in ms:
    n, d = 20000, 8
    g = torch.Generator().manual_seed(0)
    x = torch.rand(n, d, generator=g) * 2 - 1
    z = x[:m].clone()
    y = torch.sin(2 * x.sum(dim=1)) + 0.1 * torch.randn(n, generator=g)
    ls = 0.5 + torch.rand(d, generator=g)
    basis = InducingPointBasis(P.PLSKernel(P.ARDKernel(ls, 1.0), z), z, y[:m], x)
    cost = GaussianCost(0.1, y, IdentityLinkFunction())
    f = basis._chol
    for j in (1024, 2048, 8192):
        u = torch.randn(m, j, device="cuda")
        out = torch.empty_like(u)
        en = torch.empty(j, device="cuda")
        reps = 200 if j <= 2048 else 60
        flop = 2.0 * m * m * j
        res = {}
        for mode in (0, 1):
            L.check(lib.pls_set_option(L.OPT_SOLVE_MODE, mode))
            res[f"solve[{'products' if mode else 'substitution'}]"] = (region(lambda: f.solve(u), reps), flop)
            res[f"forward[{'products' if mode else 'substitution'}]"] = (region(lambda: f.forward_solve(u), reps), flop / 2)
        L.check(lib.pls_set_option(L.OPT_SOLVE_MODE, 1))
        basis.whitened = False
        basis._ws.clear()
        res["step[round-2 route]"] = (region(lambda: basis.fused_step(cost, u, 1e-6, out=out, new_state=True, noise=NoiseSpec(seed=1, step=3)), reps), 2.5 * flop)
        res["step+energy[round-2 route]"] = (region(lambda: basis.fused_step(cost, u, 1e-6, out=out, new_state=True, noise=NoiseSpec(seed=1, step=3), input_energy=en), reps), 2.5 * flop)
        L.check(lib.pls_set_option(L.OPT_SOLVE_MODE, 0))
        res["step[round-2 route, substitution]"] = (region(lambda: basis.fused_step(cost, u, 1e-6, out=out, new_state=True, noise=NoiseSpec(seed=1, step=3)), reps), 2.5 * flop)
        L.check(lib.pls_set_option(L.OPT_SOLVE_MODE, 1))
        basis.whitened = True
        basis._ws.clear()
        res["step[whitened route]"] = (region(lambda: basis.fused_step(cost, u, 1e-6, out=out, new_state=True, noise=NoiseSpec(seed=1, step=3)), reps), 2.0 * flop)
        res["step+energy[whitened route]"] = (region(lambda: basis.fused_step(cost, u, 1e-6, out=out, new_state=True, noise=NoiseSpec(seed=1, step=3), input_energy=en), reps), 2.0 * flop)
        s = basis.whiten(u)
        res["whitened step"] = (region(lambda: basis.whitened_step(cost, s, 1e-6, out=out, new_state=True, noise=NoiseSpec(seed=1, step=3)), reps), flop)
        res["whitened step+energy"] = (region(lambda: basis.whitened_step(cost, s, 1e-6, out=out, new_state=True, noise=NoiseSpec(seed=1, step=3), input_energy=en), reps), flop)
        for k, (t, fl) in res.items():
            print(f"M={m} J={j:5d} {k:36s} {t:8.1f} us   {fl / t / 1e6:6.1f} TF/s ({fl / t / 78.6e6:.3f} of peak)", flush=True)
