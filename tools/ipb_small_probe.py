"""train_pls with the inducing-point basis at the reference's curve-experiment scale (experiments/curves/{poisson_regression,
classification}/main.py build BOTH bases): us per iteration and the launches behind one step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import _lib as L
from projected_langevin_sampling_amd.basis import InducingPointBasis, OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import BernoulliCost, GaussianCost, PoissonCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction, SigmoidLinkFunction, SquareLinkFunction
from projected_langevin_sampling_amd.trainers import train_pls
torch.set_default_dtype(torch.float64)
for (n, m, j, d) in ((100, 10, 64, 1), (1000, 32, 100, 1), (4096, 128, 512, 4)):
    g = torch.Generator().manual_seed(0)
    x = torch.rand(n, d, generator=g) * 2 - 1
    z = x[torch.randperm(n, generator=g)[:m]].clone()
    y = torch.sin(2.0 * x.sum(dim=1)) + 0.1 * torch.randn(n, generator=g)
    kern = P.PLSKernel(P.ARDKernel(torch.full((d,), 0.5), 1.0), z)
    for bname, basis in (("onb", OrthonormalBasis(kern, z, x, 1e-8, verbose=False)), ("ipb", InducingPointBasis(kern, z, y[:m], x))):
        for cname, cost in (("gaussian", GaussianCost(0.1, y, IdentityLinkFunction())), ("bernoulli", BernoulliCost((y > 0).double(), SigmoidLinkFunction())),
                            ("poisson", PoissonCost(torch.poisson(y * y + 0.5, generator=g), SquareLinkFunction()))):
            pls = P.PLS(basis, cost)
            u = (1.0 + 0.1 * torch.randn(basis.approximation_dimension, j, generator=g)).cuda()
            eta = 1e-13  # (timing only; small enough that the stiffest mode of a random Z's prior drift stays stable for 2000 steps)
            train_pls(pls, u.clone(), 30, eta, 1e9)
            ws = []
            for _ in range(3):  # (median of three runs: one run in a few dozen is 1.5-2x slow as a whole, tools/sr_step_jitter.py)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                _, e = train_pls(pls, u.clone(), 2000, eta, 1e9)
                torch.cuda.synchronize(); ws.append(time.perf_counter() - t0)
            w = sorted(ws)[1]
            assert len(e) == 2000, f"the run stopped after {len(e)} iterations"
            out = torch.empty_like(u); en = torch.empty(j, device="cuda")
            with L.Timeline(64) as tl:
                basis.fused_step(cost, u, eta, out=out, new_state=True, noise=NoiseSpec(seed=1, step=0), input_energy=en)
            print(f"N={n} M={m} J={j} {bname} {cname:9s}: train_pls {w / max(len(e), 1) * 1e6:7.2f} us per iteration | launches of one step + energies: "
                  + ", ".join(f"{k} x{v['launches']} {v['total_ms'] * 1e3:.1f}us" for k, v in tl.summary().items()), flush=True)
