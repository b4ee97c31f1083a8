"""The one-launch small-rank step (csrc/small_rank_step.h, PLS_OPT_SMALL_RANK_STEP = 2) against the slab kernels + update
launch (option 0) at the sizes of the reference's own experiments: max relative difference of the new state and of the
energies, time per step in a hipGraph replay with and without the energy by-product, and the timeline's per-launch durations.
usage: sr_step_probe.py [quick]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import _lib as L
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import BernoulliCost, GaussianCost, PoissonCost
from projected_langevin_sampling_amd.link_functions import SigmoidLinkFunction, IdentityLinkFunction, SquareLinkFunction
torch.set_default_dtype(torch.float64)
lib = L.load()
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
shapes = [(100, 10, 64, 1), (1000, 32, 100, 2), (4096, 128, 512, 4)] if quick else \
    [(100, 10, 64, 1), (100, 10, 100, 1), (1000, 32, 100, 2), (1000, 100, 1000, 3), (4096, 128, 512, 4), (4096, 64, 512, 4),
     (20000, 89, 1024, 3), (4096, 128, 4096, 4)]
for (n, m, j, d) in shapes:
    g = torch.Generator().manual_seed(0)
    x = torch.rand(n, d, generator=g) * 2 - 1
    z = x[torch.randperm(n, generator=g)[:m]].clone()
    f = torch.sin(2.0 * x.sum(dim=1))
    basis = OrthonormalBasis(P.PLSKernel(P.ARDKernel(torch.full((d,), 0.5), 1.0), z.cuda()), z.cuda(), x.cuda(), 1e-8, verbose=False)
    mk = basis.approximation_dimension
    u = (1.0 + 0.1 * torch.randn(mk, j, generator=g)).cuda()
    costs = {"gaussian(generic)": (GaussianCost(0.1, f, IdentityLinkFunction()), True),
             "poisson/square": (PoissonCost(torch.poisson(f * f + 0.5, generator=g), SquareLinkFunction()), False),
             "bernoulli/sigmoid": (BernoulliCost((f > 0).double(), SigmoidLinkFunction()), False)}
    for name, (cost, fg) in costs.items():
        res = {}
        for with_e in (False, True):
            for mode in (0, 2):
                lib.pls_set_option(L.OPT_SMALL_RANK_STEP, mode)
                out = torch.empty_like(u); e = torch.full((j,), float("nan"), device="cuda")
                call = lambda: basis.fused_step(cost, u, 1e-9, out=out, new_state=True, force_generic=fg, noise=NoiseSpec(seed=1, step=2),
                                                input_energy=e if with_e else None)
                for _ in range(3): call()
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph(); side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    with torch.cuda.graph(gr, stream=side):
                        for _ in range(20): call()
                torch.cuda.current_stream().wait_stream(side)
                gr.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): gr.replay()
                e1.record(); torch.cuda.synchronize()
                with L.Timeline(256) as tl:
                    for _ in range(10): call()
                res[(with_e, mode)] = (out.clone(), e.clone(), e0.elapsed_time(e1) / 400 * 1e3,
                                       ", ".join(f"{k} {v['avg_ms'] * 1e3:.1f}" for k, v in tl.summary().items()))
        lib.pls_set_option(L.OPT_SMALL_RANK_STEP, 1)
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
        du = max(rel(res[(w, 2)][0], res[(w, 0)][0]) for w in (False, True))
        de = rel(res[(True, 2)][1], res[(True, 0)][1])
        print(f"N={n} M_k={mk} J={j} {name:18s}: |dU| {du:.1e} |dE| {de:.1e} | us/step (graph) old {res[(False, 0)][2]:6.2f} new {res[(False, 2)][2]:6.2f}"
              f" | with energies old {res[(True, 0)][2]:6.2f} new {res[(True, 2)][2]:6.2f} | launches new+E: {res[(True, 2)][3]} | old+E: {res[(True, 0)][3]}",
              flush=True)
