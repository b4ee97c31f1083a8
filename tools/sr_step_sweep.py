"""Time of the one-launch small-rank step (option 2) over N at fixed (M, J): slope = cost per round of tiles, intercept = fixed
cost of a launch.  usage: sr_step_sweep.py M J D N1 N2 ...   (env PLS_SRS_FORCE_NS forces the slab count in probe builds)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import _lib as L
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import BernoulliCost, GaussianCost
from projected_langevin_sampling_amd.link_functions import SigmoidLinkFunction, IdentityLinkFunction
torch.set_default_dtype(torch.float64)
lib = L.load()
m, j, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for n in [int(v) for v in sys.argv[4:]]:
    g = torch.Generator().manual_seed(0)
    x = torch.rand(n, d, generator=g) * 2 - 1
    z = x[torch.randperm(n, generator=g)[:m]].clone()
    f = torch.sin(2.0 * x.sum(dim=1))
    basis = OrthonormalBasis(P.PLSKernel(P.ARDKernel(torch.full((d,), 0.5), 1.0), z.cuda()), z.cuda(), x.cuda(), 1e-10, verbose=False)
    mk = basis.approximation_dimension
    u = (1.0 + 0.1 * torch.randn(mk, j, generator=g)).cuda()
    line = f"N={n} M_k={mk} J={j}:"
    for name, cost, fg in (("gauss", GaussianCost(0.1, f, IdentityLinkFunction()), True), ("bern", BernoulliCost((f > 0).double(), SigmoidLinkFunction()), False)):
        for with_e in (False, True):
            for mode in (0, 2):
                lib.pls_set_option(L.OPT_SMALL_RANK_STEP, mode)
                out = torch.empty_like(u); e = torch.empty(j, device="cuda")
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
                for _ in range(10): gr.replay()
                e1.record(); torch.cuda.synchronize()
                line += f" {name}{'+E' if with_e else ''} {'new' if mode else 'old'} {e0.elapsed_time(e1) / 200 * 1e3:6.2f}"
    lib.pls_set_option(L.OPT_SMALL_RANK_STEP, 1)
    print(line, flush=True)
