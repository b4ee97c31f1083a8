"""train_pls at the reference's own experiment scale, where an iteration is bound by the host and the launch, not the kernel:
configs[0] (N = 100, M = 10, J = 64) and a UCI-sized problem (N = 4096, M = 128, J = 512); eager and captured loops."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd.basis import OrthonormalBasis
from projected_langevin_sampling_amd.costs import BernoulliCost, GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction, SigmoidLinkFunction
from projected_langevin_sampling_amd.trainers import train_pls, train_pls_captured

torch.set_default_dtype(torch.float64)
for (n, m, j, d) in ((100, 10, 64, 1), (4096, 128, 512, 4)):
    g = torch.Generator().manual_seed(0)
    x = torch.rand(n, d, generator=g) * 2 - 1
    z = x[torch.randperm(n, generator=g)[:m]].clone()
    y = torch.sin(2.0 * x.sum(dim=1)) + 0.1 * torch.randn(n, generator=g)
    basis = OrthonormalBasis(P.PLSKernel(P.ARDKernel(torch.full((d,), 0.5), 1.0), z.cuda()), z.cuda(), x.cuda(), 1e-8, verbose=False)
    pls = P.PLS(basis, GaussianCost(0.1, y, IdentityLinkFunction()))
    u = torch.randn(basis.approximation_dimension, j, generator=g).cuda()
    eta = 0.5 * float(basis.eigenvalues.min())
    train_pls(pls, u.clone(), 50, eta, 1e9)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, e = train_pls(pls, u.clone(), 5000, eta, 1e9)
        torch.cuda.synchronize(); w = time.perf_counter() - t0
        print(f"N={n} M={m} (M_k={basis.approximation_dimension}) J={j}: train_pls {w / len(e) * 1e6:6.2f} us per iteration", flush=True)
    for k in (32, 128):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, e = train_pls_captured(pls, u.clone(), 5000, eta, 1e9, steps_per_replay=k, seed=1)
        torch.cuda.synchronize(); w = time.perf_counter() - t0
        print(f"N={n} M={m} J={j}: train_pls_captured({k}) {w / len(e) * 1e6:6.2f} us per iteration incl. capture", flush=True)
    # a cost without the Gaussian algebra: the loop launches the fused small-rank / two-GEMM step and a mean kernel per iteration
    yb = (y > 0).double()
    bpls = P.PLS(basis, BernoulliCost(yb, SigmoidLinkFunction()))
    train_pls(bpls, u.clone(), 50, eta, 1e9)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, e = train_pls(bpls, u.clone(), 3000, eta, 1e9)
        torch.cuda.synchronize(); w = time.perf_counter() - t0
        print(f"N={n} M={m} J={j}: Bernoulli/sigmoid train_pls {w / len(e) * 1e6:6.2f} us per iteration", flush=True)
