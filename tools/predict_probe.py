"""Timing of the prediction path at configs[1] sizes: N = 1e5, M = 1024, J = 8192 particles, N* = 2000 test
points, 2000 calibration points.  pls.predict = predictive noise G([Z, x]) (an (M_k + N*) eigh + (M_k + N*) x J host
normals, samplers.py:27-35) -> predict_untransformed_samples -> observation noise -> Gaussian moments; then the conformal
quantiles over J.  The eigh of the sampler on the host (the reference's CPU call, the library default) and on the device."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import samplers
from projected_langevin_sampling_amd.basis import OrthonormalBasis
from projected_langevin_sampling_amd.costs import GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction
from projected_langevin_sampling_amd.conformalise import ConformalisePLS

torch.set_default_dtype(torch.float64)
n, m, j, d, ns = 100000, 1024, 8192, 8, 2000
g = torch.Generator().manual_seed(0)
x = torch.rand(n, d, generator=g) * 2 - 1
w = torch.randn(d, generator=g)
y = torch.sin(2 * torch.pi * (x @ w)) + 0.1 * torch.randn(n, generator=g)
z = x[torch.randperm(n, generator=g)[:m]].contiguous()
xs = torch.rand(ns, d, generator=g) * 2 - 1
xc = torch.rand(ns, d, generator=g) * 2 - 1
yc = torch.sin(2 * torch.pi * (xc @ w)) + 0.1 * torch.randn(ns, generator=g)
ls = torch.linspace(0.5, 1.5, d)


def clock(f, reps=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, r


t, basis = clock(lambda: OrthonormalBasis(P.PLSKernel(P.ARDKernel(ls, 1.0), z.cuda()), z.cuda(), x.cuda(), 0.0, verbose=False, eigh_device="cuda"))
print(f"basis (device eigh): {t:.3f} s, M_k = {basis.approximation_dimension}")
cost = GaussianCost(0.1, y, IdentityLinkFunction())
pls = P.PLS(basis, cost)
torch.manual_seed(0)
u = pls.initialise_particles(number_of_particles=j, noise_only=True)
xs_d, xc_d, yc_d = xs.cuda(), xc.cuda(), yc.cuda()  # (the sampler's eigh is remembered per test-point TENSOR: keep the object)
for where, stream in (("cpu", "reference"), ("cuda", "reference"), ("cuda", "device"), ("cuda", "device")):
    samplers.DEFAULT_EIGH_DEVICE, samplers.DEFAULT_NORMAL_STREAM = where, stream
    basis.__dict__.pop("_pred_factor_cache", None)
    torch.manual_seed(1)
    t_noise, noise = clock(lambda: basis.sample_predictive_noise(u, xs_d))
    t_noise2, noise = clock(lambda: basis.sample_predictive_noise(u, xs_d))
    t_pred, f = clock(lambda: basis.predict_untransformed_samples(u, xs_d, noise=noise))
    torch.manual_seed(1)
    t_all, dist = clock(lambda: pls.predict(xs_d, u))
    print(f"sampler eigh on {where:4s}, normals from the {stream:9s} stream: predictive noise {t_noise:.3f} s first call, {t_noise2 * 1e3:.1f} ms "
          f"again (factor remembered)   predict_untransformed_samples (noise given) {t_pred * 1e3:.2f} ms   pls.predict end to end "
          f"{t_all * 1e3:.1f} ms   mean |mu| {dist.mean.abs().mean().item():.4f}", flush=True)
samplers.DEFAULT_EIGH_DEVICE, samplers.DEFAULT_NORMAL_STREAM = "cuda", "device"
torch.manual_seed(2)
t_c, conf = clock(lambda: ConformalisePLS(xc_d, yc_d, pls, u))
t_q, pred = clock(lambda: conf.predict(xs_d, 0.9))
t_q2, pred = clock(lambda: conf.predict(xs_d, 0.9))
print(f"ConformalisePLS: construct (calibration samples) {t_c:.3f} s, predict(coverage 0.9) on {ns} points {t_q:.3f} s, again {t_q2:.3f} s")
