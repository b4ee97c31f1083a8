"""Round-2 timing probe (development aid): triangular solves vs the explicit-inverse contraction, the Cholesky setup, and
the like-for-like step around the rank-128 boundary.  usage: python tools/r2_probe.py [solve] [ranks]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import _chol, _ops
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import GaussianCost, PoissonCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction, SquareLinkFunction

what = set(sys.argv[1:]) or {"solve", "ranks"}


def timeit(f, reps=10, warm=3):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if "solve" in what:
    print(f"{'M':>6s} {'J':>6s}  {'factor ms':>10s} {'solve ms':>9s} {'TF/s':>6s} {'W U ms':>8s} {'L xi ms':>8s}")
    for m, j in ((1024, 8192), (1024, 1024), (2048, 8192), (4096, 8192)):
        g = torch.Generator().manual_seed(0)
        z = torch.rand(m, 8, generator=g, dtype=torch.float64) * 2 - 1
        k = P.ARDKernel(0.5 + torch.rand(8, generator=g, dtype=torch.float64), 1.0)(z, z) + 1e-6 * torch.eye(m, dtype=torch.float64, device="cuda")
        t0 = time.perf_counter(); f = _chol.cholesky_factor(k); torch.cuda.synchronize(); t_first = time.perf_counter() - t0
        t_f = timeit(lambda: _chol.cholesky_factor(k), reps=3, warm=1)
        u = torch.randn(m, j, dtype=torch.float64, device="cuda")
        v = torch.empty_like(u)
        lib, L = P._lib.load(), P._lib
        d = f.desc()
        t_s = timeit(lambda: L.check(lib.pls_chol_solve(d, u.data_ptr(), j, j, v.data_ptr(), j, L.stream_ptr())))
        w = torch.randn(m, m, dtype=torch.float64, device="cuda")
        t_w = timeit(lambda: _ops.gemm_tn(w, u, out=v))
        t_c = timeit(lambda: L.check(lib.pls_tri_multiply(f.LcT.data_ptr(), L.ld(f.LcT), m, u.data_ptr(), j, j, v.data_ptr(), j, L.stream_ptr())))
        print(f"{m:6d} {j:6d}  {t_f:10.3f} {t_s:9.3f} {2.0 * m * m * j / t_s / 1e9:6.1f} {t_w:8.3f} {t_c:8.3f}   (first factor call {t_first * 1e3:.1f} ms)", flush=True)

if "ranks" in what:
    print(f"{'N':>8s} {'M_k':>5s} {'J':>6s}  {'gauss ms':>9s} {'TF/s':>6s}  {'poisson ms':>10s} {'TF/s':>6s}")
    for n in (20000, 100000):
        for mk in (128, 129, 144, 160, 192, 224, 256, 320):
            j = 8192
            a = torch.randn(mk, n, dtype=torch.float64, device="cuda") / mk ** 0.5
            lam = torch.rand(mk, dtype=torch.float64, device="cuda") + 0.5
            basis = OrthonormalBasis.from_projection(a, lam)
            basis.workspace_bytes = 8 << 30
            y = torch.poisson(torch.rand(n, dtype=torch.float64) * 4)
            u = torch.randn(mk, j, dtype=torch.float64, device="cuda")
            out = torch.empty_like(u)
            res = []
            for cost in (GaussianCost(0.5, y, IdentityLinkFunction()), PoissonCost(y, SquareLinkFunction())):
                f = lambda: basis.fused_step(cost, u, 1e-9, out=out, new_state=True, noise=NoiseSpec(none=True), force_generic=True)
                ms = timeit(f, reps=5 if n * mk * j > 1e10 else 20)
                res.append((ms, 4.0 * n * mk * j / ms / 1e9))
            print(f"{n:8d} {mk:5d} {j:6d}  {res[0][0]:9.3f} {res[0][1]:6.1f}  {res[1][0]:10.3f} {res[1][1]:6.1f}", flush=True)
            del a, basis, u, out
