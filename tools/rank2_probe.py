"""Like-for-like step for 129 .. 256 basis functions: the wave-pair fused kernel (csrc/small_rank2.h) against the two-GEMM
path (pls_set_option(PLS_OPT_SMALL_RANK2_MAX, 0)), N = 1e5 (and 2e4), J = 8192, Gaussian and Poisson costs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import _lib as L
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import GaussianCost, PoissonCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction, SquareLinkFunction

lib = L.load()
L.check(lib.pls_set_option(L.OPT_SMALL_RANK2_MIN, 129))  # (A/B over the whole range; the shipped default is 161 .. 240)


def timeit(f, reps=5, warm=2):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ns = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [100000]
print(f"{'N':>8s} {'M_k':>5s}  {'cost':8s} {'two-GEMM ms':>12s} {'frac':>6s} {'fused ms':>9s} {'frac':>6s}  {'fused+energy ms':>15s}")
for n in ns:
    for mk in (129, 144, 160, 176, 192, 208, 224, 240, 256):
        j = 8192
        a = torch.randn(mk, n, dtype=torch.float64, device="cuda") / mk ** 0.5
        lam = torch.rand(mk, dtype=torch.float64, device="cuda") + 0.5
        basis = OrthonormalBasis.from_projection(a, lam)
        basis.workspace_bytes = 8 << 30
        y = torch.poisson(torch.rand(n, dtype=torch.float64) * 4)
        u = torch.randn(mk, j, dtype=torch.float64, device="cuda").abs() + 0.5
        out = torch.empty_like(u)
        en = torch.empty(j, dtype=torch.float64, device="cuda")
        for cname, cost in (("gaussian", GaussianCost(0.5, y, IdentityLinkFunction())), ("poisson", PoissonCost(y, SquareLinkFunction()))):
            f = lambda: basis.fused_step(cost, u, 1e-9, out=out, new_state=True, noise=NoiseSpec(none=True), force_generic=True)
            fe = lambda: basis.fused_step(cost, u, 1e-9, out=out, new_state=True, noise=NoiseSpec(none=True), force_generic=True, input_energy=en)
            res = {}
            for name, limit in (("gemm", 0), ("fused", 256)):
                L.check(lib.pls_set_option(L.OPT_SMALL_RANK2_MAX, limit))
                basis._ws.clear()
                res[name] = timeit(f)
                if name == "fused":
                    res["fused_e"] = timeit(fe)
            fl = 4.0 * n * mk * j
            print(f"{n:8d} {mk:5d}  {cname:8s} {res['gemm']:12.3f} {fl / res['gemm'] / 78.6e9:6.3f} {res['fused']:9.3f} {fl / res['fused'] / 78.6e9:6.3f}  {res['fused_e']:15.3f}", flush=True)
        del a, basis, u, out
L.check(lib.pls_set_option(L.OPT_SMALL_RANK2_MAX, 240))
L.check(lib.pls_set_option(L.OPT_SMALL_RANK2_MIN, 161))
