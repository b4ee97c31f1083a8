"""Like-for-like step for ranks that are not a multiple of 128 (129 .. 256, 300, 1000), N = 1e5, J = 8192, Gaussian and
Poisson costs: the two-GEMM path with the back-projection as 128-row tiles + remainder launches (round 2), the same with
the back-projection in row blocks (csrc/gemm_tn_f64_rows.h, the default), and the wave-pair fused kernel
(csrc/small_rank2.h, pls_set_option(PLS_OPT_SMALL_RANK2_MAX, 256)).  -> profiles/r03_step_sweep_ranks.txt"""
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
print(f"{'N':>8s} {'M_k':>5s}  {'cost':8s} {'pieces ms':>10s} {'frac':>6s} {'row blocks ms':>13s} {'frac':>6s} {'fused ms':>9s} {'frac':>6s}  {'fused+energy ms':>15s}"
      "     (pieces: two-GEMM path, 128-row tiles + remainder launches; row blocks: two-GEMM path, gemm_tn_f64_rows.h; fused: small_rank2.h)")
for n in ns:
    for mk in (129, 144, 160, 176, 192, 208, 224, 240, 256, 300, 1000):
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
            for name, limit, rows in (("gemm", 0, 0), ("rows", 0, 1), ("fused", 256, 1)):
                if name == "fused" and mk > 256:
                    res["fused"] = res["fused_e"] = float("nan")
                    continue
                L.check(lib.pls_set_option(L.OPT_SMALL_RANK2_MAX, limit))
                L.check(lib.pls_set_option(L.OPT_ROW_BLOCKS, rows))
                basis._ws.clear()
                res[name] = timeit(f)
                if name == "fused":
                    res["fused_e"] = timeit(fe)
            fl = 4.0 * n * mk * j
            print(f"{n:8d} {mk:5d}  {cname:8s} {res['gemm']:10.3f} {fl / res['gemm'] / 78.6e9:6.3f} {res['rows']:13.3f} {fl / res['rows'] / 78.6e9:6.3f} "
                  f"{res['fused']:9.3f} {fl / res['fused'] / 78.6e9:6.3f}  {res['fused_e']:15.3f}", flush=True)
        del a, basis, u, out
L.check(lib.pls_set_option(L.OPT_SMALL_RANK2_MAX, 0))
L.check(lib.pls_set_option(L.OPT_SMALL_RANK2_MIN, 161))
