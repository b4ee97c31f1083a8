"""A/B of the fused small-rank step between library builds on one box: python tools/ab_smallrank.py libA.so libB.so ...
(each variant runs in a subprocess with PLSHIP_LIBRARY pointing at it)."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import projected_langevin_sampling_amd as P
    from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
    from projected_langevin_sampling_amd.costs import PoissonCost, GaussianCost
    from projected_langevin_sampling_amd.link_functions import SquareLinkFunction, IdentityLinkFunction
    torch.manual_seed(0)
    for (n, mk, j, kind) in [(50000, 89, 16384, "poisson"), (50000, 89, 16384, "gaussian"), (50000, 128, 16384, "poisson"), (50000, 64, 16384, "poisson")]:
        a = torch.randn(mk, n, dtype=torch.float64, device="cuda") / mk ** 0.5
        lam = torch.rand(mk, dtype=torch.float64, device="cuda") + 0.5
        basis = OrthonormalBasis.from_projection(a, lam)
        y = torch.poisson(torch.rand(n, dtype=torch.float64) * 4)
        cost = PoissonCost(y, SquareLinkFunction()) if kind == "poisson" else GaussianCost(0.5, y, IdentityLinkFunction())
        u = torch.randn(mk, j, dtype=torch.float64, device="cuda")
        out = torch.empty_like(u)
        f = lambda: basis.fused_step(cost, u, 1e-9, out=out, new_state=True, noise=NoiseSpec(none=True), force_generic=True)
        for _ in range(3): f()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        print(f"   N={n} Mk={mk} J={j} {kind:8s}: {best:.3f} ms  ({4.0 * n * mk * j / best / 1e9:.1f} TF/s)", flush=True)
    sys.exit(0)
for lib in sys.argv[1:]:
    print(lib, flush=True)
    env = dict(os.environ, PLSHIP_LIBRARY=os.path.abspath(lib))
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=False)
