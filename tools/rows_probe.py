"""Per-launch durations (library timeline) of the like-for-like step at ranks that are not a multiple of 128: forward launch,
back-projection launches (row-block kernel vs 128-row tiles + remainder pieces), update.  N = 1e5, J = 8192, Gaussian."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import _lib as L
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction

lib = L.load()

ranks = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [129, 144, 160, 176, 192, 208, 224, 240, 272, 288, 300, 320, 352, 400, 500, 1000]
n, j = 100000, 8192
for mk in ranks:
    a = torch.randn(mk, n, dtype=torch.float64, device="cuda") / mk ** 0.5
    lam = torch.rand(mk, dtype=torch.float64, device="cuda") + 0.5
    basis = OrthonormalBasis.from_projection(a, lam)
    basis.workspace_bytes = 8 << 30
    y = torch.randn(n, dtype=torch.float64)
    u = torch.randn(mk, j, dtype=torch.float64, device="cuda")
    out = torch.empty_like(u)
    cost = GaussianCost(0.5, y, IdentityLinkFunction())
    f = lambda: basis.fused_step(cost, u, 1e-9, out=out, new_state=True, noise=NoiseSpec(none=True), force_generic=True)
    for rows in (0, 1):
        L.check(lib.pls_set_option(L.OPT_ROW_BLOCKS, rows))
        basis._ws.clear()
        for _ in range(2): f()
        torch.cuda.synchronize()
        with L.Timeline(capacity=256) as tl:
            for _ in range(3): f()
        per = {}
        for name, ms in tl.records:
            per.setdefault(name, []).append(ms)
        line = "  ".join(f"{k}: {len(v) // 3} x, {sum(v) / 3:.3f} ms/step" for k, v in per.items())
        print(f"M_k {mk:5d} rows={rows}  {line}", flush=True)
    del a, basis, u, out
