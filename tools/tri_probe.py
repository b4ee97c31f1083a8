"""Triangular products on a narrow particle shard (M = 1024, J = 1024 / 2048: a rank of an 8- / 4-GPU run): forward solve
S = L_c^-1 U, full solve, un-whitening U = L_c S and the per-call inducing-point Gaussian step that chains them, with the
balanced kernel (PLS_OPT_TRI_BALANCE 1: tile rows paired, two workgroups per pair) and without.  us per call, back-to-back
launches in one timed region; interleaved repetitions so that both variants see the same clock."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import _lib as L
from projected_langevin_sampling_amd.basis import InducingPointBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction

lib = L.load()
torch.manual_seed(0)
torch.set_default_dtype(torch.float64)


def region(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


ms = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1024]
js = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1024, 2048]
for m in ms:
    n, d = 20000, 8
    g = torch.Generator().manual_seed(0)
    x = torch.rand(n, d, generator=g) * 2 - 1
    z = x[:m].clone()
    y = torch.sin(2 * x.sum(dim=1)) + 0.1 * torch.randn(n, generator=g)
    ls = 0.5 + torch.rand(d, generator=g)
    basis = InducingPointBasis(P.PLSKernel(P.ARDKernel(ls, 1.0), z), z, y[:m], x)
    cost = GaussianCost(0.1, y, IdentityLinkFunction())
    f = basis._chol
    for j in js:
        u = torch.randn(m, j, device="cuda")
        out = torch.empty_like(u)
        en = torch.empty(j, device="cuda")
        s = basis.whiten(u)
        reps = 300
        flop = 2.0 * m * m * j
        cases = {
            "forward solve  S = Lc^-1 U": (lambda: f.forward_solve(u), flop / 2),
            "solve  V = k(Z,Z)^-1 U": (lambda: f.solve(u), flop),
            "unwhiten  U = Lc S": (lambda: basis.unwhiten(s, out=out), flop / 2),
            "step per call [folded operator]": (lambda: basis.fused_step(cost, u, 1e-6, out=out, new_state=True, noise=NoiseSpec(seed=1, step=3)), 1.5 * flop),
            "step per call + energy": (lambda: basis.fused_step(cost, u, 1e-6, out=out, new_state=True, noise=NoiseSpec(seed=1, step=3), input_energy=en), 2 * flop),
            "whitened step (one contraction)": (lambda: basis.whitened_step(cost, s, 1e-6, out=out, new_state=True, noise=NoiseSpec(seed=1, step=3)), flop),
        }
        def three_launch():
            L.check(lib.pls_set_option(L.OPT_IPB_STEP_OPERATOR, 0))
            basis.fused_step(cost, u, 1e-6, out=out, new_state=True, noise=NoiseSpec(seed=1, step=3))
            L.check(lib.pls_set_option(L.OPT_IPB_STEP_OPERATOR, 1))

        cases["step per call [solve, Q S, Lc dS]"] = (three_launch, 2 * flop)
        for name, (fn, fl) in cases.items():
            t = {0: [], 1: []}
            for rep in range(3):
                for bal in (0, 1):
                    L.check(lib.pls_set_option(L.OPT_TRI_BALANCE, bal))
                    t[bal].append(region(fn, reps))
            L.check(lib.pls_set_option(L.OPT_TRI_BALANCE, 1))
            a, b = min(t[0]), min(t[1])
            print(f"M={m} J={j:5d} {name:34s} one tile per workgroup {a:7.1f} us ({fl / a / 78.6e6:.3f})   balanced {b:7.1f} us "
                  f"({fl / b / 78.6e6:.3f})   [{', '.join(f'{v:.1f}' for v in t[0])} | {', '.join(f'{v:.1f}' for v in t[1])}]", flush=True)
