"""Narrow particle shards (what a rank of a 2/4/8-GPU run holds): the Gaussian fast-path step and the bare M x M x J
contraction at M_k = 1024 (and 4096) for J = 512 .. 8192, with the k-split 64 x 64 kernel (gemm_tn_f64_kg.h) on / off.
Prints us per launch (back-to-back launches, one timed region) and the fraction of the 78.6 TF/s fp64 MFMA peak."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import _lib as L
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction

lib = L.load()
torch.manual_seed(0)


def region(fn, reps):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def graph_region(fn, reps, k=20):
    """the same launches replayed from a hipGraph of k launches (no host launch overhead between them)"""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(k):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(max(1, reps // k)):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (max(1, reps // k) * k) * 1e3


MODES = {"old": (0, 256), "ksplit": (1, 256), "ksplit<512": (1, 512), "ksplit all": (2, 256), "1 group": (3, 256)}
mks = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1024]
for mk in mks:
    n = 4096
    a = torch.randn(mk, n, dtype=torch.float64, device="cuda") / mk ** 0.5
    lam = torch.rand(mk, dtype=torch.float64, device="cuda") + 0.5
    basis = OrthonormalBasis.from_projection(a, lam)
    y = torch.randn(n, dtype=torch.float64)
    cost = GaussianCost(observation_noise=0.5, y_train=y, link_function=IdentityLinkFunction())
    basis.fused_step(cost, torch.randn(mk, 64, dtype=torch.float64, device="cuda"), 1e-6, noise=NoiseSpec(seed=1, step=3))
    B = basis._B  # (built lazily by the first Gaussian step)
    for j in (512, 1024, 2048, 4096, 8192):
        u = torch.randn(mk, j, dtype=torch.float64, device="cuda")
        out = torch.empty_like(u)
        c = torch.empty_like(u)
        en = torch.empty(j, dtype=torch.float64, device="cuda")
        flop = 2.0 * mk * mk * j
        reps = 400 if j <= 2048 else 200
        for name, (mode, mt) in MODES.items():
            L.check(lib.pls_set_option(L.OPT_KSPLIT_MODE, mode))
            L.check(lib.pls_set_option(L.OPT_KSPLIT_MAX_TILES, mt))
            basis._ws.clear()
            step = lambda: basis.fused_step(cost, u, 1e-6, out=out, new_state=True, noise=NoiseSpec(seed=1, step=3))
            step_e = lambda: basis.fused_step(cost, u, 1e-6, out=out, new_state=True, noise=NoiseSpec(seed=1, step=3), input_energy=en)
            gemm = lambda: lib.pls_gemm_tn(B.data_ptr(), L.ld(B), u.data_ptr(), j, c.data_ptr(), j, mk, j, mk, 1.0, 0.0, L.stream_ptr())
            t_s, t_g = region(step, reps), region(gemm, reps)
            t_sg = graph_region(step, reps)
            t_se = graph_region(step_e, reps)
            print(f"M_k={mk} J={j:5d} {name:12s} step {t_s:7.1f} us ({flop / t_s / 78.6e6:.3f})  graph {t_sg:7.1f} us ({flop / t_sg / 78.6e6:.3f})"
                  f"  +energy {t_se:7.1f}  bare gemm {t_g:7.1f} us ({flop / t_g / 78.6e6:.3f})", flush=True)
L.check(lib.pls_set_option(L.OPT_KSPLIT_MODE, 1))
L.check(lib.pls_set_option(L.OPT_KSPLIT_MAX_TILES, 256))
