"""What the energy by-product costs inside the fused Gaussian step (M_k = 1024): the step alone, with the by-product and a
finishing launch (PLS_OPT_ENERGY_FUSED_FINISH 0), with the by-product finished by the step launch itself (energy_sync), with
the partial rows finished by the NEXT launch at its start (energy_partials: lagged energies), as
hipGraph replays of 20 steps (no host in the loop), for the shard of an 8- / 4-GPU run and the full particle matrix."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import _lib as L
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.basis.base import BlockSpec
from projected_langevin_sampling_amd.costs import GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction

lib = L.load()
if os.environ.get("PLS_KSPLIT_MAX_TILES"):  # e.g. 512: J = 4096 (256 tiles of 128 x 128) through the k-split kernel as well
    L.check(lib.pls_set_option(L.OPT_KSPLIT_MAX_TILES, int(os.environ["PLS_KSPLIT_MAX_TILES"])))
    print("PLS_OPT_KSPLIT_MAX_TILES =", os.environ["PLS_KSPLIT_MAX_TILES"], flush=True)
torch.set_default_dtype(torch.float64)
mk, n = 1024, 20000
g = torch.Generator().manual_seed(0)
a = (torch.randn(mk, n, generator=g) / mk ** 0.5).cuda()
lam = (torch.rand(mk, generator=g) + 0.5).cuda()
basis = OrthonormalBasis.from_projection(a, lam)
cost = GaussianCost(0.5, torch.randn(n, generator=g), IdentityLinkFunction())


def graph_time(fn, steps=20, reps=30):
    fn(); torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(gr, stream=side):
            for _ in range(steps):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    best = 1e9
    for _ in range(3):
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            gr.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (reps * steps) * 1e3)
    return best


for pregen, j in [(p, int(v)) for p in (1, 0) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1024", "2048", "8192"])]:
    if pregen == 0 and j >= 4096 and not os.environ.get("PLS_KSPLIT_MAX_TILES"):
        continue  # (the 128 x 128 tiles of the full-width launch draw their noise in the epilogue either way)
    L.check(lib.pls_set_option(L.OPT_KG_NOISE_PREGEN, pregen))
    u, out = torch.randn(mk, j, device="cuda"), torch.empty(mk, j, device="cuda")
    e = torch.empty(j, device="cuda")
    nchunk = (j + 255) // 256
    sums, sync = torch.empty(nchunk, device="cuda"), torch.zeros(nchunk, dtype=torch.int32, device="cuda")
    eta = torch.full((1,), 1e-7, device="cuda")
    ws = torch.empty(64 * j, device="cuda")
    ns = NoiseSpec(seed=1, step=3)
    plain = lambda: basis.fused_step(cost, u, 1e-7, out=out, new_state=True, noise=ns)
    with_e = lambda: basis.fused_step(cost, u, 1e-7, out=out, new_state=True, noise=ns, input_energy=e, workspace=ws,
                                      blocks=BlockSpec(j, eta, energy_sums=sums.data_ptr()))
    fused = lambda: basis.fused_step(cost, u, 1e-7, out=out, new_state=True, noise=ns, input_energy=e, workspace=ws,
                                     blocks=BlockSpec(j, eta, energy_sums=sums.data_ptr(), energy_sync=sync))
    parts = [torch.empty(basis.energy_partial_rows_bytes(j) // 8, device="cuda") for _ in (0, 1)]
    turn = [0]

    def lagged():
        k = turn[0]
        turn[0] = k ^ 1
        basis.fused_step(cost, u, 1e-7, out=out, new_state=True, noise=ns, workspace=ws,
                         blocks=BlockSpec(j, eta, energy_partials=parts[k], energy_partials_prev=parts[k ^ 1], energy_prev=e,
                                          energy_sums_prev=sums.data_ptr()))

    t0, t1, t2, t3 = graph_time(plain), graph_time(with_e), graph_time(fused), graph_time(lagged)
    print(f"M_k={mk} J={j:5d} noise {'in front of the k-loop' if pregen and j < 4096 else 'in the epilogue        '}: step {t0:7.2f} us | + energy, finishing launch {t1:7.2f} us (+{t1 - t0:.2f}) | + energy, finished by the step "
          f"launch {t2:7.2f} us (+{t2 - t0:.2f}) | + energy, finished by the NEXT launch {t3:7.2f} us (+{t3 - t0:.2f})", flush=True)
