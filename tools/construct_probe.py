"""cProfile of the timed block of the reference's profiler protocol at its default point (N = 100, M = 10, T = 10, J = 100):
where the host time of construction and of the eager step loop goes."""
import cProfile, io, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd.basis import OrthonormalBasis
from projected_langevin_sampling_amd.costs import BernoulliCost, GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction, SigmoidLinkFunction
from profiler_grid import LN2, OBS_NOISE, STEP_SIZE, make_data
torch.set_default_dtype(torch.float64)
n, m, t, j = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else (100, 10, 10, 100)
x, y = make_data(n)
z = x[:: max(1, n // m)][:m].clone()
def block(cost_name):
    kernel = P.PLSKernel(P.ARDKernel(torch.full((1,), LN2), LN2), z)
    basis = OrthonormalBasis(kernel=kernel, x_induce=z, x_train=x, verbose=False)
    cost = GaussianCost(OBS_NOISE, y, IdentityLinkFunction()) if cost_name == "gaussian" else BernoulliCost((y > 0).double(), SigmoidLinkFunction())
    pls = P.PLS(basis=basis, cost=cost)
    particles = pls.initialise_particles(number_of_particles=j, noise_only=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(t):
        particles += pls.calculate_particle_update(particles=particles, step_size=STEP_SIZE)
    torch.cuda.synchronize()
    return t1
for cost_name in ("gaussian", "bernoulli"):
    for _ in range(20): block(cost_name)
    t0 = time.perf_counter(); c = 0.0
    for _ in range(200):
        s = time.perf_counter(); t1 = block(cost_name); c += t1 - s
    w = time.perf_counter() - t0
    print(f"{cost_name}: block {w / 200 * 1e3:.3f} ms = construction {c / 200 * 1e3:.3f} ms + {t} steps {(w - c) / 200 * 1e3:.3f} ms ({(w - c) / 200 / t * 1e6:.1f} us/step)")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): block(cost_name)
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
