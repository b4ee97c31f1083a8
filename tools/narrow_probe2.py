"""Narrow-shard fast path as a training loop runs it: ping-pong particle buffers (each launch reads what the previous one
wrote), advancing noise counters, replayed from a hipGraph of 20 steps.  M_k = 1024, J in {1024, 2048, 4096}."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import _lib as L
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction

lib = L.load()
torch.manual_seed(0)
mk, n = 1024, 4096
a = torch.randn(mk, n, dtype=torch.float64, device="cuda") / mk ** 0.5
lam = torch.rand(mk, dtype=torch.float64, device="cuda") + 0.5
basis = OrthonormalBasis.from_projection(a, lam)
cost = GaussianCost(observation_noise=0.5, y_train=torch.randn(n, dtype=torch.float64), link_function=IdentityLinkFunction())


def graph_time(body, k, reps):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            body()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * k) * 1e3


for j in (1024, 2048, 4096):
    flop = 2.0 * mk * mk * j
    bufs = [torch.randn(mk, j, dtype=torch.float64, device="cuda"), torch.empty(mk, j, dtype=torch.float64, device="cuda")]
    en = torch.empty(j, dtype=torch.float64, device="cuda")
    k = 20

    def fixed():
        for s in range(k):
            basis.fused_step(cost, bufs[0], 1e-6, out=bufs[1], new_state=True, noise=NoiseSpec(seed=1, step=s))

    def pingpong():
        for s in range(k):
            basis.fused_step(cost, bufs[s & 1], 1e-6, out=bufs[(s + 1) & 1], new_state=True, noise=NoiseSpec(seed=1, step=s))

    def pingpong_energy():
        for s in range(k):
            basis.fused_step(cost, bufs[s & 1], 1e-6, out=bufs[(s + 1) & 1], new_state=True, noise=NoiseSpec(seed=1, step=s), input_energy=en)

    counter = torch.zeros(1, dtype=torch.int64, device="cuda")

    def pingpong_counter():
        for s in range(k):
            basis.fused_step(cost, bufs[s & 1], 1e-6, out=bufs[(s + 1) & 1], new_state=True,
                             noise=NoiseSpec(seed=1, step=s, step_base=counter))
        L.check(lib.pls_counter_add(counter.data_ptr(), k, L.stream_ptr()))

    for name, body in (("fixed buffers", fixed), ("ping-pong", pingpong), ("ping-pong + energy", pingpong_energy),
                       ("ping-pong, device step counter", pingpong_counter)):
        t = graph_time(body, k, 50)
        print(f"J={j:5d} {name:20s} {t:7.1f} us/step ({flop / t / 78.6e6:.3f} of peak)", flush=True)
