"""Fast-path (B = A A^T) step at the C2 shape: where do the 0.30 ms go?  Times the fused kernel with Philox noise,
injected noise, no noise, with/without the input-energy by-product, and the bare pls_gemm_tn of the same shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import _lib as L
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction
torch.manual_seed(0)
mk, n, j = 1024, 20000, 8192
a = torch.randn(mk, n, dtype=torch.float64, device="cuda") / mk ** 0.5
lam = torch.rand(mk, dtype=torch.float64, device="cuda") + 0.5
basis = OrthonormalBasis.from_projection(a, lam)
y = torch.randn(n, dtype=torch.float64)
cost = GaussianCost(observation_noise=0.5, y_train=y, link_function=IdentityLinkFunction())
u = torch.randn(mk, j, dtype=torch.float64, device="cuda")
out = torch.empty_like(u)
xi = torch.randn(mk, j, dtype=torch.float64, device="cuda")
en = torch.empty(j, dtype=torch.float64, device="cuda")
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
cases = {
    "philox": lambda: basis.fused_step(cost, u, 1e-6, out=out, new_state=True, noise=NoiseSpec(seed=1, step=3)),
    "philox + input energy": lambda: basis.fused_step(cost, u, 1e-6, out=out, new_state=True, noise=NoiseSpec(seed=1, step=3), input_energy=en),
    "injected": lambda: basis.fused_step(cost, u, 1e-6, out=out, new_state=True, noise=NoiseSpec(injected=xi)),
    "no noise": lambda: basis.fused_step(cost, u, 1e-6, out=out, new_state=True, noise=NoiseSpec(none=True)),
}
for rep in range(2):  # (the first pass also warms the clocks up)
    for k, f in cases.items():
        print(f"{k:24s} {timeit(f):.4f} ms", flush=True)
B = basis._B
lib = L.load()
c = torch.empty(mk, j, dtype=torch.float64, device="cuda")
print(f"{'bare gemm_tn (store)':24s} {timeit(lambda: lib.pls_gemm_tn(B.data_ptr(), L.ld(B), u.data_ptr(), j, c.data_ptr(), j, mk, j, mk, 1.0, 0.0, None)):.4f} ms")
z = torch.empty(mk, j, dtype=torch.float64, device="cuda")
print(f"{'normal_fill alone':24s} {timeit(lambda: lib.pls_normal_fill(z.data_ptr(), j, mk, j, 1, 2, 0, None)):.4f} ms")
print("ideal MFMA time at 78.6 TF/s: %.4f ms" % (2.0 * mk * mk * j / 78.6e12 * 1e3))

# ---- round 2: is a stream of these launches slower than the launch itself?  Same kernel, (a) fixed input / output buffers,
# (b) ping-pong buffers with the noise counter advancing (what bench.py and a training loop do), 300 launches each, timed
# as one region; then (c) per-launch events (pls_timeline) around the same 300 launches.
def region(fn, reps=300):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
state = {"a": u.clone(), "b": torch.empty_like(u), "t": 0}
def pingpong():
    basis.fused_step(cost, state["a"], 1e-6, out=state["b"], new_state=True, noise=NoiseSpec(seed=1, step=state["t"]))
    state["a"], state["b"] = state["b"], state["a"]
    state["t"] += 1
for rep in range(2):
    print(f"region, fixed buffers     {region(cases['philox']):.4f} ms/launch", flush=True)
    print(f"region, ping-pong + steps {region(pingpong):.4f} ms/launch", flush=True)
with L.Timeline(capacity=400) as tl:
    for _ in range(300): pingpong()
print("per-launch events, ping-pong:", {k: round(v["avg_ms"], 4) for k, v in tl.summary().items()})
import time
t0 = time.perf_counter()
for _ in range(300): pingpong()
t_host = (time.perf_counter() - t0) / 300
torch.cuda.synchronize()
print(f"host time per enqueue (queue never drained): {t_host * 1e3:.4f} ms")
