"""Where the fixed cost of a few-tiles launch goes (profiles/r04_kg_k_sweep.txt: time = 4.9 us + 0.943 us per 32 k-rows at
M_k = 1024, J = 1024): s_memtime stamps inside gemm_tn_f64_kg_kernel<2, EpiLangevinGaussian> -- entry, first operand rows
landed, contraction done, epilogue's stores drained -- for the fused Gaussian step on one rank's shard of an 8-GPU run,
next to an EMPTY launch of the same geometry (256 workgroups x 512 threads, the same LDS) timed the same way.
Build the stamp library first (read the SHARES: the stamps serialise a little of what the kernel overlaps):
  (cd projected-langevin-sampling_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DPLS_STAMP -shared -o ../../tools/libplship_stamp.so \
     plship.hip gemm_cost.hip gemm_cost_value.hip small_rank_drift.hip small_rank_value.hip small_rank_drift_value.hip small_rank_step.hip small_rank_step_value.hip chol.hip)
  PLSHIP_LIBRARY=$PWD/tools/libplship_stamp.so python tools/kg_stamp_probe.py [M_k J]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import _lib as L
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction
torch.set_default_dtype(torch.float64)
lib = L.load()
if os.environ.get("PLS_PREGEN") is not None:
    lib.pls_set_option(L.OPT_KG_NOISE_PREGEN, int(os.environ["PLS_PREGEN"]))
raw = C.CDLL(os.environ["PLSHIP_LIBRARY"])
raw.pls_debug_set_stamp_buffer.argtypes = [C.c_void_p]
raw.pls_debug_empty_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
mk, j = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 1024)
g = torch.Generator().manual_seed(0)
n = 4096
a = (torch.randn(mk, n, generator=g) / n ** 0.5).cuda()
basis = OrthonormalBasis.from_projection(a, (torch.rand(mk, generator=g) + 0.5).cuda())
cost = GaussianCost(0.1, torch.randn(n, generator=g), IdentityLinkFunction())
u = torch.randn(mk, j, generator=g).cuda(); out = torch.empty_like(u)
step = lambda: basis.fused_step(cost, u, 1e-6, out=out, new_state=True, noise=NoiseSpec(seed=1, step=0))
for _ in range(5): step()
torch.cuda.synchronize()
ntiles = ((mk + 63) // 64) * ((j + 63) // 64)
raw.pls_debug_calibrate.argtypes = [C.c_void_p, C.c_void_p]
cal = torch.zeros(2, dtype=torch.int64, device="cuda")
raw.pls_debug_calibrate(cal.data_ptr(), L.stream_ptr()); torch.cuda.synchronize()
wall_khz = 100000  # hipDeviceAttributeWallClockRate: 100 MHz on this part
TICK_US = (cal[1].item() / (wall_khz * 1e-3)) / cal[0].item()  # microseconds per s_memtime tick
print(f"s_memtime: {1.0 / TICK_US / 1e3:.3f} GHz ({cal[0].item()} ticks over {cal[1].item()} wall-clock ticks at {wall_khz / 1e3:.0f} MHz)")
def timed(fn, reps=200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
raw.pls_debug_set_stamp_buffer(None)
print(f"M_k = {mk}, J = {j}: {ntiles} tiles of 64 x 64; fused Gaussian step {timed(step):.2f} us per launch back to back (no stamps written)")
stream = L.stream_ptr()
es = torch.zeros(ntiles, dtype=torch.int64, device="cuda")
empty = lambda: raw.pls_debug_empty_launch(ntiles, 512, 98304, None, stream)
for _ in range(5): empty()
print(f"an EMPTY launch of {ntiles} workgroups x 512 threads (96 KB LDS): {timed(empty):.2f} us per launch back to back")
# (s_memtime counters of different XCDs are not synchronised: only differences INSIDE a workgroup mean anything)
stamps = torch.zeros(ntiles * 6, dtype=torch.int64, device="cuda")
rows = []
for rep in range(20):
    stamps.zero_()
    raw.pls_debug_set_stamp_buffer(stamps.data_ptr())
    step(); torch.cuda.synchronize()
    raw.pls_debug_set_stamp_buffer(None)
    s = stamps.reshape(ntiles, 6)[:, :4].cpu().double() * TICK_US
    rows.append(torch.stack([(s[:, 1] - s[:, 0]).median(), (s[:, 2] - s[:, 1]).median(), (s[:, 3] - s[:, 2]).median(),
                             (s[:, 3] - s[:, 0]).median(), (s[:, 3] - s[:, 0]).max()]))
r = torch.stack(rows).median(dim=0).values
print(f"inside the step launch (medians over the tiles, then over 20 launches; us):\n"
      f"   entry -> first operand rows in LDS (descriptors, the noise of the tile drawn meanwhile) {r[0]:.2f} | contraction {r[1]:.2f} | "
      f"hand-over between the k-groups + epilogue + stores drained {r[2]:.2f} | a tile in all {r[3]:.2f} (slowest tile {r[4]:.2f})")
k_loop = mk / 32 * 0.943
print(f"   (the k-loop's own rate from the K sweep: {k_loop:.1f} us for {mk} k-rows)")
