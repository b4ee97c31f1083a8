"""A/B of the Gaussian fast-path launch at the full configs[1] width (M_k = 1024, J = 8192) between the shipped tiling
(128 x 128 tiles, 512 workgroups = exactly one round of two per CU) and the k-split 64 x 64 tiling of
csrc/gemm_tn_f64_kg.h (2048 workgroups of 8 waves, two resident per CU, i.e. four rounds with dynamic refill: one tile's
epilogue and Philox can overlap another's k-loop).  Interleaved rounds in ONE process, 200 launches per arm and round;
run it under `rocprofv3 --kernel-trace --stats` / `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES` for the
per-kernel durations and the MFMA-pipe utilisation of the two kernels (tools/profile_cmd.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import _lib as L
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction

lib = L.load()
torch.manual_seed(0)
mk, n, j = 1024, 4096, 8192
a = torch.randn(mk, n, dtype=torch.float64, device="cuda") / mk ** 0.5
lam = torch.rand(mk, dtype=torch.float64, device="cuda") + 0.5
basis = OrthonormalBasis.from_projection(a, lam)
cost = GaussianCost(observation_noise=0.5, y_train=torch.randn(n, dtype=torch.float64), link_function=IdentityLinkFunction())
u = torch.randn(mk, j, dtype=torch.float64, device="cuda")
out = torch.empty_like(u)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def arm(max_tiles):
    L.check(lib.pls_set_option(L.OPT_KSPLIT_MAX_TILES, max_tiles))
    step = lambda: basis.fused_step(cost, u, 1e-6, out=out, new_state=True, noise=NoiseSpec(seed=1, step=3))
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


flop = 2.0 * mk * mk * j
for rnd in range(3):
    t_big = arm(256)        # shipped: 128 x 128 tiles (512 tiles >= 256)
    t_kg = arm(1 << 30)     # every shape through the k-split 64 x 64 kernel
    print(f"round {rnd}: 128x128 tiles {t_big:7.1f} us ({flop / t_big / 78.6e6:.3f} of peak)   "
          f"64x64 k-split tiles {t_kg:7.1f} us ({flop / t_kg / 78.6e6:.3f})", flush=True)
L.check(lib.pls_set_option(L.OPT_KSPLIT_MAX_TILES, 256))
