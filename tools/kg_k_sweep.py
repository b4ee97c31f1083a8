"""Fixed cost of a few-tiles launch: C (1024 x J) = L^T R with K = 32 ... 4096 rows through pls_gemm_tn (the k-split 64 x 64
kernel, plain store epilogue), back-to-back launches in one timed region.  time(K) = c0 + c1 K / 32: c1 is the k-loop per
32-row super-step (0.85 us at the fp64 MFMA peak of one CU), c0 what a launch costs before and after it (dispatch, first
operand rows from a cold L2, hand-over between the k-groups, stores, end-of-kernel write-back)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from projected_langevin_sampling_amd import _lib as L, _ops
from projected_langevin_sampling_amd.basis.base import alloc_matrix

lib = L.load()
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


for j in (1024, 2048):
    pts = []
    for k in (32, 64, 128, 256, 512, 1024, 2048, 4096):
        l, r, c = alloc_matrix(k, 1024, "cuda"), alloc_matrix(k, j, "cuda"), alloc_matrix(1024, j, "cuda")
        l.normal_()
        r.normal_()
        t = min(region(lambda: _ops.gemm_tn(l, r, out=c), 300) for _ in range(3))
        pts.append((k / 32, t))
        print(f"I=1024 J={j} K={k:5d}: {t:7.2f} us  ({2.0 * 1024 * j * k / t / 78.6e6:.3f} of peak)", flush=True)
    (x0, y0), (x1, y1) = pts[3], pts[-1]
    c1 = (y1 - y0) / (x1 - x0)
    print(f"  -> c1 = {c1:.3f} us per 32-row super-step, c0 = {y0 - c1 * x0:.2f} us (from K = 256 and 4096); K = 32: {pts[0][1]:.2f} us")
