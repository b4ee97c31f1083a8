"""How long does a tile's prologue + epilogue take?  pls_gemm_tn at tiny K (the k-loop is a few steps)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import projected_langevin_sampling_amd as pkg
L = pkg._lib; lib = L.load(); dev = "cuda"
I, J = 32768, 8192
for K in (16, 32, 64, 128, 256, 1024):
    Lm = torch.randn(K, I, dtype=torch.float64, device=dev); Rm = torch.randn(K, J, dtype=torch.float64, device=dev)
    C = torch.empty(I, J, dtype=torch.float64, device=dev)
    f = lambda: L.check(lib.pls_gemm_tn(Lm.data_ptr(), I, Rm.data_ptr(), J, C.data_ptr(), J, I, J, K, 1.0, 0.0, L.stream_ptr()))
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    tiles = (I // 128) * (J // 128)
    print(f"K={K:5d}: {dt*1e3:7.3f} ms  {I*J*8/dt/1e12:5.2f} TB/s written  {dt*512/tiles*1e6:6.1f} us per tile-slot  mfma-only {K/16*4096*2/2.35e9*1e6:6.1f} us", flush=True)
