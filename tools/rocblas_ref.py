"""Reference point on the same device and data: torch.matmul (rocBLAS/hipBLASLt dgemm) vs pls_gemm_tn on the step's shapes."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import projected_langevin_sampling_amd as pkg
L = pkg._lib; lib = L.load(); dev = "cuda"
torch.manual_seed(0)
for (I, J, K) in [(1024, 8192, 32768), (32768, 8192, 1024)]:
    Lm = torch.randn(K, I, dtype=torch.float64, device=dev); Rm = torch.randn(K, J, dtype=torch.float64, device=dev)
    C = torch.empty(I, J, dtype=torch.float64, device=dev)
    for name, f in (("ours", lambda: L.check(lib.pls_gemm_tn(Lm.data_ptr(), I, Rm.data_ptr(), J, C.data_ptr(), J, I, J, K, 1.0, 0.0, L.stream_ptr()))),
                    ("rocblas", lambda: torch.matmul(Lm.T, Rm, out=C))):
        f(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
        print(f"{name:8s} I={I} J={J} K={K}: {dt*1e3:.3f} ms {2.0*I*J*K/dt/1e12:.2f} TF/s", flush=True)
