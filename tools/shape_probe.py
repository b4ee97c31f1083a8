"""Does the forward GEMM's efficiency depend on the row stride of A (k-rows 8*N bytes apart)?"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import projected_langevin_sampling_amd as pkg
L = pkg._lib; lib = L.load(); dev = "cuda"
J, K = 8192, 1024
for I in (32768, 65536, 100000, 100096, 131072, 16384):
    Lm = torch.randn(K, I, dtype=torch.float64, device=dev); Rm = torch.randn(K, J, dtype=torch.float64, device=dev)
    C = torch.empty(I, J, dtype=torch.float64, device=dev)
    f = lambda: L.check(lib.pls_gemm_tn(Lm.data_ptr(), I, Rm.data_ptr(), J, C.data_ptr(), J, I, J, K, 1.0, 0.0, L.stream_ptr()))
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print(f"I={I:7d}: {dt*1e3:7.3f} ms  {2.0*I*J*K/dt/1e12:6.2f} TF/s", flush=True)
    del Lm, Rm, C
