"""A/B timing of pls_gemm_tn between library builds on ONE box (boxes differ by a few % in sustained clock).
usage: python tools/ab_gemm.py libA.so libB.so ...   -- interleaved repetitions, median ms per shape."""
import ctypes as C, sys, statistics, torch
libs = []
for p in sys.argv[1:]:
    l = C.CDLL(p)
    l.pls_gemm_tn.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_void_p]
    libs.append((p, l))
dev = "cuda"
for (I, J, K) in [(100000, 8192, 1024), (1024, 8192, 100000)]:
    Lm = torch.randn(K, I, dtype=torch.float64, device=dev); Rm = torch.randn(K, J, dtype=torch.float64, device=dev)
    Cm = torch.empty(I, J, dtype=torch.float64, device=dev)
    res = {p: [] for p, _ in libs}
    for rep in range(7):
        for p, l in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                l.pls_gemm_tn(Lm.data_ptr(), I, Rm.data_ptr(), J, Cm.data_ptr(), J, I, J, K, 1.0, 0.0, None)
            e1.record(); torch.cuda.synchronize()
            if rep: res[p].append(e0.elapsed_time(e1) / 3)
    for p, _ in libs:
        v = res[p]
        fl = 2.0 * I * J * K
        print(f"I={I} J={J} K={K} {p:40s} median {statistics.median(v):8.3f} ms  min {min(v):8.3f}  {fl / statistics.median(v) / 1e9:6.2f} TF/s", flush=True)
