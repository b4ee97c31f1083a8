"""First-contact GPU sanity: MFMA layout, GEMM edges, kernel build, RNG moments, raw timings."""
import importlib.util, time, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("plslib", os.path.join(ROOT, "projected-langevin-sampling_amd", "_lib.py"))
L = importlib.util.module_from_spec(spec); spec.loader.exec_module(L)
lib = L.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
print(torch.cuda.get_device_name(0), flush=True)

def gemm(Lm, Rm, alpha=1.0, beta=0.0, C=None):
    K, I = Lm.shape; K2, J = Rm.shape; assert K == K2
    if C is None: C = torch.empty(I, J, dtype=torch.float64, device=dev)
    L.check(lib.pls_gemm_tn(Lm.data_ptr(), Lm.stride(0), Rm.data_ptr(), Rm.stride(0), C.data_ptr(), C.stride(0), I, J, K, alpha, beta, L.stream_ptr()))
    return C

ok = True
for (I, J, K) in [(16, 16, 4), (64, 64, 16), (128, 128, 64), (130, 70, 33), (1, 1, 1), (257, 3, 5), (3, 1000, 7), (2048, 2048, 512), (4099, 2051, 1027), (100, 64, 10)]:
    Lm = torch.randn(K, I, dtype=torch.float64, device=dev); Rm = torch.randn(K, J, dtype=torch.float64, device=dev)
    C = gemm(Lm, Rm); ref = (Lm.cpu().T @ Rm.cpu())
    err = (C.cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-300)
    print(f"gemm I={I} J={J} K={K} relerr={err:.2e}", flush=True); ok &= err < 1e-12
# odd leading dimensions (scalar load path) + alpha/beta
Lb = torch.randn(37, 131, dtype=torch.float64, device=dev)[:, :129]; Rb = torch.randn(37, 77, dtype=torch.float64, device=dev)[:, 1:76]
C0 = torch.randn(129, 75, dtype=torch.float64, device=dev); C = C0.clone()
gemm(Lb, Rb, 0.5, 2.0, C); ref = 0.5 * (Lb.cpu().T @ Rb.cpu()) + 2.0 * C0.cpu()
err = (C.cpu() - ref).abs().max().item(); print("gemm odd-ld alpha/beta abs err", err); ok &= err < 1e-11

# kernel gram
for (n1, n2, d) in [(5, 7, 3), (1024, 4096, 8), (33, 1001, 1), (17, 64, 20)]:
    x1 = torch.randn(n1, d, dtype=torch.float64, device=dev); x2 = torch.randn(n2, d, dtype=torch.float64, device=dev)
    ls = torch.rand(d, dtype=torch.float64, device=dev) + 0.5
    out = torch.empty(n1, n2, dtype=torch.float64, device=dev)
    L.check(lib.pls_kernel_gram(0, x1.data_ptr(), n1, x2.data_ptr(), n2, d, ls.data_ptr(), 1.7, out.data_ptr(), n2, L.stream_ptr()))
    a = (x1 / ls).cpu(); b = (x2 / ls).cpu(); ref = 1.7 * torch.exp(-0.5 * (a[:, None, :] - b[None]).square().sum(-1))
    err = (out.cpu() - ref).abs().max().item(); print(f"rbf n1={n1} n2={n2} d={d} abs err {err:.2e}"); ok &= err < 1e-13
    L.check(lib.pls_kernel_gram(1, x1.data_ptr(), n1, x2.data_ptr(), n2, d, None, 1.0, out.data_ptr(), n2, L.stream_ptr()))
    err = (out.cpu() - x1.cpu() @ x2.cpu().T).abs().max().item(); print(f"linear abs err {err:.2e}"); ok &= err < 1e-12

# rng moments
z = torch.empty(1024, 4096, dtype=torch.float64, device=dev)
L.check(lib.pls_normal_fill(z.data_ptr(), 4096, 1024, 4096, 1234, 0, 0, L.stream_ptr()))
print("normal mean %.4e var %.5f kurt %.4f" % (z.mean().item(), z.var().item(), (z**4).mean().item()))
ok &= abs(z.mean().item()) < 3e-3 and abs(z.var().item() - 1) < 5e-3 and abs((z**4).mean().item() - 3) < 3e-2

# timings
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for (I, J, K) in [(1024, 8192, 1024), (8192, 8192, 1024), (1024, 8192, 16384), (16384, 8192, 1024)]:
    Lm = torch.randn(K, I, dtype=torch.float64, device=dev); Rm = torch.randn(K, J, dtype=torch.float64, device=dev)
    C = torch.empty(I, J, dtype=torch.float64, device=dev)
    t = timeit(lambda: gemm(Lm, Rm, C=C)); fl = 2.0 * I * J * K
    t2 = timeit(lambda: torch.matmul(Lm.T, Rm, out=C))
    print(f"gemm_tn I={I} J={J} K={K}: {t*1e3:.3f} ms {fl/t/1e12:.2f} TF/s | torch(rocBLAS) {t2*1e3:.3f} ms {fl/t2/1e12:.2f} TF/s", flush=True)
x1 = torch.randn(1024, 8, dtype=torch.float64, device=dev); x2 = torch.randn(100000, 8, dtype=torch.float64, device=dev); ls = torch.ones(8, dtype=torch.float64, device=dev)
out = torch.empty(1024, 100000, dtype=torch.float64, device=dev)
t = timeit(lambda: L.check(lib.pls_kernel_gram(0, x1.data_ptr(), 1024, x2.data_ptr(), 100000, 8, ls.data_ptr(), 1.0, out.data_ptr(), 100000, L.stream_ptr())))
print(f"rbf 1024x1e5 d=8: {t*1e3:.3f} ms, {out.numel()*8/t/1e9:.0f} GB/s written")
t = timeit(lambda: L.check(lib.pls_normal_fill(z.data_ptr(), 4096, 1024, 4096, 1, 0, 0, L.stream_ptr())))
print(f"normal_fill 1024x4096: {t*1e6:.1f} us")
print("SANITY", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
