"""Where does the basis setup time go at configs[1] (N=1e5, M=1024, D=8)?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t00 = time.perf_counter()
import projected_langevin_sampling_amd as pkg
from projected_langevin_sampling_amd.basis import OrthonormalBasis
from projected_langevin_sampling_amd.costs import GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction
def T(label, t0):
    torch.cuda.synchronize(); print(f"{label:40s} {time.perf_counter() - t0:7.3f} s", flush=True)
T("import package", t00)
g = torch.Generator().manual_seed(0)
n, m, d = 100_000, 1024, 8
x = torch.rand(n, d, generator=g, dtype=torch.float64) * 2 - 1
z = x[torch.randperm(n, generator=g)[:m]].clone()
y = torch.randn(n, generator=g, dtype=torch.float64)
ls = 0.5 + torch.rand(d, dtype=torch.float64)
t0 = time.perf_counter(); torch.zeros(1, device="cuda"); T("first CUDA touch", t0)
for rep in range(2):
    print(f"-- pass {rep}")
    t0 = time.perf_counter(); kernel = pkg.PLSKernel(pkg.ARDKernel(ls, 1.0), z); T("PLSKernel", t0)
    t0 = time.perf_counter(); kzz = kernel.base_kernel(x1=z, x2=z); T("k(Z,Z) (incl. H2D)", t0)
    t0 = time.perf_counter(); kzx = kernel.base_kernel(x1=z, x2=x); T("k(Z,X) (incl. H2D of X)", t0)
    t0 = time.perf_counter(); kc = kzz.cpu(); T("k(Z,Z) D2H", t0)
    t0 = time.perf_counter(); lam, vec = torch.linalg.eigh(kc / m); T("eigh on the host (M=1024)", t0)
    t0 = time.perf_counter(); lam_g, vec_g = torch.linalg.eigh(kzz / m); T("eigh on the GPU (torch/hipSOLVER)", t0)
    del kzz, kzx
    t0 = time.perf_counter(); basis = OrthonormalBasis(kernel, z, x, verbose=False, keep_gram=False); T("OrthonormalBasis total", t0)
    cost = GaussianCost(0.01, y, IdentityLinkFunction())
    t0 = time.perf_counter(); basis.prepare_gaussian(cost.y_device()); T("prepare_gaussian (B = A A^T, c = A y)", t0)
    del basis
