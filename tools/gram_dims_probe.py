import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.getcwd())
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd.kernel import ARDKernel
torch.manual_seed(0)
for (n1, n2, d) in [(1024, 100000, 4), (1024, 100000, 8), (1024, 100000, 9), (1024, 100000, 12), (1024, 100000, 16), (1024, 100000, 17), (1024, 100000, 32), (1024, 100000, 64)]:
    z = torch.randn(n1, d, dtype=torch.float64).cuda(); x = torch.randn(n2, d, dtype=torch.float64).cuda()
    k = ARDKernel(lengthscale=torch.rand(d, dtype=torch.float64) + 0.5, outputscale=1.7)
    out = k(z, x); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = k(z, x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"gram {n1}x{n2} d={d}: {ms:.3f} ms  {8.0 * n1 * n2 / ms / 1e9:.2f} TB/s", flush=True)
