"""Time pls_kernel_gram at the C2 shape (1024 x 1e5, D = 8) and check it against torch on the CPU."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd.kernel import ARDKernel
torch.manual_seed(0)
for (n1, n2, d) in [(1024, 100000, 8), (100000, 1024, 8), (1024, 1024, 8), (512, 50000, 1), (2048, 200000, 16)]:
    z = torch.randn(n1, d, dtype=torch.float64); x = torch.randn(n2, d, dtype=torch.float64)
    ls = torch.rand(d, dtype=torch.float64) + 0.5
    k = ARDKernel(lengthscale=ls, outputscale=1.7)
    zd, xd = z.cuda(), x.cuda()
    out = k(zd, xd); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = k(zd, xd)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    sub = slice(0, min(n2, 4096))
    ref = 1.7 * torch.exp(-0.5 * torch.cdist(z / ls, x[sub] / ls) ** 2)
    d2 = (((z / ls)[:, None, :] - (x[sub] / ls)[None, :256, :]) ** 2).sum(-1)
    ref2 = 1.7 * torch.exp(-0.5 * d2)
    err = (out[:, :256].cpu() - ref2).abs().max().item()
    print(f"gram {n1}x{n2} d={d}: {ms:.3f} ms  {8.0 * n1 * n2 / ms / 1e9:.2f} TB/s written   max abs err vs torch {err:.2e}", flush=True)
