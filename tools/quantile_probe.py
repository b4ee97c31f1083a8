"""Long-row quantiles: pls_row_quantiles (radix selection beyond 16384 samples per row) vs torch.quantile on the device."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from projected_langevin_sampling_amd import _ops
torch.manual_seed(0)
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rows, cols in [(2000, 16384), (2000, 32768), (2000, 65536), (1, 200000)]:
    s = torch.randn(rows, cols, dtype=torch.float64, device="cuda")
    qs = [0.05, 0.5, 0.95]
    q = torch.tensor(qs, dtype=torch.float64, device="cuda")
    a = _ops.row_quantiles(s, qs); b = torch.quantile(s, q, dim=1).T
    print(f"rows {rows} cols {cols}: libplship {timeit(lambda: _ops.row_quantiles(s, qs)):8.3f} ms   torch.quantile {timeit(lambda: torch.quantile(s, q, dim=1)):8.3f} ms   max diff {(a - b).abs().max().item():.1e}", flush=True)
