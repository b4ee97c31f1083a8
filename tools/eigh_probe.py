import torch, time
torch.manual_seed(0)
for m in (1024, 2048, 4096):
    x = torch.randn(m, 8, dtype=torch.float64)
    d = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    g = torch.exp(-0.5 * d) / m
    t0 = time.perf_counter(); lc, vc = torch.linalg.eigh(g); t1 = time.perf_counter()
    gd = g.cuda(); torch.cuda.synchronize()
    for rep in range(2):
        t2 = time.perf_counter(); ld, vd = torch.linalg.eigh(gd); torch.cuda.synchronize(); t3 = time.perf_counter()
        print(f"M={m}: host eigh {t1 - t0:.3f} s ({torch.get_num_threads()} threads), device eigh (call {rep}) {t3 - t2:.3f} s, "
              f"max |lambda diff| {(ld.cpu() - lc).abs().max().item():.2e}, residual {((gd @ vd) - vd * ld).abs().max().item():.2e}", flush=True)
