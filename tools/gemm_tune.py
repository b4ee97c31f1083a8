"""Times pls_gemm_tn for the three shapes of the step (one subprocess per spec; the PLSHIP_GEMM_* knobs it can set
existed only while tuning -- see DESIGN.md "tuning log" -- and are ignored by the shipped library)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    import projected_langevin_sampling_amd as pkg
    L = pkg._lib; lib = L.load(); dev = "cuda"
    def run(I, J, K, reps):
        Lm = torch.randn(K, I, dtype=torch.float64, device=dev); Rm = torch.randn(K, J, dtype=torch.float64, device=dev)
        C = torch.empty(I, J, dtype=torch.float64, device=dev)
        f = lambda: L.check(lib.pls_gemm_tn(Lm.data_ptr(), I, Rm.data_ptr(), J, C.data_ptr(), J, I, J, K, 1.0, 0.0, L.stream_ptr()))
        f(); f(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            for _ in range(reps): f()
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / reps)
        ref = (Lm[:, :64].T @ Rm[:, :64]); err = (C[:64, :64] - ref).abs().max().item() / ref.abs().max().item()
        return best, 2.0 * I * J * K / best / 1e12, err
    out = []
    for (I, J, K, reps) in [(32768, 8192, 1024, 3), (1024, 8192, 32768, 3), (1024, 8192, 1024, 50)]:
        t, tf, err = run(I, J, K, reps)
        out.append(f"I={I} J={J} K={K}: {t*1e3:8.3f} ms {tf:6.2f} TF/s ({tf/78.6*100:5.1f}%) err {err:.1e}")
    print(f"cfg {os.environ.get('PLSHIP_GEMM_CFG','0')} exp {os.environ.get('PLSHIP_GEMM_EXP','0')} lds+{os.environ.get('PLSHIP_GEMM_LDS','0')}: " + " | ".join(out), flush=True)
else:
    for spec in (sys.argv[1:] or ["0", "1", "2", "4"]):
        cfg, exp, lds = (spec.split(":") + ["0", "0"])[:3]
        env = dict(os.environ, PLSHIP_GEMM_CFG=cfg, PLSHIP_GEMM_EXP=exp, PLSHIP_GEMM_LDS=lds)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
