"""Diagnostic: where does a forward-GEMM tile spend its cycles?  Uses tools/libplship_stamp.so (built with -DPLS_STAMP:
four s_memtime stamps per workgroup: start, after the prologue barrier, after the k-loop, after the epilogue, plus
HW_ID / XCC_ID so the per-CU timeline (residency, launch gaps, tail) can be rebuilt).
Build:  (cd projected-langevin-sampling_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DPLS_STAMP -shared \
         -o ../../tools/libplship_stamp.so plship.hip gemm_cost.hip gemm_cost_value.hip small_rank_drift.hip small_rank_value.hip small_rank_drift_value.hip chol.hip)
Read the SHARES, not the lengths: the stamps serialise what the real kernel overlaps."""
import ctypes as C, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "libplship_stamp.so"))
lib.pls_gemm_tn.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_void_p]
lib.pls_debug_set_stamp_buffer.argtypes = [C.c_void_p]
dev = "cuda"
shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(100000, 8192, 1024), (1024, 8192, 14288)]
for (I, J, K) in shapes:
    Lm = torch.randn(K, I, dtype=torch.float64, device=dev); Rm = torch.randn(K, J, dtype=torch.float64, device=dev)
    Cm = torch.empty(I, J, dtype=torch.float64, device=dev)
    ntiles = ((I + 127) // 128) * ((J + 127) // 128)
    stamps = torch.zeros(ntiles * 6, dtype=torch.int64, device=dev)
    lib.pls_debug_set_stamp_buffer(None)
    lib.pls_gemm_tn(Lm.data_ptr(), I, Rm.data_ptr(), J, Cm.data_ptr(), J, I, J, K, 1.0, 0.0, None)
    torch.cuda.synchronize()
    lib.pls_debug_set_stamp_buffer(stamps.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.pls_gemm_tn(Lm.data_ptr(), I, Rm.data_ptr(), J, Cm.data_ptr(), J, I, J, K, 1.0, 0.0, None); e1.record()
    torch.cuda.synchronize()
    raw = stamps.reshape(ntiles, 6).cpu()
    s = raw[:, :4].double()
    pro, loop, epi = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
    tot = s[:, 3] - s[:, 0]
    span = (s[:, 3].max() - s[:, 0].min()).item()
    print(f"I={I} J={J} K={K}: kernel {e0.elapsed_time(e1):.2f} ms, {ntiles} tiles, span {span:.3e} ticks "
          f"({span / (e0.elapsed_time(e1) * 1e-3) / 1e9:.2f} G ticks/s)")
    for name, v in (("prologue", pro), ("k-loop", loop), ("epilogue", epi), ("total", tot)):
        print(f"   {name:9s} median {v.median().item():10.0f}  p10 {v.quantile(0.1).item():10.0f}  p90 {v.quantile(0.9).item():10.0f}  share {v.sum().item() / tot.sum().item():.3f}")
    order = torch.argsort(s[:, 0])
    print("   k-loop median, first 512 started:", loop[order[:512]].median().item(), " later:", loop[order[512:]].median().item() if ntiles > 512 else None)
    # per-CU timeline
    import numpy as np
    hw, xcc = raw[:, 4].numpy(), raw[:, 5].numpy() & 0xF
    cu = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
    t0, t3 = s[:, 0].numpy(), s[:, 3].numpy()
    base = t0.min()
    res1 = res2 = res0 = 0.0
    ends, counts, gaps = [], [], []
    for c in np.unique(cu):
        m = cu == c
        ev = sorted([(a, 1) for a in t0[m]] + [(b, -1) for b in t3[m]])
        lvl, prev = 0, base
        for t, d in ev:
            dt = t - prev
            if lvl == 0: res0 += dt
            elif lvl == 1: res1 += dt
            else: res2 += dt
            lvl += d; prev = t
        ends.append(t3[m].max() - base); counts.append(m.sum())
        # launch gap: for every end, the next start on this CU
        st = np.sort(t0[m]); en = np.sort(t3[m])
        idx = np.searchsorted(st, en)
        ok = idx < len(st)
        gaps.extend((st[idx[ok]] - en[ok]).tolist())
    ends, counts, gaps = np.array(ends), np.array(counts), np.array(gaps)
    ncu = len(ends)
    print(f"   CUs seen {ncu}; tiles/CU min {counts.min()} mean {counts.mean():.1f} max {counts.max()}")
    print(f"   CU finish time: min {ends.min():.3e} median {np.median(ends):.3e} max {ends.max():.3e} (span {span:.3e})")
    tot_t = ncu * ends.max()
    print(f"   CU-time shares up to the last finish: 2 WGs {res2 / tot_t:.3f}  1 WG {res1 / tot_t:.3f}  0 WG before own end {res0 / tot_t:.3f}  idle after own end {1 - (res0 + res1 + res2) / tot_t:.3f}")
    if len(gaps) == 0: continue
    print(f"   end->next start gap on the same CU: median {np.median(gaps):.0f} p90 {np.quantile(gaps, 0.9):.0f} mean {gaps.mean():.0f} ticks")
