"""Diagnostic: where does a forward-GEMM tile spend its cycles?  Uses tools/libplship_stamp.so (built with -DPLS_STAMP:
four s_memtime stamps per workgroup: start, after the prologue barrier, after the k-loop, after the epilogue).
Build:  (cd projected-langevin-sampling_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DPLS_STAMP -shared \
         -o ../../tools/libplship_stamp.so plship.hip)
Read the SHARES, not the lengths: the stamps serialise what the real kernel overlaps."""
import ctypes as C, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "libplship_stamp.so"))
lib.pls_gemm_tn.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_void_p]
lib.pls_debug_set_stamp_buffer.argtypes = [C.c_void_p]
dev = "cuda"
for (I, J, K) in [(100000, 8192, 1024), (1024, 8192, 14288)]:
    Lm = torch.randn(K, I, dtype=torch.float64, device=dev); Rm = torch.randn(K, J, dtype=torch.float64, device=dev)
    Cm = torch.empty(I, J, dtype=torch.float64, device=dev)
    ntiles = ((I + 127) // 128) * ((J + 127) // 128)
    stamps = torch.zeros(ntiles * 4, dtype=torch.int64, device=dev)
    lib.pls_debug_set_stamp_buffer(None)
    lib.pls_gemm_tn(Lm.data_ptr(), I, Rm.data_ptr(), J, Cm.data_ptr(), J, I, J, K, 1.0, 0.0, None)
    torch.cuda.synchronize()
    lib.pls_debug_set_stamp_buffer(stamps.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.pls_gemm_tn(Lm.data_ptr(), I, Rm.data_ptr(), J, Cm.data_ptr(), J, I, J, K, 1.0, 0.0, None); e1.record()
    torch.cuda.synchronize()
    s = stamps.reshape(ntiles, 4).cpu().double()
    pro, loop, epi = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
    tot = s[:, 3] - s[:, 0]
    span = (s[:, 3].max() - s[:, 0].min()).item()
    print(f"I={I} J={J} K={K}: kernel {e0.elapsed_time(e1):.2f} ms, {ntiles} tiles, span {span:.3e} ticks "
          f"({span / (e0.elapsed_time(e1) * 1e-3) / 1e9:.2f} G ticks/s)")
    for name, v in (("prologue", pro), ("k-loop", loop), ("epilogue", epi), ("total", tot)):
        print(f"   {name:9s} median {v.median().item():10.0f}  p10 {v.quantile(0.1).item():10.0f}  p90 {v.quantile(0.9).item():10.0f}  share {v.sum().item() / tot.sum().item():.3f}")
    order = torch.argsort(s[:, 0])
    print("   k-loop median, first 512 started:", loop[order[:512]].median().item(), " later:", loop[order[512:]].median().item() if ntiles > 512 else None)
