"""The kernel build k(Z, X) at configs[1]'s shape (M = 1024, N = 1e5, D = 8) ten times, for rocprofv3 --pmc passes:
which of its two candidate bounds -- the 8 N M bytes it writes, the ~40 fp64 vector instructions per entry -- it sits on.
    tools/profile_cmd.sh gram SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- python3 tools/gram_pmc.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
torch.manual_seed(0)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 8
z = torch.randn(1024, d, dtype=torch.float64).cuda(); x = torch.randn(100000, d, dtype=torch.float64).cuda()
k = P.ARDKernel(torch.rand(d, dtype=torch.float64) + 0.5, 1.7)
for _ in range(10):
    out = k(z, x)
torch.cuda.synchronize()
