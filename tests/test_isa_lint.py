"""CPU: the gfx950 ISA of libplship must not write into the C operand of an fp64 MFMA that is still in flight.

`v_mfma_f64_16x16x4_f64 vD, vA, vB, vC` with vD != vC leaves vC dead for the register allocator, which then parks a copy
or a fragment load there -- while the instruction is still streaming C in.  Found in round 3 (a k-tail path of
csrc/gemm_tn_f64_kg.h accumulated garbage in one 16 x 16 block; ROCm 7.2's hazard recogniser does not pad the DGEMM
opcodes); the kernels are now written so that accumulators stay in place, and this test keeps it that way.  hipcc
cross-compiles without a GPU (`make -C csrc lint` emits the device ISA of every translation unit, ~1-2 minutes cold)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "projected-langevin-sampling_amd", "csrc")


def test_no_write_into_a_live_mfma_c_operand():
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    r = subprocess.run(["make", "-C", CSRC, "-j6", "lint"], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "0 suspicious write(s)" in r.stdout


def test_the_lint_sees_the_hazard(tmp_path):
    """the pattern that corrupted a sum in round 3, and its in-place twin"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import mfma_srcc_lint as lint

    bad = tmp_path / "bad.s"
    bad.write_text("\tv_mfma_f64_16x16x4_f64 v[42:49], v[52:53], v[50:51], v[82:89]\n\tv_mov_b64_e32 v[86:87], v[56:57]\n"
                   "\tv_mfma_f64_16x16x4_f64 v[58:65], v[86:87], v[82:83], v[74:81]\n\tds_read_b64 v[74:75], v116 offset:36864\n")
    good = tmp_path / "good.s"
    good.write_text("\tv_mfma_f64_16x16x4_f64 v[18:25], v[52:53], v[54:55], v[18:25]\n\tv_mov_b64_e32 v[86:87], v[56:57]\n"
                    "\tds_read_b64 v[74:75], v116 offset:36864\n\tv_mfma_f64_16x16x4_f64 v[10:17], v[52:53], v[56:57], v[10:17]\n")
    assert len(lint.lint(str(bad))) == 2 and lint.lint(str(good)) == []
