"""GPU: libplship driven from plain C++ (examples/cabi_step.cpp) -- no Python, no torch on the data path: the C ABI of
include/plship.h is the whole boundary.  The program checks one fused step (Gaussian and Poisson, energy by-product
included) against its own scalar loop and the error reporting across the boundary; here it is compiled and run."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cabi_from_plain_cpp(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    assert os.path.exists(hipcc), "hipcc is part of the image"
    libdir = os.path.join(ROOT, "projected-langevin-sampling_amd")
    assert os.path.exists(os.path.join(libdir, "libplship.so")), "build the library first (__graft_entry__.build())"
    exe = str(tmp_path / "cabi_step")
    subprocess.run([hipcc, "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "cabi_step.cpp"),
                    "-L", libdir, "-lplship", f"-Wl,-rpath,{libdir}", "-o", exe], check=True, timeout=300)
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "cabi_step OK" in run.stdout and "workspace error reported" in run.stdout
    assert "cholesky: info 0" in run.stdout and "not positive definite: info = 6" in run.stdout
