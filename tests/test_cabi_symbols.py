"""CPU: the C-ABI library loads and exports every symbol include/plship.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "plship.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(pls_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_path():
    names = declared_functions()
    for must in ("pls_onb_step", "pls_ipb_step", "pls_gemm_tn", "pls_kernel_gram", "pls_cost_derivative", "pls_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    import projected_langevin_sampling_amd as pkg

    lib = pkg._lib.load()
    raw = ctypes.CDLL(pkg._lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(raw, name), f"{name} declared in include/plship.h but not exported"
    assert set(pkg._lib.SIGNATURES) == set(declared_functions()), "ctypes table and header disagree"
    assert lib.pls_abi_version() == pkg._lib.ABI_VERSION
    assert lib.pls_last_error() is not None


def test_options_validate_without_touching_the_gpu():
    """pls_set_option / pls_get_option are host-only: range checks and unknown options behave as documented."""
    import projected_langevin_sampling_amd as pkg

    L = pkg._lib
    lib = L.load()
    assert lib.pls_get_option(L.OPT_SMALL_RANK_MAX) == 128
    assert lib.pls_set_option(L.OPT_SMALL_RANK_MAX, 129) != 0 and b"outside 0..128" in lib.pls_last_error()
    assert lib.pls_set_option(L.OPT_SMALL_RANK_MAX, -1) != 0
    assert lib.pls_set_option(99, 1) != 0 and b"unknown option" in lib.pls_last_error()
    assert lib.pls_get_option(99) == -1
    assert lib.pls_set_option(L.OPT_SMALL_RANK_MAX, 64) == 0 and lib.pls_get_option(L.OPT_SMALL_RANK_MAX) == 64
    assert lib.pls_set_option(L.OPT_SMALL_RANK_MAX, 128) == 0
    assert lib.pls_get_option(L.OPT_IPB_EXPLICIT_INVERSE) == 0
    assert lib.pls_set_option(L.OPT_IPB_EXPLICIT_INVERSE, 2) != 0
    assert lib.pls_set_option(L.OPT_IPB_EXPLICIT_INVERSE, 1) == 0 and lib.pls_get_option(L.OPT_IPB_EXPLICIT_INVERSE) == 1
    assert lib.pls_set_option(L.OPT_IPB_EXPLICIT_INVERSE, 0) == 0
    assert lib.pls_get_option(L.OPT_KSPLIT_MODE) == 1 and lib.pls_get_option(L.OPT_KSPLIT_MAX_TILES) == 256
    assert lib.pls_set_option(L.OPT_KSPLIT_MODE, 4) != 0 and lib.pls_set_option(L.OPT_KSPLIT_MAX_TILES, -1) != 0
    assert lib.pls_set_option(L.OPT_KSPLIT_MODE, 0) == 0 and lib.pls_get_option(L.OPT_KSPLIT_MODE) == 0
    assert lib.pls_set_option(L.OPT_KSPLIT_MODE, 1) == 0
    assert lib.pls_get_option(6) == -1 and lib.pls_set_option(7, 200) != 0  # (ABI 3's wave-pair kernel options: gone)
    assert lib.pls_get_option(L.OPT_ROW_BLOCKS) == 1 and lib.pls_set_option(L.OPT_ROW_BLOCKS, 2) != 0
    assert lib.pls_set_option(L.OPT_ROW_BLOCKS, 0) == 0 and lib.pls_set_option(L.OPT_ROW_BLOCKS, 1) == 0
    assert lib.pls_get_option(L.OPT_SOLVE_MODE) == 1 and lib.pls_set_option(L.OPT_SOLVE_MODE, 2) != 0
    assert lib.pls_set_option(L.OPT_SOLVE_MODE, 0) == 0 and lib.pls_set_option(L.OPT_SOLVE_MODE, 1) == 0
    assert lib.pls_get_option(L.OPT_TRI_BALANCE) == 1 and lib.pls_set_option(L.OPT_TRI_BALANCE, 2) != 0
    assert lib.pls_set_option(L.OPT_TRI_BALANCE, 0) == 0 and lib.pls_set_option(L.OPT_TRI_BALANCE, 1) == 0
    assert lib.pls_get_option(L.OPT_ENERGY_FUSED_FINISH) == 1 and lib.pls_set_option(L.OPT_ENERGY_FUSED_FINISH, 2) != 0
    assert lib.pls_get_option(L.OPT_KG_NOISE_PREGEN) == 1 and lib.pls_set_option(L.OPT_KG_NOISE_PREGEN, 2) != 0
    assert lib.pls_set_option(L.OPT_KG_NOISE_PREGEN, 0) == 0 and lib.pls_set_option(L.OPT_KG_NOISE_PREGEN, 1) == 0
    assert lib.pls_get_option(L.OPT_IPB_STEP_OPERATOR) == 1 and lib.pls_set_option(L.OPT_IPB_STEP_OPERATOR, 3) != 0
    assert lib.pls_set_option(L.OPT_IPB_STEP_OPERATOR, 0) == 0 and lib.pls_set_option(L.OPT_IPB_STEP_OPERATOR, 1) == 0
    assert lib.pls_get_option(L.OPT_SMALL_RANK_STEP) == 1 and lib.pls_set_option(L.OPT_SMALL_RANK_STEP, 3) != 0
    assert lib.pls_set_option(L.OPT_SMALL_RANK_STEP, 2) == 0 and lib.pls_set_option(L.OPT_SMALL_RANK_STEP, 1) == 0
    assert lib.pls_get_option(L.OPT_IPB_PREP) == 1 and lib.pls_set_option(L.OPT_IPB_PREP, 2) != 0
    assert lib.pls_set_option(L.OPT_IPB_PREP, 0) == 0 and lib.pls_set_option(L.OPT_IPB_PREP, 1) == 0
    assert lib.pls_step_sync_words(512) == 32 + 2 and lib.pls_step_sync_words(17) == 2 + 1 and lib.pls_step_sync_words(0) == 0
    assert lib.pls_tri_scratch_bytes(1024, 1024) == 16384 + 8 * 16 * 2 * 4096 * 8 and lib.pls_tri_scratch_bytes(0, 5) == 0


def test_struct_layouts_match_the_header():
    """sizeof / field order of the descriptor structs (plain C layout on x86-64)."""
    import projected_langevin_sampling_amd as pkg

    L = pkg._lib
    assert ctypes.sizeof(L.CostDesc) == 4 * 4 + 4 * 8 + 8
    assert ctypes.sizeof(L.NoiseDesc) == 8 + 8 + 8 + 8 + 8 + 8 + 8
    assert ctypes.sizeof(L.OnbDesc) == 10 * 8
    assert ctypes.sizeof(L.IpbDesc) == 31 * 8  # (ABI 4: + tri_scratch, tri_scratch_bytes, Pt, ldpt; ABI 5: + Awa, ldawa)
    assert ctypes.sizeof(L.CholDesc) == 15 * 8
    assert ctypes.sizeof(L.BlockDesc) == 11 * 8  # (ABI 5: + step_sync, energy_sums16)
    assert L.CostDesc.p.offset == 16 and L.CostDesc.jitter.offset == 48


def test_struct_layouts_match_what_a_c_compiler_makes_of_the_header(tmp_path):
    """sizeof and the offset of the last field of every descriptor, as gcc lays the header's structs out."""
    import shutil
    import subprocess

    import projected_langevin_sampling_amd as pkg

    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    L = pkg._lib
    structs = {"pls_cost_desc": (L.CostDesc, "jitter"), "pls_noise_desc": (L.NoiseDesc, "step_base"),
               "pls_onb_desc": (L.OnbDesc, "c"), "pls_ipb_desc": (L.IpbDesc, "q_inv_noise"),
               "pls_chol_desc": (L.CholDesc, "ldlinvt"), "pls_block_desc": (L.BlockDesc, "energy_sums16")}
    src = tmp_path / "layout.c"
    body = "".join(f'  printf("{n} %zu %zu\\n", sizeof({n}), offsetof({n}, {last}));\n' for n, (_, last) in structs.items())
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "plship.h"\nint main(void) {\n' + body + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    for k in range(0, len(out), 3):
        cls, last = structs[out[k]]
        assert ctypes.sizeof(cls) == int(out[k + 1]), out[k]
        assert getattr(cls, last).offset == int(out[k + 2]), out[k]


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import projected_langevin_sampling_amd as pkg

    L = pkg._lib
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "libplship.so"))
    with pytest.raises(L.PlsHipError, match="no CPU fallback|not found"):
        L.load()


def test_product_package_never_imports_the_oracle():
    pkg_dir = os.path.join(ROOT, "projected-langevin-sampling_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "/root/reference" not in text, f


def test_route_options_belong_to_the_calling_thread():
    """pls_set_option: a value of the CALLING thread (no global state, thread-compatible per context) -- another
    thread starts from the defaults, and what it sets stays with it."""
    import threading

    import projected_langevin_sampling_amd as pkg

    L = pkg._lib
    lib = L.load()
    assert lib.pls_get_option(L.OPT_SMALL_RANK_STEP) == 1 and lib.pls_get_option(L.OPT_TRI_BALANCE) == 1
    assert lib.pls_set_option(L.OPT_SMALL_RANK_STEP, 2) == 0 and lib.pls_set_option(L.OPT_TRI_BALANCE, 0) == 0
    seen = {}

    def other():
        seen["start"] = (lib.pls_get_option(L.OPT_SMALL_RANK_STEP), lib.pls_get_option(L.OPT_TRI_BALANCE))
        lib.pls_set_option(L.OPT_SMALL_RANK_STEP, 0)
        lib.pls_set_option(L.OPT_SMALL_RANK_MAX, 64)
        seen["end"] = (lib.pls_get_option(L.OPT_SMALL_RANK_STEP), lib.pls_get_option(L.OPT_SMALL_RANK_MAX))

    t = threading.Thread(target=other)
    t.start()
    t.join()
    try:
        assert seen == {"start": (1, 1), "end": (0, 64)}
        assert lib.pls_get_option(L.OPT_SMALL_RANK_STEP) == 2 and lib.pls_get_option(L.OPT_TRI_BALANCE) == 0
        assert lib.pls_get_option(L.OPT_SMALL_RANK_MAX) == 128
    finally:
        lib.pls_set_option(L.OPT_SMALL_RANK_STEP, 1)
        lib.pls_set_option(L.OPT_TRI_BALANCE, 1)
