"""ctypes binding of libplship.so (include/plship.h).  The product path has no fallback: if the
HIP library is missing or a call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# the in-tree build; PLSHIP_LIBRARY points at another build of the same ABI (A/B runs, tools/ab_smallrank.py)
LIB_PATH = os.environ.get("PLSHIP_LIBRARY") or os.path.join(_HERE, "libplship.so")

# enums (include/plship.h)
KERNEL_RBF_ARD, KERNEL_LINEAR = 0, 1
COST_GAUSSIAN, COST_POISSON, COST_BERNOULLI, COST_STUDENT_T, COST_MULTIMODAL = range(5)
LINK_IDENTITY, LINK_SQUARE, LINK_SIGMOID, LINK_PROBIT = range(4)
DERIV_REFERENCE, DERIV_AUTOGRAD = 0, 1
NOISE_NONE, NOISE_INJECTED, NOISE_PHILOX = 0, 1, 2
OUT_DELTA, OUT_NEW_STATE = 0, 1


class PlsHipError(RuntimeError):
    pass


class CostDesc(C.Structure):
    _fields_ = [
        ("cost", C.c_int32),
        ("link", C.c_int32),
        ("deriv_mode", C.c_int32),
        ("reserved", C.c_int32),
        ("p", C.c_double * 4),
        ("jitter", C.c_double),
    ]


class NoiseDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("reserved", C.c_int32),
        ("xi", C.c_void_p),
        ("ldxi", C.c_int64),
        ("seed", C.c_uint64),
        ("step", C.c_uint64),
        ("j_offset", C.c_int64),
        ("step_base", C.c_void_p),
    ]


class OnbDesc(C.Structure):
    _fields_ = [
        ("mk", C.c_int64),
        ("n", C.c_int64),
        ("A", C.c_void_p),
        ("lda", C.c_int64),
        ("At", C.c_void_p),
        ("ldat", C.c_int64),
        ("lam", C.c_void_p),
        ("B", C.c_void_p),
        ("ldb", C.c_int64),
        ("c", C.c_void_p),
    ]


class IpbDesc(C.Structure):
    _fields_ = [
        ("m", C.c_int64),
        ("n", C.c_int64),
        ("Kzx", C.c_void_p),
        ("ldkzx", C.c_int64),
        ("Kxz", C.c_void_p),
        ("ldkxz", C.c_int64),
        ("W", C.c_void_p),
        ("ldw", C.c_int64),
        ("LcT", C.c_void_p),
        ("ldlct", C.c_int64),
        ("B", C.c_void_p),
        ("ldb", C.c_int64),
        ("c", C.c_void_p),
        ("Sf", C.c_void_p),
        ("ldsf", C.c_int64),
        ("Sb", C.c_void_p),
        ("ldsb", C.c_int64),
        ("Linv", C.c_void_p),
        ("ldlinv", C.c_int64),
        ("LinvT", C.c_void_p),
        ("ldlinvt", C.c_int64),
        ("Q", C.c_void_p),
        ("ldq", C.c_int64),
        ("ct", C.c_void_p),
        ("q_inv_noise", C.c_double),
        ("tri_scratch", C.c_void_p),
        ("tri_scratch_bytes", C.c_size_t),
        ("Pt", C.c_void_p),
        ("ldpt", C.c_int64),
        ("Awa", C.c_void_p),
        ("ldawa", C.c_int64),
    ]


class CholDesc(C.Structure):
    _fields_ = [
        ("m", C.c_int64),
        ("Lc", C.c_void_p),
        ("ldlc", C.c_int64),
        ("LcT", C.c_void_p),
        ("ldlct", C.c_int64),
        ("Sf", C.c_void_p),
        ("ldsf", C.c_int64),
        ("Sb", C.c_void_p),
        ("ldsb", C.c_int64),
        ("Linv", C.c_void_p),
        ("ldlinv", C.c_int64),
        ("LinvT", C.c_void_p),
        ("ldlinvt", C.c_int64),
        ("tri_scratch", C.c_void_p),
        ("tri_scratch_bytes", C.c_size_t),
    ]


class BlockDesc(C.Structure):
    _fields_ = [("block_cols", C.c_int64), ("eta", C.c_void_p), ("energy_sums", C.c_void_p), ("energy_sync", C.c_void_p),
                ("energy_partials", C.c_void_p), ("energy_partials_prev", C.c_void_p), ("energy_prev", C.c_void_p),
                ("energy_sums_prev", C.c_void_p), ("energy_flush", C.c_int32), ("reserved", C.c_int32), ("step_sync", C.c_void_p),
                ("energy_sums16", C.c_void_p)]


_P, _I64, _I32, _U64, _D, _SZ = C.c_void_p, C.c_int64, C.c_int32, C.c_uint64, C.c_double, C.c_size_t
_CD, _ND, _OD, _ID = C.POINTER(CostDesc), C.POINTER(NoiseDesc), C.POINTER(OnbDesc), C.POINTER(IpbDesc)
_CHD, _BD = C.POINTER(CholDesc), C.POINTER(BlockDesc)

# name -> (restype, argtypes); every symbol include/plship.h declares
SIGNATURES = {
    "pls_last_error": (C.c_char_p, []),
    "pls_abi_version": (C.c_int, []),
    "pls_set_option": (C.c_int, [_I32, _I64]),
    "pls_get_option": (C.c_int64, [_I32]),
    "pls_debug_math": (C.c_int, [_I32, _P, _P, _I64, _P]),
    "pls_timeline_begin": (C.c_int, [_I32]),
    "pls_timeline_end": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_int32), _I32, C.POINTER(C.c_int32)]),
    "pls_kernel_gram": (C.c_int, [_I32, _P, _I64, _P, _I64, _I64, _P, _D, _P, _I64, _P]),
    "pls_gemm_tn": (C.c_int, [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _D, _D, _P]),
    "pls_cost_derivative": (C.c_int, [_CD, _P, _I64, _P, _I64, _I64, _P, _I64, _P]),
    "pls_cost_value_workspace_bytes": (_SZ, [_I64, _I64]),
    "pls_cost_value": (C.c_int, [_CD, _P, _I64, _P, _I64, _I64, _P, _P, _SZ, _P]),
    "pls_link_transform": (C.c_int, [_I32, _D, _P, _I64, _I64, _I64, _P, _P, _I64, _P]),
    "pls_row_power_sums": (C.c_int, [_P, _I64, _I64, _I64, _P, _I32, _P, _P]),
    "pls_row_quantiles": (C.c_int, [_P, _I64, _I64, _I64, _P, _I32, _P, _I64, _P]),
    "pls_counter_add": (C.c_int, [_P, _U64, _P]),
    "pls_normal_fill": (C.c_int, [_P, _I64, _I64, _I64, _U64, _U64, _I64, _P]),
    "pls_step_sync_words": (_SZ, [_I64]),
    "pls_sums16": (C.c_int, [_P, _I64, _P, _P]),
    "pls_onb_build_projection": (C.c_int, [_P, _I64, _P, _I64, _I64, _I64, _I64, _P, _I64, _P, _I64, _P]),
    "pls_onb_build_gaussian": (C.c_int, [_OD, _P, _P, _I64, _P, _P]),
    "pls_onb_forward": (C.c_int, [_OD, _P, _I64, _I64, _P, _I64, _P]),
    "pls_onb_particle_update": (C.c_int, [_OD, _P, _I64, _P, _I64, _I64, _D, _ND, _P, _I64, _P]),
    "pls_onb_step_workspace_bytes": (_SZ, [_OD, _I64, _I64]),
    "pls_onb_step": (C.c_int, [_OD, _CD, _P, _P, _I64, _I64, _D, _ND, _P, _I64, _I32, _I32, _P, _P, _SZ, _P]),
    "pls_onb_energy_workspace_bytes": (_SZ, [_OD, _I64, _I64]),
    "pls_onb_energy": (C.c_int, [_OD, _CD, _P, _P, _I64, _I64, _P, _I32, _P, _SZ, _P]),
    "pls_onb_prior_energy": (C.c_int, [_OD, _P, _I64, _I64, _P, _P, _P]),
    "pls_ipb_prior_energy": (C.c_int, [_ID, _P, _I64, _I64, _P, _P, _P, _SZ, _P]),
    "pls_select_inducing_workspace_bytes": (_SZ, [_I64, _I64]),
    "pls_select_inducing_conditional_variance": (C.c_int, [_I32, _P, _I64, _I64, _P, _D, _I64, _D, _D, _P, _P, _P, _SZ, _P]),
    "pls_ipb_forward": (C.c_int, [_ID, _P, _I64, _I64, _P, _I64, _P, _SZ, _P]),
    "pls_ipb_particle_update": (C.c_int, [_ID, _P, _I64, _P, _I64, _I64, _D, _ND, _P, _I64, _P, _SZ, _P]),
    "pls_ipb_step_workspace_bytes": (_SZ, [_ID, _I64, _I64]),
    "pls_ipb_step": (C.c_int, [_ID, _CD, _P, _P, _I64, _I64, _D, _ND, _P, _I64, _I32, _I32, _P, _P, _SZ, _P]),
    "pls_ipb_build_gaussian": (C.c_int, [_ID, _P, _P, _I64, _P, _P]),
    "pls_ipb_energy_workspace_bytes": (_SZ, [_ID, _I64, _I64]),
    "pls_ipb_energy": (C.c_int, [_ID, _CD, _P, _P, _I64, _I64, _P, _I32, _P, _SZ, _P]),
    "pls_block_means": (C.c_int, [_P, _I64, _I64, _P, _P]),
    "pls_chunk_sums": (C.c_int, [_P, _I64, _P, _P]),
    "pls_chol_factor": (C.c_int, [_P, _I64, _I64, _D, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _P]),
    "pls_chol_build_operators": (C.c_int, [_P, _I64, _P, _I64, _I64, _P, _I64, _P, _I64, _P]),
    "pls_chol_solve": (C.c_int, [_CHD, _P, _I64, _I64, _P, _I64, _P]),
    "pls_tri_multiply": (C.c_int, [_P, _I64, _I64, _P, _I64, _I64, _P, _I64, _P]),
    "pls_chol_build_inverse": (C.c_int, [_CHD, _P, _I64, _P, _I64, _P]),
    "pls_chol_forward_solve": (C.c_int, [_CHD, _P, _I64, _I64, _P, _I64, _P]),
    "pls_chol_solve_workspace_bytes": (_SZ, [_I64, _I64]),
    "pls_tri_scratch_bytes": (_SZ, [_I64, _I64]),
    "pls_energy_partials_bytes": (_SZ, [_I64, _I64]),
    "pls_chol_solve_ws": (C.c_int, [_CHD, _P, _I64, _I64, _P, _I64, _P, _SZ, _P]),
    "pls_ipb_build_whitened_workspace_bytes": (_SZ, [_I64]),
    "pls_ipb_build_whitened": (C.c_int, [_ID, _D, _P, _I64, _P, _P, _SZ, _P]),
    "pls_ipb_build_step_operator": (C.c_int, [_ID, _P, _I64, _P]),
    "pls_ipb_whiten": (C.c_int, [_ID, _P, _I64, _I64, _P, _I64, _P]),
    "pls_ipb_unwhiten": (C.c_int, [_ID, _P, _I64, _I64, _P, _I64, _P]),
    "pls_ipb_whitened_workspace_bytes": (_SZ, [_ID, _I64]),
    "pls_ipb_whitened_step": (C.c_int, [_ID, _CD, _P, _I64, _I64, _D, _ND, _P, _I64, _I32, _P, _P, _SZ, _P]),
    "pls_ipb_whitened_step_blocks": (C.c_int, [_ID, _CD, _P, _I64, _I64, _BD, _ND, _P, _I64, _I32, _P, _P, _SZ, _P]),
    "pls_ipb_whitened_energy": (C.c_int, [_ID, _CD, _P, _I64, _I64, _P, _P, _SZ, _P]),
    "pls_ipb_build_whitened_operand": (C.c_int, [_ID, _P, _I64, _P]),
    "pls_ipb_whitened_generic_applies": (C.c_int, [_ID, _P, _I64]),
    "pls_ipb_whitened_generic_workspace_bytes": (_SZ, [_ID, _I64]),
    "pls_ipb_whitened_generic_step": (C.c_int, [_ID, _CD, _P, _P, _I64, _I64, _D, _BD, _ND, _P, _I64, _I32, _P, _P, _SZ, _P]),
    "pls_onb_step_blocks": (C.c_int, [_OD, _CD, _P, _P, _I64, _I64, _BD, _ND, _P, _I64, _I32, _I32, _P, _P, _SZ, _P]),
    "pls_ipb_step_blocks": (C.c_int, [_ID, _CD, _P, _P, _I64, _I64, _BD, _ND, _P, _I64, _I32, _I32, _P, _P, _SZ, _P]),
}

ABI_VERSION = 5

_lib = None


def load() -> C.CDLL:
    """Load libplship.so (built in-tree by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PlsHipError(
            f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(or __graft_entry__.build()).  There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.pls_abi_version() != ABI_VERSION:
        raise PlsHipError(f"libplship ABI {lib.pls_abi_version()} != {ABI_VERSION}: rebuild it (make -C csrc)")
    _lib = lib
    return lib


OPT_SMALL_RANK_MAX = 1
OPT_IPB_EXPLICIT_INVERSE = 2
OPT_KSPLIT_MODE = 3
OPT_KSPLIT_MAX_TILES = 4
OPT_SOLVE_MODE = 5
OPT_ROW_BLOCKS = 8
OPT_TRI_BALANCE = 9
OPT_IPB_STEP_OPERATOR = 10
OPT_ENERGY_FUSED_FINISH = 11
OPT_KG_NOISE_PREGEN = 12
OPT_SMALL_RANK_STEP = 13
OPT_IPB_PREP = 14
TAG_NAMES = {1: "gemm_store", 2: "gemm_cost_deriv", 3: "gemm_cost_value", 4: "gemm_langevin_gaussian",
             5: "langevin_update", 6: "kernel_gram", 7: "other", 8: "small_rank_drift", 9: "small_rank_value",
             10: "tri_solve", 11: "small_rank_step", 12: "ipb_prep"}


class Timeline:
    """with Timeline(capacity) as tl: ...launches...; tl.records -> [(tag_name, ms), ...] measured with HIP events
    recorded on the launch stream around every libplship kernel launch of this thread."""

    def __init__(self, capacity: int = 4096):
        self.capacity = capacity
        self.records: list[tuple[str, float]] = []
        self.launches = 0

    def __enter__(self):
        check(load().pls_timeline_begin(self.capacity), "pls_timeline_begin")
        return self

    def __exit__(self, *exc):
        ms = (C.c_float * self.capacity)()
        tags = (C.c_int32 * self.capacity)()
        count = C.c_int32(0)
        check(load().pls_timeline_end(ms, tags, self.capacity, C.byref(count)), "pls_timeline_end")
        self.launches = count.value
        n = min(count.value, self.capacity)
        self.records = [(TAG_NAMES.get(tags[i], "other"), float(ms[i])) for i in range(n)]
        return False

    def summary(self) -> dict:
        out: dict[str, dict] = {}
        for name, t in self.records:
            d = out.setdefault(name, {"launches": 0, "total_ms": 0.0})
            d["launches"] += 1
            d["total_ms"] += t
        for d in out.values():
            d["avg_ms"] = d["total_ms"] / d["launches"]
        return out


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().pls_last_error().decode()
        raise PlsHipError(f"{what or 'libplship'} failed (status {rc}): {msg}")


#: floating dtypes a caller may hand in where the library only READS (particles, cost derivatives, injected noise ...): the
#: reference's bases compute in whatever dtype the caller uses (basis/base.py:52-63, :99-102 of the reference; its tests run
#: in float32); libplship computes in float64, so such inputs are promoted on entry -- like x / z / y in kernel._dev -- and
#: results are float64
PROMOTED_DTYPES = (torch.float32, torch.float16, torch.bfloat16)


def require_gpu_tensor(t: torch.Tensor, name: str, promote: bool = False) -> torch.Tensor:
    """``t`` as a float64 device tensor.  ``promote``: a float32 / float16 / bfloat16 device tensor the library only reads
    comes back as a float64 COPY; buffers the library writes (out=, loop state, workspaces) must be float64 already."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.device.type != "cuda":
        raise PlsHipError(
            f"{name} lives on {t.device}: the projected-Langevin hot path runs on the MI355X only "
            "(no CPU fallback); move it with .cuda()"
        )
    if t.dtype != torch.float64:
        if promote and t.dtype in PROMOTED_DTYPES:
            return t.detach().to(torch.float64)
        raise TypeError(f"{name} must be float64 on the device, got {t.dtype}: convert it with {name}.double() "
                        "(libplship computes in float64; read-only inputs in float32 are promoted automatically)")
    return t


def ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def ld(t: torch.Tensor) -> int:
    """Leading dimension (elements) of a row-major 2-D tensor whose rows are contiguous."""
    assert t.dim() == 2 and t.stride(1) == 1, "matrix rows must be contiguous"
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))


def stream_ptr() -> int:
    """Raw hipStream_t of torch's current stream (torch.cuda.current_stream() builds a Python Stream object: 8 us per
    call, more than a launch-bound step's kernels)."""
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
