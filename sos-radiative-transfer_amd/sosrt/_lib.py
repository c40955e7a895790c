"""ctypes binding of libsosrt.so (include/sosrt.h).

The product path has no CPU fallback: if the HIP library has not been built
the import of anything that computes fails loudly here.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_longlong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SOSRT_LIB") or os.path.join(os.path.dirname(_HERE), "libsosrt.so")

SOSRT_OK, E_INVALID, E_HIP, E_STATE, E_NOMEM = 0, -1, -2, -3, -4
COL_OK, COL_INDEXERROR, COL_MAXORDERS, COL_INTERNAL = 0, 1, 2, 3
GEOM_THREE_ZONE, GEOM_SINGLE_SLAB = 0, 1
SURFACE_NONE, SURFACE_SPECULAR, SURFACE_LAMBERTIAN, SURFACE_LAMBERTIAN_README = 0, 1, 2, 3
K_GEMM, K_TRANSPORT, K_FIRST, K_SMALLMU, K_ORDER_LOOP = 0, 1, 2, 3, 4
PLAN_GEMM_DENSE, PLAN_GEMM_LIVE64, PLAN_GEMM_LIVE32, PLAN_GEMM_LIVE32_DEEP, PLAN_GEMM_LIVE16_REGS = 0, 1, 2, 3, 4
PLAN_TRANSPORT_GENERAL, PLAN_TRANSPORT_FAST, PLAN_TRANSPORT_RING, PLAN_TRANSPORT_SCAN = 0, 1, 3, 4
CONTRACT_F64, CONTRACT_F32, CONTRACT_F64_FULL = 0, 1, 2
FIRST_ORDER_CODED, FIRST_ORDER_README = 0, 1
PHASE_ISO, PHASE_RAYLEIGH, PHASE_HG, PHASE_TABLE = 0, 1, 2, 3

_dp = POINTER(c_double)
_ip = POINTER(c_int)

# name -> (restype, argtypes); every symbol include/sosrt.h declares
SIGNATURES = {
    "sosrt_last_error": (c_char_p, []),
    "sosrt_version": (c_int, []),
    "sosrt_create": (c_int, [c_int, c_int, c_int, c_int, c_int, POINTER(c_void_p)]),
    "sosrt_destroy": (c_int, [c_void_p]),
    "sosrt_set_stream": (c_int, [c_void_p, c_void_p]),
    "sosrt_use_own_stream": (c_int, [c_void_p]),
    "sosrt_synchronize": (c_int, [c_void_p]),
    "sosrt_set_saved_orders": (c_int, [c_void_p, c_int]),
    "sosrt_set_order_budget": (c_int, [c_void_p, c_int]),
    "sosrt_set_contraction": (c_int, [c_void_p, c_int]),
    "sosrt_phase_asymmetry": (c_int, [c_void_p, POINTER(c_double), _ip]),
    "sosrt_set_order_loop": (c_int, [c_void_p, c_int]),
    "sosrt_order_loop_stats": (c_int, [c_void_p, _ip, _ip, POINTER(c_longlong)]),
    "sosrt_plan_launch": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, _ip]),
    "sosrt_set_first_order": (c_int, [c_void_p, c_int]),
    "sosrt_set_grid": (c_int, [c_void_p, c_void_p]),
    "sosrt_set_phase": (c_int, [c_void_p, c_void_p, c_void_p]),
    "sosrt_set_columns": (c_int, [c_void_p, c_int, c_int, c_int] + [c_void_p] * 9),
    "sosrt_set_columns_zones": (c_int, [c_void_p, c_int, c_int, c_int] + [c_void_p] * 10),
    "sosrt_first_order": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sosrt_source": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "sosrt_transport": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sosrt_solve": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sosrt_solve_dev": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sosrt_last_solve_stats": (c_int, [c_void_p, _ip, POINTER(c_longlong)]),
    "sosrt_fluxes": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "sosrt_epilogue_dev": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int] + [c_void_p] * 6),
    "sosrt_epilogue": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 6),
    "sosrt_phase_table": (c_int, [c_void_p, c_void_p, c_void_p, c_int]),
    "sosrt_phase_p0_dev": (c_int, [c_void_p, c_int, c_int, c_double, c_void_p, c_void_p]),
    "sosrt_phase_p0": (c_int, [c_void_p, c_int, c_int, c_double, c_void_p, c_void_p]),
    "sosrt_phase_matrix": (c_int, [c_void_p, c_int, c_double, c_void_p]),
    "sosrt_comm_unique_id": (c_int, [c_void_p]),
    "sosrt_comm_init": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "sosrt_gather": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "sosrt_comm_destroy": (c_int, [c_void_p]),
    "sosrt_limit_mu_down": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "sosrt_asymptotic_down": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sosrt_plan_weights": (c_int, [c_void_p, c_void_p]),
    "sosrt_plan_fold": (c_int, [c_void_p, c_int, c_void_p]),
    "sosrt_plan_fix_table": (c_int, [c_void_p, c_int, _ip, _ip, c_void_p]),
    "sosrt_plan_fix_count": (c_int, [c_double, c_int]),
    "sosrt_debug_stamps": (c_int, [c_void_p, c_void_p]),
    "sosrt_microbench": (c_int, [c_void_p, c_int, _dp]),
    "sosrt_profile_enable": (c_int, [c_void_p, c_int]),
    "sosrt_profile_reset": (c_int, [c_void_p]),
    "sosrt_profile_get": (c_int, [c_void_p, c_int, _dp, POINTER(c_longlong), _dp]),
}

_lib = None


def _preload_hip_runtime():
    """PyTorch-ROCm ships its own libamdhip64.so.7 / libhsa-runtime64 under torch/lib with the same
    soname as the system ones.  Whichever copy is loaded first serves the whole process, and mixing the
    system HIP runtime with torch's HSA runtime leaves torch without a GPU.  When torch is installed,
    load its copy first (without importing torch), so that the order of `import torch` and the first
    sosrt call does not matter."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(p):
        try:
            ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """The loaded library; raises ImportError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libsosrt.so not found at %s: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        _preload_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class SosrtError(RuntimeError):
    pass


def check(rc):
    if rc == SOSRT_OK:
        return
    msg = lib().sosrt_last_error().decode()
    if rc == E_INVALID:
        raise ValueError(msg)
    if rc == E_NOMEM:
        raise MemoryError(msg)
    raise SosrtError(msg)
