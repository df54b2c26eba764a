"""ctypes declarations for include/friedrich_amd.h (the C ABI of libfriedrich_amd.so).

This module only binds; it holds no numerical code and there is no fallback: if the HIP library has not
been built (python -m friedrich_amd.build) loading raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfriedrich_amd.so")

FR_OK = 0
FR_NOT_POSITIVE_DEFINITE = 1
FR_SINGULAR_SOLVE = 2
FR_UNSUPPORTED_KERNEL = 3
FR_SHAPE = 4
FR_INVALID_ARGUMENT = 5
FR_OUT_OF_MEMORY = 6
FR_HIP_ERROR = 7
FR_RCCL_ERROR = 8
FR_NO_DEVICE = 9

STATUS_NAMES = {
    0: "FR_OK", 1: "FR_NOT_POSITIVE_DEFINITE", 2: "FR_SINGULAR_SOLVE", 3: "FR_UNSUPPORTED_KERNEL", 4: "FR_SHAPE",
    5: "FR_INVALID_ARGUMENT", 6: "FR_OUT_OF_MEMORY", 7: "FR_HIP_ERROR", 8: "FR_RCCL_ERROR", 9: "FR_NO_DEVICE",
}

FR_PROF_GRAM, FR_PROF_POTF2, FR_PROF_GEMM_PANEL, FR_PROF_SYRK, FR_PROF_GEMM_SOLVE, FR_PROF_REDUCE, FR_PROF_COMM, FR_PROF_SYRK_CHAIN = range(8)
PROF_NAMES = ["gram", "potf2", "gemm_panel", "syrk", "gemm_solve", "reduce", "comm", "syrk_chain"]

FR_KPROG_MAX_OPS = 15
FR_COMM_ID_BYTES = 128

LEAF_KINDS = {
    "linear": (0, 1),
    "polynomial": (1, 3),
    "squared_exp": (2, 2),
    "gaussian": (2, 2),
    "exponential": (3, 2),
    "matern1": (4, 2),
    "matern2": (5, 2),
    "hyper_tan": (6, 2),
    "multiquadric": (7, 1),
    "rational_quadratic": (8, 2),
}
K_SUM, K_PROD = 100, 101


class KernelOp(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("nparams", ctypes.c_int32), ("params", ctypes.c_double * 3)]


class KProg(ctypes.Structure):
    _fields_ = [("nops", ctypes.c_int32), ("reserved", ctypes.c_int32), ("ops", KernelOp * FR_KPROG_MAX_OPS)]


# every exported symbol of include/friedrich_amd.h (tests/test_abi.py checks the list against the header)
_dp = ctypes.c_void_p  # data pointers are passed as integers: host (numpy) or device (torch .data_ptr())
_i64 = ctypes.c_int64
_int = ctypes.c_int
_dbl = ctypes.c_double
_vp = ctypes.c_void_p
_kp = ctypes.POINTER(KProg)
_pp = ctypes.POINTER(ctypes.c_void_p)
_pi64 = ctypes.POINTER(ctypes.c_int64)
_pdbl = ctypes.POINTER(ctypes.c_double)
_pint = ctypes.POINTER(ctypes.c_int)

SIGNATURES = {
    "fr_abi_version": (_int, []),
    "fr_ctx_create": (_int, [_pp, _int]),
    "fr_ctx_destroy": (None, [_vp]),
    "fr_ctx_set_stream": (_int, [_vp, _vp]),
    "fr_ctx_synchronize": (_int, [_vp]),
    "fr_last_error": (ctypes.c_char_p, [_vp]),
    "fr_ctx_set_option": (_int, [_vp, ctypes.c_char_p, _i64]),
    "fr_ctx_get_counter": (_int, [_vp, ctypes.c_char_p, _pi64]),
    "fr_ctx_profile_enable": (_int, [_vp, _int]),
    "fr_ctx_profile_reset": (_int, [_vp]),
    "fr_ctx_profile_get": (_int, [_vp, _int, _pdbl, _pi64, _pdbl, _pdbl]),
    "fr_comm_unique_id": (_int, [_vp]),
    "fr_ctx_comm_init": (_int, [_vp, _int, _int, _vp]),
    "fr_ctx_comm_init_local": (_int, [_vp, _int, _int, _int]),
    "fr_ctx_comm_info": (_int, [_vp, _pint, _pint]),
    "fr_ctx_comm_selftest": (_int, [_vp]),
    "fr_ctx_comm_finalize": (_int, [_vp, _int]),
    "fr_inputs_to_device": (_int, [_vp, _int, _vp, _i64, _i64, _i64, _pp, _pi64]),
    "fr_device_free": (None, [_vp, _vp]),
    "fr_linear_prior_fit": (_int, [_vp, _dp, _i64, _i64, _i64, _dp, _pdbl, _pdbl]),
    "fr_gram": (_int, [_vp, _kp, _dp, _i64, _i64, _dp, _i64, _i64, _i64, _dp, _i64]),
    "fr_chol_from_inputs": (_int, [_vp, _kp, _dp, _i64, _i64, _i64, _dbl, _int, _dbl, _i64, _pp]),
    "fr_chol_refactor": (_int, [_vp, _kp, _dbl, _int, _dbl]),
    "fr_chol_from_matrix": (_int, [_vp, _dp, _i64, _i64, _int, _dbl, _pp]),
    "fr_chol_add_rows": (_int, [_vp, _kp, _dp, _i64, _i64, _i64, _i64, _dbl]),
    "fr_chol_info": (_int, [_vp, _pi64, _pi64, _pi64, _pi64, _pi64]),
    "fr_chol_conditioning": (_int, [_vp, _pdbl, _pint]),
    "fr_chol_substitutions": (_int, [_vp, _pi64, _i64]),
    "fr_chol_solve": (_int, [_vp, _dp, _i64, _i64]),
    "fr_chol_solve_lower": (_int, [_vp, _dp, _i64, _i64]),
    "fr_chol_inverse": (_int, [_vp, _dp, _i64]),
    "fr_chol_download_l": (_int, [_vp, _dp, _i64, _int]),
    "fr_chol_upload_l": (_int, [_vp, _dp, _i64, _i64, _dp, _i64, _i64, _i64, _pp]),
    "fr_chol_free": (None, [_vp]),
    "fr_chol_set_targets": (_int, [_vp, _dp]),
    "fr_likelihood": (_int, [_vp, _kp, _dp, _dbl, _pdbl]),
    "fr_predict_mean": (_int, [_vp, _kp, _dp, _dp, _i64, _i64, _dp, _dp]),
    "fr_predict_variance": (_int, [_vp, _kp, _dp, _i64, _i64, _dp]),
    "fr_predict_mean_variance": (_int, [_vp, _kp, _dp, _dp, _i64, _i64, _dp, _dp, _dp]),
    "fr_predict_covariance": (_int, [_vp, _kp, _dp, _i64, _i64, _dp, _i64]),
    "fr_posterior": (_int, [_vp, _kp, _dp, _dp, _i64, _i64, _dp, _dp, _dp, _i64, _dp, _i64]),
    "fr_gemm": (_int, [_vp, _int, _int, _i64, _i64, _i64, _dbl, _dp, _i64, _dp, _i64, _dbl, _dp, _i64]),
    "fr_panel_rows_solve": (_int, [_vp, _dp, _i64, _i64, _dp, _i64, _i64, _dp]),
    "fr_mean_pairwise_distance": (_int, [_vp, _dp, _i64, _i64, _i64, _pdbl]),
    "fr_grad_terms": (_int, [_vp, _kp, _dp, _dbl, _int, _pdbl, _pdbl]),
}

_lib = None
hip_runtime = None  # "torch" or "system" once loaded


def _preload_hip_runtime():
    """libfriedrich_amd.so carries no DT_NEEDED on libamdhip64: bind it to the ONE HIP runtime of this process.
    torch wheels bundle their own libamdhip64/libhsa-runtime64/librccl; a second runtime in the same process
    cannot open the GPU, so when torch is installed its copy is used (FRIEDRICH_AMD_HIP_RUNTIME=system|torch
    overrides)."""
    global hip_runtime
    mode = os.environ.get("FRIEDRICH_AMD_HIP_RUNTIME", "auto")
    if mode in ("auto", "torch"):
        try:
            import torch  # noqa: F401  (loads its bundled ROCm libraries)

            tl = os.path.join(os.path.dirname(torch.__file__), "lib")
            p = os.path.join(tl, "libamdhip64.so")
            if os.path.exists(p):
                ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)
                rccl = os.path.join(tl, "librccl.so")
                if os.path.exists(rccl):
                    os.environ.setdefault("FRIEDRICH_AMD_RCCL_PATH", rccl)
                hip_runtime = "torch"
                return
        except ImportError:
            if mode == "torch":
                raise
    for cand in ("/opt/rocm/lib/libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so", "libamdhip64.so.7",
                 "libamdhip64.so"):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
            hip_runtime = "system"
            return
        except OSError:
            continue
    raise RuntimeError("no HIP runtime (libamdhip64) found")


def load():
    """Load libfriedrich_amd.so and declare every entry point.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP library first (python -m friedrich_amd.build). "
            "There is no CPU fallback behind this package.")
    _preload_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def flatten_spec(spec):
    name = spec[0]
    if name in ("sum", "prod"):
        return flatten_spec(spec[1]) + flatten_spec(spec[2]) + [(K_SUM if name == "sum" else K_PROD, [])]
    kind, npar = LEAF_KINDS[name]
    params = [float(v) for v in spec[1:]]
    if len(params) != npar:
        raise ValueError(f"kernel {name} takes {npar} parameters, got {len(params)}")
    return [(kind, params)]


def kprog(spec):
    """nested-tuple kernel spec (see oracle/oracle.py for the grammar) -> fr_kprog"""
    if isinstance(spec, KProg):
        return spec
    ops = flatten_spec(spec)
    if len(ops) > FR_KPROG_MAX_OPS:
        raise ValueError("kernel program too long")
    p = KProg()
    p.nops = len(ops)
    p.reserved = 0
    for i, (kind, params) in enumerate(ops):
        p.ops[i].kind = kind
        p.ops[i].nparams = len(params)
        for q in range(3):
            p.ops[i].params[q] = params[q] if q < len(params) else 0.0
    return p
