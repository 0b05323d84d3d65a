"""ctypes binding of ``liblip_hip.so`` (the C ABI declared in ``include/lip.h``).

There is no CPU fallback: if the library is missing or a symbol is absent, importing the
product path raises ``ImportError`` — loudly — so a GPU test can never pass on a silent
PyTorch path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LIP_LIB_PATH") or os.path.join(_HERE, "csrc", "liblip_hip.so")   # env: A/B builds only

LIP_OK = 0
SP_NONE, SP_THETA, SP_CONST, SP_PRIM, SP_WORK, SP_VIN, SP_YOUT, SP_HEAD = -1, 0, 1, 2, 3, 4, 5, 6
OP_IGEMM, OP_WGRAD, OP_REDUCE, OP_POOL_FWD, OP_POOL_BWD, OP_PRIMAL_POST, OP_SOFTMAX, OP_HEAD = 1, 2, 3, 4, 5, 6, 7, 8
OP_MAXPOOL_PRIMAL, OP_MAXPOOL_FWD, OP_MAXPOOL_BWD = 9, 10, 11
HEAD_GGN, HEAD_LT, HEAD_L, HEAD_OUT, HEAD_IN = 0, 1, 2, 3, 4
TAPE_PRIMAL, TAPE_TANGENT, TAPE_BACKWARD = 0, 1, 2
SEG_B_TRANS = 1


class Ref(C.Structure):
    _fields_ = [("space", C.c_int32), ("reserved", C.c_int32), ("off", C.c_int64), ("pstride", C.c_int64)]


class Seg(C.Structure):
    _fields_ = [("a", Ref), ("b", Ref),
                ("IH", C.c_int32), ("IW", C.c_int32), ("C", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32),
                ("stride", C.c_int32), ("pad_h", C.c_int32), ("pad_w", C.c_int32), ("mode", C.c_int32),
                ("flags", C.c_int32)]


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("nseg", C.c_int32), ("seg", Seg * 3),
                ("n_img", C.c_int32), ("OH", C.c_int32), ("OW", C.c_int32), ("N", C.c_int32),
                ("act", C.c_int32), ("ksplit", C.c_int32), ("M", C.c_int32), ("classifier", C.c_int32),
                ("fscale", C.c_float), ("bn_eps", C.c_float),
                ("out", Ref), ("out2", Ref), ("out3", Ref), ("scale", Ref), ("e0", Ref), ("e1", Ref),
                ("xhat", Ref), ("res", Ref), ("dphi", Ref), ("red0", Ref), ("red1", Ref), ("xhat2", Ref),
                ("aux0", Ref), ("aux1", Ref)]


REF_FIELDS = ("out", "out2", "out3", "scale", "e0", "e1", "xhat", "res", "dphi", "red0", "red1", "xhat2",
              "aux0", "aux1")

_F = C.POINTER(C.c_float)
_I = C.POINTER(C.c_int32)
_V = C.c_void_p

# name -> (restype, argtypes); every symbol include/lip.h declares
SIGNATURES = {
    "lip_abi_version": (C.c_int, []),
    "lip_last_error": (C.c_char_p, []),
    "lip_sizeof_op": (C.c_int, []),
    "lip_set_precision": (C.c_int, [C.c_int32]),
    "lip_set_split_k": (C.c_int, [C.c_int32]),
    "lip_set_winograd": (C.c_int, [C.c_int32]),
    "lip_get_winograd": (C.c_int, []),
    "lip_get_precision": (C.c_int, []),
    "lip_engine_create": (C.c_int, [C.POINTER(_V), C.c_int64, C.c_int32, C.c_int32]),
    "lip_engine_destroy": (C.c_int, [_V]),
    "lip_engine_set_tape": (C.c_int, [_V, C.c_int32, C.POINTER(Op), C.c_int32]),
    "lip_engine_bind": (C.c_int, [_V, _V, _V, _V, _V, C.c_int64, C.c_int32]),
    "lip_engine_primal": (C.c_int, [_V, _V]),
    "lip_engine_profile": (C.c_int, [_V, C.c_int32]),
    "lip_engine_profile_read": (C.c_int, [_V, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32]),
    "lip_engine_run_op": (C.c_int, [_V, C.POINTER(Op), _V, _V, _V, C.c_int32, C.c_int32, C.c_float, _V]),
    "lip_debug_run_ops": (C.c_int, [_V, C.c_int32, C.c_int32, C.c_int32, _V, _V, _V, C.c_int32, C.c_int32,
                                    C.c_float, _V]),
    "lip_ggn_vp": (C.c_int, [_V, _V, _V, C.c_int32, C.c_float, C.c_float, _V]),
    "lip_jvp": (C.c_int, [_V, _V, _V, C.c_int32, C.c_int32, C.c_float, _V]),
    "lip_vjp": (C.c_int, [_V, _V, _V, C.c_int32, C.c_int32, C.c_float, _V]),
    "lip_vjp_rows": (C.c_int, [_V, _V, _V, C.c_int32, C.c_int32, C.c_float, _V]),
    "lip_bdot": (C.c_int, [_V, _V, _V, C.c_int32, C.c_int64, _V]),
    "lip_axpby": (C.c_int, [_V, _V, _V, C.c_float, _V, C.c_float, C.c_int32, C.c_int64, _V]),
    "lip_multi_dot": (C.c_int, [_V, _V, _V, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, _V]),
    "lip_multi_axpy_norm": (C.c_int, [_V, _V, _V, _V, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, _V]),
    "lip_scale_store": (C.c_int, [_V, _V, _V, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, _V]),
    "lip_cg_update": (C.c_int, [_V, _V, _V, _V, _V, _V, _V, _V, C.c_int32, C.c_int64, _V]),
    "lip_cg_direction": (C.c_int, [_V, _V, _V, _V, _V, C.c_int32, C.c_int64, _V]),
    "lip_dot_nt_f64": (C.c_int, [_V, C.c_int64, C.c_int32, _V, C.c_int64, C.c_int32, C.c_int64, _V, _V]),
    "lip_gemm_nt": (C.c_int, [_V, C.c_int64, C.c_int32, _V, C.c_int64, C.c_int32, C.c_int64, _V, _V]),
    "lip_gemm_nn_axpy": (C.c_int, [_V, C.c_int64, C.c_int32, C.c_int32, _V, C.c_int64, C.c_int64, _V, C.c_int64, C.c_float, _V,
                                   C.c_int64, _V]),
    "lip_rows_combine": (C.c_int, [_V, _V, C.c_int64, C.c_int32, _V, C.c_int64, C.c_float, _V, C.c_int64, C.c_int32,
                                   C.c_int64, _V]),
    "lip_fill_rademacher": (C.c_int, [_V, C.c_int32, C.c_int64, C.c_uint64, _V]),
    "lip_fill_normal": (C.c_int, [_V, C.c_int32, C.c_int64, C.c_uint64, _V]),
}

_lib: Optional[C.CDLL] = None


class NativeError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the library once; raise ImportError (never fall back) if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (libamdhip64 of its ROCm build).  It must be the one resident in
    # the process before liblip_hip.so is opened, otherwise the library binds the system runtime and the
    # two disagree about devices ("no ROCm-capable device is detected" at the first launch).
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()      # enforce "torch first" (the cause of an early `no ROCm-capable device` smoke failure)
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C laplace-inducing-points_amd/csrc`. There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise ImportError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.lip_sizeof_op() != C.sizeof(Op):
        raise ImportError(f"lip_op_t layout mismatch: C {lib.lip_sizeof_op()} vs ctypes {C.sizeof(Op)}")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != LIP_OK:
        msg = load().lip_last_error().decode("utf-8", "replace")
        raise NativeError(f"{what} failed (status {rc}): {msg}")


def ptr(t) -> int:
    """Device pointer of a contiguous float32 / int32 CUDA tensor (or 0 for None)."""
    if t is None:
        return 0
    if not t.is_cuda:
        raise NativeError("the HIP engine takes device tensors only (no CPU fallback)")
    if not t.is_contiguous():
        raise NativeError("tensor must be contiguous")
    return t.data_ptr()


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


import contextlib as _contextlib


@_contextlib.contextmanager
def winograd_route(mode: int):
    """Process-wide switch of the Winograd route (``lip_set_winograd``) for the duration of a block; restores the
    previous setting.  0 = the direct implicit GEMMs everywhere: bit-for-bit fmaf chains, and an absolute rounding error
    ~4x smaller than the Winograd transforms' on products whose result cancels to far below ||A|| ||v|| (the deflated
    Krylov routes work exactly there); 1 / 2 = the route on."""
    lib = load()
    before = lib.lip_get_winograd()
    check(lib.lip_set_winograd(int(mode)), "lip_set_winograd")
    try:
        yield
    finally:
        lib.lip_set_winograd(before)
