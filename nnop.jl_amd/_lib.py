"""ctypes binding of libnnop_hip.so (the C ABI of include/nnop_hip.h).

This is the same binding, call for call, that the Julia extension in ``julia/NNopHIPExt.jl``
makes with ``ccall``.  There is NO fallback: if the shared library is missing or a symbol is
absent the import of the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NNOP_LIB_PATH: development override (timing-only ablation builds); never set in production
LIB_PATH = os.environ.get("NNOP_LIB_PATH") or os.path.join(_HERE, "lib", "libnnop_hip.so")

# NNOP_HIP_ABI_VERSION of the header this binding was written against; load() refuses another library
ABI_VERSION = 6

# nnop_dtype (include/nnop_hip.h)
NNOP_F32, NNOP_F16, NNOP_BF16 = 0, 1, 2

# nnop_status (include/nnop_hip.h)
NNOP_OK = 0
NNOP_ERR_EMB_MISMATCH = -1
NNOP_ERR_KV_SHAPE = -2
NNOP_ERR_EMB_NOT_POW2 = -3
NNOP_ERR_HEADS = -4
NNOP_ERR_DTYPE = -5
NNOP_ERR_NULL = -6
NNOP_ERR_EMB_UNSUPPORTED = -7
NNOP_ERR_SHAPE = -8
NNOP_ERR_WORKSPACE = -9
NNOP_ERR_HIP = -10
NNOP_ERR_ALIGN = -11

EXPORTED_SYMBOLS = (
    "nnop_fa_fwd",
    "nnop_fa_bwd_workspace_bytes",
    "nnop_fa_bwd_workspace_bytes_pair",
    "nnop_fa_bwd",
    "nnop_llama_rope",
    "nnop_online_softmax",
    "nnop_online_softmax_bwd",
    "nnop_rms_norm",
    "nnop_rms_norm_bwd",
    "nnop_layer_norm",
    "nnop_layer_norm_bwd",
    "nnop_norm_bwd_workspace_bytes",
    "nnop_fa_shards",
    "nnop_shared_memory",
    "nnop_strerror",
    "nnop_abi_version",
)


# Test-only hooks (csrc/nnop_debug.h): exported by the library, deliberately NOT in the public header.
DEBUG_SYMBOLS = ("nnop_debug_set", "nnop_debug_dev_build", "nnop_debug_fwd_form", "nnop_debug_bwd_form")
FWD_FORMS = {0: "fa_fwd_kernel", 1: "fa_fwd_split_kernel", 2: "fa_fwd_w64_kernel", 3: "fa_fwd_generic_kernel", 4: "fa_fwd_duo_kernel"}
# keys of nnop_debug_set == enum TuneKey (csrc/tuning.hpp)
TUNE_KEYS = {"fwd_split": 0, "fwd_nw": 1, "fwd_w64": 2, "bwd_big7": 3, "norm_bwd_cap": 4, "bwd_nw": 5,
             "fwd_exact_scale": 6, "bwd_w64": 7, "bwd_stages": 8, "fwd_persist": 9, "bwd_persist": 10, "fwd_duo": 11, "fwd_persist_asc": 12, "bwd_narrow": 13, "fwd_causal_alt": 14}


class FaShard(C.Structure):
    """struct nnop_fa_shard (declared below FaDesc; fields filled in after it)"""


class FaDesc(C.Structure):
    """struct nnop_fa_desc"""
    _fields_ = [
        ("dtype", C.c_int32), ("emb", C.c_int32), ("ql", C.c_int32), ("kl", C.c_int32),
        ("qh", C.c_int32), ("kh", C.c_int32), ("batch", C.c_int32), ("causal", C.c_int32),
        ("emb_k", C.c_int32), ("emb_v", C.c_int32), ("kl_v", C.c_int32), ("kh_v", C.c_int32),
    ]


FaShard._fields_ = [("desc", FaDesc), ("b0", C.c_int32), ("b1", C.c_int32), ("kh0", C.c_int32), ("kh1", C.c_int32),
                    ("q_off", C.c_uint64), ("kv_off", C.c_uint64), ("row_off", C.c_uint64), ("mask_off", C.c_uint64),
                    ("pair_off", C.c_int64)]


class RopeDesc(C.Structure):
    """struct nnop_rope_desc"""
    _fields_ = [
        ("dtype", C.c_int32), ("cs_dtype", C.c_int32), ("dim", C.c_int32), ("seq", C.c_int32),
        ("qh", C.c_int32), ("kh", C.c_int32), ("batch", C.c_int32),
    ]


class SoftmaxDesc(C.Structure):
    """struct nnop_softmax_desc"""
    _fields_ = [("dtype", C.c_int32), ("n", C.c_int32), ("batch", C.c_int64)]


class NormDesc(C.Structure):
    """struct nnop_norm_desc"""
    _fields_ = [("dtype", C.c_int32), ("w_dtype", C.c_int32), ("emb", C.c_int32), ("reserved", C.c_int32),
                ("n", C.c_int64)]


class NNopLibraryMissing(ImportError):
    pass


_lib = None


def load():
    """dlopen the library once and declare the prototypes.  Raises loudly when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NNopLibraryMissing(
            f"{LIB_PATH} not found: build it with `make -C nnop.jl_amd/csrc -j8` "
            "(or __graft_entry__.build()).  There is no CPU or PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    # the version first: a stale library may lack symbols this binding declares below (a bare AttributeError otherwise)
    if not hasattr(lib, "nnop_abi_version"):
        raise NNopLibraryMissing(f"{LIB_PATH} exports no nnop_abi_version: rebuild with `make -C nnop.jl_amd/csrc -j8`")
    lib.nnop_abi_version.restype = C.c_int
    lib.nnop_abi_version.argtypes = []
    if lib.nnop_abi_version() != ABI_VERSION:
        raise NNopLibraryMissing(f"{LIB_PATH} has ABI version {lib.nnop_abi_version()}, this binding needs "
                                 f"{ABI_VERSION}: rebuild with `make -C nnop.jl_amd/csrc -j8`")
    vp, u8p = C.c_void_p, C.c_void_p
    lib.nnop_fa_fwd.restype = C.c_int
    lib.nnop_fa_fwd.argtypes = [C.POINTER(FaDesc), vp, vp, vp, vp, vp, vp, vp, u8p, vp]
    lib.nnop_fa_bwd_workspace_bytes.restype = C.c_size_t
    lib.nnop_fa_bwd_workspace_bytes.argtypes = [C.POINTER(FaDesc)]
    lib.nnop_fa_bwd_workspace_bytes_pair.restype = C.c_size_t
    lib.nnop_fa_bwd_workspace_bytes_pair.argtypes = [C.POINTER(FaDesc)]
    lib.nnop_fa_bwd.restype = C.c_int
    lib.nnop_fa_bwd.argtypes = [C.POINTER(FaDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                u8p, vp, C.c_size_t, vp]
    lib.nnop_llama_rope.restype = C.c_int
    lib.nnop_llama_rope.argtypes = [C.POINTER(RopeDesc), vp, vp, vp, vp, vp, vp, C.c_float, vp]
    lib.nnop_online_softmax.restype = C.c_int
    lib.nnop_online_softmax.argtypes = [C.POINTER(SoftmaxDesc), vp, vp, vp]
    lib.nnop_online_softmax_bwd.restype = C.c_int
    lib.nnop_online_softmax_bwd.argtypes = [C.POINTER(SoftmaxDesc), vp, vp, vp, vp]
    nd = C.POINTER(NormDesc)
    lib.nnop_rms_norm.restype = C.c_int
    lib.nnop_rms_norm.argtypes = [nd, vp, vp, vp, vp, C.c_float, C.c_float, vp]
    lib.nnop_rms_norm_bwd.restype = C.c_int
    lib.nnop_rms_norm_bwd.argtypes = [nd, vp, vp, vp, vp, vp, vp, C.c_float, vp, C.c_size_t, vp]
    lib.nnop_layer_norm.restype = C.c_int
    lib.nnop_layer_norm.argtypes = [nd, vp, vp, vp, vp, vp, vp, C.c_float, vp]
    lib.nnop_layer_norm_bwd.restype = C.c_int
    lib.nnop_layer_norm_bwd.argtypes = [nd, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.nnop_norm_bwd_workspace_bytes.restype = C.c_size_t
    lib.nnop_norm_bwd_workspace_bytes.argtypes = [nd, C.c_int]
    lib.nnop_fa_shards.restype = C.c_int
    lib.nnop_fa_shards.argtypes = [C.POINTER(FaDesc), C.c_int, C.c_int, C.POINTER(FaShard)]
    lib.nnop_shared_memory.restype = C.c_int
    lib.nnop_shared_memory.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
    lib.nnop_strerror.restype = C.c_char_p
    lib.nnop_strerror.argtypes = [C.c_int]
    # test-only hooks (csrc/nnop_debug.h): not part of the ABI, bound only when the library carries them
    if hasattr(lib, "nnop_debug_set"):
        lib.nnop_debug_set.restype = C.c_int
        lib.nnop_debug_set.argtypes = [C.c_int, C.c_int]
    if hasattr(lib, "nnop_debug_dev_build"):
        lib.nnop_debug_dev_build.restype = C.c_int
        lib.nnop_debug_dev_build.argtypes = []
    for name in ("nnop_debug_fwd_form", "nnop_debug_bwd_form"):
        if hasattr(lib, name):
            getattr(lib, name).restype = C.c_int
            getattr(lib, name).argtypes = [C.POINTER(FaDesc), C.c_int, C.c_int]
    _lib = lib
    return lib


def strerror(status: int) -> str:
    return load().nnop_strerror(int(status)).decode()


def debug_set(key: str, value: int) -> int:
    """Test-only: override one launch-shape knob (csrc/tuning.hpp); -1 = automatic.  Returns the previous value.
    The hook is LOCKED unless the process environment held NNOP_DEBUG_HOOKS=1 at the library's first launch (the test-suite's
    conftest and bench.py set it): a production host cannot flip process-wide kernel selection by accident."""
    prev = load().nnop_debug_set(TUNE_KEYS[key], int(value))
    if prev == -(2 ** 31):
        if os.environ.get("NNOP_DEBUG_HOOKS", "0") in ("", "0"):
            raise RuntimeError("nnop_debug_set is locked: start the process with NNOP_DEBUG_HOOKS=1 (csrc/nnop_debug.h)")
        raise KeyError(key)
    return prev


def fa_shards(desc: FaDesc, world: int, rank: int):
    """nnop_fa_shards: rank's (batch, kv-head) unit range as <= 3 dense rectangles with their element offsets (host-only)."""
    out = (FaShard * 3)()
    n = load().nnop_fa_shards(C.byref(desc), int(world), int(rank), out)
    if n < 0:
        raise ValueError(strerror(n))
    return [out[i] for i in range(n)]


def dev_build() -> bool:
    return bool(load().nnop_debug_dev_build())


def fwd_form(desc: FaDesc, has_pair: bool = False, has_mask: bool = False) -> str:
    """Name of the forward kernel the launcher picks for this problem (reporting only; csrc/nnop_debug.h)."""
    code = load().nnop_debug_fwd_form(C.byref(desc), int(has_pair), int(has_mask))
    if code < 0:
        raise ValueError(strerror(code))
    return FWD_FORMS[code]


def bwd_kernels(desc: FaDesc, has_pair: bool = False, has_mask: bool = False):
    """Names of the (dK/dV, dQ) kernels the launcher picks for this problem (reporting only; csrc/nnop_debug.h)."""
    code = load().nnop_debug_bwd_form(C.byref(desc), int(has_pair), int(has_mask))
    if code < 0:
        raise ValueError(strerror(code))
    return ("fa_bwd_w64_kernel<dK/dV>" if code & 1 else "fa_bwd_dkdv_kernel", "fa_bwd_w64_kernel<dQ>" if code & 2 else "fa_bwd_dq_kernel")
