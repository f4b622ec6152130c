"""ctypes binding of libtadmm_hip.so (include/tadmm.h).

The library is the product; there is no CPU fallback.  ``load()`` raises
``TadmmLibraryError`` when the shared object is missing or does not export the
full C ABI, and every compute wrapper raises ``TadmmError`` when the C side
returns a negative status (with ``tadmm_last_error`` text).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

MAX_MODES = 8

KIND_TT_CONV, KIND_TT_LINEAR, KIND_SVD, KIND_TUCKER2 = 0, 1, 2, 3
FLAG_SKIP_ROTATIONS = 1

_LIB_NAME = "libtadmm_hip.so"
_HERE = os.path.dirname(os.path.abspath(__file__))


class TadmmLibraryError(RuntimeError):
    pass


class TadmmError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"tadmm status {status}: {msg}")
        self.status = status


class LayerDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("ndim", C.c_int32),
        ("dims", C.c_int64 * 4),
        ("d", C.c_int32),
        ("tt_shapes", C.c_int32 * MAX_MODES),
        ("ranks", C.c_int32 * (MAX_MODES + 1)),
        ("flags", C.c_uint32),
        ("hooi_max_iter", C.c_int32),
        ("hooi_tol", C.c_float),
    ]


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("a_rs", C.c_int64), ("a_cs", C.c_int64),
        ("b_rs", C.c_int64), ("b_cs", C.c_int64),
        ("c_rs", C.c_int64), ("c_cs", C.c_int64),
        ("alpha", C.c_float), ("beta", C.c_float),
        ("bias_n", C.c_void_p), ("bias_m", C.c_void_p),
    ]


class ChainDesc(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("Y", C.c_void_p), ("Win", C.c_void_p), ("Wout", C.c_void_p), ("bias", C.c_void_p),
        ("T", C.c_int64),
        ("Kin", C.c_int32), ("R", C.c_int32), ("Nout", C.c_int32),
        ("ldx", C.c_int64), ("ldy", C.c_int64), ("win_plane", C.c_int64), ("wout_plane", C.c_int64),
        ("x_hw", C.c_int32), ("y_hw", C.c_int32), ("dtype", C.c_int32), ("tile_tokens", C.c_int32),
    ]


class ConvChainDesc(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("Y", C.c_void_p), ("W1", C.c_void_p), ("W2", C.c_void_p), ("W3", C.c_void_p),
        ("bias", C.c_void_p),
        ("w1_plane", C.c_int64), ("w2_plane", C.c_int64), ("w3_plane", C.c_int64),
        ("B", C.c_int32), ("C", C.c_int32), ("R1", C.c_int32), ("R2", C.c_int32), ("Nout", C.c_int32),
        ("H", C.c_int32), ("W", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
        ("stride_h", C.c_int32), ("stride_w", C.c_int32), ("pad_h", C.c_int32), ("pad_w", C.c_int32),
        ("dil_h", C.c_int32), ("dil_w", C.c_int32), ("dtype", C.c_int32),
    ]


CHAIN_F32, CHAIN_BF16 = 0, 1

# name -> (restype, argtypes); this table IS the list of symbols include/tadmm.h declares
ABI = {
    "tadmm_version": (C.c_int, []),
    "tadmm_abi_sizes": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tadmm_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "tadmm_destroy": (C.c_int, [C.c_void_p]),
    "tadmm_last_error": (C.c_char_p, [C.c_void_p]),
    "tadmm_tt_clamp_ranks": (C.c_int, [C.POINTER(LayerDesc)]),
    "tadmm_plan_workspace_bytes": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(LayerDesc), C.POINTER(C.c_size_t)]),
    "tadmm_plan_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(LayerDesc), C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                    C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "tadmm_plan_run": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "tadmm_plan_singular_values": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p]),
    "tadmm_plan_last_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "tadmm_plan_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "tadmm_plan_set_jacobi": (C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_int]),
    "tadmm_plan_filter_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "tadmm_plan_filter_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "tadmm_plan_filter_timing_fast": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "tadmm_plan_jacobi_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "tadmm_plan_ranks": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32)]),
    "tadmm_plan_lanes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "tadmm_lane_split": (C.c_int, [C.c_int, C.POINTER(LayerDesc), C.POINTER(C.c_int32)]),
    "tadmm_plan_destroy": (C.c_int, [C.c_void_p]),
    "tadmm_tucker_workspace_bytes": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(LayerDesc), C.POINTER(C.c_size_t)]),
    "tadmm_tucker_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(LayerDesc), C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t,
                                      C.POINTER(C.c_void_p)]),
    "tadmm_tucker_run": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "tadmm_tucker_factors": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_void_p)]),
    "tadmm_tucker_jacobi_sweeps": (C.c_int, [C.c_void_p]),
    "tadmm_tucker_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "tadmm_tucker_last_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "tadmm_tucker_iterations": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_void_p]),
    "tadmm_tucker_destroy": (C.c_int, [C.c_void_p]),
    "tadmm_penalty_scratch_doubles": (C.c_int, []),
    "tadmm_penalty": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "tadmm_gemm_pack_bytes": (C.c_size_t, [C.c_int, C.POINTER(GemmDesc)]),
    "tadmm_gemm_pack": (C.c_int, [C.c_int, C.POINTER(GemmDesc), C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]),
    "tadmm_gemm_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "tadmm_gemm": (C.c_int, [C.c_void_p, C.POINTER(GemmDesc), C.c_void_p]),
    "tadmm_gemm_bf16_nt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "tadmm_chain_desc_bytes": (C.c_int, []),
    "tadmm_conv_chain_desc_bytes": (C.c_int, []),
    "tadmm_ttconv_fused": (C.c_int, [C.c_void_p, C.POINTER(ConvChainDesc), C.c_void_p]),
    "tadmm_ttlinear_fwd": (C.c_int, [C.c_void_p, C.POINTER(ChainDesc), C.c_void_p]),
    "tadmm_ttlinear_bwd": (C.c_int, [C.c_void_p, C.POINTER(ChainDesc), C.c_void_p]),
    "tadmm_ttconv_chain_in": (C.c_int, [C.c_void_p, C.POINTER(ChainDesc), C.c_void_p]),
    "tadmm_ttconv_chain_out": (C.c_int, [C.c_void_p, C.POINTER(ChainDesc), C.c_void_p]),
    "tadmm_tucker_1x1": (C.c_int, [C.c_void_p, C.POINTER(ChainDesc), C.c_void_p]),
    "tadmm_gram_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "tadmm_gram_ld": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tadmm_gram_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                 C.c_size_t, C.c_void_p]),
    "tadmm_eigh_scratch_bytes": (C.c_size_t, [C.c_int]),
    "tadmm_eigh_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                 C.POINTER(C.c_int), C.c_void_p]),
    "tadmm_dgemm_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "tadmm_dgemm3_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "tadmm_dgemm3_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tadmm_dgemm_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tadmm_cholqr_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "tadmm_cholqr_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                   C.POINTER(C.c_int), C.c_void_p]),
}

_lib = None
_lock = threading.Lock()


def library_path():
    # TADMM_LIB: file name of an instrumented build of the same sources next to the product library (e.g. the
    # -DTADMM_TRI_STAMPS variant scripts/stamp_tri.py uses); never a fallback -- a missing file still raises
    return os.path.join(_HERE, os.path.basename(os.environ.get("TADMM_LIB", _LIB_NAME)))


def load():
    """dlopen the HIP library and bind every symbol of the ABI table.  Raises loudly."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        # PyTorch-ROCm ships its own libamdhip64; import it FIRST so that this library's HIP calls resolve to the
        # same runtime instance that owns torch's device memory and streams (loading the system runtime first
        # leaves the process with a runtime torch then cannot see a device through)
        import torch  # noqa: F401
        path = library_path()
        if not os.path.exists(path):
            raise TadmmLibraryError(
                f"{path} not found: build it with `make -C dnn-compression-tensor-admm_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        try:
            lib = C.CDLL(path)
        except OSError as e:  # pragma: no cover
            raise TadmmLibraryError(f"cannot load {path}: {e}") from e
        missing = []
        for name, (res, args) in ABI.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
        if missing:
            raise TadmmLibraryError(f"{path} does not export: {', '.join(missing)}")
        a, b = C.c_int(), C.c_int()
        lib.tadmm_abi_sizes(C.byref(a), C.byref(b))
        if (a.value, b.value) != (C.sizeof(LayerDesc), C.sizeof(GemmDesc)):
            raise TadmmLibraryError(f"{path}: struct layout mismatch (library {a.value}/{b.value} bytes, "
                                    f"binding {C.sizeof(LayerDesc)}/{C.sizeof(GemmDesc)})")
        if lib.tadmm_chain_desc_bytes() != C.sizeof(ChainDesc):
            raise TadmmLibraryError(f"{path}: tadmm_chain_desc layout mismatch (library {lib.tadmm_chain_desc_bytes()} "
                                    f"bytes, binding {C.sizeof(ChainDesc)})")
        if lib.tadmm_conv_chain_desc_bytes() != C.sizeof(ConvChainDesc):
            raise TadmmLibraryError(f"{path}: tadmm_conv_chain_desc layout mismatch")
        _lib = lib
        return lib


class Handle:
    """One tadmm handle per device (not thread-safe, like the C object)."""

    _cache = {}

    def __init__(self, device_index: int):
        self.lib = load()
        self.ptr = C.c_void_p()
        rc = self.lib.tadmm_create(int(device_index), C.byref(self.ptr))
        if rc != 0:
            msg = self.lib.tadmm_last_error(self.ptr).decode() if self.ptr else "create failed"
            raise TadmmError(rc, msg)
        self.device_index = device_index

    @classmethod
    def get(cls, device_index: int) -> "Handle":
        h = cls._cache.get(device_index)
        if h is None:
            h = cls(device_index)
            cls._cache[device_index] = h
        return h

    def check(self, rc):
        if rc < 0:
            raise TadmmError(rc, self.lib.tadmm_last_error(self.ptr).decode())
        return rc


def make_layer_desc(kind, dims, tt_shapes=None, ranks=None, flags=0, hooi_max_iter=100, hooi_tol=1e-4):
    d = LayerDesc()
    d.kind = kind
    d.ndim = len(dims)
    for i, v in enumerate(dims):
        d.dims[i] = int(v)
    if kind == KIND_SVD:
        d.d = 2
        d.ranks[0] = int(ranks if isinstance(ranks, int) else ranks[0])
    elif kind == KIND_TUCKER2:
        d.d = 2
        d.ranks[0], d.ranks[1] = int(ranks[0]), int(ranks[1])
    else:
        if len(tt_shapes) > MAX_MODES:
            raise ValueError(f"at most {MAX_MODES} TT modes are supported")
        d.d = len(tt_shapes)
        for i, v in enumerate(tt_shapes):
            d.tt_shapes[i] = int(v)
        if len(ranks) != len(tt_shapes) + 1:
            raise ValueError("len(ranks) must be len(tt_shapes)+1")
        for i, v in enumerate(ranks):
            d.ranks[i] = int(v)
    d.flags = flags
    d.hooi_max_iter = hooi_max_iter
    d.hooi_tol = hooi_tol
    return d


def clamp_ranks(tt_shapes, ranks):
    """ttd.py:18-19 -- the rank clamp as a pure function of shapes (host only)."""
    lib = load()
    dims = [1, 1]
    d = LayerDesc()
    d.kind = KIND_TT_LINEAR
    d.ndim = 2
    d.dims[0], d.dims[1] = dims
    d.d = len(tt_shapes)
    for i, v in enumerate(tt_shapes):
        d.tt_shapes[i] = int(v)
    for i, v in enumerate(ranks):
        d.ranks[i] = int(v)
    rc = lib.tadmm_tt_clamp_ranks(C.byref(d))
    if rc < 0:
        raise TadmmError(rc, "tadmm_tt_clamp_ranks: invalid descriptor")
    return [int(d.ranks[i]) for i in range(len(ranks))]
