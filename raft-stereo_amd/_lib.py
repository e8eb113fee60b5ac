"""ctypes binding of libraftcorr_hip.so (declared in include/raftcorr.h).

The product path has no fallback: if the shared library is missing or a symbol cannot be
resolved, importing this module raises, and every operator built on it fails loudly.
"""
import ctypes as C
import os

RC_MAX_LEVELS = 4
RC_MAX_RADIUS = 8
RC_MAX_FUSED_LOOKUPS = 32
RC_CONVC1_OUT = 64
RC_CONVC1_MAX_RADIUS = 7
RC_SAMPLER_MAX_RADIUS = 7
RC_ALT_MAX_RADIUS = 6
RC_F32, RC_F16, RC_BF16 = 0, 1, 2
RC_COORD_EXACT, RC_COORD_GRIDSAMPLE = 0, 1
RC_WRAP_ROWS = 32
RC_VERSION = 200

LIB_NAME = "libraftcorr_hip.so"
# RAFTCORR_LIB: another build of the same library (kernel A/B labs under tools/); it must exist and export everything
LIB_PATH = os.environ.get("RAFTCORR_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


class Layout(C.Structure):
    """rc_layout of include/raftcorr.h."""
    _fields_ = [("B", C.c_int64), ("H", C.c_int64), ("W1", C.c_int64), ("W2", C.c_int64),
                ("L", C.c_int32), ("pitch", C.c_int32), ("rb", C.c_int32), ("dtype", C.c_int32),
                ("width", C.c_int32 * RC_MAX_LEVELS),
                ("widthp", C.c_int32 * RC_MAX_LEVELS),
                ("stored", C.c_int32 * RC_MAX_LEVELS),
                ("full", C.c_int32), ("reserved", C.c_int32),
                ("offset", C.c_int64 * RC_MAX_LEVELS),
                ("total", C.c_int64)]


class AltLayout(C.Structure):
    """rc_alt_layout of include/raftcorr.h."""
    _fields_ = [("B", C.c_int64), ("H", C.c_int64), ("W1", C.c_int64), ("W2", C.c_int64),
                ("D", C.c_int64), ("Dp", C.c_int64),
                ("L", C.c_int32),
                ("width", C.c_int32 * RC_MAX_LEVELS),
                ("f1_elems", C.c_int64),
                ("f2_offset", C.c_int64 * RC_MAX_LEVELS),
                ("f2_elems", C.c_int64)]


_P, _I, _L = C.c_void_p, C.c_int, C.c_int64
_LP, _ALP = C.POINTER(Layout), C.POINTER(AltLayout)

# name -> (restype, argtypes); must list every symbol include/raftcorr.h declares
SIGNATURES = {
    "rc_version": (_I, []),
    "rc_last_error_string": (C.c_char_p, []),
    "rc_pyramid_layout": (_I, [_L, _L, _L, _L, _I, _I, _LP]),
    "rc_grad_pyramid_layout": (_I, [_L, _L, _L, _L, _I, _LP]),
    "rc_build_pyramid": (_I, [_P, _P, _P, _LP, _L, _I, _P]),
    "rc_build_pyramid_naive": (_I, [_P, _P, _P, _LP, _L, _I, _P]),
    "rc_lookup_fwd": (_I, [_P, _LP, _P, _I, _P, _I, _I, _P]),
    "rc_lookup_fwd_vt": (_I, [_P, _LP, _P, _I, _P, _I, _I, _P]),
    "rc_lookup_fwd_update": (_I, [_P, _LP, _P, _I, _P, _P, _P, _P, _I, _I, _P]),
    "rc_lookup_bwd": (_I, [_P, _I, _P, _P, _LP, _I, _I, _P]),
    "rc_build_bwd_work_bytes": (_L, [_LP, _I]),
    "rc_build_bwd": (_I, [_P, _LP, _P, _P, _P, _P, _L, _I, _P, _P]),
    "rc_dense_grad_ld": (_L, [_LP]),
    "rc_lookup_bwd_dense": (_I, [C.POINTER(_P), C.POINTER(_P), _I, _I, _P, _LP, _L, _I, _I, _I, _P]),
    "rc_build_bwd_dense": (_I, [_P, _LP, _P, _P, _P, _P, _L, _I, _P]),
    "rc_convc1_pack": (_I, [_P, _I, _I, _I, _P, _P]),
    "rc_lookup_convc1_fwd": (_I, [_P, _LP, _P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "rc_lookup_convc1_fwd_update": (_I, [_P, _LP, _P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "rc_convc1_bwd_work_bytes": (_L, []),
    "rc_convc1_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _L, _I, _I, _P]),
    "rc_convc1_bwd_static": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _L, _I, _I, _P]),
    "rc_sampler_fwd": (_I, [_P, _P, _I, _P, _L, _L, _L, _L, _I, _I, _P]),
    "rc_sampler_bwd": (_I, [_P, _I, _P, _P, _L, _L, _L, _L, _I, _I, _P]),
    "rc_alt_layout_init": (_I, [_L, _L, _L, _L, _L, _I, _ALP]),
    "rc_alt_prepare": (_I, [_P, _P, _P, _P, _ALP, _I, _P]),
    "rc_alt_fwd": (_I, [_P, _P, _ALP, _I, _P, _I, _P, _I, _I, _P]),
    "rc_gru_gate_r": (_I, [_P, _L, _P, _L, _P, _P, _L, _L, _L, _I, _P]),
    "rc_gru_update": (_I, [_P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _L, _L, _I, _P]),
    "rc_gru_update_bwd": (_I, [_P, _P, _L, _P, _P, _L, _P, _P, _P, _P, _P, _L, _L, _I, _P]),
    "rc_gru_gate_r_bwd": (_I, [_P, _P, _L, _P, _P, _L, _P, _P, _I, _L, _L, _I, _P]),
    "rc_pool2x": (_I, [_P, _P, _L, _L, _L, _L, _L, _I, _P]),
    "rc_interp_bilinear": (_I, [_P, _P, _L, _L, _L, _L, _L, _L, _L, _I, _P]),
    "rc_pool2x_bwd": (_I, [_P, _P, _L, _L, _L, _L, _I, _P]),
    "rc_interp_bilinear_bwd": (_I, [_P, _P, _L, _L, _L, _L, _L, _L, _I, _P]),
    "rc_bias_act": (_I, [_P, _P, _I, _L, _L, _L, _I, _I, _P]),
    "rc_upsample_flow_fwd": (_I, [_P, _L, _P, _P, _L, _I, _L, _L, _I, _I, _P]),
    "rc_upsample_flow_bwd_work_bytes": (_L, [_L, _I, _L, _L]),
    "rc_upsample_flow_bwd": (_I, [_P, _L, _P, _P, _P, _P, _L, _P, _L, _I, _L, _L, _I, _I, _P]),
    "rc_sequence_loss_work_bytes": (_L, [_I]),
    "rc_sequence_loss_fwd": (_I, [C.POINTER(_P), C.POINTER(C.c_float), _I, _L, _P, _P, C.c_float, _P, _P, _L, _L, _P]),
    "rc_sequence_loss_bwd": (_I, [C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_float), _I, _L, _L, _P, _P,
                                  C.c_float, _P, _P, _L, _L, _P]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C raft-stereo_amd/csrc` "
            "(or __graft_entry__.build()); there is no fallback path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


class RaftCorrError(RuntimeError):
    pass


def check(rc, what):
    """Turn a C-ABI return code into a Python exception (RuntimeError for launch/shape errors,
    like the TORCH_CHECKs of sampler/sampler.cpp:20-22)."""
    if rc != 0:
        msg = lib.rc_last_error_string().decode("utf-8", "replace")
        raise RaftCorrError(f"{what} failed (rc={rc}): {msg}")
