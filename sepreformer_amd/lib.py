"""ctypes binding of ``libsepr_hip.so`` (C ABI declared in ``include/sepr.h``).

The library is built in-tree by ``__graft_entry__.build()`` (``make -C sepreformer_amd/csrc``).  There is no
fallback of any kind: if the shared object is missing or an entry point fails, this module raises.
ctypes releases the GIL around every foreign call, so ``torch.nn.parallel.data_parallel`` replicas
(one Python thread per device, reference ``engine.py:64``) can drive the library concurrently.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# SEPR_LIB_VARIANT=<tag> selects an A/B build (make -C csrc variants); unset = the product library
_VARIANT = os.environ.get("SEPR_LIB_VARIANT", "")
LIB_PATH = os.path.join(_HERE, "_native", f"libsepr_hip_{_VARIANT}.so" if _VARIANT else "libsepr_hip.so")

SEPR_OK, SEPR_EINVAL, SEPR_EWORKSPACE, SEPR_EHIP = 0, -1, -2, -3
_ERR = {SEPR_EINVAL: "SEPR_EINVAL (bad shape / unsupported size / null pointer)",
        SEPR_EWORKSPACE: "SEPR_EWORKSPACE (workspace too small)",
        SEPR_EHIP: "SEPR_EHIP (HIP launch failed)"}

(OP_ENCODER, OP_GCFN, OP_CLA, OP_EGA, OP_SPKATTN, OP_SPKSPLIT, OP_OUTLAYER, OP_PIT) = range(8)
(SITE_NONE, SITE_GCFN_UP, SITE_GCFN_DOWN, SITE_CLA, SITE_ATTN_PROJ, SITE_EGA_GATE, SITE_SPLIT, SITE_FUSE,
 SITE_OUT, SITE_PROJECTOR, SITE_LINEAR, SITE_WGRAD, SITE_GCFN_BWD) = range(13)

_fp = C.c_void_p  # device pointers travel as plain addresses


class X3W(C.Structure):
    """sepr_x3_w: optional bf16x3 form of one projection (wp NULL = exact f32 core)."""
    _fields_ = [("wp", _fp), ("bias", _fp)]


def _struct(name, fields, x3=(), tail=()):
    return type(name, (C.Structure,), {"_fields_": [(f, _fp) for f in fields] + [(f, X3W) for f in x3] + [(f, _fp) for f in tail]})


GcfnW = _struct("GcfnW", ["ln_g", "ln_b", "w1", "b1", "dw_w", "dw_b", "w2", "b2", "ls"], ["x3_up", "x3_down"],
                ["fused_w1p", "fused_w2p"])
ClaW = _struct("ClaW", ["ln_g", "ln_b", "w1", "b1", "dw_w", "dw_b", "w2", "b2", "w3", "b3", "ls"], ["x3_1", "x3_2", "x3_3"],
               ["fused_w1p", "fused_w2p", "fused_w3p"])
MhaW = _struct("MhaW", ["ln_g", "ln_b", "wqkv", "bqkv", "wo", "bo", "ls"], ["x3_qkv", "x3_out"],
               ["fused_qkv_p", "fused_out_p"])


class EgaW(C.Structure):
    _fields_ = [("attn", MhaW), ("gate_ln_g", _fp), ("gate_ln_b", _fp), ("gate_w", _fp), ("gate_b", _fp),
                ("pe_k", _fp), ("maxlen", C.c_int), ("x3_gate", X3W), ("fused_gate_p", _fp), ("pe_k_planes", _fp), ("fused_qkv_p", _fp), ("fused_out_p", _fp)]


DownW = _struct("DownW", ["w", "scale", "shift"])
SplitW = _struct("SplitW", ["w1", "b1", "w2", "b2", "gn_g", "gn_b"], ["x3_1", "x3_2"], ["fused_w1p", "fused_w2p"])
FuseW = _struct("FuseW", ["w", "b"], ["x3"])
OutW = _struct("OutW", ["w1", "b1", "w2", "b2", "wdec"], ["x3_1", "x3_2"], ["fused_w1p", "fused_w2p", "fold_w2p", "fold_b"])

_i, _f, _sz, _ll = C.c_int, C.c_float, C.c_size_t, C.c_longlong
_u64, _d = C.c_ulonglong, C.c_double


# ---- training path (include/sepr.h, "Training path") ----------------------------------------------------------------------
class Lin(C.Structure):
    """sepr_lin: one projection, exact-f32 form (w) or packed-bf16 form (wp); b may be NULL; planes: 0 / 3 = bf16x3 split
    arithmetic, 1 = plain bf16 operands (the hi plane of the same fragments)."""
    _fields_ = [("w", _fp), ("wp", _fp), ("b", _fp), ("planes", C.c_int)]


def _tstruct(name, spec):
    """spec: list of field names; a name starting with '@' is a Lin, '#' an int, anything else a device pointer."""
    fields = []
    for f in spec:
        if f.startswith("@"):
            fields.append((f[1:], Lin))
        elif f.startswith("#"):
            fields.append((f[1:], C.c_int))
        else:
            fields.append((f, _fp))
    return type(name, (C.Structure,), {"_fields_": fields})


GcfnTW = _tstruct("GcfnTW", ["@up", "@up_t", "@down", "@down_t", "dw_w", "dw_b", "ls", "w1", "ln_g", "ln_b", "w2", "b2",
                              "fused_w1p", "fused_w2p", "seed_salt"])
GcfnGrad = _tstruct("GcfnGrad", ["ln_g", "ln_b", "w1", "b1", "dw_w", "dw_b", "w2", "b2", "ls"])
ClaTW = _tstruct("ClaTW", ["@l1", "@l1_t", "dw_w", "dw_wf", "dw_b", "zeros", "@l2", "@l2_t", "bn_g", "bn_b", "bn_rm", "bn_rv",
                           "@l3", "@l3_t", "ls", "w1", "ln_g", "ln_b", "w3", "b3", "seed_salt"])
ClaGrad = _tstruct("ClaGrad", ["ln_g", "ln_b", "w1", "b1", "dw_w", "dw_b", "w2", "b2", "bn_g", "bn_b", "w3", "b3", "ls"])
MhaTW = _tstruct("MhaTW", ["@qkv", "@qkv_t", "@out", "@out_t", "ls", "wqkv", "ln_g", "ln_b", "wo", "bo", "seed_salt"])
MhaGrad = _tstruct("MhaGrad", ["ln_g", "ln_b", "wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "ls"])


class AdamWTables(C.Structure):
    """include/sepr.h sepr_adamw_tables (device pointers)."""
    _fields_ = [("params", _fp), ("grad_off", _fp), ("state_off", _fp), ("numel", _fp), ("blocks", _fp), ("ntensors", C.c_int),
                ("nblocks", C.c_int)]


class EgaTW(C.Structure):
    _fields_ = [("attn", MhaTW), ("gate", Lin), ("gate_t", Lin), ("gate_w", _fp), ("gate_ln_g", _fp), ("gate_ln_b", _fp),
                ("pe_k", _fp), ("maxlen", C.c_int)]


class EgaGrad(C.Structure):
    _fields_ = [("attn", MhaGrad), ("gate_ln_g", _fp), ("gate_ln_b", _fp), ("gate_w", _fp), ("gate_b", _fp), ("pe_k", _fp)]


DownTW = _tstruct("DownTW", ["w", "b", "bn_g", "bn_b", "bn_rm", "bn_rv"])
DownGrad = _tstruct("DownGrad", ["w", "b", "bn_g", "bn_b"])
SplitTW = _tstruct("SplitTW", ["@l1", "@l1_t", "@l2", "@l2_t", "gn_g", "gn_b"])
SplitGrad = _tstruct("SplitGrad", ["w1", "b1", "w2", "b2", "gn_g", "gn_b"])
FuseTW = _tstruct("FuseTW", ["@l", "@l_t"])
FuseGrad = _tstruct("FuseGrad", ["w", "b"])
OutTW = _tstruct("OutTW", ["@l1", "@l1_t", "@l2", "@l2_t", "wdec"])
OutGrad = _tstruct("OutGrad", ["w1", "b1", "w2", "b2", "wdec"])
FrontTW = _tstruct("FrontTW", ["w_enc", "proj_w", "gn_g", "gn_b", "@proj_t", "ones"])
FrontGrad = _tstruct("FrontGrad", ["w_enc", "gn_g", "gn_b", "proj_w"])
(TOP_GCFN, TOP_CLA, TOP_EGA, TOP_SPKATTN, TOP_DOWN, TOP_SPLIT, TOP_FUSE, TOP_OUT, TOP_FRONT, TOP_GCFN_FUSED, TOP_EGA_X3,
 TOP_GCFN_FUSED16) = range(12)

KNOB_X3_WIDE, KNOB_TRAIN_GCFN_PLANES, KNOB_TRAIN_ATTN_ONE, KNOB_TRAIN_CLA16, KNOB_FOLD_HEAD, KNOB_TN16 = range(6)

# name -> (restype, argtypes); must list every symbol include/sepr.h declares (tests check this)
SIGNATURES = {
    "sepr_version": (_i, []),
    "sepr_build_info": (C.c_char_p, []),
    "sepr_knob": (_i, [_i]),
    "sepr_knobs_reload": (None, []),
    "sepr_last_hip_error": (C.c_char_p, []),
    "sepr_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "sepr_encoder_fwd": (_i, [_fp, _i, _i, _fp, _i, _i, _i, _f, _fp, _fp, _fp, _sz, _fp]),
    "sepr_projector_fwd": (_i, [_fp, _i, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp, _fp]),
    "sepr_gcfn_fwd": (_i, [_fp, _fp, _i, _i, _i, C.POINTER(GcfnW), _fp, _sz, _fp]),
    "sepr_cla_fwd": (_i, [_fp, _fp, _i, _i, _i, _i, C.POINTER(ClaW), _fp, _sz, _fp]),
    "sepr_ega_fwd": (_i, [_fp, _fp, _i, _i, _i, _i, _i, C.POINTER(EgaW), _fp, _sz, _fp]),
    "sepr_spkattn_fwd": (_i, [_fp, _fp, _i, _i, _i, _i, _i, C.POINTER(MhaW), _fp, _sz, _fp]),
    "sepr_gcfn_fwd_st": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, C.POINTER(GcfnW), _fp, _sz, _fp]),
    "sepr_cla_fwd_st": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, C.POINTER(ClaW), _fp, _sz, _fp]),
    "sepr_ega_fwd_st": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, C.POINTER(EgaW), _fp, _sz, _fp]),
    "sepr_spkattn_fwd_st": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, C.POINTER(MhaW), _fp, _sz, _fp]),
    "sepr_downconv_fwd": (_i, [_fp, _fp, _i, _i, _i, _i, C.POINTER(DownW), _fp]),
    "sepr_spksplit_fwd": (_i, [_fp, _fp, _i, _i, _i, _i, _f, C.POINTER(SplitW), _fp, _sz, _fp]),
    "sepr_fuse_fwd": (_i, [_fp, _fp, _fp, _i, _i, _i, C.POINTER(FuseW), _fp]),
    "sepr_outlayer_decoder_fwd": (_i, [_fp, _i, _i, _i, _i, _fp, _fp, _i, _i, _i, _i, C.POINTER(OutW), _fp, _fp, _sz, _fp]),
    "sepr_groupnorm_stats": (_i, [_fp, _i, _ll, _f, _fp, _fp, _sz, _fp]),
    "sepr_linear_fwd": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _fp]),
    "sepr_linear_x3_fwd": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _fp]),
    "sepr_pit_sisnr_mag_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "sepr_pit_sisnr_mag_fwd": (_i, [_fp, _fp, _i, _i, _i, _fp, _i, _i, C.c_double, _fp, _fp, _fp, _sz, _fp]),
    "sepr_pit_sisnr_fwd": (_i, [_fp, _fp, _fp, _i, _i, _i, C.c_double, C.c_double, C.c_double, _fp, _fp, _fp, _fp,
                                _fp, _sz, _fp]),
    "sepr_train_ctx_bytes": (_sz, [_i] * 8),
    "sepr_train_ws_bytes": (_sz, [_i] * 9),
    "sepr_gcfn_train_fwd": (_i, [_fp, _fp, _i, _i, _i, C.POINTER(GcfnTW), _fp, _sz, _fp, _sz, _f, _u64, _fp]),
    "sepr_gcfn_bwd": (_i, [_fp, _fp, _fp, _i, _i, _i, C.POINTER(GcfnTW), C.POINTER(GcfnGrad), _fp, _sz, _fp, _sz, _f, _u64, _fp]),
    "sepr_cla_train_fwd": (_i, [_fp, _fp, _i, _i, _i, _i, C.POINTER(ClaTW), _fp, _sz, _fp, _sz, _f, _u64, _fp]),
    "sepr_cla_bwd": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, C.POINTER(ClaTW), C.POINTER(ClaGrad), _fp, _sz, _fp, _sz, _f, _u64, _fp]),
    "sepr_ega_train_fwd": (_i, [_fp, _fp, _i, _i, _i, _i, _i, C.POINTER(EgaTW), _fp, _sz, _fp, _sz, _f, _u64, _fp]),
    "sepr_ega_bwd": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, C.POINTER(EgaTW), C.POINTER(EgaGrad), _fp, _sz, _fp, _sz, _f, _u64, _fp]),
    "sepr_spkattn_train_fwd": (_i, [_fp, _fp, _i, _i, _i, _i, _i, C.POINTER(MhaTW), _fp, _sz, _fp, _sz, _f, _u64, _fp]),
    "sepr_spkattn_bwd": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, C.POINTER(MhaTW), C.POINTER(MhaGrad), _fp, _sz, _fp, _sz, _f, _u64, _fp]),
    "sepr_downconv_train_fwd": (_i, [_fp, _fp, _i, _i, _i, _i, C.POINTER(DownTW), _fp, _sz, _fp, _sz, _fp]),
    "sepr_downconv_bwd": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, C.POINTER(DownTW), C.POINTER(DownGrad), _fp, _sz, _fp, _sz, _fp]),
    "sepr_spksplit_train_fwd": (_i, [_fp, _fp, _i, _i, _i, _i, _f, C.POINTER(SplitTW), _fp, _sz, _fp, _sz, _fp]),
    "sepr_spksplit_bwd": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, C.POINTER(SplitTW), C.POINTER(SplitGrad), _fp, _sz, _fp, _sz, _fp]),
    "sepr_fuse_bwd": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, C.POINTER(FuseTW), C.POINTER(FuseGrad), _fp, _sz, _fp]),
    "sepr_outlayer_decoder_train_fwd": (_i, [_fp, _i, _i, _i, _i, _fp, _fp, _i, _i, _i, _i, C.POINTER(OutTW), _fp, _fp, _sz, _fp, _sz, _fp]),
    "sepr_outlayer_decoder_bwd": (_i, [_fp, _fp, _fp, _i, _fp, _i, _i, _i, _i, _fp, _fp, _fp, _i, _i, _i, _i, C.POINTER(OutTW),
                                       C.POINTER(OutGrad), _fp, _sz, _fp, _sz, _fp]),
    "sepr_front_train_fwd": (_i, [_fp, _i, _i, _i, _i, _i, _i, _i, _f, C.POINTER(FrontTW), _fp, _fp, _fp, _sz, _fp, _sz, _fp]),
    "sepr_front_bwd": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, C.POINTER(FrontTW), C.POINTER(FrontGrad), _fp, _sz, _fp, _sz, _fp]),
    "sepr_linear_wgrad_workspace": (_sz, [_i, _i, _i]),
    "sepr_linear_wgrad": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _sz, _fp]),
    "sepr_linear_wgrad_norm": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _sz, _fp]),
    "sepr_linear_wgrad_bf16": (_i, [_fp, _i, _fp, _i, _fp, _fp, _i, _i, _i, _i, _fp, _sz, _fp]),
    "sepr_train_pack_lin": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _i, _i, _fp, _fp]),
    "sepr_train_fold_bias": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _fp, _fp]),
    "sepr_train_defer_begin": (_i, [_fp, _sz]),
    "sepr_train_defer_flush": (_i, [_i, _fp]),
    "sepr_train_defer_parts": (_i, [_fp, _sz]),
    "sepr_train_wgrad_stream": (_i, [_fp]),
    "sepr_train_wgrad_join": (_i, [_fp]),
    "sepr_train_wgrad_mark": (_i, [_i]),
    "sepr_train_wgrad_wait": (_i, [_i, _fp]),
    "sepr_train_pack_gcfn_fused": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _fp, _fp, _fp]),
    "sepr_pit_sisnr_bwd": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _d, _d, _fp, _fp, _sz, _fp]),
    "sepr_pit_sisnr_mag_bwd_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "sepr_pit_sisnr_mag_bwd": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _fp, _fp, _i, _i, _d, _fp, _fp, _sz, _fp]),
    "sepr_adamw_block_elems": (_i, []),
    "sepr_adamw_workspace": (_sz, []),
    "sepr_adamw_step": (_i, [C.POINTER(AdamWTables), _fp, _ll, _fp, _fp, _fp, _fp, _d, _d, _d, _d, _d, _fp, _fp, _sz, _fp]),
    "sepr_prof_start": (_i, [_i, _i]),
    "sepr_prof_stop": (_i, [C.POINTER(_ll), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "sepr_prof_last_bytes": (C.c_double, []),
}

_lib: Optional[C.CDLL] = None
_lock = threading.Lock()


ABI_VERSION = 412          # include/sepr.h SEPR_VERSION this binding mirrors (tests/test_boundary_cpu.py keeps the two equal)


class SeprLibraryError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen the library once and type every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise SeprLibraryError(
                f"{LIB_PATH} is missing: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C sepreformer_amd/csrc`). "
                "There is no CPU fallback for the separator path.")
        # torch bundles its own HIP runtime (torch/lib/libamdhip64.so).  It must be the one already in
        # the process when libsepr_hip.so is dlopen'ed, otherwise the loader binds this library to
        # /opt/rocm's copy and the two runtimes do not share devices or streams
        # ("no ROCm-capable device is detected" at the first launch).
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError -> a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        got = int(lib.sepr_version())
        if got != ABI_VERSION:      # the ctypes struct mirrors below are written against ONE header version: a stale .so would read short structs
            raise SeprLibraryError(f"{LIB_PATH} reports ABI {got}, this binding is written against include/sepr.h SEPR_VERSION {ABI_VERSION}: "
                                   "rebuild the library (`make -C sepreformer_amd/csrc`)")
        _lib = lib
        return lib


def check(rc: int, what: str) -> None:
    if rc == SEPR_OK:
        return
    msg = _ERR.get(rc, f"status {rc}")
    if rc == SEPR_EHIP:
        msg += ": " + load().sepr_last_hip_error().decode(errors="replace")
    raise RuntimeError(f"{what} failed: {msg}")


def version() -> int:
    return int(load().sepr_version())


def build_info() -> str:
    return load().sepr_build_info().decode()
