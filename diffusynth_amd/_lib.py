"""ctypes binding of libdiffusynth_hip.so (include/diffusynth_hip.h).

The parameter structs are generated from the header itself at import time, so the Python
mirror cannot drift from the C ABI.  There is no fallback: if the shared library is missing
(or a call returns a DS_E* code) a RuntimeError is raised.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_HEADER = os.path.join(os.path.dirname(_HERE), "include", "diffusynth_hip.h")
_LIBNAME = "libdiffusynth_hip.so"

DS_F32, DS_BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_SILU, ACT_RELU = 0, 1, 2, 3
TILE_128x192, TILE_256x96, TILE_128x32, TILE_64x192 = 0, 1, 2, 3
TILE_HALO3_256x96, TILE_QUAD_HALO3, TILE_HALO3_N16, TILE_INIT7 = 11, 12, 13, 14          # (4 .. 10: retired halo kernel generations)

_SCALARS = {"int32_t": C.c_int32, "int": C.c_int, "float": C.c_float, "double": C.c_double, "int64_t": C.c_int64,
            "uint64_t": C.c_uint64, "size_t": C.c_size_t}


def _parse_structs(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for body, name in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            if "*" in decl:
                names = decl.rsplit("*", 1)[1]
                ctype = C.c_void_p
            else:
                tname, names = decl.replace("const ", "").split(" ", 1)
                ctype = _SCALARS[tname]
            for n in names.split(","):
                fields.append((n.strip(), ctype))
        out[name] = type(name, (C.Structure,), {"_fields_": fields})
    return out


with open(_HEADER) as _f:
    _STRUCTS = _parse_structs(_f.read())
ConvParams = _STRUCTS["ds_conv_params"]
PackConvParams = _STRUCTS["ds_pack_conv_params"]
DwconvParams = _STRUCTS["ds_dwconv_params"]
GnApplyParams = _STRUCTS["ds_gn_apply_params"]
AttnParams = _STRUCTS["ds_attn_params"]
StepParams = _STRUCTS["ds_step_params"]
AttnFusedParams = _STRUCTS["ds_attn_fused_params"]
AttnX3Params = _STRUCTS["ds_attn_x3_params"]
VqAttnParams = _STRUCTS["ds_vq_attn_params"]

_P, _I, _F, _D, _SZ, _U64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t, C.c_uint64
_PROTOS = {  # name: (restype, argtypes); restype int => checked
    "ds_abi_version": (C.c_int, []),
    "ds_conv_igemm": (C.c_int, [C.POINTER(ConvParams), _P]),
    "ds_conv_splitk_reduce": (C.c_int, [C.POINTER(ConvParams), _P]),
    "ds_conv1x1_x3": (C.c_int, [C.POINTER(ConvParams), _P]),
    "ds_conv1x1_x3_stats_parts": (C.c_int, [C.POINTER(ConvParams)]),
    "ds_conv1x1_x3_weight_elems": (C.c_size_t, [_I, _I]),
    "ds_conv_stats_parts": (C.c_int, [C.POINTER(ConvParams)]),
    "ds_conv_tile_bn": (C.c_int, [_I]),
    "ds_pack_conv_weight": (C.c_int, [C.POINTER(PackConvParams), _P]),
    "ds_pack_conv_elems": (_SZ, [_I, _I, _I, _I, _I]),
    "ds_conv_fold_tables": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "ds_dwconv7": (C.c_int, [C.POINTER(DwconvParams), _P]),
    "ds_dwconv_stats_parts": (C.c_int, [C.POINTER(DwconvParams)]),
    "ds_pack_dw_weight": (C.c_int, [_P, _I, _P, _P]),
    "ds_pack_dw_weight_mfma": (C.c_int, [_P, _I, _P, _P]),
    "ds_split_planes": (C.c_int, [_P, _P, C.c_longlong, _I, _P]),
    "ds_gn_finalize": (C.c_int, [_P, _I, _I, _D, _F, _P, _P]),
    "ds_gn_stats": (C.c_int, [_P, _I, _I, _I, _I, _I, _F, _P, _P]),
    "ds_gn_stats_stream": (C.c_int, [_P, _I, _I, _I, _I, _I, _F, _P, _P, _P]),
    "ds_gn_stats_ws_floats": (_SZ, [_I, _I, _I]),
    "ds_gn_apply": (C.c_int, [C.POINTER(GnApplyParams), _P]),
    "ds_linattn_context": (C.c_int, [C.POINTER(AttnParams), _P]),
    "ds_linattn_output": (C.c_int, [C.POINTER(AttnParams), _P]),
    "ds_linattn_part_floats": (_SZ, [_I, _I, _I]),
    "ds_pack_attn_fused": (C.c_int, [_P, _P, _P, _P, _P, _I, _P]),
    "ds_attn_fused_context": (C.c_int, [C.POINTER(AttnFusedParams), _P]),
    "ds_attn_fused_output": (C.c_int, [C.POINTER(AttnFusedParams), _P]),
    "ds_attn_fused_stats_parts": (C.c_int, [C.POINTER(AttnFusedParams)]),
    "ds_attn_fused_segments": (C.c_int, [_I, _I, _I]),
    "ds_attn_fused_segments_gen": (C.c_int, [_I, _I, _I, _I]),
    "ds_pack_attn_x3": (C.c_int, [_P, _P, _P, _I, _P]),
    "ds_attn_x3_context": (C.c_int, [C.POINTER(AttnX3Params), _P]),
    "ds_attn_x3_output": (C.c_int, [C.POINTER(AttnX3Params), _P]),
    "ds_attn_x3_stats_parts": (C.c_int, [C.POINTER(AttnX3Params)]),
    "ds_attn_x3_segments": (C.c_int, [_I, _I, _I]),
    "ds_vq_attn_segments": (C.c_int, [_I, _I, _I]),
    "ds_vq_attn_wfold_bytes": (_SZ, [_I, _I]),
    "ds_vq_attn_context": (C.c_int, [C.POINTER(VqAttnParams), _P]),
    "ds_vq_attn_output": (C.c_int, [C.POINTER(VqAttnParams), _P]),
    "ds_attn_x3_qplane_bytes": (_SZ, [_I, _I]),
    "ds_attn_x3_mfold_bytes": (_SZ, [_I, _I]),
    "ds_sinusoid": (C.c_int, [_P, _P, _I, _I, _P, _P]),
    "ds_linear": (C.c_int, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _I, _P]),
    "ds_conv3x3_f32_n4_weight_floats": (C.c_size_t, [_I]),
    "ds_pack_conv3x3_f32_n4": (C.c_int, [_P, _P, _I, _I, _P, _P]),
    "ds_conv3x3_f32_n4": (C.c_int, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "ds_activation": (C.c_int, [_P, C.c_size_t, _I, _P, _P]),
    "ds_add_layernorm": (C.c_int, [_P, _P, _P, _P, _I, _I, _F, _P, _P]),
    "ds_dup_batch": (C.c_int, [_P, _P, _SZ, _P]),
    "ds_nchw_to_nhwc": (C.c_int, [_P, _I, _I, _I, _I, _P, _I, _I, _P]),
    "ds_conv1x1_in_nchw": (C.c_int, [_P, _I, _I, _I, _P, _P, _I, _P, _P]),
    "ds_nhwc_to_nchw": (C.c_int, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "ds_ddim_step": (C.c_int, [C.POINTER(StepParams), _P]),
    "ds_philox_normal": (C.c_int, [_P, _SZ, _U64, _U64, _P]),
    "ds_gather_cols": (C.c_int, [_P, _I, _I, _P, _I, _P, _P]),
    "ds_vq_nearest": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "ds_vq_stats_ws_bytes": (_SZ, [_I]),
    "ds_vq_stats": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _F, _I, _P, _P, _P]),
    "ds_decoder_tail": (C.c_int, [_P, _I, _I, _I, _I, _P, _P]),
    "ds_dec_final": (C.c_int, [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ds_conv7x7_c4_weight_elems": (C.c_size_t, []),
    "ds_pack_conv7x7_c4": (C.c_int, [_P, _I, _I, _P, _P]),
    "ds_conv7x7_c4": (C.c_int, [_P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "ds_pack_conv7x7_c4_x3": (C.c_int, [_P, _I, _I, _P, _P]),
    "ds_conv7x7_c4_x3": (C.c_int, [_P, _I, _I, _I, _P, _P, _P, _P]),
    "ds_conv3x3_c80_weight_elems": (C.c_size_t, []),
    "ds_pack_conv3x3_c80": (C.c_int, [_P, _I, _I, _P, _P]),
    "ds_conv3x3_c80": (C.c_int, [_P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _I, _I, _P, _P]),
    "ds_conv3x3_c80_stats_slots": (C.c_int, [_I, _I, _I]),
    "ds_conv3x3_c80_res": (C.c_int, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "ds_convt4x4_c80_stats_slots": (C.c_int, [_I, _I, _I, _I]),
    "ds_gn_stats_finish": (C.c_int, [_P, _I, _I, _I, _I, _I, _F, _P, _P]),
    "ds_convt4x4_c80_weight_elems": (C.c_size_t, [_I]),
    "ds_pack_convt4x4_c80": (C.c_int, [_P, _I, _I, _P, _P]),
    "ds_convt4x4_c80": (C.c_int, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "ds_istft_plus": (C.c_int, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "ds_istft_ws_floats": (_SZ, [_I, _I, _I]),
    "ds_stft_plus": (C.c_int, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "ds_bounds_report": (C.c_int, [C.c_char_p, _I, _I]),
}
_UNCHECKED = {"ds_conv3x3_f32_n4_weight_floats", "ds_bounds_report", "ds_abi_version", "ds_conv_stats_parts", "ds_conv1x1_x3_stats_parts", "ds_conv_tile_bn", "ds_dwconv_stats_parts", "ds_attn_fused_stats_parts", "ds_attn_fused_segments", "ds_attn_fused_segments_gen", "ds_attn_x3_stats_parts", "ds_attn_x3_segments", "ds_vq_attn_segments", "ds_conv3x3_c80_stats_slots", "ds_convt4x4_c80_stats_slots"}
EXPORTS = sorted(list(_PROTOS) + ["ds_last_error_string"])

_lib = None


def lib_path():
    return os.path.join(_HERE, os.environ.get("DS_LIB", _LIBNAME))   # DS_LIB: diagnostic (ablation) builds only


class DsError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise DsError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950).  diffusynth_amd has no CPU fallback.")
    lib = C.CDLL(path)
    lib.ds_last_error_string.restype = C.c_char_p
    lib.ds_last_error_string.argtypes = []
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().ds_last_error_string().decode("utf-8", "replace")
        raise DsError(f"{what} failed with code {rc}: {msg}")


def call(name, *args):
    """Call an int-returning entry point and raise on a DS_E* code."""
    rc = getattr(load(), name)(*args)
    if name not in _UNCHECKED:
        check(rc, name)
    return rc


def ptr(t):
    """Device (or host) address of a torch tensor as c_void_p-compatible int; None -> NULL."""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
