"""ctypes binding of libdrba_hip.so (C ABI declared in include/drba_hip.h).

There is exactly one compute backend.  If the library is missing or a call returns an
error code this module raises — nothing in drba_amd falls back to torch or CPU math.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdrba_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "drba_hip.h")


def _header_abi_version():
    """DRBA_ABI_VERSION of include/drba_hip.h: the header is the one place the number is written down (the library
    returns the macro, the entry point and the tests compare with this)."""
    import re
    if not os.path.exists(HEADER_PATH):  # the package used without the repo's include/ tree (copied / installed): the
        return None                       # library describes itself, load() takes its number (with a warning)
    with open(HEADER_PATH) as f:
        m = re.search(r"^#define\s+DRBA_ABI_VERSION\s+(\d+)", f.read(), flags=re.M)
    if m is None:
        raise RuntimeError(f"{HEADER_PATH} does not define DRBA_ABI_VERSION")
    return int(m.group(1))


ABI_VERSION = _header_abi_version()

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_d = C.c_double
_z = C.c_size_t

# name -> (restype, argtypes); mirrors include/drba_hip.h one to one
class ConvLayer(C.Structure):
    """drba_conv_layer_t (include/drba_hip.h)."""
    _fields_ = [("packed_w", C.c_void_p), ("bias", C.c_void_p), ("beta", C.c_void_p), ("cin", C.c_int), ("cout", C.c_int),
                ("stride", C.c_int), ("act", C.c_int), ("cfg", C.c_int), ("residual", C.c_int), ("deconv", C.c_int),
                ("pixel_shuffle", C.c_int)]


SIGNATURES = {
    "drba_abi_version": (_i, []),
    "drba_rife_splat_ws_floats": (_z, [_i, _i, _i, _i]),
    "drba_set_range_check": (_i, [_i]),
    "drba_status_word": (_i, [C.POINTER(C.c_void_p)]),
    "drba_status_clear": (_i, []),
    "drba_stream_create_cu_mask": (_i, [C.POINTER(C.c_uint32), _i, C.POINTER(C.c_void_p)]),
    "drba_stream_destroy": (_i, [_p]),
    "drba_conv_state_reset": (_i, [_p]),
    "drba_trace_begin": (_i, []),
    "drba_trace_end": (_i, []),
    "drba_trace_resume": (_i, []),
    "drba_trace_count": (_i, []),
    "drba_trace_get": (_i, [_i, C.POINTER(C.c_char_p), C.POINTER(C.c_uint), C.POINTER(C.c_float)]),
    "drba_trace_get_start": (_i, [_i, C.POINTER(C.c_float), C.POINTER(C.c_ulonglong)]),
    "drba_error_string": (C.c_char_p, [_i]),
    "drba_softsplat": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "drba_softsplat_ws_floats": (_z, [_i, _i, _i, _i]),
    "drba_softsplat_again": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "drba_quad_interleave": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "drba_softsplat_index": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "drba_softsplat_gather_quad": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "drba_backwarp": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "drba_flow_distance": (_i, [_p, _p, _i, _i, _i, _p]),
    "drba_flow_reverse": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "drba_drm_rife_linear": (_i, [_p, _p, _f, _p, _f, _p, _p, _i, _i, _i, _p]),
    "drba_drm_rife_linear_batch": (_i, [_p, _i, _f, _p, _i, _i, _p]),
    "drba_drm_ratio": (_i, [_p, _p, _f, _p, _p, _i, _i, _i, _p]),
    "drba_affine": (_i, [_p, _f, _f, _p, _z, _p]),
    "drba_mul_map": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "drba_fill_holes": (_i, [_p, _p, _p, _p, _z, _p]),
    "drba_drm_retime": (_i, [_p, _p, _d, _d, _z, _p]),
    "drba_resize_bilinear": (_i, [_p, _p, _i, _i, _i, _i, _i, _f, _f, _p]),
    "drba_u8hwc_to_f32nchw": (_i, [_p, _p, _i, _i, _p]),
    "drba_f32nchw_to_u8hwc": (_i, [_p, _p, _i, _i, _p]),
    "drba_to_inp": (_i, [_p, _p, _i, _i, _i, _i, _f, _f, _p]),
    "drba_to_inp_x4": (_i, [_p, _p, _p, _i, _i, _i, _i, _f, _f, _p]),
    "drba_to_out": (_i, [_p, _p, _i, _i, _i, _i, _f, _f, _i, _p]),
    "drba_ssim3d_32": (_i, [_p, _p, _p, _p]),
    "drba_conv3x3_pick_cfg": (_i, [_i, _i, _i, _i, _i]),
    "drba_conv3x3_num_cfgs": (_i, []),
    "drba_conv3x3_cfg_stride": (_i, [_i]),
    "drba_conv3x3_cfg_family": (_i, [_i]),
    "drba_deconv4x4_cfg_family": (_i, [_i]),
    "drba_deconv4x4_num_cfgs": (_i, []),
    "drba_conv3x3_packed_floats": (_z, [_i, _i, _i]),
    "drba_conv3x3_pack": (_i, [_p, _p, _i, _i, _i]),
    "drba_conv3x3": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _i, _f, _i, _p]),
    "drba_conv3x3_shuffle": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "drba_deconv4x4_pick_cfg": (_i, [_i, _i, _i, _i]),
    "drba_deconv4x4_packed_floats": (_z, [_i, _i, _i]),
    "drba_deconv4x4_pack": (_i, [_p, _p, _i, _i, _i]),
    "drba_conv_chain": (_i, [_p, _p, _p, _p, C.POINTER(ConvLayer), _i, _i, _i, _i, _p]),
    "drba_deconv4x4s2": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "drba_ifblock_input": (_i, [_p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _i, _i, _f, _p, _i, _i, _i, _i, _f, _p]),
    "drba_ifblock_input_lds": (_i, [_p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _i, _i, _f, _p, _p, _i, _i, _i, _i, _f, _p]),
    "drba_ifblock_input_batch": (_i, [_p, _i, _i, _i, _f, _i, _i, _i, _i, _f, _p]),
    "drba_ifblock_input_lds_batch": (_i, [_p, _i, _i, _i, _f, _i, _i, _i, _i, _f, _p]),
    "drba_stage_conv0_packed_floats": (_z, []),
    "drba_stage_conv0_pack": (_i, [_p, _p]),
    "drba_stage_conv0_supported": (_i, [_i, _i, _f, _f, _i]),
    "drba_stage_conv0_batch": (_i, [_p, _i, _p, _i, _i, _f, _i, _i, _p, _p, _p]),
    "drba_rgbx": (_i, [_p, _p, _i, _i, _p]),
    "drba_stage_conv16_packed_floats": (_z, [_i]),
    "drba_stage_conv16_pack": (_i, [_p, _i, _p]),
    "drba_stage_conv16_supported": (_i, [_i, _i, _f, _f, _i]),
    "drba_stage_conv16_batch": (_i, [_p, _i, _p, _i, _i, _f, _i, _i, _f, _i, _p, _p, _p]),
    "drba_head_fused_packed_floats": (_z, []),
    "drba_head_fused_pack": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "drba_head_fused": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "drba_head_fused16_packed_floats": (_z, []),
    "drba_head_fused16_pack": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "drba_head_fused16": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "drba_ifblock_input_lazy_batch": (_i, [_p, _i, _p, _i, _i, _f, _i, _i, _i, _i, _f, _p]),
    "drba_warp_blend_lazy_batch": (_i, [_p, _i, _p, _i, _i, _f, _i, _i, _p]),
    "drba_ifblock_update_batch": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "drba_warp_blend_fold": (_i, [_p, _p, _p, _p, _i, _i, _f, _p, _i, _i, _p]),
    "drba_pair_interleave": (_i, [_p, _p, _i, _i, _i, _p]),
    "drba_ifblock_update": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "drba_metric_input": (_i, [_p, _p, _p, _p, _p, _i, _i, _p]),
    "drba_pixel_shuffle2": (_i, [_p, _p, _i, _i, _i, _p]),
    "drba_timestep_fix": (_i, [_p, _p, _p, _p, _p, _p, _z, _p]),
    "drba_swap_select": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p]),
    "drba_clamp": (_i, [_p, _p, _f, _f, _z, _p]),
    "drba_conv_direct": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "drba_instance_norm": (_i, [_p, _p, _p, _i, _z, _f, _i, _p]),
    "drba_instance_norm_ws_floats": (_z, [_i]),
    "drba_add_act": (_i, [_p, _p, _p, _z, _i, _p]),
    "drba_channel_normalize3": (_i, [_p, _p, _i, _z, _p, _p, _p]),
    "drba_layernorm": (_i, [_p, _p, _p, _p, _p, _z, _i, _f, _p]),
    "drba_gelu": (_i, [_p, _p, _z, _p]),
    "drba_window_attention": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _p, _i, _p]),
    "drba_window_attention_ws_floats": (_z, [_i, _i, _i, _i]),
    "drba_linear_split_packed_floats": (_z, [_i, _i, _i]),
    "drba_linear_split_pack": (_i, [_p, _p, _i, _i, _i]),
    "drba_linear_split": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "drba_linear_split_cat": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "drba_linear_split_layernorm": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p]),
    "drba_softmax_rows": (_i, [_p, _p, _z, _i, _i, _i, _f, _p]),
    "drba_global_expect2": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _i, _p]),
    "drba_global_expect2_ws_floats": (_z, [_i]),
    "drba_bmm": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "drba_local_corr_flow": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "drba_local_attn_flow": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "drba_convex_upsample": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "drba_flow_warp": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "drba_resize_bilinear_ac": (_i, [_p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "drba_warp_blend": (_i, [_p, _p, _p, _p, _i, _i, _f, _p, _i, _i, _p]),
}

class StageItem(C.Structure):
    """include/drba_hip.h: drba_stage_item_t"""
    _fields_ = [("img0", C.c_void_p), ("img1", C.c_void_p), ("f0", C.c_void_p), ("f1", C.c_void_p), ("f0_pair", C.c_void_p),
                ("f1_pair", C.c_void_p), ("timestep_map", C.c_void_p), ("timestep_scalar", C.c_float), ("flow", C.c_void_p),
                ("tmp_prev", C.c_void_p), ("flow_out", C.c_void_p), ("out", C.c_void_p), ("term", C.c_void_p * 4),
                ("img0_x4", C.c_void_p), ("img1_x4", C.c_void_p)]


class DrmJob(C.Structure):
    """include/drba_hip.h: drba_drm_job_t"""
    _fields_ = [("flow_self", C.c_void_p), ("flow_other", C.c_void_p), ("t", C.c_float), ("out", C.c_void_p)]


MAX_FLOW_TERMS = 4  # DRBA_MAX_FLOW_TERMS


class FlowTerms(C.Structure):
    """include/drba_hip.h: drba_flow_terms_t"""
    _fields_ = [("n", C.c_int), ("h", C.c_int * 4), ("w", C.c_int * 4), ("scale", C.c_float * 4)]


MAX_STAGE_ITEMS = 8  # DRBA_MAX_STAGE_ITEMS

_lib = None


class DrbaHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library with all prototypes set."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel bundles its own libamdhip64 / libhsa-runtime64, the library links the image's /opt/rocm copies
    # under the same sonames.  Whichever is mapped first serves both; with the library's copy mapped first torch's streams
    # and allocations belong to a runtime it was not built against and every launch fails (hipGetLastError after the
    # launch: __graft_entry__.build() followed by smoke() in one process did exactly that).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise DrbaHipError(
            f"{LIB_PATH} not found: the HIP kernel library is not built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C drba_amd/csrc`). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    got = lib.drba_abi_version()
    global ABI_VERSION
    if ABI_VERSION is None:
        import warnings
        warnings.warn(f"{HEADER_PATH} not found: taking ABI version {got} from {LIB_PATH} unchecked")
        ABI_VERSION = got
    if got != ABI_VERSION:
        raise DrbaHipError(f"{LIB_PATH} reports ABI version {got}, include/drba_hip.h declares {ABI_VERSION}: "
                           "stale build, run `make -C drba_amd/csrc`")
    if os.environ.get("DRBA_CHECK_RANGE", "0") not in ("", "0"):  # debug: family-4 outputs are scanned for inf / NaN (drba_hip.h)
        lib.drba_set_range_check(1)
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().drba_error_string(code).decode()
        raise DrbaHipError(f"{what} failed: {msg} ({code})")
