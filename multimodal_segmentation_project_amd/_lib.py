"""ctypes binding of libmi3d.so (include/mi3d.h).  Plumbing only: raw pointers in, error codes -> RuntimeError.

The product path has NO fallback: if the HIP library is missing or a tensor is not on the GPU the call raises.
"""
import contextlib
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI3D_LIB_PATH") or os.path.join(_HERE, "libmi3d.so")   # override: A/B-test another build
MAX_LEVELS = 6
LOSS_COEF_FLOATS = 20
DTYPE_F32, DTYPE_BF16 = 0, 1

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t


class UNetDesc(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("out_channels", C.c_int32), ("n_levels", C.c_int32),
                ("features", C.c_int32 * MAX_LEVELS), ("N", C.c_int32), ("D", C.c_int32), ("H", C.c_int32),
                ("W", C.c_int32), ("dtype", C.c_int32), ("bn_momentum", C.c_float), ("bn_eps", C.c_float),
                ("prepacked_from", C.c_int32)]


class LossCfg(C.Structure):
    _fields_ = [("w_ce", C.c_float), ("region_kind", C.c_int32), ("w_reg", C.c_float), ("alpha", C.c_float),
                ("beta", C.c_float), ("eps", C.c_float), ("w_kd", C.c_float), ("temperature", C.c_float)]


AUG_MAX_COEFF, AUG_MAX_CP, AUG_MAX_HOLES = 20, 16, 8


class AugParams(C.Structure):                      # mi3d_aug_params
    _fields_ = [("do_bias", C.c_int32), ("bias_degree", C.c_int32), ("bias_coeff", C.c_double * AUG_MAX_COEFF),
                ("do_noise", C.c_int32), ("noise_mean", C.c_float), ("noise_std", C.c_float), ("noise_seed", C.c_uint64),
                ("do_contrast", C.c_int32), ("gamma", C.c_float), ("do_hist", C.c_int32), ("n_cp", C.c_int32),
                ("ref_cp", C.c_float * AUG_MAX_CP), ("flt_cp", C.c_float * AUG_MAX_CP), ("n_holes", C.c_int32),
                ("hole_size", C.c_int32 * 3), ("hole_lo", (C.c_int32 * 3) * AUG_MAX_HOLES), ("fill_value", C.c_float)]


_DP, _LP, _AP = C.POINTER(UNetDesc), C.POINTER(LossCfg), C.POINTER(AugParams)

# name -> (restype, argtypes); one line per symbol declared in include/mi3d.h
_SIGS = {
    "mi3d_last_error": (C.c_char_p, []),
    "mi3d_abi_version": (i32, []),
    "mi3d_unet_num_params": (i32, [_DP]),
    "mi3d_unet_num_buffers": (i32, [_DP]),
    "mi3d_unet_num_segments": (i32, [_DP]),
    "mi3d_unet_workspace_bytes": (sz, [_DP]),
    "mi3d_unet_dropout_count": (i64, [_DP]),
    "mi3d_unet_segment_params": (i32, [_DP, i32, C.POINTER(C.c_int)]),
    "mi3d_unet_forward": (i32, [_DP, vp, vp, vp, vp, i32, vp, vp, vp, sz, vp]),
    "mi3d_unet_infer": (i32, [_DP, vp, vp, vp, vp, vp, vp, sz, vp]),
    "mi3d_unet_bn_apply_deferred": (i32, [_DP, vp, vp, vp]),
    "mi3d_unet_backward": (i32, [_DP, vp, vp, vp, vp, vp, vp, f32, i32, i32, i32, vp, sz, vp, vp, vp, i32]),
    "mi3d_unet_chain_tail_blocks": (i32, [_DP]),
    "mi3d_unet_pack_from": (i32, [_DP, C.POINTER(vp), vp, sz, i32, vp]),
    "mi3d_unet_backward_marks": (i32, [C.POINTER(C.c_int), C.POINTER(vp), i32]),
    "mi3d_stream_wait_event": (i32, [vp, vp]),
    "mi3d_flag_set": (i32, [vp, C.c_int64, vp]),
    "mi3d_flag_wait": (i32, [vp, C.c_int64, C.c_int64, vp]),
    "mi3d_event_create": (i32, [C.POINTER(vp)]),
    "mi3d_event_destroy": (i32, [vp]),
    "mi3d_stream_create": (i32, [i32, C.POINTER(vp)]),
    "mi3d_stream_create_masked": (i32, [i32, i32, C.POINTER(vp)]),
    "mi3d_stream_destroy": (i32, [vp]),
    "mi3d_debug_set_route": (i32, [C.c_char_p, i32]),
    "mi3d_debug_get_route": (i32, [C.c_char_p, C.POINTER(C.c_int)]),
    "mi3d_debug_route_count": (i32, []),
    "mi3d_debug_experiments": (i32, []),
    "mi3d_debug_route_name": (C.c_char_p, [i32]),
    "mi3d_debug_occupy_cus": (i32, [i32, i32, vp, i64, vp]),
    "mi3d_set_cu_budget": (i32, [i32]),
    "mi3d_timing_event_create": (i32, [C.POINTER(vp)]),
    "mi3d_time_next_conv3_kernel": (i32, [vp, vp, i32, i32, i32]),
    "mi3d_time_hook_fired": (i32, []),
    "mi3d_event_elapsed_ms": (i32, [vp, vp, C.POINTER(C.c_float)]),
    "mi3d_seg_loss_workspace_bytes": (sz, [i32]),
    "mi3d_seg_loss_forward": (i32, [vp, vp, vp, i32, i32, i64, _LP, vp, vp, vp, vp]),
    "mi3d_seg_loss_backward": (i32, [vp, vp, vp, i32, i32, i64, _LP, vp, vp, vp, vp]),
    "mi3d_seg_loss_metrics_forward": (i32, [vp, vp, vp, i32, i32, i32, i64, _LP, vp, vp, vp, vp, vp, vp]),
    "mi3d_unet_head_loss_supported": (i32, [_DP, _LP]),
    "mi3d_head_loss_supported": (i32, [i32, i32, i32, _LP]),
    "mi3d_head_loss_forward": (i32, [vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, i64, _LP, vp, vp, vp, vp, vp, vp, vp]),
    "mi3d_head_loss_backward": (i32, [vp, i32, i32, vp, vp, vp, vp, i32, i32, i64, _LP, vp, vp, vp, i32, vp, vp, i32, vp, sz, vp]),
    "mi3d_unet_forward_loss": (i32, [_DP, vp, vp, vp, vp, i32, vp, vp, _LP, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "mi3d_unet_head_loss_forward": (i32, [_DP, vp, vp, vp, _LP, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "mi3d_unet_backward_loss": (i32, [_DP, vp, vp, vp, vp, vp, vp, _LP, vp, vp, vp, f32, i32, i32, i32, vp, sz, vp, vp, vp, i32]),
    "mi3d_seg_metrics_workspace_bytes": (sz, [i32]),
    "mi3d_seg_metrics": (i32, [vp, vp, i32, i32, i32, i64, vp, vp, vp]),
    "mi3d_seg_class_counts": (i32, [vp, vp, i32, i32, i64, vp, vp, vp]),
    "mi3d_linear_forward": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "mi3d_linear_backward": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, f32, vp, vp]),
    "mi3d_scale": (i32, [vp, vp, i64, f32, vp, vp]),
    "mi3d_softmax_ce_rows": (i32, [vp, vp, i32, i32, vp, vp, f32, vp]),
    "mi3d_adamw_step": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, vp, vp]),
    "mi3d_adamw_apply": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, vp, i32, vp]),
    "mi3d_dropout_scales": (i32, [vp, i64, f32, vp, vp]),
    "mi3d_preprocess_ct": (i32, [vp, vp, i64, f32, f32, vp]),
    "mi3d_preprocess_mri_workspace_bytes": (sz, []),
    "mi3d_preprocess_mri": (i32, [vp, vp, i64, f32, f32, vp, vp]),
    "mi3d_remap_labels": (i32, [vp, vp, i64, i32, vp]),
    "mi3d_augment_workspace_bytes": (sz, []),
    "mi3d_augment": (i32, [vp, vp, vp, i32, i32, i32, i32, _AP, vp, sz, vp]),
    "mi3d_fill_boxes_i64": (i32, [vp, i32, i32, i32, i32, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), i64, vp]),
    "mi3d_conv3_workspace_bytes": (sz, [i32, i32, i32, i32, i32, i32]),
    "mi3d_conv3_forward": (i32, [i32, i32, vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "mi3d_conv3_backward": (i32, [i32, i32, vp, i32, i32, vp, vp, i32, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32,
                                  vp, sz, vp]),
    "mi3d_bn_workspace_bytes": (sz, [i32]),
    "mi3d_bn_relu_drop_forward": (i32, [i32, vp, i32, i32, i64, i64, vp, vp, vp, vp, vp, f32, f32, i32, vp, vp, i32,
                                        vp, vp, vp]),
    "mi3d_bn_relu_drop_backward": (i32, [i32, vp, i32, vp, i32, i32, i64, i64, vp, vp, vp, i32, vp, vp, i32, vp, vp]),
    "mi3d_bn_relu_drop_pool_forward": (i32, [i32, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, f32, f32, vp, vp, i32,
                                             vp, i32, vp, vp, vp]),
    "mi3d_conv1_workspace_bytes": (sz, [i32, i32]),
    "mi3d_conv1_forward": (i32, [i32, vp, i32, i32, vp, vp, vp, i32, i32, i64, vp]),
    "mi3d_conv1_backward": (i32, [i32, vp, i32, i32, vp, vp, i32, vp, i32, vp, vp, i32, i32, i64, vp, sz, vp]),
    "mi3d_maxpool2_forward": (i32, [i32, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp]),
    "mi3d_maxpool2_backward": (i32, [i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mi3d_upconv2_workspace_bytes": (sz, [i32, i32, i32, i32, i32, i32]),
    "mi3d_upconv2_forward": (i32, [i32, vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "mi3d_upconv2_backward": (i32, [i32, vp, i32, i32, vp, vp, i32, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp,
                                    sz, vp]),
    "mi3d_ncdhw_to_ndhwc": (i32, [i32, vp, vp, i32, i32, i32, i64, vp]),
    "mi3d_ndhwc_to_ncdhw": (i32, [i32, vp, i32, vp, i32, i32, i64, vp]),
    "mi3d_graph_begin": (i32, [vp]),
    "mi3d_graph_end": (i32, [vp, C.POINTER(vp)]),
    "mi3d_graph_launch": (i32, [vp, vp]),
    "mi3d_graph_destroy": (i32, [vp]),
}

_lib = None


class Mi3dError(RuntimeError):
    pass


def lib():
    """Load libmi3d.so (built by __graft_entry__.build() / csrc/Makefile).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Mi3dError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)       # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().mi3d_last_error()
        raise Mi3dError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


launches = 0      # bumped by every library call: lets caches notice device buffers rewritten behind torch's back


def call(name, *args):
    global launches
    launches += 1
    check(getattr(lib(), name)(*args), name)


def get_route(name):
    """Current value of a kernel-selection switch (include/mi3d.h mi3d_debug_get_route; the table is in INTEGRATION.md)."""
    v = C.c_int()
    check(lib().mi3d_debug_get_route(name.encode(), C.byref(v)), "mi3d_debug_get_route")
    return v.value


def set_route(name, value):
    check(lib().mi3d_debug_set_route(name.encode(), int(value)), "mi3d_debug_set_route")


def route_names():
    l = lib()
    return [l.mi3d_debug_route_name(i).decode() for i in range(l.mi3d_debug_route_count())]


@contextlib.contextmanager
def routes(**kw):
    """Temporarily change route switches: `with _lib.routes(no_fused_bwd=1): ...` (tests / tools; the library reads the
    environment only once, at first use).  A hipGraph captured inside keeps the routes it was captured with."""
    old = {k: get_route(k) for k in kw}
    try:
        for k, v in kw.items():
            set_route(k, v)
        yield
    finally:
        for k, v in old.items():
            set_route(k, v)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def require_cuda(t, what):
    if not t.is_cuda:
        raise Mi3dError(f"{what}: tensor is on {t.device}; the MI355X HIP path needs GPU tensors (no CPU fallback)")


def ptr_table(ptrs):
    """Host array of device pointers (void* const*)."""
    arr = (C.c_void_p * len(ptrs))(*[p if p else None for p in ptrs])
    return arr


def dtype_code(dt):
    if dt == torch.float32:
        return DTYPE_F32
    if dt == torch.bfloat16:
        return DTYPE_BF16
    raise Mi3dError(f"unsupported compute dtype {dt} (float32 or bfloat16)")
