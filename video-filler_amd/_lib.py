"""ctypes binding of include/vf_hip.h (the C-ABI of the gfx950 backend).

The library is built in-tree by build.py (hipcc, --offload-arch=gfx950).  There is NO fallback:
if the shared object is missing or no GPU is visible, loading / context creation raises.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_HEADER = os.path.join(_HERE, "..", "include", "vf_hip.h")

vp, f32, i32, i64, f64, sz, u64 = C.c_void_p, C.c_float, C.c_int, C.c_int64, C.c_double, C.c_size_t, C.c_uint64

# name -> (restype, argtypes); mirrors include/vf_hip.h one to one
SIGNATURES = {
    "vf_last_error": (C.c_char_p, []),
    "vf_version": (i32, []),
    "vf_ctx_create": (i32, [C.POINTER(vp), i32, vp]),
    "vf_ctx_destroy": (i32, [vp]),
    "vf_ctx_set_mfma_mode": (i32, [vp, i32]),
    "vf_ctx_set_stream": (i32, [vp, vp]),
    "vf_ctx_set_workspace": (i32, [vp, vp, sz]),
    "vf_workspace_bytes_hint": (sz, []),
    "vf_stream_synchronize": (i32, [vp]),
    "vf_malloc": (i32, [C.POINTER(vp), sz]),
    "vf_free": (i32, [vp]),
    "vf_memcpy_h2d": (i32, [vp, vp, vp, sz]),
    "vf_memcpy_d2h": (i32, [vp, vp, vp, sz]),
    "vf_zero": (i32, [vp, vp, sz]),
    "vf_zero_segments": (i32, [vp, vp, vp, vp, i32]),
    "vf_nchw_to_nhwc": (i32, [vp, vp, vp, i32, i32, i32, i32]),
    "vf_nhwc_to_nchw": (i32, [vp, vp, vp, i32, i32, i32, i32]),
    "vf_conv2d_fwd": (i32, [vp, vp, vp, vp, vp] + [i32] * 8 + [i32, f32]),
    "vf_conv2d_bwd_data": (i32, [vp, vp, vp, vp] + [i32] * 8),
    "vf_conv2d_bwd_weight": (i32, [vp, vp, vp, vp, vp] + [i32] * 8 + [f32]),
    "vf_deconv2d_fwd": (i32, [vp, vp, vp, vp, vp] + [i32] * 8 + [i32, f32]),
    "vf_deconv2d_bwd_data": (i32, [vp, vp, vp, vp] + [i32] * 8),
    "vf_deconv2d_bwd_weight": (i32, [vp, vp, vp, vp, vp] + [i32] * 8 + [f32]),
    "vf_planes_split": (i32, [vp, vp, vp, i64]),
    "vf_conv2d_fwd_planes": (i32, [vp, vp, vp, vp, vp, vp] + [i32] * 8 + [i32, f32]),
    "vf_weight_planes": (i32, [vp, vp, vp, vp, i32, i32]),
    "vf_weight_planes_multi": (i32, [vp, vp, i32, i32]),
    "vf_pconv_supported": (i32, [i32] * 9),
    "vf_pconv_supported_in_mode": (i32, [i32] * 10),
    "vf_pconv_set_routing": (i32, [i32, i32]),
    "vf_pconv_gather": (i32, [vp, vp, vp, vp, vp] + [i32] * 5 + [i32, f32]),
    "vf_pconv_scatter": (i32, [vp, vp, vp, vp, vp] + [i32] * 5 + [i32, f32, vp, i32, f32]),
    "vf_bn_stats": (i32, [vp, vp, vp, vp, i64, i32]),
    "vf_bn_finalize": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, f32, f32]),
    "vf_bn_apply": (i32, [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, f32]),
    "vf_bn_train_fwd": (i32, [vp] * 10 + [i64, i32, f32, f32, i32, f32]),
    "vf_bn_eval_fwd": (i32, [vp] * 7 + [i64, i32, f32, i32, f32]),
    "vf_bn_bwd_stats": (i32, [vp] * 6 + [i64, i32, i32, f32]),
    "vf_bn_bwd_apply": (i32, [vp] * 11 + [i64, i64, i32, i32, f32, f32]),
    "vf_bn_bwd": (i32, [vp] * 11 + [i64, i32, i32, f32, f32]),
    "vf_bn_train_fwd_groups": (i32, [vp] * 10 + [i64, i32, i32, f32, f32, i32, f32]),
    "vf_bn_bwd_groups": (i32, [vp] * 11 + [i64, i32, i32, i32, f32, f32]),
    "vf_bn_train_fwd_planes": (i32, [vp] * 10 + [i64, i32, i32, f32, f32, i32, f32, vp]),
    "vf_bn_bwd_planes": (i32, [vp] * 11 + [i64, i32, i32, i32, f32, f32, vp]),
    "vf_bn_fuse_next_fwd": (i32, [vp, vp, vp, i32, i32]),
    "vf_bn_fuse_next_bwd": (i32, [vp, vp, vp, i32, f32, vp, vp, i32, i32]),
    "vf_bn_fuse_result": (i32, [vp, C.POINTER(i32)]),
    "vf_bn_train_fwd_pre": (i32, [vp, vp, i32] + [vp] * 9 + [i64, i32, i32, f32, f32, i32, f32, vp]),
    "vf_bn_bwd_pre": (i32, [vp, vp, i32] + [vp] * 9 + [i64, i32, i32, f32, vp]),
    "vf_act_fwd": (i32, [vp, vp, vp, i64, i32, f32]),
    "vf_act_bwd": (i32, [vp, vp, vp, vp, i64, i32, f32]),
    "vf_axpby": (i32, [vp, f32, vp, f32, vp, i64]),
    "vf_cmul": (i32, [vp, vp, vp, i64]),
    "vf_scale_shift": (i32, [vp, vp, f32, f32, i64]),
    "vf_masked_compose": (i32, [vp, vp, vp, vp, vp, i64]),
    "vf_conv2d_bwd_data_act": (i32, [vp, vp, vp, vp, vp, i32, f32, i32, i32, i32, i32, i32, i32, i32, i32]),
    "vf_wgrad_group_begin": (i32, [vp]),
    "vf_wgrad_group_end": (i32, [vp]),
    "vf_wgrad_group_abort": (i32, [vp]),
    "vf_bias_grad_plan": (i32, [i64, i32, vp, vp, vp, vp]),
    "vf_bias_grad_multi": (i32, [vp, vp, i32, i32, i32]),
    "vf_center_prepare": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32]),
    "vf_clip_prepare": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, i32, i32, vp, vp]),
    "vf_tiles_gather": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "vf_tiles_scatter": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "vf_channel_copy": (i32, [vp, vp, i32, i32, vp, i32, i32, i32, i64]),
    "vf_noise_fill": (i32, [vp, vp, i64, u64, vp, u64, i32]),
    "vf_bce_fwd": (i32, [vp, vp, f32, i32, vp]),
    "vf_bce_bwd": (i32, [vp, vp, f32, vp, i32]),
    "vf_bce_fwd_bwd": (i32, [vp, vp, f32, f32, i32, i32, vp, vp, vp]),
    "vf_mse_fwd": (i32, [vp, vp, vp, i64, vp]),
    "vf_mse_bwd": (i32, [vp, vp, vp, vp, i64]),
    "vf_recon_grad_mix": (i32, [vp, vp, vp, vp, vp, f32, f32, f32, i32, i32, i32, i64, vp]),
    "vf_gdl_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "vf_gdl_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32]),
    "vf_masked_mse_fwd": (i32, [vp, vp, vp, vp, f32, i64, vp]),
    "vf_masked_mse_bwd": (i32, [vp, vp, vp, vp, f32, vp, i64]),
    "vf_adam_step": (i32, [vp, vp, vp, vp, vp, i64, f64, f64, f64, f64, vp]),
    "vf_adam_prep": (i32, [vp, f64, f64, f64, vp]),
    "vf_adam_apply": (i32, [vp, vp, vp, vp, vp, i64, f64, f64, f64, vp]),
    "vf_adam_apply_ranges": (i32, [vp, vp, vp, vp, vp, C.POINTER(i64), C.POINTER(i64), i32, f64, f64, f64, vp]),
    "vf_wgrad_adam_outer_supported": (i32, [i32, i32, i32]),
    "vf_wgrad_adam_outer": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, f64, f64, f64, vp]),
    "vf_wgrad_adam_outer_gathered": (i32, [vp, vp, vp, i32, i32, i64, i32, i32, vp, vp, vp, vp, C.c_float, f64, f64, f64, vp]),
    "vf_wgrad_adam_outer_rows": (i32, [vp, vp, vp, i32, i32, i64, i32, i32, i32, i32, vp, vp, vp, vp, C.c_float, f64, f64, f64, vp]),
    "vf_conv_is_fast": (i32, [i32, i32, i32, i32, i32]),
    "vf_conv2d_bwd_weight_planes": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32]),
    "vf_deconv2d_bwd_weight_planes": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32]),
    "vf_comm_available": (i32, []),
    "vf_comm_unique_id": (i32, [vp]),
    "vf_comm_init": (i32, [C.POINTER(vp), vp, i32, i32]),
    "vf_comm_world": (i32, [vp]),
    "vf_comm_rank": (i32, [vp]),
    "vf_comm_allreduce_async": (i32, [vp, vp, vp, i64, i32, i32, C.POINTER(i32)]),
    "vf_comm_allreduce_avg_async": (i32, [vp, vp, vp, i64, C.POINTER(i32)]),
    "vf_comm_reduce_scatter_avg_async": (i32, [vp, vp, vp, i64, C.POINTER(i32)]),
    "vf_comm_allgather_async": (i32, [vp, vp, vp, i64, C.POINTER(i32)]),
    "vf_comm_broadcast_async": (i32, [vp, vp, vp, i64, i32, C.POINTER(i32)]),
    "vf_comm_wait": (i32, [vp, vp, i32]),
    "vf_comm_allreduce_inline": (i32, [vp, vp, vp, i64, i32, i32]),
    "vf_comm_broadcast": (i32, [vp, vp, vp, i64, i32, i32]),
    "vf_comm_barrier": (i32, [vp, vp]),
    "vf_comm_destroy": (i32, [vp]),
    "vf_net_create": (i32, [vp, C.POINTER(vp), vp, i32, i32, i32, i32, i32]),
    "vf_net_destroy": (i32, [vp]),
    "vf_net_parameters": (i32, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64)]),
    "vf_net_param_offset": (i64, [vp, i32, i32, C.POINTER(i64)]),
    "vf_net_bn_running": (i32, [vp, i32, C.POINTER(vp), C.POINTER(vp)]),
    "vf_net_training": (i32, [vp, i32]),
    "vf_net_zero_grad": (i32, [vp]),
    "vf_net_forward": (i32, [vp, vp, C.POINTER(vp)]),
    "vf_net_backward": (i32, [vp, vp, vp, C.POINTER(vp)]),
    "vf_net_update_grad_input": (i32, [vp, vp, vp, C.POINTER(vp)]),
    "vf_net_layer_output": (i32, [vp, i32, C.POINTER(vp)]),
    "vf_net_layer_grad_input": (i32, [vp, i32, C.POINTER(vp)]),
    "vf_net_layer_shape": (i32, [vp, i32] + [C.POINTER(i32)] * 7),
    "vf_net_reshape": (i32, [vp, i32, i32, i32, i32]),
    "vf_net_bind_parameters": (i32, [vp, vp, vp, i64]),
    "vf_net_bind_bn_running": (i32, [vp, i32, vp, vp]),
    "vf_net_bn_saved": (i32, [vp, i32, C.POINTER(vp), C.POINTER(vp)]),
    "vf_net_zero_conv_biases": (i32, [vp, vp]),
    "vf_net_set_skip_input_grad": (i32, [vp, i32]),
    "vf_net_set_batch_groups": (i32, [vp, i32]),
    "vf_net_update_grad_input_group": (i32, [vp, vp, vp, i32, i32, C.POINTER(vp)]),
    "vf_net_plan_size": (i32, [vp]),
    "vf_net_layer_has_act_bits": (i32, [vp, i32]),
    "vf_net_bucket_split": (i32, [vp, f64, C.POINTER(i32), C.POINTER(i64)]),
    "vf_net_backward_range": (i32, [vp, vp, vp, i32, i32, i32, C.POINTER(vp)]),
    "vf_net_backward_split": (i32, [vp, vp, vp, i32, i32, C.POINTER(vp)]),
    "vf_net_backward_finish": (i32, [vp]),
    "vf_net_set_fused_adam": (i32, [vp, i32, C.POINTER(i32)]),
    "vf_net_fused_adam_range": (i32, [vp, i32, C.POINTER(i64), C.POINTER(i64)]),
    "vf_net_adam_fused": (i32, [vp, vp, vp, f64, f64, f64, vp, i32]),
    "vf_net_fused_adam_pack_size": (i32, [vp, C.POINTER(i64)]),
    "vf_net_fused_adam_pack": (i32, [vp, vp]),
    "vf_net_adam_fused_gathered": (i32, [vp, vp, i32, i64, vp, vp, f64, f64, f64, vp, i32]),
    "vf_net_fused_adam_rows_ok": (i32, [vp, i32]),
    "vf_net_fused_adam_row_range": (i32, [vp, i32, i32, i32, C.POINTER(i64), C.POINTER(i64)]),
    "vf_net_forward_wait_fused": (i32, [vp, vp, i32]),
    "vf_net_adam_fused_gathered_rows": (i32, [vp, vp, i32, i64, vp, vp, f64, f64, f64, vp, i32, i32, i32]),
    "vf_net_set_sync_bn": (i32, [vp, vp, i32, i32]),
    "vf_net_set_weight_planes_managed": (i32, [vp, i32]),
    "vf_net_refresh_weight_planes": (i32, [vp]),
    "vf_net_set_planes_gate": (i32, [f64, i32]),
    "vf_net_bind_output": (i32, [vp, i32, vp]),
    "vf_net_set_act_observer": (i32, [vp, vp, vp]),
    "vf_trace_available": (i32, []),
    "vf_trace_enable": (i32, [i32]),
    "vf_range_push": (i32, [C.c_char_p]),
    "vf_range_pop": (i32, []),
    "vf_mark": (i32, [C.c_char_p]),
    "vf_range_depth": (i32, []),
    "vf_wgrad_group_count": (i32, [vp, C.POINTER(i32)]),
    "vf_wgrad_group_end_partial": (i32, [vp, i32]),
    "vf_prof_begin": (i32, [vp]),
    "vf_prof_end": (i32, [vp]),
    "vf_prof_count": (i32, []),
    "vf_prof_get": (i32, [i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(f64), C.POINTER(f64), C.POINTER(f64)]),
}


def header_symbols():
    """Every function name declared in include/vf_hip.h."""
    with open(_HEADER) as fh:
        text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(vf_[a-z0-9_]+)\s*\(", text)))


def lib_path():
    """VF_HIP_LIB names another build of the same library (A/B timing of kernel versions; the Lua binding honours it too)."""
    return os.environ.get("VF_HIP_LIB") or os.path.join(_HERE, "lib", "libvf_hip.so")


_LIB = None


def load():
    """dlopen the HIP library and attach signatures.  Raises if it is not built."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                "video-filler_amd: %s is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % path)
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


class VfError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise VfError(load().vf_last_error().decode())
