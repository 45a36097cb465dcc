"""ctypes binding of the gfx950 C-ABI library (include/locate_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails, this module raises.
PyTorch must be imported first so that the library's libamdhip64.so.7 dependency resolves to the HIP runtime
already loaded by torch (one runtime per process: streams and device pointers are then interchangeable)."""
import ctypes
import os

import torch  # noqa: F401  (loads the HIP runtime the extension shares)

_HERE = os.path.dirname(os.path.abspath(__file__))
# LOCATE_HIP_DEBUG_LIBRARY=1 (read here, on the Python side, once): load the debug variant, the only one that has the
# LOCATE_DISABLE kernel-flavour switch compiled in (build.py)
LIB_PATH = os.path.join(_HERE, "csrc", "liblocate_hip_dbg.so" if os.environ.get("LOCATE_HIP_DEBUG_LIBRARY") == "1"
                        else "liblocate_hip.so")

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_i64 = ctypes.c_int64
c_f = ctypes.c_float
c_d = ctypes.c_double
c_sz = ctypes.c_size_t
c_ip = ctypes.POINTER(ctypes.c_int)

# name -> (restype, argtypes); mirrors include/locate_hip.h one to one
PROTOTYPES = {
    "locate_last_error": (ctypes.c_char_p, []),
    "locate_abi_version": (c_i, []),
    "locate_device_info": (c_i, [ctypes.c_char_p, c_i, c_ip, c_ip]),
    "locate_roottanh_fwd": (c_i, [c_p, c_p, c_i64, c_p, c_p]),
    "locate_roottanh_bwd": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_p, c_p]),
    "locate_act_cat_rows_fwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "locate_act_rows_bwd": (c_i, [c_p, c_p, c_i64, c_p, c_p, c_i, c_i, c_p]),
    "locate_add3": (c_i, [c_p, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_i, c_i64, c_p]),
    "locate_absmax_words": (c_i, []),
    "locate_wgrad_batch_record_bytes": (c_sz, []),
    "locate_wgrad_batch_max": (c_i, []),
    "locate_wgrad_batch_record": (c_i, [c_ip, c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_i, c_i, c_p, c_p]),
    "locate_wgrad_batch": (c_i, [c_p, c_i, c_p]),
    "locate_absmax": (c_i, [c_p, c_i64, c_p, c_p]),
    "locate_tanh_fwd": (c_i, [c_p, c_p, c_i64, c_p]),
    "locate_tanh_bwd": (c_i, [c_p, c_p, c_p, c_i64, c_p]),
    "locate_norm_stats_workspace_bytes": (c_sz, []),
    "locate_norm_stats": (c_i, [c_p, c_i64, c_p, c_p, c_p]),
    "locate_norm_bwd_workspace_bytes": (c_sz, [c_i, c_i]),
    "locate_norm_fwd": (c_i, [c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "locate_norm_bwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "locate_norm_bwd_fused_workspace_bytes": (c_sz, [c_i, c_i]),
    "locate_norm_bwd_fused_plane_offset": (c_sz, []),
    "locate_norm_bwd_fused": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "locate_fin_norm_channels": (c_i, [c_p, c_i, c_p]),
    "locate_channel_sum_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "locate_channel_sum": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i64, c_p, c_p]),
    "locate_gate_fwd": (c_i, [c_p, c_p, c_i, c_p, c_p, c_i64, c_i, c_p]),
    "locate_gate_fwd_stats": (c_i, [c_p, c_p, c_i, c_p, c_p, c_i64, c_i, c_i, c_p, c_p]),
    "locate_gate_bwd_workspace_bytes": (c_sz, [c_i64]),
    "locate_gate_bwd": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_p, c_i, c_p, c_p]),
    "locate_softmax_fwd": (c_i, [c_p, c_p, c_i64, c_i, c_p]),
    "locate_softmax_bwd": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_p]),
    "locate_upsample2x_fwd": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_p]),
    "locate_upsample2x_bwd": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_p]),
    "locate_pool2_upsample2x_fwd": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_p]),
    "locate_pool2_upsample2x_bwd": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_i, c_p]),
    "locate_avgpool2_fwd": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_p]),
    "locate_avgpool2_bwd": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_i, c_p]),
    "locate_feature_pool_fwd": (c_i, [c_p, c_p, c_i64, c_i, c_p]),
    "locate_feature_pool_bwd": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_p]),
    "locate_copy_channels": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i64, c_i64, c_i, c_p]),
    "locate_sn_workspace_bytes": (c_sz, [c_i, c_i]),
    "locate_sn_power_iter": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p]),
    "locate_sn_table_record_bytes": (c_sz, []),
    "locate_sn_power_iter_batched": (c_i, [c_p, c_i, c_i, c_i, c_p]),
    "locate_sn_weight_bwd": (c_i, [c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p]),
    "locate_sn_group_workspace_bytes": (c_sz, []),
    "locate_sn_weight_bwd_grouped": (c_i, [c_p, c_i64, c_p, c_i64, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_i64, c_p,
                                           c_p, c_p, c_i, c_i, c_p, c_p]),
    "locate_sn_dv_batched": (c_i, [c_p, c_i, c_i, c_i, c_p]),
    "locate_conv_panel_bytes": (c_sz, [c_ip, c_i]),
    "locate_conv_pack_panel": (c_i, [c_ip, c_i, c_p, c_p, c_p]),
    "locate_conv_pack_job_bytes": (c_sz, []),
    "locate_conv_pack_job": (c_i, [c_ip, c_i, c_p, c_p, c_i, c_p, c_ip, c_i, c_p]),
    "locate_conv_pack_panels": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p]),
    "locate_conv_pack_job_is_direct": (c_i, [c_p]),
    "locate_conv_pack_job_is_window": (c_i, [c_p]),
    "locate_conv_win_ok": (c_i, [c_ip, c_i, c_i64, c_p]),
    "locate_conv_win_workspace_bytes": (c_sz, [c_ip, c_i]),
    "locate_conv_fwd_workspace_bytes": (c_sz, [c_ip]),
    "locate_conv_counter_bytes": (c_sz, []),
    "locate_conv_fwd": (c_i, [c_ip, c_p, c_i64, c_p, c_p, c_i, c_i, c_p, c_p, c_i64, c_p, c_p, c_i, c_p, c_p, c_p]),
    "locate_conv_dgrad_workspace_bytes": (c_sz, [c_ip]),
    "locate_conv_dgrad": (c_i, [c_ip, c_p, c_i64, c_p, c_p, c_i, c_i, c_p, c_p, c_i64, c_p, c_p, c_i, c_p, c_p, c_p]),
    "locate_conv_wgrad_workspace_bytes": (c_sz, [c_ip]),
    "locate_conv_wgrad_partials": (c_i, [c_ip]),
    "locate_conv_wgrad": (c_i, [c_ip, c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p]),
    "locate_conv_wgrad_group_partials": (c_i, [c_ip, c_i]),
    "locate_conv_wgrad_group_workspace_bytes": (c_sz, [c_ip, c_i]),
    "locate_slab_reduce_record_bytes": (c_sz, []),
    "locate_slab_reduce_max": (c_i, []),
    "locate_slab_reduce_record_blocks": (c_i, [c_p]),
    "locate_slab_reduce_batch": (c_i, [c_p, c_i, c_p]),
    "locate_dwconv_fwd": (c_i, [c_ip, c_p, c_i64, c_p, c_p, c_i, c_i, c_p, c_i64, c_p]),
    "locate_dwconv_dgrad": (c_i, [c_ip, c_p, c_i64, c_p, c_p, c_i, c_i, c_p, c_i64, c_p]),
    "locate_dwconv_wgrad_workspace_bytes": (c_sz, [c_ip]),
    "locate_dwconv_wgrad_partials": (c_i, [c_ip]),
    "locate_dwconv_wgrad": (c_i, [c_ip, c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p]),
    "locate_groupdot_fwd": (c_i, [c_p, c_i64, c_p, c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_p]),
    "locate_groupdot_dgrad": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p, c_i64, c_i, c_i, c_i, c_p]),
    "locate_groupdot_wgrad_partials": (c_i, [c_i, c_i]),
    "locate_groupdot_wgrad": (c_i, [c_p, c_i64, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_p]),
    "locate_fin_record_bytes": (c_sz, []),
    "locate_fin_sn_dot_partials": (c_i, [c_i, c_i, c_i]),
    "locate_fin_sn_dots": (c_i, [c_p, c_i, c_p]),
    "locate_fin_sn_rank1": (c_i, [c_p, c_i, c_p]),
    "locate_fin_sums": (c_i, [c_p, c_i, c_p]),
    "locate_fin_channel_slices": (c_i, [c_i, c_i, c_i]),
    "locate_fin_channel_sums": (c_i, [c_p, c_i, c_p]),
    "locate_gate_bwd_partials": (c_i, [c_i64, c_i]),
    "locate_multi_copy_record_bytes": (c_sz, []),
    "locate_multi_copy_chunk_elems": (c_i, []),
    "locate_multi_copy": (c_i, [c_p, c_p, c_i, c_i, c_f, c_p]),
    "locate_nadam_tensor_record_bytes": (c_sz, []),
    "locate_nadam_chunk_elems": (c_i, []),
    "locate_nadam_step": (c_i, [c_p, c_p, c_p, c_i, c_i, c_d, c_d, c_d, c_d, c_d, c_d, c_p]),
    "locate_d_loss": (c_i, [c_p, c_p, c_p, c_i, c_f, c_p, c_p, c_p, c_p, c_p]),
    "locate_g_loss": (c_i, [c_p, c_i, c_p, c_p, c_p]),
}


class LocateError(RuntimeError):
    pass


# bumped together with locate_abi_version() in csrc/runtime.hip whenever a prototype above changes: a stale .so that still
# exports every NAME would otherwise be called with shifted arguments
EXPECTED_ABI = 10


_lib = None


def lib():
    """The loaded library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LocateError("%s is missing: run `python -m locate_amd.build` (hipcc, gfx950). "
                              "There is no CPU fallback for the MI355X kernels." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)   # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        got = handle.locate_abi_version()
        if got != EXPECTED_ABI:
            raise LocateError("%s reports ABI version %d, this package binds version %d: rebuild it (`python -m locate_amd.build`)"
                              % (LIB_PATH, got, EXPECTED_ABI))
        _lib = handle
    return _lib


def check(status, what=""):
    if status != 0:
        msg = lib().locate_last_error()
        raise LocateError("%s failed (status %d): %s" % (what, status, msg.decode() if msg else "?"))


def require_gpu():
    """Fail loudly unless the current device is an MI355X-class (gfx950) GPU."""
    if not torch.cuda.is_available():
        raise LocateError("no HIP device: locate_amd computes only on MI355X (gfx950); there is no CPU path")
    name = ctypes.create_string_buffer(64)
    cu, wave = ctypes.c_int(0), ctypes.c_int(0)
    check(lib().locate_device_info(name, 64, ctypes.byref(cu), ctypes.byref(wave)), "locate_device_info")
    arch = name.value.decode()
    if not arch.startswith("gfx950"):
        raise LocateError("device architecture %r: the kernels are built for gfx950 only" % arch)
    return arch, cu.value, wave.value
