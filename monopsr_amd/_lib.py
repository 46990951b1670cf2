"""ctypes binding of libmonopsr_hip.so (C ABI declared in include/monopsr_hip.h).

There is no CPU fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPSR_LIB_PATH: development knob for A/B-ing two builds of the library inside one GPU session
LIB_PATH = os.environ.get("MPSR_LIB_PATH") or os.path.join(_HERE, "libmonopsr_hip.so")
ABI_VERSION = 6

_lib = None


class MpsrError(RuntimeError):
    """A libmonopsr_hip.so call returned a non-zero status."""


class InvalidArgumentError(MpsrError, ValueError):
    """Shape/argument check failed -- the analogue of the tf.errors.InvalidArgumentError the reference's
    OP_REQUIRES checks raise (tf_nndistance.cpp:51-58, tf_approxmatch.cpp:152-165)."""


c_f = ctypes.c_void_p  # device pointers travel as integers
c_i = ctypes.c_int
c_sz = ctypes.c_size_t


class Layer(ctypes.Structure):
    """struct mpsr_layer"""
    _fields_ = [("cin", ctypes.c_int32), ("cout", ctypes.c_int32), ("kh", ctypes.c_int32), ("kw", ctypes.c_int32),
                ("dilation", ctypes.c_int32), ("relu", ctypes.c_int32), ("w_off", ctypes.c_int64),
                ("b_off", ctypes.c_int64)]


class HeadConsts(ctypes.Structure):
    """struct mpsr_head_consts"""
    _fields_ = [("image_h", ctypes.c_float), ("image_w", ctypes.c_float), ("max_depth", ctypes.c_float),
                ("cen_y_norm", ctypes.c_float), ("cen_y_class_offset", ctypes.c_float),
                ("num_classes", ctypes.c_int32), ("num_alpha_bins", ctypes.c_int32)]


class HeadOutputs(ctypes.Structure):
    """struct mpsr_head_outputs"""
    _fields_ = [(k, ctypes.c_void_p) for k in
                ("lwh", "lwh_offs", "alpha_bins", "alpha_regs", "prop_cen_z", "cen_y", "cen_y_offs", "cen_z",
                 "cen_z_offs", "cen_x", "centroids")]


class NetOpts(ctypes.Structure):
    """struct mpsr_net_opts"""
    _fields_ = [("filter_cache", ctypes.c_void_p), ("filter_cache_floats", ctypes.c_size_t),
                ("filter_cache_valid", ctypes.c_int32), ("ready_event", ctypes.c_void_p),
                ("filter_cache_tags", ctypes.POINTER(ctypes.c_int32)), ("math", ctypes.c_int32),
                ("winograd_policy", ctypes.c_int32)]


class ConvOpts(ctypes.Structure):
    """struct mpsr_conv_opts"""
    _fields_ = [("math", ctypes.c_int32), ("winograd_policy", ctypes.c_int32)]


class PackJob(ctypes.Structure):
    """struct mpsr_pack_job"""
    _fields_ = [("w", ctypes.c_void_p), ("wd", ctypes.c_void_p), ("N", ctypes.c_int32), ("Nd", ctypes.c_int32),
                ("T", ctypes.c_int32), ("C", ctypes.c_int32), ("chunk0", ctypes.c_int64)]


# per-call option values (MPSR_CALL_MATH_*, MPSR_CALL_WINOGRAD_*): None / "inherit" = the process-wide default
CALL_MATH = {None: 0, "inherit": 0, "fp32": 1, "bf16x3": 2}
CALL_WINOGRAD = {None: 0, "inherit": 0, "auto": 1, "off": 2, "accurate": 3}


# name -> (restype, argtypes); must list every symbol include/monopsr_hip.h declares (tests/test_cabi.py checks).
SIGNATURES = {
    "mpsr_last_error": (ctypes.c_char_p, []),
    "mpsr_abi_version": (c_i, []),
    "mpsr_set_conv_math": (c_i, [c_i]),
    "mpsr_get_conv_math": (c_i, []),
    "mpsr_set_winograd_policy": (c_i, [c_i]),
    "mpsr_get_winograd_policy": (c_i, []),
    "mpsr_crc32c": (ctypes.c_uint32, [ctypes.c_uint32, ctypes.c_void_p, c_sz]),
    "mpsr_nn_distance_fwd": (c_i, [c_i, c_i, c_f, c_i, c_f, c_f, c_f, c_f, c_f, c_f]),
    "mpsr_nn_distance_bwd": (c_i, [c_i, c_i, c_f, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "mpsr_approx_match_temp_floats": (c_sz, [c_i, c_i, c_i]),
    "mpsr_emd_temp_floats": (c_sz, [c_i, c_i, c_i, c_i]),
    "mpsr_emd_loss_temp_floats": (c_sz, [c_i, c_i, c_i, c_i]),
    "mpsr_approx_match": (c_i, [c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_sz, c_f]),
    "mpsr_approx_match_ex": (c_i, [c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_sz, c_i, c_f]),
    "mpsr_emd_loss": (c_i, [c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_i, c_f]),
    "mpsr_match_cost": (c_i, [c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f]),
    "mpsr_match_cost_grad": (c_i, [c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f]),
    "mpsr_crop_and_resize": (c_i, [c_f, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_i, c_i, ctypes.c_float, c_f, c_f]),
    "mpsr_resize_bilinear": (c_i, [c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f]),
    "mpsr_max_pool": (c_i, [c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f]),
    "mpsr_conv2d_nhwc_f32": (c_i, [c_f, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f,
                                   c_sz, c_f]),
    "mpsr_conv2d_nhwc_f32_ex": (c_i, [c_f, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f,
                                      c_sz, ctypes.POINTER(ConvOpts), c_f]),
    "mpsr_conv2d_scratch_floats": (c_sz, [c_i, c_i, c_i, c_i]),
    "mpsr_conv2d_plan": (c_i, [c_i] * 8 + [ctypes.POINTER(c_i), ctypes.POINTER(ctypes.c_double)]),
    "mpsr_im2col_root": (c_i, [c_f, c_i, c_i, c_i, c_f, c_i, c_f]),
    "mpsr_conv2d_wgrad_f32": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f]),
    "mpsr_conv2d_wgrad_scratch_floats": (c_sz, [c_i] * 8),
    "mpsr_conv2d_wgrad_ws_f32": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_sz, c_f]),
    "mpsr_conv2d_dgrad_pack": (c_i, [c_f, c_i, c_i, c_i, c_i, c_f, c_f]),
    "mpsr_dgrad_pack_table_bytes": (c_sz, [ctypes.POINTER(PackJob), c_i]),
    "mpsr_dgrad_pack_table_build": (c_i, [ctypes.POINTER(PackJob), c_i, ctypes.c_void_p,
                                          ctypes.POINTER(ctypes.c_longlong)]),
    "mpsr_conv2d_dgrad_pack_batch": (c_i, [c_f, ctypes.c_longlong, c_f]),
    "mpsr_act_bias_grad": (c_i, [c_f, c_f, c_f, c_f, ctypes.c_longlong, c_i, c_f]),
    "mpsr_bias_grad": (c_i, [c_f, ctypes.c_longlong, c_i, c_f, c_f]),
    "mpsr_relu_grad": (c_i, [c_f, c_f, c_f, ctypes.c_longlong, c_f]),
    "mpsr_conv2d_relu_masked_f32": (c_i, [c_f, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_sz, c_f]),
    "mpsr_relu_bitmask_words": (ctypes.c_longlong, [ctypes.c_longlong, c_i]),
    "mpsr_relu_bitmask": (c_i, [c_f, ctypes.c_longlong, c_i, c_f, c_f]),
    "mpsr_conv1x1_masked_applies": (c_i, [ctypes.c_longlong, c_i, c_i]),
    "mpsr_conv1x1_masked_f32": (c_i, [c_f, ctypes.c_longlong, c_i, c_f, c_f, c_f, c_f, c_f, c_i, c_f]),
    "mpsr_conv1x1_relu_bitmask_f32": (c_i, [c_f, ctypes.c_longlong, c_i, c_f, c_f, c_f, c_i, c_f, c_f, c_i, c_f]),
    "mpsr_max_pool_grad": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f]),
    "mpsr_resize_bilinear_grad": (c_i, [c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f]),
    "mpsr_adam_step": (c_i, [c_f, c_f, c_f, c_f, ctypes.c_longlong, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                             ctypes.c_float, c_i, ctypes.c_float, c_f]),
    "mpsr_adam_step_lr_dev": (c_i, [c_f, c_f, c_f, c_f, ctypes.c_longlong, c_f, ctypes.c_float, ctypes.c_float,
                                    ctypes.c_float, ctypes.c_float, c_f]),
    "mpsr_crop_and_resize_grad": (c_i, [c_f, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_i, c_i, c_f, c_f]),
    "mpsr_batch_norm_stats": (c_i, [c_f, ctypes.c_longlong, c_i, c_f, c_f, c_f]),
    "mpsr_batch_norm_finalize": (c_i, [c_f, c_f, c_f, ctypes.c_longlong, c_i, ctypes.c_float, ctypes.c_float, c_f, c_f,
                                       c_f, c_f, c_f]),
    "mpsr_batch_norm_grad_finalize": (c_i, [c_f, c_f, ctypes.c_double, c_i, c_f, c_f, c_f, c_f]),
    "mpsr_batch_norm_apply": (c_i, [c_f, ctypes.c_longlong, c_i, c_f, c_f, c_f, c_i, c_f, c_f]),
    "mpsr_batch_norm_grad_sums": (c_i, [c_f, c_f, c_f, ctypes.c_longlong, c_i, c_f, c_f, c_f, c_f, c_f]),
    "mpsr_batch_norm_grad_sums_z": (c_i, [c_f, c_f, ctypes.c_longlong, c_i, c_f, c_f, c_f, c_f, c_f, c_f]),
    "mpsr_batch_norm_grad_z": (c_i, [c_f, c_f, ctypes.c_longlong, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "mpsr_batch_norm_grad": (c_i, [c_f, c_f, c_f, ctypes.c_longlong, c_i, c_f, c_f, c_f, c_f, c_f, c_f]),
    "mpsr_clip_by_norm_segments": (c_i, [c_f, c_f, c_f, c_f, c_i, c_f, ctypes.c_size_t, c_i, ctypes.c_float, c_f]),
    "mpsr_clip_adam_ema_step": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_f, ctypes.c_size_t, c_i, ctypes.c_float,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_i, ctypes.c_float,
                                      c_f]),
    "mpsr_xyz_map_local_to_global": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_f]),
    "mpsr_xyz_map_local_to_global_grad": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_f]),
    "mpsr_proj_err_norm": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    "mpsr_proj_err_norm_grad": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    "mpsr_depth_map_local_to_global": (c_i, [c_f, c_i, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "mpsr_depth_map_local_to_global_grad": (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "mpsr_huber_loss_sums": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, ctypes.c_float, c_f, c_f, c_f]),
    "mpsr_huber_loss_grad": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, ctypes.c_float, c_f, c_f]),
    "mpsr_format_boxes": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i,
                                ctypes.c_float, c_f, c_f, c_f]),
    "mpsr_trunk_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "mpsr_trunk_fwd": (c_i, [c_f, c_i, c_i, c_i, c_f, ctypes.POINTER(Layer), c_i, c_f, c_f, c_sz, c_f]),
    "mpsr_conv3x3_upsampled_scratch_floats": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "mpsr_conv3x3_upsampled_f32": (c_i, [c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_f, c_i, c_f, c_sz, c_f]),
    "mpsr_conv3x3_upsampled_applies": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "mpsr_conv3x3_upsampled_bwd_scratch_floats": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "mpsr_conv3x3_upsampled_bwd_f32": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_f, c_f, c_sz,
                                            c_f]),
    "mpsr_filter_cache_floats": (c_sz, [ctypes.POINTER(Layer), c_i]),
    "mpsr_trunk_fwd_ex": (c_i, [c_f, c_i, c_i, c_i, c_f, ctypes.POINTER(Layer), c_i, c_f, c_f, c_sz,
                                ctypes.POINTER(NetOpts), c_f]),
    "mpsr_squash_decoder_fwd_ex": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f, ctypes.POINTER(Layer), c_i, c_f, c_f,
                                         c_f, c_f, c_sz, ctypes.POINTER(NetOpts), c_f]),
    "mpsr_decoder_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "mpsr_squash_decoder_plan": (c_i, [c_i, c_i, c_i, c_i, c_i, ctypes.POINTER(Layer), c_i, ctypes.POINTER(c_i),
                                       ctypes.POINTER(ctypes.c_double)]),
    "mpsr_squash_decoder_fwd": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f, ctypes.POINTER(Layer), c_i, c_f, c_f,
                                      c_f, c_f, c_sz, c_f]),
    "mpsr_heads_workspace_bytes": (c_sz, [c_i, c_i]),
    "mpsr_heads_fwd": (c_i, [c_f, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f, ctypes.POINTER(HeadConsts), c_f,
                             ctypes.POINTER(Layer), c_i, ctypes.POINTER(HeadOutputs), c_f, c_sz, c_f]),
    "mpsr_heads_fwd_cams": (c_i, [c_f, c_i, c_i, c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_f, ctypes.POINTER(HeadConsts), c_f,
                                  ctypes.POINTER(Layer), c_i, ctypes.POINTER(HeadOutputs), c_f, c_sz, c_f]),
}


def lib():
    """Load libmonopsr_hip.so once -- AFTER torch, so that both resolve to ONE HIP runtime (torch ships its own
    libamdhip64; loaded second it would be a second runtime in the process, and a kernel launched through the first one
    on torch's memory fails with "no ROCm-capable device is detected": seen in r06 when build() loaded the library before
    smoke() imported torch)."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (the order matters, see above)
        if not os.path.exists(LIB_PATH):
            raise MpsrError(
                "libmonopsr_hip.so is not built (%s). Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C monopsr_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        missing = [name for name in SIGNATURES if not hasattr(handle, name)]
        if missing and not os.environ.get("MPSR_PARTIAL_LIB"):
            raise MpsrError("libmonopsr_hip.so lacks declared symbols %s; rebuild (make -C monopsr_amd/csrc)"
                            % missing)
        for name, (res, args) in SIGNATURES.items():
            if name in missing:
                continue
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        got = handle.mpsr_abi_version()
        if got != ABI_VERSION:
            raise MpsrError("libmonopsr_hip.so ABI %d != expected %d; rebuild" % (got, ABI_VERSION))
        _lib = handle
    return _lib


MATH_MODES = {"fp32": 0, "bf16x3": 1}


def get_conv_math():
    return {v: k for k, v in MATH_MODES.items()}[lib().mpsr_get_conv_math()]


def set_conv_math(mode):
    """Process-wide arithmetic of the convolution / FC contractions: "fp32" (default, exact) or "bf16x3" (opt-in:
    split-bfloat16 products with fp32 accumulation, see include/monopsr_hip.h).  Returns the previous mode."""
    names = {v: k for k, v in MATH_MODES.items()}
    prev = names[lib().mpsr_get_conv_math()]
    if mode not in MATH_MODES:
        raise InvalidArgumentError("unknown conv math mode %r (choose from %s)" % (mode, sorted(MATH_MODES)))
    check(lib().mpsr_set_conv_math(MATH_MODES[mode]))
    return prev


WINOGRAD_POLICIES = {"auto": 0, "off": 1, "accurate": 2}


def set_winograd_policy(policy):
    """Process-wide: "auto" (default: Winograd kernels wherever they are faster), "off" (direct / implicit-GEMM kernels
    everywhere: tightest element-wise error on heavy-tailed activations) or "accurate" (only the transform-domain forms
    that keep an element-wise 1e-3 on such maps: sixteen-product tiles, F(2x2,3x3); include/monopsr_hip.h).  Returns the
    previous policy."""
    names = {v: k for k, v in WINOGRAD_POLICIES.items()}
    prev = names[lib().mpsr_get_winograd_policy()]
    if policy not in WINOGRAD_POLICIES:
        raise InvalidArgumentError("unknown Winograd policy %r (choose from %s)" % (policy, sorted(WINOGRAD_POLICIES)))
    check(lib().mpsr_set_winograd_policy(WINOGRAD_POLICIES[policy]))
    return prev


def check(status):
    if status == 0:
        return
    msg = lib().mpsr_last_error().decode("utf-8", "replace")
    if status == 1:
        raise InvalidArgumentError(msg)
    raise MpsrError("status %d: %s" % (status, msg))


def ptr(t):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MpsrError("expected a tensor on the GPU; monopsr_amd has no CPU path")
    if not t.is_contiguous():
        raise MpsrError("expected a contiguous tensor")
    return t.data_ptr()


def stream(device_index=None):
    """The raw handle of torch's current stream on the current (or the given) device.  (Through the C binding: building a
    torch.cuda.Stream object per launch -- torch.cuda.current_stream() -- was the largest single item of the host's ~8 us
    per launch, r06 tools/host_profile.py: 11.6 -> 10.5 ms per training step of 8 instances, which is bound by the host.)"""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is None:
        return torch.cuda.current_stream(device_index).cuda_stream
    return raw(torch._C._cuda_getDevice() if device_index is None else device_index)
