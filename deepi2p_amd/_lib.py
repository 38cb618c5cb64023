"""ctypes binding of libdeepi2p_hip.so (include/deepi2p_hip.h).

There is NO CPU fallback: if the HIP library cannot be loaded every operator raises.  PyTorch is
used only for device memory and streams; every compute kernel is ours.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DI2P_LIB: development knob -- load another build of the SAME library (A/B runs of kernel variants on one GPU box); still no fallback
LIB_PATH = os.environ.get("DI2P_LIB") or os.path.join(_HERE, "lib", "libdeepi2p_hip.so")
_lib = None

c_void_p, c_int, c_float, c_double, c_ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_longlong


class DeepI2PHipError(RuntimeError):
    pass


class SrcT(ctypes.Structure):
    _fields_ = [("ptr", c_void_p), ("gidx", c_void_p), ("batch_stride", c_ll), ("row_stride", c_int),
                ("channels", c_int), ("mode", c_int), ("group", c_int), ("pad_", c_int)]


class EpilogueT(ctypes.Structure):
    _fields_ = [("scale", c_void_p), ("shift", c_void_p), ("batch_bias", c_void_p), ("relu", c_int),
                ("group_max", c_int), ("g_table", c_void_p * 2), ("g_idx", c_void_p * 2), ("g_w", c_void_p * 2),
                ("g_nodes", c_int * 2), ("g_k", c_int * 2), ("transpose_out", c_int), ("group_max_out", c_void_p),
                ("planes_out", c_void_p)]


class HeadX3T(ctypes.Structure):      # di2p_head_x3_t
    _fields_ = [("src", c_void_p * 2), ("batch_stride", c_ll * 2), ("row_stride", c_int * 2), ("channels", c_int * 2),
                ("W0p", c_void_p), ("W1p", c_void_p), ("scale_shift", c_void_p), ("relu0", c_int), ("relu1", c_int),
                ("tab", c_void_p * 2), ("idx", c_void_p * 2), ("w", c_void_p * 2), ("nodes", c_int * 2),
                ("W2t", c_void_p), ("scale2", c_void_p), ("shift2", c_void_p), ("relu2", c_int), ("P", c_int)]


SRC_DENSE, SRC_GATHER, SRC_GROUP = 0, 1, 2

# name -> argtypes (all return int)
_SIGS = {
    "di2p_index_max_forward": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "di2p_index_max_values": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "di2p_ball_query_forward": [c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int, c_void_p],
    "di2p_knn_nodes": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "di2p_cluster_stats": [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "di2p_build_point_input": [c_void_p] * 7 + [c_int, c_int, c_int, c_void_p],
    "di2p_interpolate": [c_void_p] * 4 + [c_int] * 5 + [c_void_p],
    "di2p_gather_neighbors": [c_void_p] * 4 + [c_int] * 4 + [c_void_p],
    "di2p_argmax_channels": [c_void_p, c_void_p, c_int, c_int, c_int, c_ll, c_void_p],
    "di2p_pointwise_gemm": [ctypes.POINTER(SrcT), c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                            ctypes.POINTER(EpilogueT), c_void_p],
    "di2p_pointwise_gemm_x3": [ctypes.POINTER(SrcT), c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                               ctypes.POINTER(EpilogueT), c_void_p],
    "di2p_pointwise_gemm_x3p": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, ctypes.POINTER(EpilogueT), c_void_p],
    "di2p_bf16x3_pack": [c_void_p, c_int, c_int, c_void_p, c_void_p],
    "di2p_bf16x3_pack_conv3x3": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "di2p_batch_gemv": [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p],
    "di2p_batch_gemv2": [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p],
    "di2p_attention_pool": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "di2p_point_head": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                        c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "di2p_point_chain": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                         c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p],
    "di2p_conv2d": [c_void_p] * 6 + [c_int] * 11 + [c_void_p],
    "di2p_conv2d_ws": [c_void_p] * 6 + [c_int] * 11 + [c_void_p, c_ll, c_void_p],
    "di2p_maxpool3x3s2": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "di2p_global_avgpool": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "di2p_channel_max": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "di2p_initial_guess": [c_void_p] * 5 + [c_int, c_int, c_void_p],
    "di2p_solve_batched": [c_void_p] * 6 + [c_double, c_double, ctypes.POINTER(c_double), ctypes.POINTER(c_double)]
                          + [c_int] * 5 + [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "di2p_solve_batched_f32": [c_void_p] * 6 + [c_double, c_double, ctypes.POINTER(c_double), ctypes.POINTER(c_double)]
                              + [c_int] * 5 + [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "di2p_select_best": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "di2p_solver_residuals": [c_void_p] * 4 + [c_double, c_double, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "di2p_f32_to_f64": [c_void_p, c_void_p, c_ll, c_void_p],
    "di2p_farthest_point_sampling": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "di2p_gather_points": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "di2p_project_labels": [c_void_p, c_void_p, c_int, c_void_p, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "di2p_label_accuracy": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "di2p_pnp_ransac": [c_void_p] * 5 + [c_int, c_void_p, c_int, c_double, c_int, c_int, c_int, c_int] + [c_void_p] * 7,
    "di2p_draw_restarts": [ctypes.c_ulonglong, c_int, c_int, c_double, c_double, c_void_p, c_void_p, c_void_p],
    "di2p_random_choice": [ctypes.c_ulonglong, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "di2p_pnp_pack": [c_void_p] * 4 + [c_int, c_int, c_int] + [c_void_p] * 3,
    "di2p_pnp_ransac_epnp": [c_void_p] * 5 + [c_int, c_void_p, c_int, c_double, c_int, c_int] + [c_void_p] * 7,
    "di2p_classifier_loss": [c_void_p] * 4 + [c_int, c_int, c_int, c_float, c_float, c_float] + [c_void_p] * 5,
    "di2p_adam_step": [c_void_p] * 4 + [c_ll, c_int, c_float, c_float, c_float, c_float, c_void_p],
    "di2p_pack_pc_label": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "di2p_stem_pack": [c_void_p, c_void_p, c_void_p],
    "di2p_conv7x7s2_stem": [c_void_p] * 5 + [c_int] * 4 + [c_void_p],
    "di2p_winograd_weight_transform": [c_void_p, c_void_p, c_int, c_int, c_void_p],
    "di2p_winograd_weight_transform_dgrad": [c_void_p, c_void_p, c_int, c_int, c_void_p],
    "di2p_conv3x3_winograd": [c_void_p] * 6 + [c_int] * 6 + [c_void_p],
    "di2p_conv3x3_x3": [c_void_p] * 6 + [c_int] * 7 + [c_void_p] * 5,
    "di2p_conv3x3_x3_supported": [c_int] * 6,
    "di2p_head_x3_pack": [c_void_p, c_int, c_void_p, c_void_p],
    "di2p_stem_x3_pack": [c_void_p, c_void_p, c_void_p],
    "di2p_stem_x3_supported": [c_int, c_int],
    "di2p_stem_x3": [c_void_p] * 5 + [c_int] * 3 + [c_void_p],
    "di2p_point_head_x3": [ctypes.POINTER(HeadX3T), c_void_p, c_int, c_int, c_void_p],
    "di2p_bn_train_forward": [c_void_p] * 9 + [c_float, c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "di2p_bn_train_backward": [c_void_p] * 6 + [c_int] + [c_void_p] * 4 + [c_int, c_int, c_int, c_void_p, c_void_p],
    "di2p_channel_sum": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "di2p_bmm_rc": [c_void_p, c_ll, c_ll, c_void_p, c_ll, c_ll, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_ll, c_void_p],
    "di2p_bmm_km": [c_void_p, c_int, c_ll, c_void_p, c_int, c_ll, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p],
    "di2p_gather_backward": [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_ll, c_void_p],
    "di2p_conv2d_wgrad": [c_void_p] * 3 + [c_int] * 9 + [c_void_p, c_ll, c_void_p],
    "di2p_conv2d_dgrad": [c_void_p] * 3 + [c_int] * 9 + [c_void_p],
    "di2p_maxpool3x3s2_backward": [c_void_p] * 3 + [c_int] * 4 + [c_void_p],
    "di2p_segment_max_backward": [c_void_p] * 4 + [c_int] * 4 + [c_void_p],
    "di2p_group_max_forward": [c_void_p, c_void_p, c_void_p, c_ll, c_int, c_void_p],
    "di2p_group_max_backward": [c_void_p, c_void_p, c_void_p, c_ll, c_int, c_void_p],
    "di2p_dropout_mask": [ctypes.c_ulonglong, c_int, c_float, c_ll, c_void_p, c_void_p],
    "di2p_apply_mask": [c_void_p, c_void_p, c_float, c_void_p, c_ll, c_void_p],
}
_WS_SIGS = {        # <name>_workspace_bytes helpers returning long long
    "di2p_channel_reduce_workspace_bytes": [c_int] * 3,
    "di2p_bmm_rc_workspace_bytes": [c_int] * 4,
    "di2p_gather_backward_workspace_bytes": [c_int] * 4,
    "di2p_conv2d_wgrad_workspace_bytes": [c_int] * 9,
    "di2p_bf16x3_packed_bytes": [c_int] * 2,
    "di2p_bf16x3_planes_bytes": [c_int] * 3,
    "di2p_head_x3_packed_bytes": [c_int],
    "di2p_stem_x3_packed_bytes": [],
}
EXPORTS = sorted(list(_SIGS) + ["di2p_last_error", "di2p_version", "di2p_solve_workspace_bytes", "di2p_solver_set_profile_buffer", "di2p_pnp_workspace_bytes",
                 "di2p_conv2d_workspace_bytes", "di2p_set_option", "di2p_get_option", "di2p_random_choice_workspace_bytes", "di2p_classifier_loss_workspace_bytes"] + list(_WS_SIGS))


def load():
    """Load (once) and return the ctypes library; raises DeepI2PHipError if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DeepI2PHipError(
                "libdeepi2p_hip.so not found at %s -- build it with `python -m deepi2p_amd.build` "
                "(there is no CPU fallback)" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = c_int
        lib.di2p_last_error.restype = ctypes.c_char_p
        lib.di2p_version.restype = c_int
        lib.di2p_solve_workspace_bytes.restype = c_ll
        lib.di2p_solve_workspace_bytes.argtypes = [c_int, c_int, c_int]
        lib.di2p_pnp_workspace_bytes.restype = c_ll
        lib.di2p_pnp_workspace_bytes.argtypes = [c_int, c_int, c_int]
        lib.di2p_conv2d_workspace_bytes.restype = c_ll
        lib.di2p_conv2d_workspace_bytes.argtypes = [c_int] * 10
        lib.di2p_classifier_loss_workspace_bytes.restype = c_ll
        lib.di2p_classifier_loss_workspace_bytes.argtypes = [c_int, c_int]
        lib.di2p_random_choice_workspace_bytes.restype = c_ll
        lib.di2p_random_choice_workspace_bytes.argtypes = [c_int, c_int]
        for name, args in _WS_SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = c_ll
        lib.di2p_set_option.restype = c_int
        lib.di2p_set_option.argtypes = [ctypes.c_char_p, c_ll]
        lib.di2p_get_option.restype = c_ll
        lib.di2p_get_option.argtypes = [ctypes.c_char_p]
        lib.di2p_solver_set_profile_buffer.restype = None
        lib.di2p_solver_set_profile_buffer.argtypes = [c_void_p]
        _lib = lib
    return _lib


# Optional per-launch timing (bench.py): {kernel name: [(start_event, end_event, tag), ...]}.  Events are
# recorded on the stream the kernel is launched on (torch's current stream).
TIMED = None
TIMED_TAG = None
# Optional executed-work counters (bench.py): {kernel name: multiply-accumulates issued}, filled by ops.* when not None.
WORK = None


def call(name, *args):
    lib = load()
    rec = TIMED is not None and name in TIMED
    if rec:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = getattr(lib, name)(*args)
    if rec:
        e1.record()
        TIMED[name].append((e0, e1, TIMED_TAG))
    if rc != 0:
        raise DeepI2PHipError("%s failed (%d): %s" % (name, rc, lib.di2p_last_error().decode()))


def set_option(name, value):
    """Set a cached library knob ("conv_nosplit", "pw_novec", "solver_nocull", ...; include/deepi2p_hip.h)."""
    if load().di2p_set_option(name.encode(), int(value)) != 0:
        raise DeepI2PHipError("unknown option " + name)


def get_option(name):
    return int(load().di2p_get_option(name.encode()))


class option:
    """``with _lib.option("solver_nocull", 1): ...`` -- temporarily override a knob (tests compare code paths)."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = get_option(self.name)
        set_option(self.name, self.value)

    def __exit__(self, *exc):
        set_option(self.name, self.old)


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    """Mirror of the reference's CHECK_INPUT (index_max.cpp:119-121): CUDA + contiguous, RuntimeError otherwise."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("tensor must be a CUDA tensor/variable")
        if not t.is_contiguous():
            raise RuntimeError("tensor must be contiguous")
