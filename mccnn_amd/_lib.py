"""ctypes binding of the C-ABI in include/mccnn.h (libmccnn_hip.so).

The product path has NO fallback: if the HIP library is missing or cannot be loaded this
module raises, and every op in MCConvModule fails loudly.
"""
import ctypes as C
import os

import torch  # must be imported first: libmccnn_hip.so binds to the HIP runtime torch already loaded

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", os.environ.get("MCCNN_LIB_NAME", "libmccnn_hip.so"))  # override only for A/B experiments

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/mccnn.h declaration by declaration
SIGNATURES = {
    "mccnn_block_size": (_i, []),
    "mccnn_abi_version": (_i, []),
    "mccnn_arch": (C.c_char_p, []),
    "mccnn_error_string": (C.c_char_p, [_i]),
    "mccnn_check_batch_ids": (_i, [_vp, _i, _i, _vp, _vp]),
    "mccnn_debug_conv_impl": (_i, [_i]),
    "mccnn_debug_launch_count": (C.c_longlong, []),
    "mccnn_debug_wait_ns": (C.c_longlong, []),
    "mccnn_debug_wait_accounting": (_i, [_i]),
    "mccnn_debug_small_kernels": (_i, [_i]),
    "mccnn_debug_f1_x4_min_edges": (_i, [_i]),
    "mccnn_background_launches": (_i, [_i]),
    "mccnn_compute_aabb_workspace_bytes": (_sz, [_i]),
    "mccnn_compute_aabb": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "mccnn_num_cells": (_i, [_vp, _vp, _i, _f, _i, C.POINTER(_i), _vp]),
    "mccnn_aabb_extent": (_i, [_vp, _vp, C.POINTER(_f), _vp]),
    "mccnn_sort_step1_workspace_bytes": (_sz, [_i, _i, _i]),
    "mccnn_sort_step1": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "mccnn_sort_step2_workspace_bytes": (_sz, [_i]),
    "mccnn_sort_step2": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mccnn_sort_step1_dn": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "mccnn_sort_step2_dn": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mccnn_hierarchy_level_workspace_bytes": (_sz, [_i, _i, _i]),
    "mccnn_hierarchy_level": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _f, _i, _i] + [_vp] * 9 + [_vp, _sz, _vp]),
    "mccnn_geometry_build_batch": (_i, [_vp, _i, _vp]),
    "mccnn_geometry_prebuild_batch_ws_bytes": (_sz, [_vp, _vp, _i]),
    "mccnn_geometry_prebuild_batch": (_i, [_vp, _vp, _i, _i, _vp, _sz, _vp]),
    "mccnn_build_grid_workspace_bytes": (_sz, [_i, _i, _i]),
    "mccnn_build_grid": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mccnn_transform_indexs_dn": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "mccnn_permute_gather": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "mccnn_permute_scatter": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "mccnn_transform_indexs_workspace_bytes": (_sz, [_i]),
    "mccnn_transform_indexs": (_i, [_vp, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "mccnn_find_neighbors_workspace_bytes": (_sz, [_i, _i]),
    "mccnn_find_neighbors_count": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mccnn_find_neighbors_count2": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mccnn_find_neighbors_fill": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _f, _i, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "mccnn_invert_permutation": (_i, [_vp, _i, _vp, _vp]),
    "mccnn_compute_pdf_workspace_bytes": (_sz, [_i, _i]),
    "mccnn_compute_pdf": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _f, _f, _i, _i, _vp, _vp, _sz, _vp]),
    "mccnn_compute_pdf_dn": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _f, _f, _i, _vp, _vp, _sz, _vp]),
    "mccnn_poisson_sampling_workspace_bytes": (_sz, [_i, _i, _i]),
    "mccnn_poisson_sampling_count": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _f, _i, _i, _vp, _vp, _sz, _vp]),
    "mccnn_poisson_sampling_fill": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mccnn_spatial_conv_fwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "mccnn_spatial_conv_state_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "mccnn_spatial_conv_fwd": (_i, [_vp] * 15 + [_i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "mccnn_spatial_conv_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "mccnn_spatial_conv_bwd": (_i, [_vp] * 16 + [_i, _i, _i, _i, _i, _i, _i, _f, _i, _i] + [_vp] * 10 + [_vp, _sz, _vp]),
    "mccnn_spatial_conv_fwd_bf16": (_i, [_vp] * 15 + [_i, _i, _i, _i, _i, _f, _i, _i, _vp, _vp, _sz, _vp]),
    "mccnn_spatial_conv_bwd_bf16": (_i, [_vp] * 16 + [_i, _i, _i, _i, _i, _f, _i, _i] + [_vp] * 9 + [_vp, _sz, _vp]),
    "mccnn_rowplan_sizes": (_i, [_i, _i, C.POINTER(_i), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "mccnn_rowplan_workspace_bytes": (_sz, [_i, _i]),
    "mccnn_rowplan_layout": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mccnn_edge_records": (_i, [_vp] * 8 + [_i, _i, _i, _i, _f, _i, _i, _vp, _vp]),
    "mccnn_rowplan_fill": (_i, [_i, _vp, _vp, _i, _i] + [_vp] * 6 + [_vp, _vp, _vp]),
    "mccnn_rowplan_buffer": (_i, [_i, _i, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(_i),
                                  C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "mccnn_rowplan_build_workspace_bytes": (_sz, [_i, _i, _i]),
    "mccnn_rowplan_bound": (_i, [_i, _i, _i, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "mccnn_rowplan_inline_records": (_i, [_i, _i]),
    "mccnn_rowplan_build": (_i, [_i] + [_vp] * 8 + [_i, _i, _i, _i, _f, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "mccnn_spatial_conv_fwd_rows": (_i, [_vp] * 15 + [_i, _i, _i, _i, _i, _f, _i, _i, _i] + [_vp] * 6 + [_vp, _vp, _vp, _vp]),
    "mccnn_spatial_conv_bwd_rows_workspace_bytes": (_sz, [_i, _i, _i]),
    "mccnn_spatial_conv_bwd_rows": (_i, [_vp] * 16 + [_i, _i, _i, _i, _i, _f, _i, _i, _i] + [_vp] * 7 + [_vp] * 8 + [_vp, _vp, _sz, _vp]),
    "mccnn_transpose_neighbors_workspace_bytes": (_sz, [_i, _i]),
    "mccnn_transpose_neighbors": (_i, [_vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    # native step executor (exec.hip)
    "mccnn_geometry_create": (_vp, []),
    "mccnn_geometry_destroy": (None, [_vp]),
    "mccnn_geometry_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "mccnn_geometry_build": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _f, _i, _f, _i, _i, _vp, _vp, _sz, _vp, _vp]),
    "mccnn_geometry_edges": (_i, [_vp, _i]),
    "mccnn_geometry_info": (_i, [_vp, C.POINTER(C.c_longlong)]),
    "mccnn_geometry_attach": (_i, [_vp, _i, _vp, _sz]),
    "mccnn_geometry_piece_bytes": (_i, [_vp, _i, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "mccnn_geometry_piece_bound": (_i, [_i, _i, _i, _i, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "mccnn_geometry_prebuild": (_i, [_vp, _i, _i, _vp, _sz, _vp]),
    "mccnn_conv_prepare": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(C.c_longlong),
                                C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(_i)]),
    "mccnn_conv_forward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i] + [_vp] * 6 + [_vp, _vp, _sz, _vp, _sz, _vp]),
    "mccnn_conv_backward": (_i, [_vp, _vp, _vp, _sz, _vp, _i, _i, _i, _i, _i, _i] + [_vp] * 6 + [_vp] * 7 + [_vp, _sz, _vp]),
}


class MCCNNError(RuntimeError):
    """Raised for any non-zero return code of the C-ABI (the reference raised
    errors::InvalidArgument for shape errors and exit()ed on CUDA errors)."""


_lib = None


def load():
    """Load libmccnn_hip.so and type every exported symbol. Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MCCNNError(
            "libmccnn_hip.so not found at %s -- build it with `python -m mccnn_amd.build` "
            "(there is no CPU / PyTorch fallback for the MC-convolution ops)" % LIB_PATH)
    # An A/B library (MCCNN_LIB_NAME) is loaded with RTLD_GLOBAL: the torch extension links libmccnn_hip.so by name, and only
    # symbols of global scope, loaded BEFORE it, take precedence over that library's -- without this every layer that goes
    # through the extension runs the DEFAULT library's kernels whatever MCCNN_LIB_NAME says (round 5: a dozen void A/Bs).
    mode = C.RTLD_GLOBAL if os.environ.get("MCCNN_LIB_NAME") else C.DEFAULT_MODE
    lib = C.CDLL(LIB_PATH, mode=mode)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().mccnn_error_string(rc).decode()
        raise MCCNNError("%s failed: %s (code %d)" % (what, msg, rc))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_handle():
    """hipStream_t of torch's current stream. Called once per C-ABI launch (10-20 times per step): the two raw C calls
    cost ~0.5 us, torch.cuda.current_stream().cuda_stream ~9 us (device-index resolution, a Stream object) -- at a
    sub-millisecond step that difference was 10 % of the host path."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream
