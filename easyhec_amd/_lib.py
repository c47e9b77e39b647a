"""ctypes binding of libehr_hip.so (include/ehr.h).  There is NO fallback: if the library is missing, or a tensor
is not on a HIP device, the call raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EHR_LIB") or os.path.join(_HERE, "libehr_hip.so")  # EHR_LIB: A/B builds of the same ABI
_lib = None

c_void_p, c_int, c_size_t, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_float

# name -> (restype, argtypes); mirrors include/ehr.h one to one
SIGNATURES = {
    "ehr_version": (c_int, []),
    "ehr_last_error": (ctypes.c_char_p, []),
    "ehr_device_count": (c_int, []),
    "ehr_device_arch": (ctypes.c_char_p, [c_int]),
    "ehr_ctx_create": (c_int, [c_int, ctypes.POINTER(c_void_p)]),
    "ehr_ctx_destroy": (c_int, [c_void_p]),
    "ehr_ctx_scratch_bytes": (c_size_t, [c_void_p]),
    "ehr_tile_flags_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ehr_rasterize_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                  c_void_p, c_void_p, c_void_p]),
    "ehr_rasterize_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p]),
    "ehr_rasterize_grad_db": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_void_p, c_void_p]),
    "ehr_interpolate_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                    c_void_p, c_void_p, c_void_p]),
    "ehr_interpolate_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ehr_interpolate_da_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p, c_void_p]),
    "ehr_interpolate_da_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 8 +
                                [c_void_p, c_void_p, c_void_p]),
    "ehr_topology_scratch_bytes": (c_size_t, [c_int]),
    "ehr_antialias_topology": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ehr_antialias_work_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ehr_antialias_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ehr_antialias_fwd_zg": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                     c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ehr_antialias_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ehr_fused_plan": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    "ehr_render_mask_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    "ehr_fused_status": (c_int, [c_void_p]),
    "ehr_fused_bind_ref": (c_int, [c_void_p, c_void_p, c_void_p]),
    "ehr_pose_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ehr_pose_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                  c_float, c_float, c_void_p, c_void_p]),
    "ehr_pose_adam": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float,
                              c_float, c_void_p, c_void_p, c_void_p]),
    "ehr_solver_step": (c_int, [c_void_p] * 9 + [c_int] * 6 + [c_float] * 2 + [c_void_p] * 5 + [c_int, c_void_p] +
                        [c_float] * 5 + [c_void_p] * 8 + [c_int, c_void_p]),
    "ehr_mask_variance": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p, c_void_p, c_int, c_void_p]),
    "ehr_comm_unique_id": (c_int, [c_void_p]),
    "ehr_comm_init": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "ehr_comm_allreduce": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "ehr_comm_destroy": (c_int, [c_void_p]),
    "ehr_comm_p2p_export": (c_int, [c_void_p, c_void_p]),
    "ehr_comm_p2p_open": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "ehr_comm_p2p_step": (c_int, [c_void_p] * 6 + [c_float] * 5 + [c_void_p, c_void_p, c_void_p]),
    "ehr_comm_p2p_close": (c_int, [c_void_p]),
    "ehr_graph_begin": (c_int, [c_void_p, ctypes.POINTER(c_void_p)]),
    "ehr_graph_end": (c_int, [c_void_p]),
    "ehr_graph_launch": (c_int, [c_void_p, c_void_p]),
    "ehr_graph_release": (c_int, [c_void_p]),
    "ehr_fused_timing": (c_int, [c_void_p, c_int]),
    "ehr_fused_timing_read": (c_int, [c_void_p, ctypes.POINTER(c_float), ctypes.POINTER(c_int)]),
}


def lib():
    """Load libehr_hip.so; raises (loudly) if it has not been built -- there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m easyhec_amd.build` (hipcc --offload-arch=gfx950). "
                "easyhec_amd has no CPU or PyTorch fallback for the render path.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the .so does not export a symbol the header declares
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


EHR_ERR_OVERFLOW = -3  # include/ehr.h
EHR_ERR_RETRY = -4


def check(rc, what):
    if rc != 0:
        msg = lib().ehr_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what}: {msg} (code {rc})")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())
