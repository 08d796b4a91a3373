"""ctypes binding of libnerf_amd.so (C ABI: include/nerf_amd.h).

There is deliberately NO fallback: if the shared library is missing or a call
fails, a RuntimeError is raised.  The library is built in-tree by
``make -C nerf-simple_amd/csrc`` (or ``__graft_entry__.build()``).
"""
import ctypes
import os
import threading

# torch must be imported BEFORE the library is dlopen'ed: torch ships its own
# HIP runtime (torch/lib/libamdhip64.so) and device pointers only mean something
# inside the runtime that allocated them.  Loaded first, that runtime also
# satisfies libnerf_amd.so's libamdhip64 dependency; loaded second, the process
# would hold two runtimes and every launch would fail with hipErrorNoDevice.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnerf_amd.so")

F32, BF16, FP16, BF16_BWD = 0, 1, 2, 3
FLAG_TS_GIVEN, FLAG_DEVICE_RNG, FLAG_SEED_IN_MEMORY = 1, 2, 4
FLAG_STORE_E4M3 = 8          # nerf_amd_mlp_forward_train: the 8-bit storage form of the saved activations
STATUS_NONFINITE, STATUS_WEIGHT_RANGE = 1, 2
_PRECISIONS = {"fp32": F32, "f32": F32, "float32": F32, F32: F32,
               "bf16": BF16, "bfloat16": BF16, BF16: BF16,
               "fp16": FP16, "f16": FP16, "float16": FP16, "half": FP16, FP16: FP16,
               "bf16_bwd": BF16_BWD, BF16_BWD: BF16_BWD}     # the backward kernel's transposed image

_lib = None
_lock = threading.Lock()

_vp, _i64, _i32, _u32, _u64 = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                               ctypes.c_uint32, ctypes.c_uint64)
_SIGNATURES = {
    # name: (restype, argtypes)
    "nerf_amd_abi_version": (_i32, []),
    "nerf_amd_param_count": (_i64, []),
    "nerf_amd_packed_bytes": (_i64, [_i32]),
    "nerf_amd_render_workspace_bytes": (_i64, [_i32, _i64, _i32]),
    "nerf_amd_packed_status_offset": (_i64, [_i32]),
    "nerf_amd_query_points": (_i32, [_vp, _vp, _vp, _u32, _u64, _i64, _vp, _vp, _i64, _i32, _vp]),
    "nerf_amd_grad_bucket_range": (_i32, [_i32, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "nerf_amd_param_gradients_finish_bucket": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "nerf_amd_layout_selfcheck": (_i32, []),
    "nerf_amd_layout_src_col": (_i32, [_i32, _i32, _i32, _i32, _i32]),
    "nerf_amd_pack_weights": (_i32, [_vp, _vp, _i32, _vp]),
    "nerf_amd_gamma": (_i32, [_vp, _i64, _vp, _i64, _i32, _vp]),
    "nerf_amd_positional_encoder": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "nerf_amd_mlp_forward": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "nerf_amd_volume_render": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "nerf_amd_volume_render_pixels": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "nerf_amd_volume_render_backward": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "nerf_amd_volume_render_rays": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "nerf_amd_volume_render_rays_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "nerf_amd_volume_render_mse_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "nerf_amd_param_gradients_begin": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "nerf_amd_param_gradients_finish": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "nerf_amd_pack_weights_train": (_i32, [_vp, _vp, _vp, _vp]),
    "nerf_amd_mse_loss": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "nerf_amd_mlp_forward_train_points": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "nerf_amd_encode_points_bf16": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "nerf_amd_sample_encode": (_i32, [_vp, _vp, _vp, _u32, _u64, _i64, _vp, _vp, _vp, _i64, _i32, _vp]),
    "nerf_amd_generate_rays": (_i32, [_vp, _i32, _i32, ctypes.c_float, _i64, _i64, _vp, _vp]),
    "nerf_amd_render_image_workspace_bytes": (_i64, [_i32, _i64, _i32]),
    "nerf_amd_render_image_forward": (_i32, [_vp, _i32, _i32, ctypes.c_float, _i64, _i64, _vp, _vp, _vp, _i32,
                                             _u32, _u64, _vp, _vp, _i32, _vp]),
    "nerf_amd_sample_pdf": (_i32, [_vp, _vp, _vp, _u32, _u64, _i64, _vp, _i64, _i32, _i32, _vp]),
    "nerf_amd_render_hierarchical_workspace_bytes": (_i64, [_i64, _i32, _i32]),
    "nerf_amd_render_hierarchical_forward": (_i32, [_vp, _i32, _i32, ctypes.c_float, _i64, _i64, _vp, _vp, _vp, _vp, _vp,
                                                    _i32, _u32, _u64, _vp, _vp, _i32, _i32, _vp]),
    "nerf_amd_train_activation_bytes": (_i64, [_i64]),
    "nerf_amd_mlp_forward_train": (_i32, [_vp, _vp, _vp, _vp, _u32, _u64, _i64, _vp, _vp, _vp, _i64, _i32, _vp]),
    "nerf_amd_mlp_backward": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "nerf_amd_train_activation_bytes_e4m3": (_i64, [_i64]),
    "nerf_amd_train_gradient_bytes_e4m3": (_i64, [_i64]),
    "nerf_amd_param_gradients_scratch_e4m3_bytes": (_i64, [_i64]),
    "nerf_amd_mlp_backward_e4m3": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "nerf_amd_param_gradients_convert_e4m3": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "nerf_amd_param_gradients_finish_e4m3": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "nerf_amd_sample_encode_bf16": (_i32, [_vp, _vp, _vp, _u32, _u64, _i64, _vp, _vp, _vp, _i64, _i32, _vp]),
    "nerf_amd_param_gradients_scratch_bytes": (_i64, [_i64]),
    "nerf_amd_param_gradients": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "nerf_amd_adam_step": (_i32, [_vp, _vp, _vp, _vp, _i64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                  ctypes.c_float, _i64, _vp]),
    "nerf_amd_mt19937_uniform": (_i32, [_vp, _i32, _vp, _i64, _vp, _vp]),
    "nerf_amd_mt19937_segments": (_i64, [_i32, _i64, _i64]),
    "nerf_amd_mt19937_uniform_par": (_i32, [_vp, _i32, _vp, _i64, _vp, _vp, _i32, _i64, _vp, _vp]),
    "nerf_amd_adam_step_hyper": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "nerf_amd_hyper_fetch": (_i32, [_vp, _i32, _vp, _vp, _vp]),
    "nerf_amd_pinned_device_address": (_i64, [_vp]),
    "nerf_amd_range_check": (_i32, [_vp, _vp, _vp, _vp, _u32, _u64, _i64, _vp, _i64, _i32, _vp]),
    "nerf_amd_mt19937_raw": (_i32, [_vp, _i32, _vp, _i64, _vp, _vp]),
    "nerf_amd_mt19937_jump_poly": (_i32, [_i64, _vp, _vp]),
    "nerf_amd_mt19937_advance": (_i32, [_vp, _vp, _vp, _vp]),
    "nerf_amd_mt19937_uniform_after": (_i32, [_vp, _vp, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp]),
    "nerf_amd_select_workspace_bytes": (_i64, [_i64]),
    "nerf_amd_select_rays": (_i32, [_vp, _u64, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "nerf_amd_linear_f32": (_i32, [_vp, _i64, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _u32, _vp]),
    "nerf_amd_render_forward": (_i32, [_vp, _vp, _vp, _vp, _i32, _u32, _u64, _i64,
                                       _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "nerf_amd_render_pixels_forward": (_i32, [_vp, _vp, _vp, _vp, _i32, _u32, _u64, _i64,
                                              _vp, _vp, _i64, _i32, _vp]),
    "nerf_amd_mlp_forward_rays": (_i32, [_vp, _vp, _vp, _vp, _i32, _u32, _u64, _i64,
                                         _vp, _vp, _i64, _i32, _vp]),
}
EXPORTS = tuple(_SIGNATURES)


def precision_code(p):
    try:
        return _PRECISIONS[p]
    except KeyError:
        raise ValueError(f"precision must be 'fp32', 'bf16' or 'fp16', got {p!r}") from None


def lib():
    """The loaded library (loads on first use; raises if it is not built)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"{LIB_PATH} is missing: build the HIP library with "
                        "`make -C nerf-simple_amd/csrc` (there is no CPU fallback)")
                h = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in _SIGNATURES.items():
                    fn = getattr(h, name)          # AttributeError if an export is missing
                    fn.restype, fn.argtypes = res, args
                if h.nerf_amd_abi_version() != 5:
                    raise RuntimeError("libnerf_amd.so ABI version mismatch")
                _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "unsupported configuration"}.get(rc, f"hipError_t {rc}")
        raise RuntimeError(f"{what} failed: {kind}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda_f32(t, name):
    """The kernels take fp32, contiguous, device-resident tensors; anything
    else is an error (the reference's callers do .cuda() themselves,
    utils/rendering.py:102, train.py:51)."""
    if not torch.is_tensor(t):
        raise AssertionError(f"{name} needs to be a torch tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got a {t.device} tensor); "
                           "this package has no CPU path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32, got {t.dtype}")
    return t
