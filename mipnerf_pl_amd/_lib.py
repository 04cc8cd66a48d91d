"""ctypes binding of libmipnerf_hip.so (the C ABI declared in include/mipnerf_hip.h).

The library is built in-tree by `python -m mipnerf_pl_amd.build` (hipcc --offload-arch=gfx950).
There is no CPU fallback: if the shared object is missing, loading fails loudly."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIPNERF_LIB", os.path.join(HERE, "csrc", "libmipnerf_hip.so"))   # override: A/B builds

OK, E_INVALID, E_UNSUPPORTED, E_HIP, E_WORKSPACE = 0, 1, 2, 3, 4
PREC_FP32, PREC_BF16 = 0, 1
OUT_BF16_FRAGMENTS = 2      # out_dtype of mipnerf_cast_ipe_360 only (include/mipnerf_hip.h)
FLAG_WHITE_BKGD, FLAG_DISPARITY = 1, 2
NUM_PARAM_TENSORS = 24
MAX_SAMPLES = 1024


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "num_samples", "num_levels", "min_deg_point", "max_deg_point", "deg_view", "use_viewdirs",
        "disparity", "disable_integration", "net_depth", "net_width", "net_depth_condition",
        "net_width_condition", "skip_index", "num_rgb_channels", "num_density_channels")] + [
        ("resample_padding", C.c_float), ("density_bias", C.c_float), ("rgb_padding", C.c_float),
        ("density_noise", C.c_float), ("unbounded", C.c_int32)]


class RaysPtrs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("origins", "directions", "viewdirs", "radii", "lossmult", "near", "far")]


class LrSchedule(C.Structure):
    _fields_ = [("lr_init", C.c_double), ("lr_final", C.c_double), ("lr_delay_mult", C.c_double), ("constant_lr", C.c_double),
                ("max_steps", C.c_int64), ("lr_delay_steps", C.c_int64), ("beta1", C.c_double), ("beta2", C.c_double),
                ("eps", C.c_double), ("grad_scale", C.c_float), ("reserved", C.c_int32)]


class LevelOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("comp_rgb", "distance", "acc", "weights", "t_samples")]


_P, _I32, _I64, _F, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); every symbol include/mipnerf_hip.h declares
SIGNATURES = {
    "mipnerf_last_error": (C.c_char_p, []),
    "mipnerf_abi_version": (C.c_int, []),
    "mipnerf_num_param_tensors": (C.c_int, [_P]),
    "mipnerf_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "mipnerf_destroy": (C.c_int, [_P]),
    "mipnerf_compiled_arch": (C.c_int, [C.POINTER(Config)]),
    "mipnerf_num_variants": (C.c_int, []),
    "mipnerf_variant_arch": (C.c_int, [C.c_int, C.POINTER(Config), C.POINTER(C.c_int)]),
    "mipnerf_set_params": (C.c_int, [_P, C.POINTER(_P), _P]),
    "mipnerf_workspace_bytes": (_SZ, [_P, _I64]),
    "mipnerf_forward": (C.c_int, [_P, _I64, C.POINTER(RaysPtrs), _P, _P, _P, C.c_uint32, C.c_int, _P, _SZ,
                                  C.POINTER(LevelOut), _P]),
    "mipnerf_sample_along_rays": (C.c_int, [_I64, _I32, _P, _P, _P, _I32, _P, _P]),
    "mipnerf_cast_rays": (C.c_int, [_I64, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "mipnerf_cast_ipe": (C.c_int, [_I64, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, C.c_int, _P]),
    "mipnerf_integrated_pos_enc": (C.c_int, [_I64, _I32, _I32, _P, _P, _P, C.c_int, _P]),
    "mipnerf_pos_enc": (C.c_int, [_I64, _I32, _P, _P, _I32, C.c_int, _P]),
    "mipnerf_mlp_forward": (C.c_int, [_P, _I64, _I32, _P, _P, C.c_int, _P, _P, _P]),
    "mipnerf_volumetric_rendering": (C.c_int, [_I64, _I32, _P, _P, _P, _I32, _P, _P, _P, _P, _P]),
    "mipnerf_resample_along_rays": (C.c_int, [_I64, _I32, _P, _P, _P, _F, _P, _P]),
    "mipnerf_sorted_piecewise_constant_pdf": (C.c_int, [_I64, _I32, _P, _P, _I32, _P, _P, _P]),
    "mipnerf_sample_along_rays_360": (C.c_int, [_I64, _I32, _P, _P, _P, _P, _P, _P]),
    "mipnerf_cast_ipe_360": (C.c_int, [_I64, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P]),
    "mipnerf_gauss_360": (C.c_int, [_I64, _I32, _I32, _I32, _P, _P, _P, C.c_int, _P, _P, _P]),
    "mipnerf_generate_rays": (C.c_int, [_I64, _P, _P, _P, C.POINTER(RaysPtrs), _P]),
    "mipnerf_generate_rays_f64": (C.c_int, [_I64, _P, _P, _P, C.POINTER(RaysPtrs), _P]),
    "mipnerf_eval_workspace_floats": (_I64, [_I32, _I32]),
    "mipnerf_eval_errors": (C.c_int, [_I32, _I32, _P, _P, _P, _P, _P]),
    "mipnerf_activate": (C.c_int, [_I64, _P, _F, _F, _P, _F, _P, _P]),
    "mipnerf_volumetric_rendering_bwd": (C.c_int, [_I64, _I32, _P, _P, _P, _I32, _P, _P, _P, _P, _F, _P, _P]),
    "mipnerf_distloss": (C.c_int, [_I64, _I32, _P, _P, _P, _P, _P, _P]),
    "mipnerf_volumetric_rendering_bwd_t": (C.c_int, [_I64, _I32, _P, _P, _P, _I32, _P, _P, _P, _P, _F, _P, _P, _P]),
    "mipnerf_distloss_bwd": (C.c_int, [_I64, _I32, _P, _P, _P, _P, _P, _P]),
    "mipnerf_cast_ipe_bwd": (C.c_int, [_I64, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "mipnerf_resample_along_rays_bwd": (C.c_int, [_I64, _I32, _P, _P, _P, _F, _P, _P, _P]),
    "mipnerf_mlp_train_sizes": (C.c_int, [_P, _I64, C.POINTER(_SZ), C.POINTER(_SZ), C.POINTER(_SZ), C.POINTER(_SZ)]),
    "mipnerf_mlp_forward_train": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "mipnerf_mlp_forward_train_fragments": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "mipnerf_mlp_backward": (C.c_int, [_P, _I64, _P, _P, _P, _P, _P, _P, _I32, _P]),
    "mipnerf_adam_step": (C.c_int, [_I64, _P, _P, _P, _P, C.c_double, C.c_double, C.c_double, C.c_double, _I32, _P]),
    "mipnerf_adam_step_scheduled": (C.c_int, [_I64, _P, _P, _P, _P, C.POINTER(LrSchedule), _P, _P, _P]),
    "mipnerf_mlp_dgrad": (C.c_int, [_P, _I64, _P, _P, _P, _P]),
    "mipnerf_mlp_wgrad": (C.c_int, [_P, _I64, _P, _P, _P, _P, _I32, _P]),
    "mipnerf_set_wgrad_splits": (C.c_int, [_P, _P]),
    "mipnerf_mlp_train_f32_bytes": (_SZ, [_P, _I64, C.POINTER(_SZ), C.POINTER(_SZ)]),
    "mipnerf_mlp_forward_train_f32": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _P, _P]),
    "mipnerf_mlp_backward_f32": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _P, _P, _I32, _P]),
    "mipnerf_mlp_backward_f32_enc": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _P, _P, _I32, _P, _P]),
    "mipnerf_train_workspace_bytes": (_SZ, [_P, _I64]),
    "mipnerf_train_step": (C.c_int, [_P, _I64, C.POINTER(RaysPtrs), _P, _P, _P, _P, C.c_uint32, _F, _F, _I32, _P, _SZ, _P, _I32, _P,
                                     C.POINTER(LevelOut), _P]),
    "mipnerf_time_mlp": (C.c_int, [_P, _I64, _I32, _P, _P, C.c_int, _P, C.c_int, C.POINTER(_F), _P]),
    "mipnerf_selftest": (C.c_int, [_P]),
    "mipnerf_set_option": (C.c_int, [_P, C.c_int, C.c_int]),
    "mipnerf_mlp_launch_stats": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(_I64)]),
    "mipnerf_debug_table": (_I64, [C.c_int, _P, _I64]),
    "mipnerf_debug_table_variant": (_I64, [C.c_int, C.c_int, _P, _I64]),
    "mipnerf_debug_f32net": (_I64, [_P, _I64]),
}

# include/mipnerf_diag.h: measurement tooling in its own library (bench.py, scripts/handoff_probe.py); never needed by the product path
DIAG_LIB_PATH = os.environ.get("MIPNERF_DIAG_LIB", os.path.join(HERE, "csrc", "libmipnerf_diag.so"))
DIAG_SIGNATURES = {
    "mipnerf_diag_last_error": (C.c_char_p, []),
    "mipnerf_mfma_ceiling": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double), _P]),
    "mipnerf_handoff_probe": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), _P]),
}

_lib = None
_diag = None


def diag_lib():
    """libmipnerf_diag.so or None when it was not built (callers report the diagnostics as absent)."""
    global _diag
    if _diag is None and os.path.exists(DIAG_LIB_PATH):
        try:
            import torch  # noqa: F401  (same HIP runtime instance, see lib())
        except ImportError:
            pass
        h = C.CDLL(DIAG_LIB_PATH)
        for name, (res, args) in DIAG_SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _diag = h
    return _diag


def diag_check(rc: int, what: str = "") -> None:
    if rc != OK:
        msg = (diag_lib().mipnerf_diag_last_error() or b"").decode(errors="replace")
        raise (ValueError if rc == E_INVALID else RuntimeError)(f"{what}: {msg} (code {rc})")


def lib():
    """Load (once) and return the ctypes handle with argtypes set."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the MI355X-native Mip-NeRF path has no CPU fallback. "
                "Build it with `python -m mipnerf_pl_amd.build` (needs hipcc / ROCm).")
        try:
            # PyTorch-ROCm bundles its own HIP runtime (soname-less libamdhip64.so): map it FIRST so the
            # library's DT_NEEDED "libamdhip64.so" binds to the same runtime instance torch uses
            # (one runtime per process => torch streams / synchronize() cover our kernels).
            import torch  # noqa: F401
        except ImportError:
            pass
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)       # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def last_error() -> str:
    return (lib().mipnerf_last_error() or b"").decode(errors="replace")


def check(rc: int, what: str = "") -> None:
    """Turn a C error code into the exception the reference would raise."""
    if rc == OK:
        return
    msg = f"{what}: {last_error()}" if what else last_error()
    if rc == E_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == E_INVALID:
        raise ValueError(msg)
    raise RuntimeError(f"{msg} (code {rc})")
