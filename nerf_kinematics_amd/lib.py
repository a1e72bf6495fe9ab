"""ctypes binding of ``libnerfkin.so`` (the C ABI declared in ``include/nerf_kinematics.h``).

The library is the product: there is no Python or CPU fallback.  If the shared object is
missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NK_LIB (development only): another build of the library, e.g. to compare two builds bit by bit
LIB_PATH = os.environ.get("NK_LIB") or os.path.join(_HERE, "libnerfkin.so")

NK_LINEAGE_RUN_NERF = 0
NK_ENC_NO_INPUT_XYZ, NK_ENC_NO_INPUT_DIR, NK_ENC_LINEAR_XYZ, NK_ENC_LINEAR_DIR = 1, 2, 4, 8
NK_LINEAGE_FLEXIBLE = 1

# nk_render_rays_bwd runs the weight-gradient products of both networks in one shared launch up to
# this many fine-network points per call (NK_PAIR_GEMM_MAX_POINTS in csrc/nk_api.cpp)
PAIR_GEMM_MAX_POINTS = 1 << 20

# nk_render_opts.act_format: formats of the tensors saved for backward (experiments beyond bf16)
ACT_FORMATS = {"bf16": 0, "fp8": 1, "fp8bf8": 2}

STREAM_STRATIFIED, STREAM_PDF, STREAM_NOISE_COARSE, STREAM_NOISE_FINE, STREAM_PIXELS = 0, 1, 2, 3, 4


ABI_VERSION = 103          # nk_version() of the library these ctypes declarations describe


class NkArch(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("depth", "width", "skip", "dir_width", "n_freq_xyz", "n_freq_dir",
                                         "relu_first", "relu_feat", "lineage", "no_viewdirs", "enc_flags")]


class NkStepState(C.Structure):
    """Per-iteration scalars of a training step, read by the kernels from device memory (graph replay)."""
    _fields_ = [("render_seed", C.c_uint32), ("pixel_seed", C.c_uint32), ("pixel_counter", C.c_uint64),
                ("view", C.c_int32), ("adam_step_size", C.c_float), ("adam_bc2_sqrt", C.c_float), ("reserved", C.c_int32)]


class NkRenderOpts(C.Structure):
    _fields_ = [("num_coarse", C.c_int32), ("num_fine", C.c_int32), ("perturb", C.c_int32),
                ("lindisp", C.c_int32), ("white_background", C.c_int32),
                ("radiance_field_noise_std", C.c_float), ("seed", C.c_uint32), ("act_format", C.c_int32),
                ("state", C.c_void_p), ("skip_empty_eps", C.c_float), ("reserved", C.c_int32)]


class NkCamera(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("c2w", C.c_float * 12), ("near", C.c_float),
                ("far", C.c_float), ("ndc", C.c_int32), ("ndc_near", C.c_float)]


class NkRenderOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("rgb_fine", "disp_fine", "acc_fine", "depth_fine",
                                          "rgb_coarse", "disp_coarse", "acc_coarse", "depth_coarse")]


# name -> (restype, argtypes); every symbol include/nerf_kinematics.h declares
_P, _I64, _I32, _F = C.c_void_p, C.c_int64, C.c_int32, C.c_float
_PA, _PO, _PC, _PR = C.POINTER(NkArch), C.POINTER(NkRenderOpts), C.POINTER(NkCamera), C.POINTER(NkRenderOut)
SIGNATURES = {
    "nk_version": (C.c_int, []),
    "nk_last_error": (C.c_char_p, []),
    "nk_num_params": (_I64, [_PA]),
    "nk_packed_bytes": (_I64, [_PA]),
    "nk_pack_weights": (C.c_int, [_PA, _P, _P, _P]),
    "nk_get_rays": (C.c_int, [_PC, _P, _I64, _I64, _P, _P]),
    "nk_sample_training_rays": (C.c_int, [_PC, _P, _I32, _I32, C.c_uint32, C.c_uint64, _I64, _P, _P, _P, _P]),
    "nk_sample_training_rays_dev": (C.c_int, [_P, _I32, _P, _I64, _I32, _I32, _P, _I64, _P, _P, _P, _P]),
    "nk_set_step_state": (C.c_int, [C.POINTER(NkStepState), _P, _P]),
    "nk_adam_step_dev": (C.c_int, [_P, _P, _P, _P, _I64, _P, C.c_double, C.c_double, C.c_double, _F, _P]),
    "nk_stratified_z": (C.c_int, [_P, _I64, _PO, _P, _P]),
    "nk_act_bytes": (_I64, [_PA, _I64, _I32]),
    "nk_mlp_forward": (C.c_int, [_PA, _P, _P, _P, _I64, _I32, _P, _P, _P]),
    "nk_raw2outputs": (C.c_int, [_P, _P, _P, _I64, _I32, _PO, _I32, _P, _P, _P, _P, _P, _P]),
    "nk_sample_pdf": (C.c_int, [_P, _P, _I64, _PO, _P, _P, _P]),
    "nk_raw2outputs_bwd": (C.c_int, [_P, _P, _P, _I64, _I32, _PO, _I32, _P, _P, _P, _P, _P]),
    "nk_dz_bytes": (_I64, [_PA, _I64, _I32]),
    "nk_mlp_backward": (C.c_int, [_PA, _P, _P, _P, _I64, _I32, _P, _P, _P]),
    "nk_mse_loss": (C.c_int, [_P, _P, _P, _I64, _F, _P, _P, _P, _P]),
    "nk_adam_step": (C.c_int, [_P, _P, _P, _P, _I64, _I64, C.c_double, C.c_double, C.c_double, C.c_double, _F, _P]),
    "nk_render_workspace_bytes": (_I64, [_PA, _PA, _PO, _I64, C.c_int]),
    "nk_render_rays_fwd": (C.c_int, [_PA, _P, _PA, _P, _P, _I64, _PO, C.c_int, _P, _I64, _PR, _P]),
    "nk_render_rays_bwd": (C.c_int, [_PA, _P, _PA, _P, _P, _I64, _PO, _P, _I64, _P, _P, _P, _P, _P]),
    "nk_set_stage_events": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "nk_render_rays_fwd_noalloc": (C.c_int, [_PA, _P, _PA, _P, _P, _I64, _PO, _PR, _P]),
}

_lib = None


class NerfKinError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library, or fail loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(or __graft_entry__.build()); there is no CPU fallback for the render path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    # the structs above mirror include/nerf_kinematics.h of ONE library version (nk_arch grew fields in
    # 102 and 103); a library built from other sources would read them wrongly.  NK_LIB (a developer comparing two
    # builds) takes the responsibility itself.
    if lib.nk_version() != ABI_VERSION and "NK_LIB" not in os.environ:
        raise ImportError(f"{LIB_PATH} reports version {lib.nk_version()}, these bindings are for {ABI_VERSION}: "
                          f"rebuild it with `make -C {os.path.join(_HERE, 'csrc')}`")
    _lib = lib
    return lib


def check(rc: int) -> int:
    """Raise on a negative nk_status (sizes are returned through the same channel)."""
    if rc < 0:
        msg = load().nk_last_error().decode()
        if rc == -1:
            raise ValueError(msg)
        raise NerfKinError(f"nk_status {rc}: {msg}")
    return rc
