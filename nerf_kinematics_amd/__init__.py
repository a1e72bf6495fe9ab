"""MI355X-native NeRF volume-rendering path (render_rays) behind the nerf-pytorch interface.

The compute lives in ``libnerfkin.so`` (hand-written HIP for gfx950, C ABI in
``include/nerf_kinematics.h``); this package is the host-side mirror of the reference's
Python interface for that path.
"""
from .lib import load, NerfKinError, LIB_PATH  # noqa: F401
from .model import Arch, NeRFModel, ARCH_RUN_NERF, ARCH_LEGO_CKPT, ARCH_WHEEL_CKPT  # noqa: F401
from .render import (get_rays, stratified_z, run_network, raw2outputs, raw2outputs_backward,  # noqa: F401
                     sample_pdf, render_rays, make_opts, alloc_activations, mlp_backward, mse_loss,
                     adam_step)

__version__ = "0.1.0"
