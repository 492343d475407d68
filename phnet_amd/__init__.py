"""phnet_amd: MI355X-native (gfx950) implementation of PHNet's per-clip forward/backward hot path.

Host side = Python on PyTorch-ROCm (device memory, streams, torch.distributed); compute = hand-written HIP
kernels behind the C-ABI in include/phnet_hip.h (phnet_amd/lib/libphnet_hip.so).
`phnet_amd.install()` registers drop-in modules under the reference's import paths
(libs.models.Router4OL, libs.utils.loss4OLV3, libs.ops) - see INTEGRATION.md.
"""
__version__ = "0.1.0"


def install():
    from .compat import install as _install
    return _install()
