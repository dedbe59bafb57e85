"""nautilus_amd -- MI355X (gfx950) implementation of nautilus's loop-closure correlative scan
matcher and batched Ceres-residual evaluation, behind the reference's own interfaces.

The product is libnautilus_hip.so (HIP kernels + extern "C" shim, include/nautilus_hip.h);
this package is the thin host-side mirror used by tests and bench.  No CPU fallback exists.
"""
from . import _lib  # noqa: F401
from ._lib import NhipError, device_count  # noqa: F401
