"""manytor_amd -- MI355X-native batched manipulator-environment step engine.

Keeps the reset()/step()/get_observations() surface of victorkich/ManyTor's
``Environment`` / ``Multienv`` (manytor.py) and evaluates N lock-stepped arms per
HIP launch.  See DESIGN.md / INTEGRATION.md.
"""
from ._lib import ManytorError, device_count  # noqa: F401
from .api import (DEVICE_ACTIONS, HOST, PORT, BatchView, Environment, Multienv, dh, fk, r_theta)  # noqa: F401
from .engine import (DH7_TABLE, REF_DH_TABLE, StepEngine, comm_unique_id, fk_batch, r_theta_batch, stream_probe,  # noqa: F401
                     route_trace)
from .viewer import ViewerLink  # noqa: F401
from . import _lib as lib  # noqa: F401

__all__ = [
    "Environment", "Multienv", "StepEngine", "dh", "fk", "r_theta", "fk_batch", "r_theta_batch", "route_trace", "ViewerLink", "HOST", "PORT",
    "REF_DH_TABLE", "DH7_TABLE", "DEVICE_ACTIONS", "BatchView", "ManytorError", "device_count", "comm_unique_id",
]
