"""Drop-in module name: ``import manytor as tor`` (test_multi.py:1, test_single.py:1 of the reference)
resolves to the HIP engine's surface."""
from manytor_amd.api import *  # noqa: F401,F403
from manytor_amd.api import DEVICE_ACTIONS, HOST, PORT, Environment, Multienv, dh, fk, r_theta  # noqa: F401
