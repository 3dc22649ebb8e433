"""MI355X-native deferred-PBR shading path (drop-in for the DeferredRendering HLSL passes of
zrlhahaha/Direct12PBRRenderer).  The product is csrc/ (hand-written gfx950 kernels behind the
C ABI of include/pbr_hip.h); this package is the thin host side used by tests and bench.py.
"""
from . import structs  # noqa: F401
from .scene import Camera, make_global, make_lights, attenuation_coefficients  # noqa: F401

__all__ = ["structs", "Camera", "make_global", "make_lights", "attenuation_coefficients"]
