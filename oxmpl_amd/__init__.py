"""oxmpl_amd -- MI355X-native batched RRT hot path for oxmpl (rossng/oxmpl), with the rows built next to it:
RRTConnect (R^n and SE(2)), RRT* and PRM.

The product is `lib/liboxmpl_hip.so` (hand-written HIP for gfx950 behind the C ABI of
include/oxmpl_hip.h).  This package is the thin Python host side: a ctypes binding
(`oxmpl_amd.capi`) and a mirror of oxmpl's Python class surface (`oxmpl_amd.base`,
`oxmpl_amd.geometric`).  There is no CPU fallback: importing works anywhere, computing
needs the built library and a GPU and fails loudly otherwise.
"""
from . import capi  # noqa: F401
from .capi import OxhipError, PRMRoadmap, RRTBatch, build_library, library_path  # noqa: F401

__all__ = ["capi", "OxhipError", "RRTBatch", "PRMRoadmap", "build_library", "library_path"]
