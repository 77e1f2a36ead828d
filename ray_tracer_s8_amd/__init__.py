"""MI355X-native tile renderer for the ray-tracer-s8 slave hot path.

Product code: the HIP kernel + C-ABI (csrc/, include/rt_tile.h) and this thin host-side
mirror of the reference's RenderInfo/ImageSlice surface.  Nothing here imports oracle/.
"""
from ._abi import (RT_FLAG_EXACT_SCAN, RT_FLAG_NO_BVH_CULL, RT_FLAG_NONE, SPHERE_DTYPE, TRIANGLE_DTYPE, RtError, TileRequest,
                   TileStats, default_request)
from .interface import (Controller, FrameContext, ImageSlice, RenderInfo, RenderMeta, RenderSettings, Scene, Slave, World, init,
                        render_frame_native)

__all__ = [
    "RT_FLAG_EXACT_SCAN", "RT_FLAG_NO_BVH_CULL", "RT_FLAG_NONE", "SPHERE_DTYPE", "TRIANGLE_DTYPE", "RtError", "TileRequest", "TileStats",
    "default_request", "Controller", "FrameContext", "ImageSlice", "RenderInfo", "RenderMeta", "RenderSettings", "Scene", "Slave",
    "World", "init", "render_frame_native",
]
