"""ctypes binding of include/rt_tile.h — the same stub a cgo / Rust `extern "C"` user
would write.  Loading fails loudly when the HIP library is missing:
there is no CPU fallback in the product path."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from . import build as _build

RT_ABI_VERSION = 4          # include/rt_tile.h RT_ABI_VERSION; load() refuses a library of another version

# ---- status codes (rt_status)
RT_OK = 0
RT_ERR_BAD_ARG = -1
RT_ERR_NOT_INITIALIZED = -2
RT_ERR_NO_DEVICE = -3
RT_ERR_BAD_DEVICE = -4
RT_ERR_BUFFER_TOO_SMALL = -5
RT_ERR_FRAME_SIZE = -6
RT_ERR_HIP = -7
RT_ERR_LIMIT = -8
RT_ERR_OOM = -9

RT_FLAG_NONE = 0
RT_FLAG_EXACT_SCAN = 1
RT_FLAG_NO_BVH_CULL = 2
RT_FLAG_OC_BROAD_PHASE = 4
RT_FLAG_FULL_CHAIN = 8
RT_FLAG_BVH_TRAVERSE = 16
RT_FLAG_LINEAR_SCAN = 32
RT_FLAG_EXACT_NODES = 64
RT_FLAG_QUANT_NODES = 128
RT_FLAG_NO_LDS_TREE = 256
RT_FLAG_COUNT_STEPS = 512
RT_FLAG_CULL_WALK = 1024
RT_FLAG_NO_CULL_WALK = 2048
RT_FLAG_FRAME_QUEUE = 4096          # frame-level (rt_render_frame / rt_frame_ctx_render): dynamic strip queue
RT_FLAG_FRAME_NO_PIN = 8192         # frame-level: do not page-lock the caller's frame buffer
RT_FLAG_FRAME_STATIC = 16384        # frame-level: strip k -> devices[k % n] instead of the cost-balanced assignment
RT_MAX_BOUNCES = 62
RT_MAX_SPP = 4096                 # samples per pixel limit (rt_tile.h)

# numpy dtypes with the exact layout of rt_sphere / rt_triangle (no padding)
SPHERE_DTYPE = np.dtype(
    [("cx", "<f4"), ("cy", "<f4"), ("cz", "<f4"), ("radius", "<f4"),
     ("albedo_r", "<f4"), ("albedo_g", "<f4"), ("albedo_b", "<f4"),
     ("roughness", "<f4"), ("emission", "<f4")]
)
TRIANGLE_DTYPE = np.dtype(
    [("a", "<f4", 3), ("b", "<f4", 3), ("c", "<f4", 3),
     ("albedo_r", "<f4"), ("albedo_g", "<f4"), ("albedo_b", "<f4"),
     ("roughness", "<f4"), ("emission", "<f4")]
)
assert SPHERE_DTYPE.itemsize == 36 and TRIANGLE_DTYPE.itemsize == 56


class TileRequest(C.Structure):
    """rt_tile_request = RenderMeta + division_no + the knobs the reference hard-codes."""
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("divisions", C.c_uint32),
        ("division_no", C.c_uint32), ("spp", C.c_uint32), ("max_bounces", C.c_uint32),
        ("aperture", C.c_float), ("focus_distance", C.c_float), ("fov", C.c_float),
        ("focal_length", C.c_float), ("t_min", C.c_float), ("t_max", C.c_float),
        ("seed", C.c_uint64), ("flags", C.c_uint32), ("reserved", C.c_uint32),
    ]

    def copy(self) -> "TileRequest":
        r = TileRequest()
        C.memmove(C.byref(r), C.byref(self), C.sizeof(TileRequest))
        return r


class TileStats(C.Structure):
    _fields_ = [
        ("ray_segments", C.c_uint64), ("primary_rays", C.c_uint64),
        ("broad_candidates", C.c_uint64), ("exact_fallbacks", C.c_uint64),
        ("kernel_ms", C.c_float), ("h2d_ms", C.c_float), ("d2h_ms", C.c_float),
        ("n_launches", C.c_uint32), ("engine", C.c_uint32), ("broad_form", C.c_uint32),
        ("node_steps", C.c_uint64),
    ]


class FrameStats(C.Structure):
    """rt_frame_stats: where the wall time of one rt_frame_ctx_render call went (ms)."""
    _fields_ = [
        ("totals", TileStats), ("wall_ms", C.c_float), ("pin_ms", C.c_float), ("scene_ms", C.c_float),
        ("kernel_ms", C.c_float), ("d2h_exposed_ms", C.c_float), ("host_ms", C.c_float),
        ("n_devices", C.c_uint32), ("pinned", C.c_uint32),
        ("assignment", C.c_uint32), ("balance_max_over_mean", C.c_float), ("entry_segments", C.c_uint64 * 16),
    ]


assert C.sizeof(TileRequest) == 64
assert C.sizeof(TileStats) == 64
assert C.sizeof(FrameStats) == 232


def default_request(**kw) -> TileRequest:
    """Reference literals (slave main.rs:39-51, shapes/mod.rs:12-13, controller main.rs:33-39).
    Pure Python so that host-side code and CPU tests need no GPU library."""
    r = TileRequest(width=1920, height=1080, divisions=20, division_no=0, spp=100, max_bounces=10,
                    aperture=0.1, focus_distance=1.0, fov=float(np.float32(np.pi) / np.float32(2.0)),
                    focal_length=1.0, t_min=0.001, t_max=1000.0, seed=0, flags=0, reserved=0)
    for k, v in kw.items():
        if not hasattr(r, k):
            raise AttributeError(k)
        setattr(r, k, v)
    return r


class RtError(RuntimeError):
    def __init__(self, status: int, what: str, detail: str):
        super().__init__(f"{what}: status {status} ({detail})")
        self.status = status


_lib = None


def lib_path() -> Path:
    return _build.LIB_PATH


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load librt_s8.so (building it with hipcc if absent).  Raises if it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if build_if_missing:
        path = _build.build()
    _lib = _bind(path)
    return _lib


_dbg_lib = None


def load_debug() -> C.CDLL:
    """The TEST library lib/librt_s8_dbg.so: the product sources plus the rt_debug_* hooks (build.py).  A second, independent
    instance of the library in the process (its own rt_init state)."""
    global _dbg_lib
    if _dbg_lib is None:
        _dbg_lib = _bind(_build.build_debug())
        _dbg_lib.rt_debug_set.argtypes = [C.c_char_p, C.c_int]
        _dbg_lib.rt_debug_set.restype = C.c_int
    return _dbg_lib


class debug_library:
    """`with _abi.debug_library():` — everything the package does inside the block (rt.init(), Scene, FrameContext, debug_set)
    goes to the test library instead of the product library.  Tests and tools only."""

    def __enter__(self):
        global _lib
        self._prev = _lib
        _lib = load_debug()
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._prev
        return False


def _bind(path: Path) -> C.CDLL:
    if not path.exists():
        raise FileNotFoundError(
            f"{path} not found: the HIP extension is required (run `python -m ray_tracer_s8_amd.build`); "
            "there is no CPU fallback")
    lib = C.CDLL(str(path))
    u8p, f32p, vp = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.c_void_p
    lib.rt_init.argtypes = [C.POINTER(C.c_int)]
    lib.rt_init.restype = C.c_int
    lib.rt_shutdown.argtypes = []
    lib.rt_shutdown.restype = None
    lib.rt_abi_version.argtypes = []
    lib.rt_abi_version.restype = C.c_uint32
    if lib.rt_abi_version() != RT_ABI_VERSION:
        raise RuntimeError(f"{path} has ABI version {lib.rt_abi_version()}, this binding is written for "
                           f"{RT_ABI_VERSION}: rebuild it (python -m ray_tracer_s8_amd.build)")
    lib.rt_strerror.argtypes = [C.c_int]
    lib.rt_strerror.restype = C.c_char_p
    lib.rt_last_error.argtypes = []
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_tile_request_defaults.argtypes = [C.POINTER(TileRequest)]
    lib.rt_tile_request_defaults.restype = None
    lib.rt_tile_bytes.argtypes = [C.POINTER(TileRequest)]
    lib.rt_tile_bytes.restype = C.c_size_t
    lib.rt_render_tile.argtypes = [C.c_int, C.POINTER(TileRequest), vp, C.c_uint32, vp, C.c_uint32, vp,
                                   vp, C.c_size_t, vp, C.POINTER(TileStats)]
    lib.rt_render_tile.restype = C.c_int
    lib.rt_scene_create.argtypes = [C.c_int, vp, C.c_uint32, vp, C.c_uint32, vp, C.POINTER(vp)]
    lib.rt_scene_create.restype = C.c_int
    lib.rt_scene_destroy.argtypes = [vp]
    lib.rt_scene_destroy.restype = None
    lib.rt_scene_render_tile.argtypes = [vp, C.POINTER(TileRequest), vp, C.c_size_t, vp, C.POINTER(TileStats)]
    lib.rt_scene_render_tile.restype = C.c_int
    lib.rt_scene_render_tile_device.argtypes = [vp, C.POINTER(TileRequest), vp, C.c_size_t, vp, vp]
    lib.rt_scene_render_tile_device.restype = C.c_int
    lib.rt_scene_render_tiles_device.argtypes = [vp, C.POINTER(TileRequest), C.c_uint32, C.POINTER(vp), C.c_size_t,
                                                 C.POINTER(vp), vp]
    lib.rt_scene_render_tiles_device.restype = C.c_int
    lib.rt_scene_render_tiles.argtypes = [vp, C.POINTER(TileRequest), C.c_uint32, C.POINTER(vp), C.c_size_t,
                                          C.POINTER(vp), C.POINTER(TileStats)]
    lib.rt_scene_render_tiles.restype = C.c_int
    lib.rt_scene_collect.argtypes = [vp, C.POINTER(TileStats)]
    lib.rt_scene_collect.restype = C.c_int
    lib.rt_render_frame.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(TileRequest), vp, C.c_uint32,
                                    vp, C.c_uint32, vp, vp, C.c_size_t, C.POINTER(TileStats)]
    lib.rt_render_frame.restype = C.c_int
    lib.rt_frame_ctx_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    lib.rt_frame_ctx_create.restype = C.c_int
    lib.rt_frame_ctx_set_world.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, vp]
    lib.rt_frame_ctx_set_world.restype = C.c_int
    lib.rt_frame_ctx_render.argtypes = [vp, C.POINTER(TileRequest), vp, C.c_size_t, C.POINTER(FrameStats)]
    lib.rt_frame_ctx_render.restype = C.c_int
    lib.rt_frame_ctx_release_buffer.argtypes = [vp]
    lib.rt_frame_ctx_release_buffer.restype = C.c_int
    lib.rt_frame_ctx_destroy.argtypes = [vp]
    lib.rt_frame_ctx_destroy.restype = None
    return lib


def check(status: int, what: str) -> None:
    if status != RT_OK:
        lib = load()
        raise RtError(status, what, f"{lib.rt_strerror(status).decode()}: {lib.rt_last_error().decode()}")


def as_spheres(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=SPHERE_DTYPE) if a is not None else np.zeros(0, SPHERE_DTYPE)
    return a


def as_triangles(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=TRIANGLE_DTYPE) if a is not None else np.zeros(0, TRIANGLE_DTYPE)
    return a


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def as_world_index(a, n: int):
    """world_index argument: None, or one uint32 position per primitive (spheres, then triangles)."""
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.uint32)
    if a.size != n:
        raise ValueError(f"world_index: {a.size} entries for {n} primitives")
    return a


def hip_runtime_path() -> str:
    """The libamdhip64 file the library's HIP calls are bound to (rt_hip_runtime_path).  PyTorch wheels bundle their own
    runtime; whichever of torch / this library is loaded FIRST decides which one this library uses
    (profiles/README.md, "two HIP runtimes in one process")."""
    lib = load()
    lib.rt_hip_runtime_path.argtypes = [C.c_char_p, C.c_size_t]
    lib.rt_hip_runtime_path.restype = C.c_size_t
    buf = C.create_string_buffer(4096)
    return buf.value.decode() if lib.rt_hip_runtime_path(buf, 4096) else ""


def hip_runtime() -> C.CDLL:
    """ctypes handle of THAT runtime (already mapped: the same instance the library uses), for callers that make their own
    streams or device buffers (tests)."""
    p = hip_runtime_path()
    if not p:
        raise RuntimeError("cannot tell which HIP runtime librt_s8.so is bound to")
    return C.CDLL(p)


def check_single_hip_runtime() -> None:
    """Refuse a process in which torch and this library would drive the GPU through DIFFERENT HIP runtimes: the second one
    to initialise finds "no ROCm-capable device" (round 2: torch.cuda.Stream() after 122 tests of this library), and a stream
    or event made by one is garbage to the other.  Safe orders: import torch first (the library then binds to torch's
    runtime), or never import torch in the process."""
    import sys
    if "torch" not in sys.modules:
        return
    import os
    torch = sys.modules["torch"]
    tdir = os.path.realpath(os.path.join(os.path.dirname(torch.__file__), "lib"))
    mine = os.path.realpath(hip_runtime_path())
    bundled = os.path.exists(os.path.join(tdir, "libamdhip64.so"))
    if bundled and mine and os.path.dirname(mine) != tdir:
        raise RuntimeError(f"librt_s8.so is bound to {mine} but torch brings its own HIP runtime in {tdir}: two HIP runtimes in one "
                           "process cannot share the device.  Import torch BEFORE loading ray_tracer_s8_amd, or keep torch out "
                           "of this process.")


def debug_set(name: str, value: int) -> int:
    """Set a launch-path knob (RT_FORCE_CAPPED, RT_STACK_LDS, RT_CULL_WALK, ...: csrc/rt_api.hip DebugKnob); returns the
    previous value.  Tests and tools only."""
    lib = load()
    if not hasattr(lib, "rt_debug_set"):
        raise RuntimeError("debug_set needs the test library: use it inside `with _abi.debug_library():` (the product library "
                           "exports no rt_debug_* hooks)")
    prev = lib.rt_debug_set(name.encode(), int(value))
    if prev == -2**31:
        raise KeyError(name)
    return prev
