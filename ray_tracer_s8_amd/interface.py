"""Host-side mirror of the reference's controller<->slave surface, on top of the C-ABI.

Reference (Rust, ray-tracer-slave/src/lib.rs:10-30):
    RenderInfo { world: Vec<Object>, render_meta: RenderMeta, division_no: u32 }
    RenderMeta { height, width, divisions, id: Uuid }
    ImageSlice { division_no, image: Vec<u8>, id: Uuid }
`Slave.render(info)` replaces the body of the slave's worker (main.rs:37-90); `Controller`
replaces dispatch + assembly (controller main.rs:47-75, 109-119) with GPUs as the slaves.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import threading
import uuid
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _abi
from ._abi import FrameStats, TileRequest, TileStats, default_request
from .dispatch import assemble, strips_for_worker


@dataclass
class RenderMeta:
    height: int = 1080
    width: int = 1920
    divisions: int = 20
    id: uuid.UUID = field(default_factory=uuid.uuid4)


@dataclass
class World:
    """`world: Vec<Object>` (lib.rs:11) as the C-ABI carries it: the spheres, the triangles, and `world_index` — the
    position of every sphere, then of every triangle, in the list (None: the spheres in order, then the triangles).
    The order is observable (BVH::build numbers shapes by position, bvh_impl.rs:421-427: leaf order, tie winners)."""
    spheres: np.ndarray = field(default_factory=lambda: np.zeros(0, _abi.SPHERE_DTYPE))
    triangles: np.ndarray = field(default_factory=lambda: np.zeros(0, _abi.TRIANGLE_DTYPE))
    world_index: Optional[np.ndarray] = None

    def __post_init__(self):
        self.spheres = _abi.as_spheres(self.spheres)
        self.triangles = _abi.as_triangles(self.triangles)
        self.world_index = _abi.as_world_index(self.world_index, len(self.spheres) + len(self.triangles))

    def objects(self):
        """The list in the reference's order: ("Sphere" | "Triangle", record) per position."""
        n = len(self.spheres) + len(self.triangles)
        wi = self.world_index if self.world_index is not None else np.arange(n, dtype=np.uint32)
        out = [None] * n
        for i in range(n):
            out[int(wi[i])] = (("Sphere", self.spheres[i]) if i < len(self.spheres)
                               else ("Triangle", self.triangles[i - len(self.spheres)]))
        return out


@dataclass
class RenderSettings:
    """The knobs the reference hard-codes (defaults = its literals) + the job seed."""
    spp: int = 100
    max_bounces: int = 10
    aperture: float = 0.1
    focus_distance: float = 1.0
    fov: float = float(np.float32(np.pi) / np.float32(2.0))
    focal_length: float = 1.0
    t_min: float = 0.001
    t_max: float = 1000.0
    seed: int = 0
    flags: int = 0


@dataclass
class RenderInfo:
    world: World
    render_meta: RenderMeta
    division_no: int
    settings: RenderSettings = field(default_factory=RenderSettings)

    def request(self) -> TileRequest:
        s, m = self.settings, self.render_meta
        return default_request(width=m.width, height=m.height, divisions=m.divisions, division_no=self.division_no,
                               spp=s.spp, max_bounces=s.max_bounces, aperture=s.aperture,
                               focus_distance=s.focus_distance, fov=s.fov, focal_length=s.focal_length,
                               t_min=s.t_min, t_max=s.t_max, seed=s.seed, flags=s.flags)


@dataclass
class ImageSlice:
    division_no: int
    image: np.ndarray          # uint8, (H/div)*W*3, RGB, top row first
    id: uuid.UUID
    stats: Optional[TileStats] = None


_initialised_libs = set()                    # (the product library, and the test library where a test swapped it in)


def _is_initialised() -> bool:
    return id(_abi.load()) in _initialised_libs


def init() -> int:
    """rt_init(); returns the device count.  Raises RtError(RT_ERR_NO_DEVICE) without a GPU."""
    lib = _abi.load()
    n = C.c_int(0)
    _abi.check(lib.rt_init(C.byref(n)), "rt_init")
    _abi.check_single_hip_runtime()          # torch imported after the library was bound to ROCm's runtime: say so now
    _initialised_libs.add(id(lib))
    return n.value


class Scene:
    """rt_scene handle: the world resident in one GPU's HBM."""

    def __init__(self, device: int, world: World):
        self._lib = _abi.load()
        if not _is_initialised():
            init()
        self.world = world
        self.device = device
        h = C.c_void_p()
        _abi.check(self._lib.rt_scene_create(device, _abi.ptr(world.spheres), len(world.spheres),
                                             _abi.ptr(world.triangles), len(world.triangles),
                                             _abi.ptr(world.world_index), C.byref(h)),
                   "rt_scene_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_scene_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def render_tile(self, req: TileRequest, want_f32: bool = False):
        n = self._lib.rt_tile_bytes(C.byref(req))
        out = np.empty(n, np.uint8)
        outf = np.empty(n, np.float32) if want_f32 else None
        st = TileStats()
        _abi.check(self._lib.rt_scene_render_tile(self._h, C.byref(req), out.ctypes.data_as(C.c_void_p), n,
                                                  outf.ctypes.data_as(C.c_void_p) if want_f32 else None,
                                                  C.byref(st)), "rt_scene_render_tile")
        return out, outf, st

    def render_tile_device(self, req: TileRequest, d_out_ptr: int, out_len: int, d_f32_ptr: int = 0, stream: int = 0):
        _abi.check(self._lib.rt_scene_render_tile_device(self._h, C.byref(req), C.c_void_p(d_out_ptr), out_len,
                                                         C.c_void_p(d_f32_ptr) if d_f32_ptr else None,
                                                         C.c_void_p(stream) if stream else None),
                   "rt_scene_render_tile_device")

    def render_tiles(self, reqs: Sequence[TileRequest], want_f32: bool = False, out=None):
        """Batched: strips of one frame, host buffers out (rt_scene_render_tiles).  `out`: a list of uint8 arrays
        from an earlier call to write into again (a caller that renders frame after frame keeps its pages warm)."""
        n = len(reqs)
        nb = self._lib.rt_tile_bytes(C.byref(reqs[0]))
        arr = (TileRequest * n)(*reqs)
        if out is not None:
            if len(out) != n or any(o.dtype != np.uint8 or o.size < nb or not o.flags.c_contiguous for o in out):
                raise ValueError("out: need one contiguous uint8 array of at least rt_tile_bytes per request")
        outs = out if out is not None else [np.empty(nb, np.uint8) for _ in range(n)]
        outf = [np.empty(nb, np.float32) for _ in range(n)] if want_f32 else None
        po = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        pf = (C.c_void_p * n)(*[o.ctypes.data for o in outf]) if want_f32 else None
        st = TileStats()
        _abi.check(self._lib.rt_scene_render_tiles(self._h, arr, n, po, nb, pf, C.byref(st)), "rt_scene_render_tiles")
        return outs, outf, st

    def render_tiles_device(self, reqs: Sequence[TileRequest], d_out_ptrs: Sequence[int], out_len_each: int,
                            stream: int = 0):
        """Batched, asynchronous, device-resident output (rt_scene_render_tiles_device)."""
        n = len(reqs)
        # marshalled afresh on every call (a few dozen structs): a cache keyed on id(reqs) served stale requests to a
        # caller that changed seed or division_no in place
        arr = (TileRequest * n)(*reqs)
        po = (C.c_void_p * n)(*d_out_ptrs)
        _abi.check(self._lib.rt_scene_render_tiles_device(self._h, arr, n, po, out_len_each, None,
                                                          C.c_void_p(stream) if stream else None),
                   "rt_scene_render_tiles_device")

    def collect(self) -> TileStats:
        st = TileStats()
        _abi.check(self._lib.rt_scene_collect(self._h, C.byref(st)), "rt_scene_collect")
        return st


class Slave:
    """One GPU playing the reference's `ray-tracer-slave` worker."""

    def __init__(self, device: int = 0):
        self.device = device
        self._scene: Optional[Scene] = None
        self._scene_key = None
        # The reference slave drains its requests through ONE worker thread (slave main.rs:32-36, 159-160), so two
        # jobs never overlap on a slave.  Callers here may be threads of different jobs (controller_shim starts a set
        # per upload): the whole of render() — key check, scene swap, render — runs under this lock, or one job would
        # destroy the scene the other is rendering from.
        self._lock = threading.Lock()

    def render(self, info: RenderInfo) -> ImageSlice:
        with self._lock:
            return self._render_locked(info)

    def _render_locked(self, info: RenderInfo) -> ImageSlice:
        # the reference rebuilds the BVH per strip; the scene stays resident while the world's CONTENT is the same
        # (a slave behind HTTP gets a freshly decoded world object with every strip of a job)
        w = info.world
        key = (len(w.spheres), len(w.triangles), hashlib.blake2b(w.spheres.tobytes(), digest_size=16).digest(),
               hashlib.blake2b(w.triangles.tobytes(), digest_size=16).digest(),
               None if w.world_index is None else hashlib.blake2b(w.world_index.tobytes(), digest_size=16).digest())
        if self._scene is None or self._scene_key != key:
            if self._scene:
                self._scene.close()
            self._scene = Scene(self.device, info.world)
            self._scene_key = key
        img, _, st = self._scene.render_tile(info.request())
        return ImageSlice(division_no=info.division_no, image=img, id=info.render_meta.id, stats=st)

    def close(self):
        with self._lock:
            if self._scene:
                self._scene.close()
                self._scene = None


class Controller:
    """Dispatch strips to GPUs instead of docker slaves and assemble the frame."""

    def __init__(self, devices: Optional[Sequence[int]] = None):
        n = init()
        self.devices = list(devices) if devices is not None else list(range(n))
        self.slaves = [Slave(d) for d in self.devices]

    def render_frame(self, world: World, meta: RenderMeta, settings: RenderSettings) -> np.ndarray:
        """Strip k goes to slave k mod n; the slaves work concurrently, one dispatcher thread each (the controller fires
        its requests with join_all, controller main.rs:47-75; ctypes releases the GIL inside the render call)."""
        results: list = [None] * len(self.slaves)

        def run(w: int, slave: Slave):
            try:
                results[w] = [slave.render(RenderInfo(world, meta, k, settings))
                              for k in strips_for_worker(meta.divisions, w, len(self.slaves))]
            except BaseException as e:          # handed to the caller below
                results[w] = e

        threads = [threading.Thread(target=run, args=(w, s)) for w, s in enumerate(self.slaves)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        slices = []
        for r in results:
            if isinstance(r, BaseException):
                raise r
            slices += [(s.division_no, s.image) for s in r]
        return assemble(slices, meta.width, meta.height, meta.divisions)

    def close(self):
        for s in self.slaves:
            s.close()


class FrameContext:
    """rt_frame_ctx: the controller's state for a job — dispatcher threads, the world resident on every device, streams,
    strip buffers and the page-locked registration of the frame buffer, all made once and reused frame after frame
    (replaces controller main.rs:47-75, 109-115)."""

    def __init__(self, devices: Optional[Sequence[int]] = None, world: Optional[World] = None):
        self._lib = _abi.load()
        if not _is_initialised():
            init()
        if devices is None:
            dv, nd = None, 0
        else:
            dv, nd = (C.c_int * len(devices))(*devices), len(devices)
        h = C.c_void_p()
        _abi.check(self._lib.rt_frame_ctx_create(dv, nd, C.byref(h)), "rt_frame_ctx_create")
        self._h = h
        self._buf = None
        self._pinned = None                  # the array whose page-locked registration the context holds (kept alive here)
        if world is not None:
            self.set_world(world)

    def set_world(self, world: World):
        _abi.check(self._lib.rt_frame_ctx_set_world(self._h, _abi.ptr(world.spheres), len(world.spheres),
                                                    _abi.ptr(world.triangles), len(world.triangles),
                                                    _abi.ptr(world.world_index)), "rt_frame_ctx_set_world")

    def render(self, req: TileRequest, out: Optional[np.ndarray] = None):
        """One frame.  `out`: a uint8 array of H*W*3 bytes to write into (the SAME array frame after frame keeps its
        page-locked registration: only the first frame pays pin_ms); default: the context's own buffer.
        Returns (H x W x 3 view of the buffer, FrameStats)."""
        n = req.width * req.height * 3
        if out is None:
            if self._buf is None or self._buf.size != n:
                self._buf = np.empty(n, np.uint8)
            out = self._buf
        if out.dtype != np.uint8 or out.size < n or not out.flags.c_contiguous:
            raise ValueError("out: need a contiguous uint8 array of at least H*W*3 bytes")
        # rt_tile.h: "the caller must not free a buffer the context still holds".  The context recognises a registered buffer by
        # its ADDRESS, and a freed array's address may come back with the next allocation — over pages that are no longer the
        # registered ones.  So the wrapper keeps the registered array alive (self._pinned) and drops the registration BEFORE
        # another array takes its place.
        if self._pinned is not None and self._pinned is not out:
            self.release_buffer()
        fs = FrameStats()
        try:
            _abi.check(self._lib.rt_frame_ctx_render(self._h, C.byref(req), out.ctypes.data_as(C.c_void_p), out.size,
                                                     C.byref(fs)), "rt_frame_ctx_render")
        finally:
            self._pinned = out               # (registered or not: harmless to hold, and an error return may have left it registered)
        return out.reshape(-1)[:n].reshape(req.height, req.width, 3), fs

    def release_buffer(self):
        _abi.check(self._lib.rt_frame_ctx_release_buffer(self._h), "rt_frame_ctx_release_buffer")
        self._pinned = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_frame_ctx_destroy(self._h)      # (drops the registration before the buffer can go away)
            self._h = None
            self._buf = None
            self._pinned = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def render_frame_native(world: World, req: TileRequest, devices: Optional[Sequence[int]] = None):
    """rt_render_frame: the one-shot form of FrameContext (create, set world, one frame, destroy)."""
    lib = _abi.load()
    if not _is_initialised():
        init()
    n = req.width * req.height * 3
    out = np.empty(n, np.uint8)
    st = TileStats()
    if devices is None:
        dv, nd = None, 0
    else:
        dv, nd = (C.c_int * len(devices))(*devices), len(devices)
    _abi.check(lib.rt_render_frame(dv, nd, C.byref(req), _abi.ptr(world.spheres), len(world.spheres),
                                   _abi.ptr(world.triangles), len(world.triangles), _abi.ptr(world.world_index),
                                   out.ctypes.data_as(C.c_void_p), n, C.byref(st)), "rt_render_frame")
    return out.reshape(req.height, req.width, 3), st
