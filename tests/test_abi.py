"""The C-ABI library loads and exports every symbol include/rt_tile.h declares (no compute
calls: this container has no GPU), argument checks that need no device, struct layouts."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

from ray_tracer_s8_amd import _abi

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "rt_tile.h").read_text()


def declared_functions():
    return re.findall(r"RT_API\s+[\w\s\*]+?\b(rt_\w+)\s*\(", HEADER)


def test_header_declares_expected_surface():
    names = declared_functions()
    for must in ("rt_init", "rt_shutdown", "rt_strerror", "rt_last_error", "rt_render_tile", "rt_scene_create",
                 "rt_scene_destroy", "rt_scene_render_tile", "rt_scene_render_tile_device",
                 "rt_scene_render_tiles_device", "rt_scene_render_tiles", "rt_scene_collect", "rt_render_frame",
                 "rt_frame_ctx_create", "rt_frame_ctx_set_world", "rt_frame_ctx_render", "rt_frame_ctx_release_buffer",
                 "rt_frame_ctx_destroy", "rt_tile_request_defaults", "rt_tile_bytes", "rt_abi_version"):
        assert must in names, must


def test_library_exports_every_declared_symbol():
    lib = _abi.load()
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in rt_tile.h but not exported by librt_s8.so"
    assert lib.rt_abi_version() == 4 == _abi.RT_ABI_VERSION


def _exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", str(path)], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("rt_")}


def test_product_library_exports_exactly_the_header():
    """`nm -D librt_s8.so`: the rt_* functions of rt_tile.h and nothing else — no rt_debug_* hook (those live in the test
    library lib/librt_s8_dbg.so, the same sources compiled with -DRT_DEBUG_HOOKS)."""
    from ray_tracer_s8_amd import build
    _abi.load()
    assert _exported(build.LIB_PATH) == set(declared_functions())
    _abi.load_debug()
    dbg = _exported(build.DEBUG_LIB_PATH)
    assert set(declared_functions()) < dbg
    assert {n for n in dbg if n.startswith("rt_debug_")} == {"rt_debug_read_counters", "rt_debug_sqrt_selftest", "rt_debug_set",
                                                              "rt_debug_throw"}


def test_abi_version_is_one_number_everywhere():
    # header, binding and the driver's build check must agree (build() once asserted a stale literal)
    m = re.search(r"#define\s+RT_ABI_VERSION\s+(\d+)u", HEADER)
    assert m and int(m.group(1)) == _abi.RT_ABI_VERSION
    entry = (Path(__file__).resolve().parents[1] / "__graft_entry__.py").read_text()
    assert "rt_abi_version() == _abi.RT_ABI_VERSION" in entry


def test_header_cites_reference_lines():
    # every entry point names the reference code it replaces
    for cite in ("main.rs:37-83", "main.rs:108-146", "camera.rs:109-129", "lib.rs:10-15", "sphere.rs:12-20",
                 "mesh.rs:14-23", "controller main.rs:47-75", "shapes/mod.rs:12"):
        assert cite in HEADER, cite


def test_struct_layouts_match_header():
    assert C.sizeof(_abi.TileRequest) == 64
    assert _abi.SPHERE_DTYPE.itemsize == 36 and _abi.TRIANGLE_DTYPE.itemsize == 56
    assert C.sizeof(_abi.TileStats) == 64
    assert C.sizeof(_abi.FrameStats) == 232 and _abi.FrameStats.wall_ms.offset == 64 and _abi.FrameStats.pinned.offset == 92
    assert _abi.FrameStats.assignment.offset == 96 and _abi.FrameStats.entry_segments.offset == 104
    offs = {n: getattr(_abi.TileRequest, n).offset for n, _ in _abi.TileRequest._fields_}
    assert offs["width"] == 0 and offs["spp"] == 16 and offs["aperture"] == 24 and offs["seed"] == 48 and offs["flags"] == 56


def test_defaults_are_the_reference_literals():
    lib = _abi.load()
    r = _abi.TileRequest()
    lib.rt_tile_request_defaults(C.byref(r))
    p = _abi.default_request()
    assert bytes(r) == bytes(p)
    assert (r.width, r.height, r.divisions) == (1920, 1080, 20)       # controller main.rs:33-39
    assert (r.spp, r.max_bounces) == (100, 10)                         # slave main.rs:39,51
    assert np.float32(r.aperture) == np.float32(0.1) and r.focus_distance == 1.0 and r.focal_length == 1.0
    assert np.float32(r.fov) == np.float32(np.pi) / np.float32(2)
    assert np.float32(r.t_min) == np.float32(0.001) and r.t_max == 1000.0
    assert lib.rt_tile_bytes(C.byref(r)) == 54 * 1920 * 3             # (1080/20)*1920*3, main.rs:53-59


def test_strerror_and_status_codes():
    lib = _abi.load()
    for code in range(0, -10, -1):
        s = lib.rt_strerror(code).decode()
        assert s and s != "unknown status"
    assert lib.rt_strerror(-99).decode() == "unknown status"


def test_fails_loudly_without_device_or_init():
    """No CPU fallback: without a device / without rt_init every compute entry point errors."""
    lib = _abi.load()
    n = C.c_int(-1)
    rc = lib.rt_init(C.byref(n))
    if rc == _abi.RT_OK:                       # running on a GPU box
        assert n.value >= 1
        return
    assert rc == _abi.RT_ERR_NO_DEVICE and n.value == 0
    assert "hipGetDeviceCount" in lib.rt_last_error().decode()
    sph = np.zeros(1, _abi.SPHERE_DTYPE)
    h = C.c_void_p()
    assert lib.rt_scene_create(0, _abi.ptr(sph), 1, None, 0, None, C.byref(h)) == _abi.RT_ERR_NOT_INITIALIZED
    rq = _abi.default_request(width=8, height=8, divisions=1, spp=1)
    out = np.zeros(8 * 8 * 3, np.uint8)
    st = _abi.TileStats()
    assert lib.rt_render_tile(0, C.byref(rq), _abi.ptr(sph), 1, None, 0, None, _abi.ptr(out), out.size, None,
                              C.byref(st)) == _abi.RT_ERR_NOT_INITIALIZED
    assert lib.rt_render_frame(None, 0, C.byref(rq), _abi.ptr(sph), 1, None, 0, None, _abi.ptr(out), out.size,
                               C.byref(st)) == _abi.RT_ERR_NOT_INITIALIZED
    fc = C.c_void_p()
    assert lib.rt_frame_ctx_create(None, 0, C.byref(fc)) == _abi.RT_ERR_NOT_INITIALIZED and not fc.value


def test_argument_checks_before_any_device_work():
    lib = _abi.load()
    sph = np.zeros(1, _abi.SPHERE_DTYPE)
    out = np.zeros(64, np.uint8)
    st = _abi.TileStats()
    rq = _abi.default_request(width=8, height=8, divisions=1, spp=1)
    call = lambda r, o=out, n=None: lib.rt_render_tile(0, C.byref(r), _abi.ptr(sph), 1, None, 0, None, _abi.ptr(o),
                                                       o.size if n is None else n, None, C.byref(st))
    bad = rq.copy(); bad.division_no = 1
    assert call(bad) == _abi.RT_ERR_BAD_ARG
    bad = rq.copy(); bad.spp = 0
    assert call(bad) == _abi.RT_ERR_BAD_ARG
    bad = rq.copy(); bad.max_bounces = _abi.RT_MAX_BOUNCES + 1
    assert call(bad) == _abi.RT_ERR_LIMIT
    bad = rq.copy(); bad.reserved = 1
    assert call(bad) == _abi.RT_ERR_BAD_ARG
    assert call(rq) == _abi.RT_ERR_BUFFER_TOO_SMALL                   # 64 < 8*8*3
    assert lib.rt_render_tile(0, None, None, 0, None, 0, None, None, 0, None, None) == _abi.RT_ERR_BAD_ARG
    assert lib.rt_tile_bytes(None) == 0


def test_product_package_never_imports_the_oracle():
    pkg = ROOT / "ray_tracer_s8_amd"
    for f in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")):
        txt = f.read_text()
        assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, re.M), f
        assert not re.search(r"#\s*include\s*[<\"][^>\"]*oracle", txt), f
        assert "librt_oracle" not in txt and "rt_oracle_" not in txt, f


def test_header_is_valid_c99_and_c_client_links(tmp_path):
    """The boundary is plain C: the header compiles as C99 and a C client links against librt_s8.so."""
    import shutil, subprocess
    gcc = shutil.which("gcc")
    assert gcc
    r = subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c",
                        str(ROOT / "include" / "rt_tile.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    _abi.load()
    exe = tmp_path / "render_frame"
    r = subprocess.run([gcc, "-std=c99", "-O2", "-Wall", f"-I{ROOT / 'include'}", str(ROOT / "examples" / "render_frame.c"),
                        f"-L{_abi.lib_path().parent}", "-lrt_s8", f"-Wl,-rpath,{_abi.lib_path().parent}",
                        "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([str(exe), str(tmp_path / "f.ppm")], capture_output=True, text=True)
    if run.returncode == 2:                       # CPU container: rt_init reports no device, loudly
        assert "no HIP device" in run.stderr
    else:
        assert run.returncode == 0 and "C_CLIENT_OK" in run.stdout, run.stdout + run.stderr


def test_nothing_unwinds_across_the_boundary():
    """rt_tile.h: "never throws or aborts across the boundary".  Every exported entry point runs its body through one
    guard (rt_api.hip: guarded()); rt_debug_throw raises inside such a body — the status comes back, with a message."""
    lib = _abi.load_debug()                  # (same sources, same guard; the hook that throws exists in the test library only)
    lib.rt_debug_throw.restype = C.c_int
    lib.rt_debug_throw.argtypes = [C.c_int]
    lib.rt_last_error.restype = C.c_char_p
    assert lib.rt_debug_throw(0) == -9                       # std::bad_alloc -> RT_ERR_OOM
    assert b"allocation" in lib.rt_last_error()
    assert lib.rt_debug_throw(1) == -7                       # std::exception -> RT_ERR_HIP, what() kept
    assert b"rt_debug_throw" in lib.rt_last_error()
    assert lib.rt_debug_throw(2) == -7                       # anything else
    assert lib.rt_debug_throw(3) in (-9, -7)                 # a real allocation failure (bad_alloc or length_error)
    assert lib.rt_debug_throw(4) == 0


def test_scene_size_limit_is_checked_before_anything_is_allocated():
    """More primitives than the kernels' 32-bit byte offsets can address: RT_ERR_LIMIT from rt_scene_create and
    rt_render_frame, on a box without a GPU too (argument checks come first)."""
    lib = _abi.load()
    sph = np.zeros(1, _abi.SPHERE_DTYPE)
    out = C.c_void_p()
    too_many = 0x3ffffff + 1
    rc = lib.rt_scene_create(0, sph.ctypes.data_as(C.c_void_p), C.c_uint32(too_many), None, C.c_uint32(0), None, C.byref(out))
    assert rc == -8 and not out.value
    rq = _abi.default_request(width=8, height=8, divisions=1, spp=1)
    buf = np.zeros(8 * 8 * 3, np.uint8)
    rc = lib.rt_render_frame(None, 0, C.byref(rq), sph.ctypes.data_as(C.c_void_p), C.c_uint32(too_many), None, C.c_uint32(0),
                             None, buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), None)
    assert rc == -8


def test_world_index_must_be_a_permutation():
    """world_index (rt_tile.h "the world's order") is checked before any device work: out of range, repeated."""
    lib = _abi.load()
    sph = np.zeros(2, _abi.SPHERE_DTYPE)
    tri = np.zeros(1, _abi.TRIANGLE_DTYPE)
    h = C.c_void_p()
    for bad in ([0, 1, 3], [0, 0, 1], [2, 2, 2]):
        wi = np.array(bad, np.uint32)
        assert lib.rt_scene_create(0, _abi.ptr(sph), 2, _abi.ptr(tri), 1, _abi.ptr(wi), C.byref(h)) == _abi.RT_ERR_BAD_ARG
        assert b"permutation" in lib.rt_last_error() and not h.value
    rq = _abi.default_request(width=8, height=8, divisions=1, spp=1)
    buf = np.zeros(8 * 8 * 3, np.uint8)
    wi = np.array([5, 0, 1], np.uint32)
    assert lib.rt_render_frame(None, 0, C.byref(rq), _abi.ptr(sph), 2, _abi.ptr(tri), 1, _abi.ptr(wi), _abi.ptr(buf),
                               buf.size, None) == _abi.RT_ERR_BAD_ARG
    assert lib.rt_render_tile(0, C.byref(rq), _abi.ptr(sph), 2, _abi.ptr(tri), 1, _abi.ptr(wi), _abi.ptr(buf), buf.size,
                              None, None) == _abi.RT_ERR_BAD_ARG
    with pytest.raises(ValueError):
        from ray_tracer_s8_amd.interface import World
        World(sph, tri, world_index=[0, 1])                      # one entry per primitive


def test_frame_context_lifecycle_and_error_paths():
    """rt_frame_ctx_*: argument errors come back as statuses (nothing dereferenced, nothing thrown), destroy(NULL) is a
    no-op; on a GPU box: create / destroy without a world, render before set_world, bad devices, frame-size errors."""
    lib = _abi.load()
    st = _abi.FrameStats()
    rq = _abi.default_request(width=8, height=8, divisions=1, spp=1)
    buf = np.zeros(8 * 8 * 3, np.uint8)
    sph = np.zeros(1, _abi.SPHERE_DTYPE)
    assert lib.rt_frame_ctx_create(None, 0, None) == _abi.RT_ERR_BAD_ARG
    assert lib.rt_frame_ctx_set_world(None, _abi.ptr(sph), 1, None, 0, None) == _abi.RT_ERR_BAD_ARG
    assert lib.rt_frame_ctx_render(None, C.byref(rq), _abi.ptr(buf), buf.size, C.byref(st)) == _abi.RT_ERR_BAD_ARG
    assert lib.rt_frame_ctx_release_buffer(None) == _abi.RT_ERR_BAD_ARG
    lib.rt_frame_ctx_destroy(None)
    n = C.c_int(0)
    if lib.rt_init(C.byref(n)) != _abi.RT_OK:                    # CPU container: no context can exist
        fc = C.c_void_p()
        assert lib.rt_frame_ctx_create(None, 0, C.byref(fc)) == _abi.RT_ERR_NOT_INITIALIZED
        return
    fc = C.c_void_p()
    bad_dev = (C.c_int * 1)(99)
    assert lib.rt_frame_ctx_create(bad_dev, 1, C.byref(fc)) == _abi.RT_ERR_BAD_DEVICE and not fc.value
    dev0 = (C.c_int * 2)(0, 0)
    assert lib.rt_frame_ctx_create(dev0, 2, C.byref(fc)) == _abi.RT_OK and fc.value
    assert lib.rt_frame_ctx_render(fc, C.byref(rq), _abi.ptr(buf), buf.size, C.byref(st)) == _abi.RT_ERR_BAD_ARG
    assert b"set_world" in lib.rt_last_error()
    assert lib.rt_frame_ctx_set_world(fc, None, 1, None, 0, None) == _abi.RT_ERR_BAD_ARG        # count without a pointer
    assert lib.rt_frame_ctx_set_world(fc, _abi.ptr(sph), 1, None, 0, None) == _abi.RT_OK
    assert lib.rt_frame_ctx_render(fc, None, _abi.ptr(buf), buf.size, None) == _abi.RT_ERR_BAD_ARG
    assert lib.rt_frame_ctx_render(fc, C.byref(rq), None, buf.size, None) == _abi.RT_ERR_BAD_ARG
    assert lib.rt_frame_ctx_render(fc, C.byref(rq), _abi.ptr(buf), 10, None) == _abi.RT_ERR_BUFFER_TOO_SMALL
    odd = rq.copy(); odd.divisions = 3
    assert lib.rt_frame_ctx_render(fc, C.byref(odd), _abi.ptr(buf), buf.size, None) == _abi.RT_ERR_FRAME_SIZE
    assert lib.rt_frame_ctx_render(fc, C.byref(rq), _abi.ptr(buf), buf.size, C.byref(st)) == _abi.RT_OK
    assert st.n_devices == 2 and st.totals.primary_rays == 64 and st.wall_ms > 0
    assert lib.rt_frame_ctx_release_buffer(fc) == _abi.RT_OK
    lib.rt_frame_ctx_destroy(fc)


def test_library_records_its_compile_flags(monkeypatch):
    """An A/B script that dies before restoring the default build must not leave a variant that looks fresh (round-2
    advisor): the compile lines are stored next to the library, a library built with other flags than the ones asked for
    now is stale whatever its mtime, and the driver's build check insists on the default flags."""
    from ray_tracer_s8_amd import build
    _abi.load()
    monkeypatch.delenv("RT_EXTRA_HIPCC_FLAGS", raising=False)
    assert build.built_with_default_flags() and not build.needs_build()
    monkeypatch.setenv("RT_EXTRA_HIPCC_FLAGS", "-DRT_PROFILE_TIME")
    assert build.needs_build()                                     # a variant is asked for: the default library is stale for it
    assert build.flags_record() != build.flags_record([]) and "-DRT_PROFILE_TIME" in build.flags_record()
    entry = (ROOT / "__graft_entry__.py").read_text()
    assert "built_with_default_flags()" in entry


def test_two_hip_runtimes_in_one_process_are_refused():
    """Round 2 left an unexplained `torch.cuda.Stream()` -> "no ROCm-capable device" after 122 tests of this library.  Cause
    (profiles/README.md): the torch wheel bundles its own libamdhip64.so (no SONAME), the library asks for ROCm's
    libamdhip64.so.7; the loader binds the library's hip* symbols to whichever runtime is in the global scope FIRST.  torch
    first: one runtime serves both (bench.py's order).  Library first, torch later: two runtimes, the second finds no
    device.  rt.init() now refuses that order with a message instead of leaving torch to fail later."""
    import subprocess, sys
    code_ok = ("import torch; from ray_tracer_s8_amd import _abi; import os; "
               "p = _abi.hip_runtime_path(); _abi.check_single_hip_runtime(); "
               "print('BOUND', os.path.dirname(os.path.realpath(p)) == os.path.realpath(os.path.join(os.path.dirname(torch.__file__), 'lib')))")
    r = subprocess.run([sys.executable, "-c", code_ok], capture_output=True, text=True, cwd=str(ROOT), timeout=300)
    assert r.returncode == 0 and "BOUND True" in r.stdout, r.stdout + r.stderr
    code_bad = ("from ray_tracer_s8_amd import _abi; p = _abi.hip_runtime_path(); import torch\n"
                "try:\n    _abi.check_single_hip_runtime(); print('ACCEPTED', p)\n"
                "except RuntimeError as e:\n    print('REFUSED', 'two HIP runtimes' in str(e), p)")
    r = subprocess.run([sys.executable, "-c", code_bad], capture_output=True, text=True, cwd=str(ROOT), timeout=300)
    assert r.returncode == 0 and "REFUSED True" in r.stdout and "/opt/rocm" in r.stdout, r.stdout + r.stderr


def test_frame_context_wrapper_releases_a_registration_before_switching_buffers():
    """interface.FrameContext against a recording stand-in for the library (no GPU): the array handed to rt_frame_ctx_render
    stays referenced by the wrapper, and rt_frame_ctx_release_buffer is called BEFORE a different array is passed —
    including the context's own buffer being replaced when the frame size changes."""
    from ray_tracer_s8_amd import interface

    calls = []

    class FakeLib:
        def rt_frame_ctx_render(self, h, rq, ptr, n, fs):
            calls.append(("render", C.cast(ptr, C.c_void_p).value))
            return 0

        def rt_frame_ctx_release_buffer(self, h):
            calls.append(("release",))
            return 0

        def rt_frame_ctx_destroy(self, h):
            calls.append(("destroy",))

    fc = interface.FrameContext.__new__(interface.FrameContext)
    fc._lib, fc._h, fc._buf, fc._pinned = FakeLib(), C.c_void_p(1), None, None
    rq = _abi.default_request(width=8, height=4, divisions=2, spp=1)
    a = np.zeros(8 * 4 * 3, np.uint8)
    fc.render(rq, out=a)
    fc.render(rq, out=a)
    assert [c[0] for c in calls] == ["render", "render"] and fc._pinned is a          # same array: registration kept
    b = np.zeros(8 * 4 * 3, np.uint8)
    fc.render(rq, out=b)
    assert [c[0] for c in calls[2:]] == ["release", "render"] and fc._pinned is b     # a stays alive until the release
    fc.render(rq)                                                                      # the context's own buffer
    own = fc._buf
    assert [c[0] for c in calls[4:]] == ["release", "render"] and fc._pinned is own
    rq2 = _abi.default_request(width=16, height=4, divisions=2, spp=1)
    fc.render(rq2)                                                                     # frame size changes: a new own buffer
    assert [c[0] for c in calls[6:]] == ["release", "render"] and fc._pinned is fc._buf and fc._buf is not own
    fc.close()
    assert calls[-1] == ("destroy",) and fc._pinned is None
