"""The whole user-visible loop on localhost: upload OBJ+MTL -> controller -> HTTP slave(s) -> /result ->
/poll -> JPEG, with the reference's reply texts.  The renderer is injected (CPU oracle) in this CPU test."""
import io
import time
import urllib.request
import uuid

import numpy as np
import pytest

from ray_tracer_s8_amd import obj
from ray_tracer_s8_amd.controller_shim import ControllerService
from ray_tracer_s8_amd.interface import ImageSlice, RenderInfo, RenderSettings
from ray_tracer_s8_amd.slave_shim import SlaveService
from test_obj import MTL, OBJ

PIL = pytest.importorskip("PIL.Image")


def _post(url, data):
    req = urllib.request.Request(url, data=data, method="POST")
    with urllib.request.urlopen(req, timeout=60) as r:
        return r.read()


def _oracle_render(oracle):
    def render(info: RenderInfo) -> ImageSlice:
        rgb, _, _ = oracle.render(info.request(), info.world.spheres, info.world.triangles, backend=1, nthreads=2)
        return ImageSlice(info.division_no, rgb, info.render_meta.id)
    return render


@pytest.mark.parametrize("via_http_slaves", [False, True])
def test_upload_poll_round_trip(oracle, via_http_slaves):
    settings = RenderSettings(spp=4, max_bounces=3, seed=5)
    slave = None
    if via_http_slaves:
        ctl = ControllerService(slave_urls=["placeholder"], host="127.0.0.1", port=0, width=48, height=32,
                                divisions=4, settings=settings)
        slave = SlaveService(master_url=f"http://127.0.0.1:{ctl.port}/result", host="127.0.0.1", port=0,
                             settings=settings, render_fn=_oracle_render(oracle), fixed_seed=5).start()
        ctl.slave_urls = [f"http://127.0.0.1:{slave.port}/"]
    else:
        ctl = ControllerService(host="127.0.0.1", port=0, width=48, height=32, divisions=4, settings=settings,
                                render_fn=_oracle_render(oracle))
    ctl.start()
    base = f"http://127.0.0.1:{ctl.port}"
    try:
        assert _post(base + "/poll", b"not-a-uuid") == b"Invalid Uuid"
        assert _post(base + "/poll", str(uuid.uuid4()).encode()) == b"No such job"
        job = _post(base + f"/upload/{len(OBJ)}/", OBJ + MTL).decode()
        uuid.UUID(job)
        out = b""
        for _ in range(200):
            out = _post(base + "/poll", job.encode())
            if out[:2] == b"\xff\xd8":
                break
            assert out.startswith(b"Job not finished yet ")
            time.sleep(0.05)
        assert out[:2] == b"\xff\xd8", out[:60]
        img = np.asarray(PIL.open(io.BytesIO(out)).convert("RGB"))
        assert img.shape == (32, 48, 3)
        tris = obj.build_world(OBJ + MTL, len(OBJ))
        rq = RenderInfo.__new__(RenderInfo)
        from ray_tracer_s8_amd._abi import default_request
        r = default_request(width=48, height=32, divisions=1, spp=4, max_bounces=3, seed=5)
        ref, _, _ = oracle.render(r, None, tris, backend=1)
        ref = ref.reshape(32, 48, 3).astype(np.float64)
        assert np.abs(img.astype(np.float64) - ref).mean() < 6.0          # JPEG q90 is lossy
        assert _post(base + "/poll", job.encode()) == b"No such job"       # a finished job is removed (main.rs:122)
    finally:
        ctl.stop()
        if slave:
            slave.stop()


def test_two_overlapping_jobs_share_the_slaves_safely(oracle):
    """Two different worlds uploaded before the first poll (the reference keeps a list of concurrent jobs,
    controller main.rs:13-20): the in-process slaves are shared by both jobs' dispatcher threads.  The fake slave below
    checks what the per-Slave lock of interface.Slave guarantees: never two renders inside one slave at a time."""
    import threading
    from ray_tracer_s8_amd import interface
    settings = RenderSettings(spp=2, max_bounces=2, seed=5)
    inside = [0]
    overlap = []
    guard = threading.Lock()

    class FakeScene:
        def __init__(self, device, world):
            self.world = world

        def render_tile(self, req):
            with guard:
                inside[0] += 1
                overlap.append(inside[0])
            time.sleep(0.01)
            rgb, _, _ = oracle.render(req, self.world.spheres, self.world.triangles, backend=1, nthreads=1)
            with guard:
                inside[0] -= 1
            return rgb, None, None

        def close(self):
            pass

    real = interface.Scene
    interface.Scene = FakeScene
    try:
        slave = interface.Slave(0)
        ctl = ControllerService(host="127.0.0.1", port=0, width=32, height=24, divisions=6, settings=settings,
                                render_fn=slave.render)
        ctl._render_fns = [slave.render, slave.render, slave.render]       # three dispatcher threads, ONE slave
        ctl.start()
        base = f"http://127.0.0.1:{ctl.port}"
        obj2 = b"v -2 -1 -4\nv 2 -1 -4\nv 0 2 -4\nusemtl red\nf 1 2 3\n"
        try:
            jobs = [_post(base + f"/upload/{len(o)}/", o + MTL).decode() for o in (OBJ, obj2, OBJ)]
            for job, o in zip(jobs, (OBJ, obj2, OBJ)):
                out = b""
                for _ in range(400):
                    out = _post(base + "/poll", job.encode())
                    if out[:2] == b"\xff\xd8":
                        break
                    time.sleep(0.02)
                assert out[:2] == b"\xff\xd8", out[:60]
                img = np.asarray(PIL.open(io.BytesIO(out)).convert("RGB")).astype(np.float64)
                from ray_tracer_s8_amd._abi import default_request
                r = default_request(width=32, height=24, divisions=1, spp=2, max_bounces=2, seed=5)
                ref, _, _ = oracle.render(r, None, obj.build_world(o + MTL, len(o)), backend=1)
                assert np.abs(img - ref.reshape(24, 32, 3)).mean() < 8.0
        finally:
            ctl.stop()
        assert overlap and max(overlap) == 1
    finally:
        interface.Scene = real
