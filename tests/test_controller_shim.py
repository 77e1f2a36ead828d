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
