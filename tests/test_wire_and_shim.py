"""JSON wire codec (serde_json forms of RenderInfo / ImageSlice) and the HTTP slave shim.
The shim is exercised end to end on localhost with a fake master; the renderer is injected (the CPU
oracle stands in for the GPU in this CPU test — the product default is the GPU and has no fallback)."""
import json
import threading
import uuid
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer

import numpy as np
import pytest

from ray_tracer_s8_amd import dispatch, scenes, wire
from ray_tracer_s8_amd.interface import ImageSlice, RenderInfo, RenderMeta, RenderSettings, World
from ray_tracer_s8_amd.slave_shim import REPLY, SlaveService

# a RenderInfo exactly as serde_json prints it (keys of `json!` objects come out sorted)
REFERENCE_STYLE = ('{"division_no":3,"render_meta":{"divisions":20,"height":1080,'
                   '"id":"67e55044-10b1-426f-9247-bb680e5fe0c8","width":1920},"world":['
                   '{"Sphere":{"center":[0.0,-101.0,-20.0],"node_index":0,"p_albedo_at":{"b":0.5,"g":0.5,"r":0.5},'
                   '"p_emission_at":0.0,"p_roughness_at":0.0,"radius":100.0}},'
                   '{"Triangle":{"a":[-1.0,-1.0,-3.0],"b":[1.0,-1.0,-3.0],"c":[0.0,1.0,-3.0],"node_index":7,'
                   '"p_albedo_at":{"b":0.10000000149011612,"g":0.800000011920929,"r":0.20000000298023224},'
                   '"p_emission_at":0.0,"p_roughness_at":0.30000001192092896}}]}')


def test_decode_reference_style_render_info():
    info = wire.decode_render_info(REFERENCE_STYLE)
    assert info.division_no == 3
    assert (info.render_meta.width, info.render_meta.height, info.render_meta.divisions) == (1920, 1080, 20)
    assert info.render_meta.id == uuid.UUID("67e55044-10b1-426f-9247-bb680e5fe0c8")
    s, t = info.world.spheres[0], info.world.triangles[0]
    assert (s["cy"], s["radius"], s["albedo_g"]) == (-101.0, 100.0, 0.5)
    assert t["albedo_b"] == np.float32(0.1) and t["roughness"] == np.float32(0.3)      # f32 -> f64 text -> f32 exact
    assert t["b"].tolist() == [1.0, -1.0, -3.0]
    rq = info.request()
    assert (rq.spp, rq.max_bounces, rq.division_no) == (100, 10, 3)                    # hard-coded knobs default


def test_render_info_round_trip_is_bit_exact():
    sph, tri = scenes.quad_room()
    info = RenderInfo(World(scenes.rand1024(n=40), tri), RenderMeta(64, 96, 4), 2, RenderSettings())
    back = wire.decode_render_info(wire.encode_render_info(info))
    assert back.world.spheres.tobytes() == info.world.spheres.tobytes()
    assert back.world.triangles.tobytes() == info.world.triangles.tobytes()
    assert back.render_meta == info.render_meta and back.division_no == 2
    # floats are printed as the f32 widened to f64, like serde_json's Value::from(f32)
    assert '"p_roughness_at":0.30000001192092896' in wire.encode_render_info(
        RenderInfo(World(np.array([(0, 0, -3, 1, .5, .5, .5, .3, 0)], dtype=sph.dtype)), RenderMeta(), 0))


def test_image_slice_codec():
    sl = ImageSlice(5, np.arange(256, dtype=np.uint8), uuid.uuid4())
    txt = wire.encode_image_slice(sl)
    d = json.loads(txt)
    assert d["image"][:4] == [0, 1, 2, 3] and d["id"] == str(sl.id) and d["division_no"] == 5
    back = wire.decode_image_slice(txt)
    assert np.array_equal(back.image, sl.image) and back.id == sl.id
    with pytest.raises(ValueError):
        wire.decode_image_slice('{"division_no":0,"image":[256],"id":"%s"}' % sl.id)


def test_bad_messages_are_rejected():
    with pytest.raises(ValueError):
        wire.decode_render_info('{"world":[],"render_meta":{"height":1,"width":1,"divisions":1,"id":"x"}}')
    with pytest.raises(ValueError):
        wire.decode_render_info(REFERENCE_STYLE.replace('"Sphere"', '"Cube"'))


from _fakes import FakeMaster as _FakeMaster


def test_slave_shim_end_to_end_with_fake_master(oracle):
    import urllib.request
    master = _FakeMaster()

    def render(info: RenderInfo) -> ImageSlice:          # injected renderer: the oracle (CPU test only)
        rgb, _, _ = oracle.render(info.request(), info.world.spheres, info.world.triangles, backend=1, nthreads=2)
        return ImageSlice(info.division_no, rgb, info.render_meta.id)

    svc = SlaveService(master_url=f"http://127.0.0.1:{master.port}/result", host="127.0.0.1", port=0,
                       settings=RenderSettings(spp=2, max_bounces=3), render_fn=render, fixed_seed=77).start()
    try:
        meta = RenderMeta(height=24, width=32, divisions=3)
        world = World(scenes.cornell16())
        for k in (2, 0, 1):                               # the controller fires all divisions at once
            body = wire.encode_render_info(RenderInfo(world, meta, k)).encode()
            req = urllib.request.Request(f"http://127.0.0.1:{svc.port}/", data=body,
                                         headers={"Content-Type": "application/json"}, method="POST")
            with urllib.request.urlopen(req, timeout=30) as r:
                assert r.read() == REPLY
        svc.wait_idle()
        assert len(master.got) == 3 and all(p == "/result" and ct == "application/json" for p, ct, _ in master.got)
        slices = [wire.decode_image_slice(b) for _, _, b in master.got]
        assert [s.division_no for s in slices] == [2, 0, 1]          # FIFO worker
        assert all(s.id == meta.id for s in slices)
        frame = dispatch.assemble([(s.division_no, s.image) for s in slices], 32, 24, 3)
        rq = RenderInfo(world, meta, 0, RenderSettings(spp=2, max_bounces=3, seed=77)).request()
        rq.divisions = 1
        ref, _, _ = oracle.render(rq, world.spheres, backend=1)
        assert np.array_equal(frame.reshape(-1), ref)
        # malformed body -> 400, wrong path -> 404, service keeps running
        for path, data, code in (("/", b"{not json", 400), ("/nope", b"{}", 404)):
            req = urllib.request.Request(f"http://127.0.0.1:{svc.port}{path}", data=data, method="POST")
            with pytest.raises(urllib.error.HTTPError) as e:
                urllib.request.urlopen(req, timeout=30)
            assert e.value.code == code
    finally:
        svc.stop()
        master.stop()
