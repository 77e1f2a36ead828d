"""JSON wire codec of the reference's controller<->slave messages.

serde_json forms (ray-tracer-slave/src/lib.rs:10-30, shapes/mod.rs:23-27, sphere.rs:12-20,
mesh.rs:14-23, color.rs:5-10):
  RenderInfo  {"world":[Object...], "render_meta":{"height","width","divisions","id"}, "division_no":n}
  Object      externally tagged: {"Sphere":{"radius","center":[x,y,z],"node_index","p_albedo_at":{"r","g","b"},
              "p_roughness_at","p_emission_at"}} | {"Triangle":{"a":[..],"b":[..],"c":[..],"node_index",
              "p_albedo_at":{..},"p_roughness_at","p_emission_at"}}   (glam Vec3A serialises as [x,y,z])
  ImageSlice  {"division_no":n, "image":[u8 as decimal numbers...], "id":"hyphenated-uuid"}
f32 values travel as the shortest decimal of the f32 widened to f64 (serde_json `Value::from(f32)`), so the
round trip is exact; that is what `float(np.float32(x))` + Python's repr produce as well.

The reference `world` is one ordered list mixing both variants; the GPU ABI takes two typed arrays plus
`world_index`, the position of every primitive in that list (include/rt_tile.h "the world's order"), so decoding splits
the list and records where each entry stood, and encoding writes the entries back at their positions.
"""
from __future__ import annotations

import json
import uuid
from typing import Any

import numpy as np

from ._abi import SPHERE_DTYPE, TRIANGLE_DTYPE
from .interface import ImageSlice, RenderInfo, RenderMeta, RenderSettings, World


def _f(x) -> float:
    return float(np.float32(x))


def _color(rec) -> dict:
    return {"r": _f(rec["albedo_r"]), "g": _f(rec["albedo_g"]), "b": _f(rec["albedo_b"])}


def sphere_to_obj(s) -> dict:
    return {"Sphere": {"radius": _f(s["radius"]), "center": [_f(s["cx"]), _f(s["cy"]), _f(s["cz"])], "node_index": 0,
                       "p_albedo_at": _color(s), "p_roughness_at": _f(s["roughness"]),
                       "p_emission_at": _f(s["emission"])}}


def triangle_to_obj(t) -> dict:
    return {"Triangle": {"a": [_f(v) for v in t["a"]], "b": [_f(v) for v in t["b"]], "c": [_f(v) for v in t["c"]],
                         "node_index": 0, "p_albedo_at": _color(t), "p_roughness_at": _f(t["roughness"]),
                         "p_emission_at": _f(t["emission"])}}


def world_to_json_obj(world: World) -> list:
    return [sphere_to_obj(rec) if tag == "Sphere" else triangle_to_obj(rec) for tag, rec in world.objects()]


def world_from_json_obj(objs: list) -> World:
    sph, tri, pos_s, pos_t = [], [], [], []
    for pos, o in enumerate(objs):
        if not isinstance(o, dict) or len(o) != 1:
            raise ValueError("Object must be an externally tagged enum: {\"Sphere\":{..}} or {\"Triangle\":{..}}")
        (tag, v), = o.items()
        col = v["p_albedo_at"]
        if tag == "Sphere":
            c = v["center"]
            sph.append((c[0], c[1], c[2], v["radius"], col["r"], col["g"], col["b"], v["p_roughness_at"],
                        v["p_emission_at"]))
            pos_s.append(pos)
        elif tag == "Triangle":
            tri.append((tuple(v["a"]), tuple(v["b"]), tuple(v["c"]), col["r"], col["g"], col["b"],
                        v["p_roughness_at"], v["p_emission_at"]))
            pos_t.append(pos)
        else:
            raise ValueError(f"unknown variant `{tag}`, expected `Sphere` or `Triangle`")
    # world_index only when the list is NOT already "spheres, then triangles" (None means exactly that order)
    wi = np.array(pos_s + pos_t, dtype=np.uint32)
    interleaved = bool(len(wi)) and not np.array_equal(wi, np.arange(len(wi), dtype=np.uint32))
    return World(np.array(sph, dtype=SPHERE_DTYPE) if sph else np.zeros(0, SPHERE_DTYPE),
                 np.array(tri, dtype=TRIANGLE_DTYPE) if tri else np.zeros(0, TRIANGLE_DTYPE),
                 wi if interleaved else None)


def world_to_json_text(world: World) -> str:
    """The `world` array as compact JSON text, the same text json.dumps(world_to_json_obj(world)) gives (floats as the
    shortest repr of the f32 value widened to f64), assembled with one format call per primitive."""
    if world.world_index is not None:
        return json.dumps(world_to_json_obj(world), separators=(",", ":"))
    parts = []
    if len(world.spheres):
        s = world.spheres
        cols = [s[k].astype(np.float64).tolist() for k in ("radius", "cx", "cy", "cz", "albedo_b", "albedo_g", "albedo_r",
                                                            "roughness", "emission")]
        parts += ['{"Sphere":{"radius":%r,"center":[%r,%r,%r],"node_index":0,"p_albedo_at":{"r":%r,"g":%r,"b":%r},'
                  '"p_roughness_at":%r,"p_emission_at":%r}}' % (r, x, y, z, ar, ag, ab, ro, em)
                  for r, x, y, z, ab, ag, ar, ro, em in zip(*cols)]
    if len(world.triangles):
        parts += [json.dumps(triangle_to_obj(t), separators=(",", ":")) for t in world.triangles]
    return "[" + ",".join(parts) + "]"


def encode_render_info(info: RenderInfo, world_text: str | None = None) -> str:
    """What the controller POSTs to a slave (controller main.rs:47-63), fields in the struct's order (lib.rs:10-14).
    `world_text`: world_to_json_text(info.world) from an earlier call — the controller sends the same world with every
    strip of a job."""
    m = info.render_meta
    if world_text is None:
        world_text = world_to_json_text(info.world)
    return '{"world":%s,"render_meta":{"height":%d,"width":%d,"divisions":%d,"id":"%s"},"division_no":%d}' % (
        world_text, int(m.height), int(m.width), int(m.divisions), str(m.id), int(info.division_no))


_WORLD_CACHE: dict = {}          # digest of the world's JSON text -> World (a job's strips all carry the same text)


def _world_from_text(raw: bytes) -> World:
    import hashlib
    key = (len(raw), hashlib.blake2b(raw, digest_size=16).digest())
    w = _WORLD_CACHE.get(key)
    if w is None:
        w = world_from_json_obj(json.loads(raw))
        if len(_WORLD_CACHE) >= 4:
            _WORLD_CACHE.pop(next(iter(_WORLD_CACHE)))
        _WORLD_CACHE[key] = w
    return w


def decode_render_info(text: str | bytes, settings: RenderSettings | None = None) -> RenderInfo:
    """What the slave's `web::Json<RenderInfo>` extractor accepts (slave main.rs:148-152).  The world array is cut out
    of the text and decoded once per distinct text (serde_json writes `world` first; this module's older form wrote
    it last); anything else goes through the plain parser."""
    raw = text.encode() if isinstance(text, str) else bytes(text)
    world = None
    i = raw.find(b'"world":[')
    if i >= 0 and raw.count(b'"world":') == 1:
        a = i + len(b'"world":')
        b = raw.rfind(b'],"render_meta"')
        if b < a:
            b = raw.rfind(b"]") if raw.rstrip().endswith(b"]}") else -1
        if b >= a:
            try:
                rest = json.loads(raw[:a] + b"null" + raw[b + 1:])
                world = _world_from_text(raw[a:b + 1])
                d = rest
            except ValueError:
                world = None
    if world is None:
        d = json.loads(raw)
    for k in ("world", "render_meta", "division_no"):
        if k not in d:
            raise ValueError(f"missing field `{k}`")
    if world is None:
        world = world_from_json_obj(d["world"])
    rm = d["render_meta"]
    meta = RenderMeta(height=int(rm["height"]), width=int(rm["width"]), divisions=int(rm["divisions"]),
                      id=uuid.UUID(rm["id"]))
    return RenderInfo(world, meta, int(d["division_no"]), settings or RenderSettings())


_U8_TEXT = np.frombuffer("".join("%3d," % v for v in range(256)).encode(), dtype=np.uint8).reshape(256, 4).copy()


def encode_image_slice(s: ImageSlice) -> str:
    """What the slave POSTs to master:8080/result (slave main.rs:85-90): `image` is a JSON number array.
    Every element is written 3 wide, right-aligned ("  7, 42,255"): leading blanks are JSON whitespace, which
    serde_json skips, and the text is assembled with one numpy gather instead of a million str() calls."""
    img = np.asarray(s.image, dtype=np.uint8).reshape(-1)
    body = _U8_TEXT[img].reshape(-1)[:-1].tobytes().decode("ascii") if img.size else ""
    return '{"division_no":%d,"id":"%s","image":[%s]}' % (int(s.division_no), str(s.id), body)


def decode_image_slice(text: str | bytes) -> ImageSlice:
    """`ImageSlice` as serde_json writes it (key order of the struct: division_no, image, id) or as written above."""
    raw = text.encode() if isinstance(text, str) else bytes(text)
    lo, hi = raw.find(b'"image"'), -1
    if lo >= 0:
        lo = raw.find(b"[", lo)
        hi = raw.find(b"]", lo) if lo >= 0 else -1
    if lo >= 0 and hi > lo:
        # fast path: the array is parsed by numpy, the rest (two scalars) by json
        inner = raw[lo + 1:hi]
        d = json.loads(raw[:lo + 1] + raw[hi:])
        if inner.strip():
            img = np.fromstring(inner.decode("ascii"), dtype=np.int64, sep=",")
            if img.size != inner.count(b",") + 1:
                raise ValueError("image is not a flat array of integers")
        else:
            img = np.zeros(0, np.int64)
    else:
        d = json.loads(raw)
        img = np.asarray(d["image"], dtype=np.int64)
    if img.size and (img.min() < 0 or img.max() > 255):
        raise ValueError("image element out of range for u8")
    return ImageSlice(division_no=int(d["division_no"]), image=img.astype(np.uint8), id=uuid.UUID(d["id"]))
