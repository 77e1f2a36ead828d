"""OBJ + MTL -> triangle list, as the controller does it (ray-tracer-controller/src/obj.rs:10-53).

The reference calls `tobj::load_obj_buf(.., &LoadOptions::default(), ..)` and for every model walks
`mesh.indices` three at a time: triangle (a, b, c) from `mesh.positions`, roughness =
`material.shininess / 1000`, albedo = `material.diffuse`, emission 0 (obj.rs:43-45).  The upload body is
the OBJ bytes followed by the MTL bytes, split at `obj_size` (controller main.rs:22-30, obj.rs:12,16).

This is a small parser for that subset: `v`, `f` (triangles; `v`, `v/vt`, `v/vt/vn`, `v//vn`, negative
indices), `o`/`g` (new model), `usemtl`, and in the MTL `newmtl`, `Kd`, `Ns`.  Faces with more than three
vertices are rejected: tobj's default options do not triangulate and the controller would mis-group them.
A mesh without `usemtl` panics in the reference (`material_id.unwrap()`): ValueError here.
"""
from __future__ import annotations

import numpy as np

from ._abi import TRIANGLE_DTYPE


def parse_mtl(text: str) -> dict[str, dict]:
    mats: dict[str, dict] = {}
    cur = None
    for line in text.splitlines():
        p = line.split("#", 1)[0].split()
        if not p:
            continue
        if p[0] == "newmtl":
            cur = mats.setdefault(" ".join(p[1:]), {"Kd": (0.0, 0.0, 0.0), "Ns": 0.0})   # tobj defaults
        elif cur is not None and p[0] == "Kd":
            cur["Kd"] = (float(p[1]), float(p[2]), float(p[3]))
        elif cur is not None and p[0] == "Ns":
            cur["Ns"] = float(p[1])
    return mats


def build_world(data: bytes, obj_size: int) -> np.ndarray:
    """= `obj::build_world(data, obj_size)`: triangles in model order, face order."""
    obj_text = data[:obj_size].decode("utf-8", errors="replace")
    mats = parse_mtl(data[obj_size:].decode("utf-8", errors="replace"))
    verts: list[tuple[float, float, float]] = []
    tris = []
    material = None
    for ln, line in enumerate(obj_text.splitlines(), 1):
        p = line.split("#", 1)[0].split()
        if not p:
            continue
        if p[0] == "v":
            verts.append((float(p[1]), float(p[2]), float(p[3])))
        elif p[0] == "usemtl":
            name = " ".join(p[1:])
            if name not in mats:
                raise ValueError(f"line {ln}: material `{name}` not in the MTL (tobj: Failed to load obj or mtl file)")
            material = mats[name]
        elif p[0] == "f":
            if len(p) != 4:
                raise ValueError(f"line {ln}: only triangular faces are supported (tobj default: no triangulation)")
            if material is None:
                raise ValueError(f"line {ln}: face without material (the reference unwraps mesh.material_id)")
            idx = []
            for tok in p[1:]:
                i = int(tok.split("/")[0])
                i = i - 1 if i > 0 else len(verts) + i
                if not (0 <= i < len(verts)):
                    raise ValueError(f"line {ln}: vertex index out of range")
                idx.append(i)
            a, b, c = (verts[i] for i in idx)
            kd = material["Kd"]
            tris.append((a, b, c, kd[0], kd[1], kd[2], np.float32(material["Ns"]) / np.float32(1000.0), 0.0))
    return np.array(tris, dtype=TRIANGLE_DTYPE) if tris else np.zeros(0, TRIANGLE_DTYPE)
