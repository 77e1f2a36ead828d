"""Minimal HTTP controller: the reference's job surface with GPUs as the slaves.

Reference behaviour reproduced (ray-tracer-controller/src/main.rs):
  POST /upload/{obj_size}/   body = OBJ bytes ++ MTL bytes -> new job (UUID v4), world from obj.rs rules,
                             one RenderInfo per division_no in 0..divisions; replies with the job id (:22-77)
  POST /result               JSON ImageSlice from a slave -> stored with its job (:79-93)
  POST /poll                 body = job id -> JPEG (quality 90) of the assembled frame once every division is in,
                             else "Job not finished yet k/n"; "No such job"; "Invalid Uuid" (:95-142)
The reference hard-codes 1920x1080 and 20 divisions (:33-39); they are constructor arguments here with those
defaults.  Strips go either to in-process GPU slaves (`devices=[0, 1, ...]`, the default: the C-ABI renderer) or,
like the reference, to HTTP slaves (`slave_urls=[...]`, e.g. ray_tracer_s8_amd.slave_shim instances), which
answer asynchronously on /result.  JPEG encoding uses Pillow (the reference uses the `image` crate; the bytes
are not claimed to be identical — off the hot path).
"""
from __future__ import annotations

import io
import logging
import threading
import urllib.request
import uuid
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer
from typing import Callable, Optional, Sequence

import numpy as np

from . import obj, wire
from .dispatch import assemble
from .interface import ImageSlice, RenderInfo, RenderMeta, RenderSettings, Slave, World

log = logging.getLogger("ray_tracer_s8_amd.controller")


def encode_jpeg(frame: np.ndarray, quality: int = 90) -> bytes:
    from PIL import Image            # Pillow ships in the image; raise loudly if it does not
    buf = io.BytesIO()
    Image.fromarray(frame, "RGB").save(buf, format="JPEG", quality=quality)
    return buf.getvalue()


class ControllerService:
    def __init__(self, devices: Optional[Sequence[int]] = None, slave_urls: Optional[Sequence[str]] = None,
                 host: str = "0.0.0.0", port: int = 8080, width: int = 1920, height: int = 1080, divisions: int = 20,
                 settings: Optional[RenderSettings] = None,
                 render_fn: Optional[Callable[[RenderInfo], ImageSlice]] = None):
        self.width, self.height, self.divisions = width, height, divisions
        self.settings = settings or RenderSettings()
        self.slave_urls = list(slave_urls) if slave_urls else None
        self._render_fns: list[Callable[[RenderInfo], ImageSlice]] = []
        self._slaves: list[Slave] = []
        if self.slave_urls is None:
            if render_fn is not None:
                self._render_fns = [render_fn]
            else:
                from .interface import init
                devs = list(devices) if devices is not None else list(range(init()))
                self._slaves = [Slave(d) for d in devs]
                self._render_fns = [s.render for s in self._slaves]
        self._jobs: dict[uuid.UUID, dict] = {}
        self._lock = threading.Lock()            # RwLock<AppState>, main.rs:147
        svc = self

        class Handler(BaseHTTPRequestHandler):
            def log_message(self, fmt, *a):
                log.info("%s " + fmt, self.address_string(), *a)

            def _reply(self, body: bytes, ctype: str = "text/plain; charset=utf-8"):
                self.send_response(200)
                self.send_header("Content-Type", ctype)
                self.send_header("Content-Length", str(len(body)))
                self.end_headers()
                self.wfile.write(body)

            def do_POST(self):
                body = self.rfile.read(int(self.headers.get("Content-Length", "0")))
                parts = [p for p in self.path.split("/") if p]
                try:
                    if len(parts) == 2 and parts[0] == "upload":
                        self._reply(svc.upload(body, int(parts[1])).encode())
                    elif parts == ["result"]:
                        self._reply(svc.result(body).encode())
                    elif parts == ["poll"]:
                        out = svc.poll(body.decode(errors="replace"))
                        self._reply(out, "image/jpeg" if out[:2] == b"\xff\xd8" else "text/plain; charset=utf-8")
                    else:
                        self.send_error(404)
                except Exception as e:        # the reference panics (500 / dropped connection)
                    log.exception("request failed")
                    self.send_error(500, str(e))

        self._httpd = ThreadingHTTPServer((host, port), Handler)
        self.port = self._httpd.server_address[1]
        self._thread = threading.Thread(target=self._httpd.serve_forever, daemon=True)

    # ---- /upload/{obj_size}/ (main.rs:22-77)
    def upload(self, body: bytes, obj_size: int) -> str:
        job_id = uuid.uuid4()
        meta = RenderMeta(height=self.height, width=self.width, divisions=self.divisions, id=job_id)
        world = World(triangles=obj.build_world(body, obj_size))      # panics -> 500 on a bad OBJ/MTL
        with self._lock:
            self._jobs[job_id] = {"meta": meta, "result": []}
        infos = [RenderInfo(world, meta, k, self.settings) for k in range(self.divisions)]
        if self.slave_urls is not None:
            world_text = wire.world_to_json_text(world)                # the same world goes out with every strip
            for k, info in enumerate(infos):                            # one POST per division (main.rs:47-75)
                url = self.slave_urls[k % len(self.slave_urls)]
                req = urllib.request.Request(url, data=wire.encode_render_info(info, world_text).encode(),
                                             headers={"Content-Type": "application/json"}, method="POST")
                with urllib.request.urlopen(req, timeout=60) as r:
                    log.info("Response from slave: %s", r.read().decode(errors="replace"))
        else:
            def work(w: int):
                for k in range(w, self.divisions, len(self._render_fns)):
                    sl = self._render_fns[w](infos[k])
                    self._store(ImageSlice(sl.division_no, sl.image, job_id))
            for w in range(len(self._render_fns)):
                threading.Thread(target=work, args=(w,), daemon=True).start()
        return str(job_id)

    def _store(self, sl: ImageSlice) -> bool:
        with self._lock:
            job = self._jobs.get(sl.id)
            if job is None:
                log.info("result not saved. ID wrong? : %s", sl.id)
                return False
            job["result"].append(sl)
            return True

    # ---- /result (main.rs:79-93)
    def result(self, body: bytes) -> str:
        self._store(wire.decode_image_slice(body))
        return "slice saved. thank you slave."

    # ---- /poll (main.rs:95-142)
    def poll(self, text: str) -> bytes:
        try:
            job_id = uuid.UUID(text.strip())
        except ValueError:
            return b"Invalid Uuid"
        with self._lock:
            job = self._jobs.get(job_id)
            if job is None:
                return b"No such job"
            meta: RenderMeta = job["meta"]
            have = {s.division_no for s in job["result"]}
            done = sum(1 for i in range(meta.divisions) if i in have)
            if done < meta.divisions:
                return f"Job not finished yet {done}/{meta.divisions}".encode()
            first = {}
            for s in job["result"]:
                first.setdefault(s.division_no, s.image)
            frame = assemble(first.items(), meta.width, meta.height, meta.divisions)   # sort + concat, :109-119
            del self._jobs[job_id]                                                    # state.jobs.remove(idx)
        return encode_jpeg(frame, 90)

    def start(self):
        self._thread.start()
        return self

    def stop(self):
        self._httpd.shutdown()
        self._httpd.server_close()
        for s in self._slaves:
            s.close()


def main():
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--port", type=int, default=8080)
    ap.add_argument("--slave-url", action="append", help="HTTP slave(s); default: in-process GPU slaves")
    ap.add_argument("--spp", type=int, default=100)
    a = ap.parse_args()
    logging.basicConfig(level=logging.INFO)
    ControllerService(slave_urls=a.slave_url, port=a.port, settings=RenderSettings(spp=a.spp)).start()
    threading.Event().wait()


if __name__ == "__main__":
    main()
