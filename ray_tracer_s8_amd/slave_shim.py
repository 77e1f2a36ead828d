"""Minimal HTTP slave: a literal drop-in for a docker `ray-tracer-slave`.

Reference behaviour reproduced (ray-tracer-slave/src/main.rs:148-174, 32-106):
  POST /  (JSON RenderInfo)  -> replies "i'll get you a slice at once" immediately, queues the job;
  ONE worker thread drains the queue FIFO, renders the strip, POSTs JSON ImageSlice to
  http://master:8080/result.
Here the worker renders on a GPU through the C-ABI (`Slave`), so the unmodified controller can dispatch
strips to GPUs instead of CPU slaves.  `python -m ray_tracer_s8_amd.slave_shim --device 0 --port 8081`.

`render_fn` can be injected (tests use it to run the service without a GPU); the default is the GPU path and
it fails loudly if the HIP library or device is missing.
"""
from __future__ import annotations

import argparse
import logging
import queue
import secrets
import threading
import urllib.request
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer
from typing import Callable, Optional

from . import wire
from .interface import ImageSlice, RenderInfo, RenderSettings, Slave

log = logging.getLogger("ray_tracer_s8_amd.slave")
REPLY = b"i'll get you a slice at once"              # slave main.rs:153
MAX_BODY = 500_000_000                                # JsonConfig limit, slave main.rs:167


class SlaveService:
    def __init__(self, device: int = 0, master_url: str = "http://master:8080/result", host: str = "0.0.0.0",
                 port: int = 8081, settings: Optional[RenderSettings] = None,
                 render_fn: Optional[Callable[[RenderInfo], ImageSlice]] = None, fixed_seed: Optional[int] = None):
        self.master_url = master_url
        self.settings = settings or RenderSettings()
        self.fixed_seed = fixed_seed
        self._slave = None
        if render_fn is None:
            self._slave = Slave(device)               # GPU path (rt_init happens on first render)
            render_fn = self._slave.render
        self._render = render_fn
        self._q: "queue.Queue[Optional[RenderInfo]]" = queue.Queue()   # unbounded channel, main.rs:159
        self._worker = threading.Thread(target=self._work, daemon=True)
        svc = self

        class Handler(BaseHTTPRequestHandler):
            def log_message(self, fmt, *a):            # route to logging, like RUST_LOG=info
                log.info("%s " + fmt, self.address_string(), *a)

            def do_POST(self):
                if self.path != "/":
                    self.send_error(404)
                    return
                n = int(self.headers.get("Content-Length", "0"))
                if n > MAX_BODY:
                    self.send_error(413)
                    return
                try:
                    info = wire.decode_render_info(self.rfile.read(n), svc._job_settings())
                except Exception as e:                 # actix answers 400 on a JSON extractor error
                    self.send_error(400, str(e))
                    return
                log.info("Slave Got request")
                svc._q.put(info)
                self.send_response(200)
                self.send_header("Content-Type", "text/plain; charset=utf-8")
                self.send_header("Content-Length", str(len(REPLY)))
                self.end_headers()
                self.wfile.write(REPLY)

        self._httpd = ThreadingHTTPServer((host, port), Handler)
        self.port = self._httpd.server_address[1]
        self._server_thread = threading.Thread(target=self._httpd.serve_forever, daemon=True)

    def _job_settings(self) -> RenderSettings:
        s = RenderSettings(**self.settings.__dict__)
        # the reference seeds from entropy per row (main.rs:69); per job here, unless pinned
        s.seed = self.fixed_seed if self.fixed_seed is not None else secrets.randbits(64)
        return s

    def _work(self):
        while True:
            info = self._q.get()
            if info is None:
                return
            try:
                log.info("Got job")
                sl = self._render(info)
                log.info("render finished")
                body = wire.encode_image_slice(sl).encode()
                req = urllib.request.Request(self.master_url, data=body,
                                             headers={"Content-Type": "application/json"}, method="POST")
                with urllib.request.urlopen(req, timeout=60) as resp:
                    log.info("master responded to result:  %s", resp.read().decode(errors="replace"))
            except Exception:                          # the reference unwrap()s and dies; keep serving
                log.exception("job failed")
            finally:
                self._q.task_done()

    def start(self):
        self._worker.start()
        self._server_thread.start()
        return self

    def wait_idle(self):
        self._q.join()

    def stop(self):
        self._httpd.shutdown()
        self._httpd.server_close()
        self._q.put(None)
        if self._slave:
            self._slave.close()


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--port", type=int, default=8081)
    ap.add_argument("--master-url", default="http://master:8080/result")
    ap.add_argument("--spp", type=int, default=100)
    ap.add_argument("--max-bounces", type=int, default=10)
    ap.add_argument("--seed", type=int, default=None)
    a = ap.parse_args()
    logging.basicConfig(level=logging.INFO)
    svc = SlaveService(a.device, a.master_url, port=a.port,
                       settings=RenderSettings(spp=a.spp, max_bounces=a.max_bounces), fixed_seed=a.seed).start()
    log.info("GPU slave listening on :%d, results to %s", svc.port, a.master_url)
    threading.Event().wait()


if __name__ == "__main__":
    main()
