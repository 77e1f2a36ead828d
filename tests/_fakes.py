"""Test doubles shared by CPU and GPU tests."""
import threading
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer


class FakeMaster:
    """POST /result collector with the controller's reply text (controller main.rs:79-93)."""

    def __init__(self):
        got = self.got = []

        class H(BaseHTTPRequestHandler):
            def log_message(self, *a):
                pass

            def do_POST(self):
                body = self.rfile.read(int(self.headers["Content-Length"]))
                got.append((self.path, self.headers.get("Content-Type"), body))
                msg = b"slice saved. thank you slave."
                self.send_response(200)
                self.send_header("Content-Length", str(len(msg)))
                self.end_headers()
                self.wfile.write(msg)

        self.httpd = ThreadingHTTPServer(("127.0.0.1", 0), H)
        self.port = self.httpd.server_address[1]
        threading.Thread(target=self.httpd.serve_forever, daemon=True).start()

    def stop(self):
        self.httpd.shutdown()
        self.httpd.server_close()


