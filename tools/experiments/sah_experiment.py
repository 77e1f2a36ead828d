"""Driver of sah_experiment.cpp: reference 6-bucket tree vs a 3-axis binned SAH tree on the c3 / c5 fields (CPU)."""
import ctypes as C, subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from ray_tracer_s8_amd import scenes
so = Path("/tmp/libsah_exp.so")
subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", f"-I{ROOT / 'ray_tracer_s8_amd' / 'csrc'}", "-o", str(so),
                str(Path(__file__).with_suffix(".cpp"))], check=True)
lib = C.CDLL(str(so))
for name in ("c3", "c5"):
    sph, _ = scenes.config(name)
    c = np.stack([sph["cx"], sph["cy"], sph["cz"]], 1).astype(np.float32)
    r = sph["radius"].astype(np.float32)[:, None]
    boxes = np.ascontiguousarray(np.concatenate([c - r, c + r], 1), np.float32)
    for nb in (6, 16, 32):
        out = np.zeros(8)
        lib.sah_experiment(boxes.ctypes.data_as(C.c_void_p), C.c_uint32(len(boxes)), C.c_int(nb), out.ctypes.data_as(C.c_void_p))
        print(f"{name} bins {nb:2d}: SAH ref {out[0]:8.2f} alt {out[1]:8.2f} | camera rays: visits ref {out[2]:6.1f} alt {out[3]:6.1f}"
              f" | bounce rays: ref {out[4]:6.1f} alt {out[5]:6.1f} | leaves/ray ref {out[6]:.2f} alt {out[7]:.2f}")
