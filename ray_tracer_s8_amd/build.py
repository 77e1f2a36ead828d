"""Build recipe for the HIP C-ABI library (gfx950 only).

`hipcc --offload-arch=gfx950` cross-compiles without a GPU, so this runs in the CPU
container too.  The .so is built in-tree (ray_tracer_s8_amd/lib/) so it travels to the
GPU box with the repo snapshot; it is git-ignored, not gpurun-ignored.
"""
from __future__ import annotations

import os
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB_DIR = PKG / "lib"
LIB_PATH = LIB_DIR / "librt_s8.so"

# -ffp-contract=off: the kernel must perform the reference's IEEE binary32 operations one
# by one (Rust never contracts a*b+c); FMA appears only where written as __builtin_fmaf.
# Correctly rounded f32 sqrt/div is hipcc's default (-fhip-fp32-correctly-rounded-divide-sqrt).
# -fno-unroll-loops: only the loops marked `#pragma unroll` are unrolled; the heuristic unrolling of the divergent
# loops (samplers, root tests, path product) cost c3 3 % (tools/variants_all.sh).
HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",
    "-fno-unroll-loops",
    "-fPIC",
    "-shared",
    "-fvisibility=hidden",
]


def _sources() -> list[Path]:
    return [CSRC / "rt_api.hip"]


def _deps() -> list[Path]:
    return [CSRC / "rt_api.hip", CSRC / "rt_kernel.hip.h", CSRC / "rt_bvh.h", ROOT / "include" / "rt_tile.h", Path(__file__)]


def needs_build() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    return any(d.stat().st_mtime > t for d in _deps())


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile librt_s8.so if missing or stale.  Raises on failure (no fallback)."""
    if not force and not needs_build():
        return LIB_PATH
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "hipcc")
    extra = os.environ.get("RT_EXTRA_HIPCC_FLAGS", "").split()          # experiments only (e.g. -DRT_MAXC=12)
    cmd = [hipcc, *HIPCC_FLAGS, *extra, f"-I{ROOT / 'include'}", f"-I{CSRC}", "-o", str(LIB_PATH)]
    cmd += [str(s) for s in _sources()]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
