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
# RT_LIB_VARIANT=<name> (experiments): build and load lib/librt_s8_<name>.so with RT_EXTRA_HIPCC_FLAGS, objects in their own
# directory — the product library lib/librt_s8.so is never rebuilt in place by an A/B run (round-2 advisor)
VARIANT = os.environ.get("RT_LIB_VARIANT", "")
LIB_PATH = LIB_DIR / (f"librt_s8_{VARIANT}.so" if VARIANT else "librt_s8.so")

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


# translation units and the flags each one adds: the host side, the linear-scan kernels, the traversal kernels
# (SLP-vectorised packed FP32 pairs: linear kernels +3 %, traversal kernels -1.5 %, tools/variants_all.sh)
UNITS = [
    ("rt_api.hip", []),
    ("rt_kernels_lin.hip", []),
    ("rt_kernels_trav.hip", ["-fno-slp-vectorize"]),
]


def _deps() -> list[Path]:
    return [CSRC / u for u, _ in UNITS] + [CSRC / "rt_kernel.hip.h", CSRC / "rt_cull.h", CSRC / "rt_bvh.h", CSRC / "rt_assign.h", ROOT / "include" / "rt_tile.h",
                                            Path(__file__)]


FLAGS_PATH = LIB_PATH.with_suffix(".flags")      # the exact compile lines of the library next to it

# The TEST library: the product sources plus -DRT_DEBUG_HOOKS, which compiles the rt_debug_* entry points (launch-path knobs,
# counter read-back, a throwing body, the sqrt self-test) that tests and tools use.  The product library exports exactly
# include/rt_tile.h (tests/test_abi.py checks both).
DEBUG_VARIANT = "dbg"
DEBUG_FLAGS = ["-DRT_DEBUG_HOOKS"]
DEBUG_LIB_PATH = LIB_DIR / f"librt_s8_{DEBUG_VARIANT}.so"


def _extra_flags() -> list[str]:
    return os.environ.get("RT_EXTRA_HIPCC_FLAGS", "").split()          # experiments only (e.g. -DRT_MAXC=12)


def flags_record(extra: list[str] | None = None) -> str:
    """What the library was (or would be) compiled with: one line per translation unit.  A library whose record differs
    from the flags asked for NOW is stale whatever its mtime — an A/B script that died before restoring the default
    build (tools/ab*.sh, phase_time.py: -DRT_PROFILE_TIME, experimental -D knobs) would otherwise leave a variant that
    looks fresh, travels to the GPU box and gets benchmarked."""
    extra = _extra_flags() if extra is None else extra
    common = [f for f in HIPCC_FLAGS if f != "-shared"] + extra
    return "".join(f"{unit}: {' '.join(common + flags)}\n" for unit, flags in UNITS)


def built_with_default_flags() -> bool:
    """True iff the library on disk was compiled with exactly HIPCC_FLAGS (no RT_EXTRA_HIPCC_FLAGS)."""
    return LIB_PATH.exists() and FLAGS_PATH.exists() and FLAGS_PATH.read_text() == flags_record([])


def needs_build() -> bool:
    if not LIB_PATH.exists() or not FLAGS_PATH.exists():
        return True
    if VARIANT and "RT_EXTRA_HIPCC_FLAGS" not in os.environ:
        return False          # a prebuilt experiment variant is used as it was built (its flags are in its record)
    if FLAGS_PATH.read_text() != flags_record():
        return True
    t = LIB_PATH.stat().st_mtime
    return any(d.stat().st_mtime > t for d in _deps())


def _compile(lib_path: Path, flags_path: Path, obj_dir: Path, extra: list[str], verbose: bool) -> Path:
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "hipcc")
    if flags_path.exists():
        flags_path.unlink()                       # no record while the objects are in flux
    obj_dir.mkdir(exist_ok=True)
    common = [f for f in HIPCC_FLAGS if f != "-shared"] + extra + [f"-I{ROOT / 'include'}", f"-I{CSRC}"]
    cmds, objs = [], []
    for unit, flags in UNITS:
        obj = obj_dir / (Path(unit).stem + ".o")
        objs.append(str(obj))
        cmds.append([hipcc, *common, *flags, "-c", str(CSRC / unit), "-o", str(obj)])
    cmds.append([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib_path), *objs])
    # the three compiles are independent: run them side by side, then link
    procs = []
    for cmd in cmds[:-1]:
        if verbose:
            print(" ".join(cmd))
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for cmd, pr in zip(cmds[:-1], procs):
        so, se = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed ({pr.returncode}): {' '.join(cmd)}\n{so}\n{se}")
    if verbose:
        print(" ".join(cmds[-1]))
    proc = subprocess.run(cmds[-1], capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc link failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    flags_path.write_text(flags_record(extra))
    return lib_path


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile librt_s8.so (or the RT_LIB_VARIANT named) if missing or stale.  Raises on failure (no fallback)."""
    if not force and not needs_build():
        return LIB_PATH
    return _compile(LIB_PATH, FLAGS_PATH, LIB_DIR / (f"obj_{VARIANT}" if VARIANT else "obj"), _extra_flags(), verbose)


def build_debug(force: bool = False, verbose: bool = False) -> Path:
    """Compile the test library lib/librt_s8_dbg.so (product sources + -DRT_DEBUG_HOOKS) if missing or stale."""
    fp = DEBUG_LIB_PATH.with_suffix(".flags")
    stale = (not DEBUG_LIB_PATH.exists() or not fp.exists() or fp.read_text() != flags_record(DEBUG_FLAGS)
             or any(d.stat().st_mtime > DEBUG_LIB_PATH.stat().st_mtime for d in _deps()))
    if not force and not stale:
        return DEBUG_LIB_PATH
    return _compile(DEBUG_LIB_PATH, fp, LIB_DIR / f"obj_{DEBUG_VARIANT}", DEBUG_FLAGS, verbose)


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    if not VARIANT:
        print(build_debug(force=True, verbose=True))
