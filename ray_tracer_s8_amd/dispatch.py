"""Strip -> worker assignment and frame assembly (host logic, no GPU needed).

Mirrors the controller: one request per `division_no in 0..divisions`
(ray-tracer-controller/src/main.rs:47-75) and assembly by sorting on division_no and
concatenating the strips (main.rs:109-115).  Strips are independent, so multi-GPU is pure
sharding: no collective is on the data path.
"""
from __future__ import annotations

from typing import Iterable, Sequence

import numpy as np


def strip_rows(height: int, divisions: int) -> int:
    """Rows per strip = H / divisions, integer division (slave main.rs:55-56, 66)."""
    if divisions <= 0:
        raise ValueError("divisions must be positive")
    return height // divisions


def strips_for_worker(divisions: int, worker: int, n_workers: int) -> list[int]:
    """division_no values rendered by `worker`: k with k % n_workers == worker."""
    if not (0 <= worker < n_workers):
        raise ValueError("worker out of range")
    return list(range(worker, divisions, n_workers))


def job_shards(n_frames: int, divisions: int, rank: int, world: int) -> list[tuple[int, int]]:
    """(frame, division_no) units of a multi-frame job owned by `rank`.

    Unit (f, d) goes to rank (f + d) % world, so that over `world` frames every rank gets
    every strip position once: sky-heavy top strips and ground strips are balanced."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return [(f, d) for f in range(n_frames) for d in range(divisions) if (f + d) % world == rank]


def assemble(slices: Iterable[tuple[int, np.ndarray]], width: int, height: int, divisions: int) -> np.ndarray:
    """Sort by division_no, concatenate, view as H x W x 3 (controller main.rs:109-119).

    Raises like the controller would fail: missing strips ('Job not finished yet k/n') or a
    byte count that does not fill width*height*3 (`ImageBuffer::from_vec(...).unwrap()`)."""
    got = sorted(slices, key=lambda s: s[0])
    nos = [s[0] for s in got]
    if nos != list(range(divisions)):
        have = len(set(nos) & set(range(divisions)))
        raise ValueError(f"Job not finished yet {have}/{divisions}")
    data = np.concatenate([np.asarray(s[1], dtype=np.uint8).reshape(-1) for s in got])
    if data.size != width * height * 3:
        raise ValueError("strips do not tile the frame (height % divisions != 0)")
    return data.reshape(height, width, 3)
