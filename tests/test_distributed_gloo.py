"""world_size-2 run of the strip-sharding path on CPU (gloo backend), launched the way the driver
launches bench.py for N > 1."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_job_matches_single_process(world):
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29600 + world), str(ROOT / "tests" / "_gloo_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=str(ROOT))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert f"GLOO_OK world={world}" in p.stdout
