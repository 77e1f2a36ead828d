"""world_size-2 run of the strip-sharding path on CPU (gloo backend), launched the way the driver
launches bench.py for N > 1."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("world,mode", [(2, "weak"), (3, "weak"), (2, "strong"), (3, "strong")])
def test_sharded_job_matches_single_process(world, mode):
    """weak: N frames, unit (f, d) -> rank (f + d) mod N (bench.py default); strong: ONE frame, strip d -> rank d mod N
    (bench.py --strong = the controller's split that BASELINE configs 4 and 5 name)."""
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29600 + world + (10 if mode == "strong" else 0)),
           str(ROOT / "tests" / "_gloo_worker.py")] + (["strong"] if mode == "strong" else [])
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=str(ROOT))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert f"GLOO_OK world={world}" in p.stdout and f"mode={mode}" in p.stdout
