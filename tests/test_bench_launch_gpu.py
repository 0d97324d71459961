"""`python3 bench.py --gpus N` as the round-end driver calls it (no launcher in front): bench.py starts its own ranks.
Two ranks share the one GPU of the box over the stand-in wire (tests/c_abi/standin_rccl.cpp); every line of the C++
driver and every kernel is the product's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_runs_from_a_plain_python_command(tmp_path):
    from tests.test_halo_gpu import _standin_rccl
    lib = _standin_rccl(tmp_path)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["SF_RCCL_LIB"] = lib
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-gpu", "--particles", "20000",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900,
                       env=env)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    line = json.loads(lines[0])
    assert r.returncode == 0, (line, r.stderr[-2000:])
    assert line["n_gpus"] == 2 and line["value"] and line["value"] > 0
    assert line["parity"]["ok"] is True
