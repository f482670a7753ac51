"""bench.py started the way the driver starts it for N > 1 - `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...` - on a small graph with both ranks sharing the test box's one GPU over gloo (DISGAT_BENCH_REHEARSAL=1): the
row-sharded workload construction, the per-layer exchanges, the global loss reductions, the max-over-ranks timing and the
one JSON line, weak and strong.  A functional check of the N-rank path; its numbers are not scaling numbers."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_under_torchrun(scaling):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(DISGAT_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--nodes", "20000", "--edges", "400000", "--feat", "64", "--no-secondary", "--scaling", scaling]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # rank 0 alone prints the result
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["steps"] == 2
    assert out["config"]["ranks_seen"] == 2 and out["config"]["exchanges_per_step"]
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert out["cpu_baseline"] is None or "value" in out["cpu_baseline"]
    assert out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1
