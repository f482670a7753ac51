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


@pytest.mark.parametrize("scaling,gnn", [("weak", "AT"), ("strong", "AT"), ("weak", "GCN")])       # (weak, GCN) = configs[4]'s form
def test_bench_two_ranks_under_torchrun(scaling, gnn):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(DISGAT_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--nodes", "20000", "--edges", "400000", "--feat", "64", "--no-secondary", "--scaling", scaling, "--gnn_type", gnn]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # rank 0 alone prints the result
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["steps"] == 2 and out["config"]["gnn_type"] == gnn
    assert out["config"]["ranks_seen"] == 2 and out["config"]["exchanges_per_step"]
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert out["cpu_baseline"] is None or "value" in out["cpu_baseline"]
    assert out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1


def test_bench_one_gpu_json_contract():
    """`python bench.py` (N = 1) on a small graph: ONE JSON line carrying every field of the driver's contract, the
    roofline object measured with HIP events on the launch stream and the CPU baseline (the oracle, timed on a sample)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--nodes", "20000", "--edges", "400000",
           "--feat", "64", "--no-secondary", "--cpu-nodes", "1024"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True
    assert out["vs_baseline"] is None and "workload" in out["config"] and "model" not in out["config"]
    roof = out["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    cpu = out["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]
    assert abs(out["value"] - out["config"]["nnz_total"] / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-3
