"""`python bench.py --gpus N` started bare (the way the driver invokes it) must start its own N ranks before touching
the GPU, relay rank 0's JSON line and report n_gpus from the process group (Missing #1).  Runs on the CPU
with the launcher's probe mode: the ranks rendezvous over gloo, do one collective and exit."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(DISGAT_BENCH_LAUNCH_PROBE="1", **(extra_env or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def test_bare_launch_starts_n_ranks():
    r = _run(["--gpus", "3", "--scaling", "strong"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["rank_sum"] == 3.0 and out["scaling"] == "strong"


def test_under_torchrun_env_this_process_is_a_rank():
    """With WORLD_SIZE in the environment (torch.distributed.run) nothing is spawned."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = _run(["--gpus", "1"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_failing_rank_fails_the_launch():
    r = _run(["--gpus", "2"], {"DISGAT_BENCH_PROBE_FAIL_RANK": "1"})
    assert r.returncode != 0


def test_rank_dying_before_the_rendezvous_ends_the_launch_promptly():
    """A rank that exits before it joins the process group leaves the others waiting in the rendezvous (RCCL: in the
    first collective, for the ~10 min process-group timeout); the launcher watches every rank and ends them."""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2"], {"DISGAT_BENCH_PROBE_DIE_EARLY_RANK": "1"})
    assert r.returncode != 0
    assert time.time() - t0 < 120
