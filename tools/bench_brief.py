#!/usr/bin/env python3
"""bench.py's T_iter, briefly: ms per step and per-label kernel ms per step (for the A/B scripts).  Arguments go to bench.py."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-secondary", "--steps", "6", "--warmup", "2"]
                     + sys.argv[1:], capture_output=True, text=True)
if out.returncode != 0:
    sys.stderr.write(out.stderr[-2000:])
    sys.exit(out.returncode)
d = json.loads(out.stdout.strip().splitlines()[-1])
print("T_iter", round(d["ms_per_step"], 2), {k: round(v["ms_total"] / d["steps"], 2) for k, v in d["roofline"]["all_kernels"].items()})
