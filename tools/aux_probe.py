#!/usr/bin/env python3
"""T_iter aux-scorer / edge-pass rates at C4 from bench.py's own HIP events (for same-box A/B builds)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary", "--steps", "3"],
                     capture_output=True, text=True).stdout
d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
k = d["roofline"]["all_kernels"]
print(round(d["ms_per_step"], 1), "ms  aux", k["aux_score_att3"]["GB/s"], "GB/s", round(k["aux_score_att3"]["ms_total"] / 3, 1),
      "ms  edge", k["edge_fwd_att3"]["GB/s"], "GB/s")
