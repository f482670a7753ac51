#!/usr/bin/env python3
"""Fill DESIGN.md's @PLACEHOLDER@ fields from bench.py records (ranges over the runs given, the last one named):
   python tools/fill_design.py gpurun_out/r04/bench_1.json gpurun_out/r04/bench_3.json ..."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs = [json.loads(open(p).read().strip().splitlines()[-1]) for p in sys.argv[1:]]


def rng(vals, fmt):
    vals = [v for v in vals if v is not None]
    lo, hi = min(vals), max(vals)
    return fmt.format(lo) if fmt.format(lo) == fmt.format(hi) else fmt.format(lo) + "–" + fmt.format(hi)


sec = [r["secondary"] for r in runs]
sub = {
    "T_ITER": rng([r["ms_per_step"] for r in runs], "{:.0f}"),
    "EPS": rng([r["value"] / 1e7 for r in runs], "{:.2f}") + "e7",
    "AUX_GBS": rng([r["roofline"]["achieved"] for r in runs], "{:.0f}"),
    "FRAC": rng([r["roofline"]["frac"] for r in runs], "{:.2f}"),
    "AUX_MS": rng([r["roofline"]["avg_launch_ms"] for r in runs], "{:.1f}"),
    "EDGE_GBS": rng([r["roofline"]["all_kernels"]["edge_fwd_att3"]["GB/s"] for r in runs], "{:.0f}"),
    "TFWD": rng([s["T_fwd_ms"] for s in sec], "{:.1f}"),
    "TFWD_EPS": rng([s["T_fwd_edges_per_s"] / 1e8 for s in sec], "{:.2f}") + "e8",
    "TRAIN": rng([s["train_step_ms"] for s in sec], "{:.0f}"),
    "PEAK": rng([s["train_step_peak_GiB"] for s in sec], "{:.1f}"),
    "TRAIN_S": rng([s.get("train_step_with_sampling_ms") for s in sec[-1:]], "{:.0f}"),
    "SAMP": rng([s.get("sampler_ms_per_list") for s in sec], "{:.2f}"),
    "E_CH": rng([s["small_graph_epoch_ms"]["chameleon"] for s in sec], "{:.1f}"),
    "E_CO": rng([s["small_graph_epoch_ms"]["cora"] for s in sec], "{:.1f}"),
    "E_CF": rng([s["small_graph_epoch_ms"]["cora_full"] for s in sec], "{:.1f}"),
    "CPU": rng([r["cpu_baseline"]["value"] / 1e3 for r in runs], "{:.1f}") + "e3",
}
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
for k, v in sub.items():
    s = s.replace("@" + k + "@", v)
open(p, "w").write(s)
print(sub)
left = [w for w in s.split() if w.startswith("@") and w.endswith("@")]
print("unfilled:", left)
