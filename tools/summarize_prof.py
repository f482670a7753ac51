#!/usr/bin/env python3
"""Condense a gpurun_out/prof/{kt,fetch,write} rocprofv3 run into profiles/<round>/ (tracked)."""
import csv
import glob
import shutil
import sys


def main(src, dst, tag, note):
    out = open(f"{dst}/{tag}_summary.md", "w")
    P = lambda *a: print(*a, file=out)
    P(f"# rocprofv3 summary: {tag}\n")
    P(note + "\n")
    f = glob.glob(f"{src}/kt/*/*_kernel_stats.csv")[0]
    shutil.copy(f, f"{dst}/{tag}_kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    P(f"Total kernel time {tot / 1e6:.1f} ms\n")
    P("| kernel | calls | total ms | avg ms | % |")
    P("|---|---|---|---|---|")
    for r in rows[:14]:
        P(f"| `{r['Name'][:90]}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e6:.3f} | {float(r['Percentage']):.2f} |")
    P("\n## HBM traffic (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes)\n")
    P("rocprofv3 reports KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies the 128-B "
      "requests of 16-B-per-lane loads at 64 B, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact for "
      "16-B-per-lane stores.\n")
    vals = {}
    for nm in ("fetch", "write"):
        fs = glob.glob(f"{src}/{nm}/*/*_counter_collection.csv")
        if not fs:
            continue
        for r in csv.DictReader(open(fs[0])):
            k = r["Kernel_Name"]
            if "disgat" not in k:
                continue
            vals.setdefault(k.split("(")[0], {}).setdefault(nm, []).append(float(r["Counter_Value"]))
    P("| kernel | launches | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | corrected HBM bytes/launch (2F+W) |")
    P("|---|---|---|---|---|")
    for k, v in vals.items():
        fl, wl = v.get("fetch", [0]), v.get("write", [0])
        fa, wa = sum(fl) / len(fl), sum(wl) / len(wl)
        P(f"| `{k}` | {len(fl)} | {fa:.4g} | {wa:.4g} | {(2 * fa + wa) * 1024 / 1e9:.1f} GB |")
    out.close()
    print(open(f"{dst}/{tag}_summary.md").read())


if __name__ == "__main__":
    main(*sys.argv[1:5])
