#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output tree (gpurun_out/<round>/prof) into tracked summaries under profiles/<round>/.

  python tools/summarize_round.py gpurun_out/r02/prof profiles/r02
"""
import csv
import glob
import os
import shutil
import sys

NOTES = {
    "kt_c4_att3": "bench.py default: C4 graph (1M nodes / 20M edges, F=256, H=8), att 3, AT, T_iter, 1 warm-up + 3 steps",
    "kt_c4_att1": "bench.py --att 1 (same graph)",
    "kt_c4_att2": "bench.py --att 2 (same graph)",
    "kt_c3_sage": "bench.py --nodes 100000 --edges 2000000 --feat 128 --gnn_type SAGE (BASELINE configs[2])",
    "kt_c4_fwd": "bench.py --fwd-only (T_fwd = one get_em per step)",
    "kt_c4_train": "tools/train_bench.py --nodes 1000000 --edges 20000000: 4 forward-only T_iter (eval) + 4 full training "
                   "iterations (SupEdge + DisEdge + DifHead: forward, backward, multi-tensor Adam; attention dropout 0.1)",
}


def newest(pattern):
    """Files of the most recent run only (a re-run merges next to the older process ids' files)."""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:] if fs else []


def one(path):
    fs = newest(os.path.join(path, "*", "*_kernel_stats.csv"))
    return list(csv.DictReader(open(fs[0]))) if fs else []


def counters(path):
    """kernel -> counter -> list of per-dispatch values"""
    out = {}
    for f in newest(os.path.join(path, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = kname_of(r["Kernel_Name"])
            out.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return out


def kname_of(full):
    """Kernel name without its argument list (kernels in anonymous namespaces carry a parenthesis inside the name)."""
    return full.replace("(anonymous namespace)::", "").split("(")[0]


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    for tag, note in NOTES.items():
        rows = one(os.path.join(src, tag))
        if not rows:
            continue
        for f in newest(os.path.join(src, tag, "*", "*_kernel_stats.csv")):
            shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))
        with open(os.path.join(dst, f"{tag}_summary.md"), "w") as out:
            P = lambda *a: print(*a, file=out)      # noqa: E731
            P(f"# rocprofv3 --kernel-trace --stats: {tag}\n\n{note}\n")
            tot = sum(int(r["TotalDurationNs"]) for r in rows)
            P(f"Total kernel time {tot / 1e6:.1f} ms over all steps of the run (bench.py: 1 warm-up + 3 timed)\n")
            for line in open(os.path.join(src, tag + ".log"), errors="ignore"):
                if "ms/iter" in line:
                    P("    " + line.rstrip())
            P("")
            P("| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|")
            for r in rows[:16]:
                P(f"| `{r['Name'][:100]}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.2f} | "
                  f"{float(r['AverageNs']) / 1e6:.3f} | {float(r['Percentage']):.2f} |")
            if tag == "kt_c4_att3":
                P("\n## HBM traffic (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes of the same command)\n")
                P("rocprofv3 reports KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies the 128-B "
                  "requests of 16-B-per-lane loads at 64 B, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact for "
                  "16-B-per-lane stores.\n")
                fe, wr = counters(os.path.join(src, "fetch_c4_att3")), counters(os.path.join(src, "write_c4_att3"))
                P("| kernel | launches | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | corrected HBM bytes/launch (2F+W) |\n|---|---|---|---|---|")
                for k in fe:
                    if "disgat" not in k:
                        continue
                    f = fe[k].get("FETCH_SIZE", [0])
                    w = wr.get(k, {}).get("WRITE_SIZE", [0])
                    fa, wa = sum(f) / len(f), sum(w) / len(w)
                    P(f"| `{k}` | {len(f)} | {fa:.4g} | {wa:.4g} | {(2 * fa + wa) * 1024 / 1e9:.2f} GB |")
    # GEMM PMC
    with open(os.path.join(dst, "gemm_f16x3_pmc.md"), "w") as out:
        P = lambda *a: print(*a, file=out)      # noqa: E731
        P("# f16x3 GEMM kernels: PMC counters (tools/gemm_one.py pq | proj | fuser | b2b | logits, M = 1,000,000, 4 launches, per-launch averages)\n")
        P("Separate rocprofv3 passes per counter set (SQ set 1, SQ set 2, FETCH_SIZE, WRITE_SIZE) plus a kernel-trace pass "
          "for the duration.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES "
          "counts cycles (MI355X_MICROARCH.md).\n")
        for shape, what, flop, kname in (
                ("pq", "P / Q operands x[1e6,256] x [256,2048], fp32 operand: gemm_f16x3_rs_kernel<0,8> (A register-stationary, weights through an LDS-DMA ring; csrc/gemm_rs.hip)", 2.0e6 * 256 * 2048, "gemm_f16x3"),
                ("pq_as", "the same product on round 3's kernel, gemm_f16x3_as_kernel<0> (A tile in LDS; DISGAT_GEMM_AS=1)", 2.0e6 * 256 * 2048, "gemm_f16x3"),
                ("proj", "per-head projection, Z planes [1e6,8,256] x [8,256,256] -> ELU -> head planes: gemm_planes_kernel<1,false,true>", 2.0e6 * 8 * 256 * 256, "gemm_planes"),
                ("fuser", "FuseLayer, head planes [1e6,2048] x [2048,256] + bias, leaky ReLU -> fp32: gemm_planes_kernel<2,true,false>", 2.0e6 * 2048 * 256, "gemm_planes"),
                ("b2b", "projection + fuser back to back, Z planes [1e6,8,256] -> fp32 [1e6,256] in one launch: proj_fuse_kernel<8,16,false> (csrc/gemm_b2b.hip)", 2.0e6 * 8 * 256 * 256 * 2, "proj_fuse"),
                ("logits", "DifHead classifier, head planes [1e6,8,256] x [256,256] + shared, leaky ReLU, x [256,8] in the epilogue -> [8e6,8] logits: gemm_planes_kernel<2,false,false,true>", 2.0e6 * 8 * 256 * (256 + 16), "gemm_planes")):
            P(f"\n## {what}\n")
            kt = [r for r in one(os.path.join(src, f"gemm_{shape}_kt")) if kname in r["Name"]]
            dur = None
            for r in kt:
                dur = float(r["AverageNs"]) / 1e6
                P(f"`{r['Name'][:80]}`: {r['Calls']} calls, avg **{dur:.3f} ms** "
                  f"({flop / (dur * 1e-3) / 1e12:.0f} TFLOP/s fp32-equivalent, x3 fp16 MFMA products inside)\n")
            P("| counter | per launch |\n|---|---|")
            vals = {}
            for sub in ("sq1", "sq2", "fetch", "write"):
                for k, cs in counters(os.path.join(src, f"gemm_{shape}_{sub}")).items():
                    if kname not in k:
                        continue
                    for c, v in cs.items():
                        vals[c] = sum(v) / len(v)
            for c, v in vals.items():
                P(f"| {c} | {v:.4g} |")
            if "SQ_INSTS_VALU" in vals and "SQ_INSTS_MFMA" in vals:
                P(f"\nVALU : MFMA instructions = {vals['SQ_INSTS_VALU'] / vals['SQ_INSTS_MFMA']:.2f}")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and dur:
                # 1024 SIMDs (256 CUs x 4); the counter sums busy cycles over all SIMDs
                for ghz in (2.4, 2.0):
                    P(f"\nMFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x {ghz} GHz x {dur:.3f} ms) = "
                      f"**{vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * ghz * 1e9 * dur * 1e-3) * 100:.0f} %**")
            if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
                P(f"\nHBM bytes per launch (2 x FETCH + WRITE, KiB -> bytes): read {2 * vals['FETCH_SIZE'] * 1024 / 1e9:.2f} GB, "
                  f"write {vals['WRITE_SIZE'] * 1024 / 1e9:.2f} GB")
            if all(k in vals for k in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")):
                w = vals["SQ_WAVE_CYCLES"]
                P(f"\nWave time: {vals['SQ_ACTIVE_INST_ANY'] / w * 100:.0f} % issuing, {vals['SQ_WAIT_INST_ANY'] / w * 100:.0f} % "
                  f"issue-stalled, {vals['SQ_WAIT_ANY'] / w * 100:.0f} % parked at s_waitcnt / barrier")
    if os.path.isdir(os.path.join(src, "sampler_kt")):
        with open(os.path.join(dst, "sampler_pmc.md"), "w") as out:
            P = lambda *a: print(*a, file=out)      # noqa: E731
            P("# SSL pair sampler (csrc/pair_sample.hip) at C4: one SupEdge list of 66.5M pairs over 1M x 1M entries (tools/sampler_time.py)\n")
            P("Kernel-trace pass for the durations, separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes for the traffic "
              "(2 x FETCH + WRITE, KiB -> bytes).  Algorithmic bytes of a list: 20 B written per pair (two int64 planes + a float "
              "label) = 1.33 GB, 80 MB of positive columns + 32 MB of work items read per pass.\n")
            P("| kernel | calls | avg ms | HBM GB / launch |\n|---|---|---|---|")
            fe, wr = counters(os.path.join(src, "sampler_fetch")), counters(os.path.join(src, "sampler_write"))
            for r in one(os.path.join(src, "sampler_kt")):
                if "pair_sample" not in r["Name"]:
                    continue
                k = kname_of(r["Name"])
                f = fe.get(k, {}).get("FETCH_SIZE", [0])
                w = wr.get(k, {}).get("WRITE_SIZE", [0])
                P(f"| `{k[:70]}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.3f} | {(2 * sum(f) / len(f) + sum(w) / len(w)) * 1024 / 1e9:.2f} |")
            for line in open(os.path.join(src, "sampler_kt.log"), errors="ignore"):
                if "sample" in line and "ms" in line:
                    P("    " + line.rstrip())
    with open(os.path.join(dst, "att2_pmc.md"), "w") as out:
        P = lambda *a: print(*a, file=out)      # noqa: E731
        P("# att 2 (the reference's default --att): edge pass and aux scorer, PMC counters (tools/kbench.py --att 2 --what edge aux)\n")
        P("C4 graph (1M nodes / 20M edges, F = 256, H = 8); per-launch averages; separate rocprofv3 passes per counter set.\n")
        kt = {kname_of(r["Name"]): r for r in one(os.path.join(src, "att2_kt")) if "disgat" in r["Name"]}
        vals = {}
        for sub in ("sq1", "sq2", "fetch", "write"):
            for k, cs in counters(os.path.join(src, f"att2_{sub}")).items():
                if "disgat" in k:
                    for c, v in cs.items():
                        vals.setdefault(k, {})[c] = sum(v) / len(v)
        for k, cs in vals.items():
            name = next((n for n in kt if n.startswith(k[:60])), None)
            dur = float(kt[name]["AverageNs"]) / 1e6 if name else None
            P(f"\n## `{k[:110]}`" + (f" - avg {dur:.3f} ms" if dur else "") + "\n")
            P("| counter | per launch |\n|---|---|")
            for c, v in cs.items():
                P(f"| {c} | {v:.4g} |")
            if all(c in cs for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")):
                w = cs["SQ_WAVE_CYCLES"]
                P(f"\nWave time: {cs['SQ_ACTIVE_INST_ANY'] / w * 100:.0f} % issuing, {cs['SQ_WAIT_INST_ANY'] / w * 100:.0f} % issue-stalled, "
                  f"{cs['SQ_WAIT_ANY'] / w * 100:.0f} % parked at s_waitcnt")
            if "SQ_ACTIVE_INST_VALU" in cs and "SQ_BUSY_CYCLES" in cs:
                P(f"\nVALU-active quad-cycles / SQ busy cycles = {cs['SQ_ACTIVE_INST_VALU'] / cs['SQ_BUSY_CYCLES']:.2f}")
            if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs and dur:
                by = (2 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024
                P(f"\nHBM bytes per launch (2 x FETCH + WRITE): {by / 1e9:.2f} GB -> {by / (dur * 1e-3) / 1e12:.2f} TB/s moved")
    if os.path.isdir(os.path.join(src, "train_fetch")):
        with open(os.path.join(dst, "train_pmc.md"), "w") as out:
            P = lambda *a: print(*a, file=out)      # noqa: E731
            P("# Training step (tools/train_bench.py, C4: 1M nodes / 20M edges, F = 256, H = 8, att 3, dropout 0.1): PMC counters of the backward kernels\n")
            P("Per-launch averages, one row per (kernel, grid size): the 20M-edge lists and the 66.5M-pair lists share kernels.  Durations "
              "are the dispatch timestamps of the FETCH_SIZE pass (kernels serialised by the profiler; `kt_c4_train_summary.md` has the "
              "untraced averages per kernel name).  Separate rocprofv3 passes: FETCH_SIZE, "
              "WRITE_SIZE, SQ set.  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes; gfx950 correction as in kt_c4_att3_summary.md).\n")
            vals, calls, durs = {}, {}, {}
            for sub in ("fetch", "write", "sq1"):
                for f in newest(os.path.join(src, f"train_{sub}", "*", "*_counter_collection.csv")):
                    for r in csv.DictReader(open(f)):
                        if "disgat" not in r["Kernel_Name"]:
                            continue
                        k = (kname_of(r["Kernel_Name"]), int(r["Grid_Size"]))
                        vals.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                        if sub == "fetch":
                            durs.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
            P("| kernel | grid (threads) | launches | avg ms (FETCH pass) | HBM GB / launch | TB/s moved | issuing % | issue-stalled % | parked % |\n|---|---|---|---|---|---|---|---|---|")
            rows = []
            for k, cs in vals.items():
                cs = {c: sum(v) / len(v) for c, v in cs.items()}
                d = durs.get(k)
                dur = sum(d) / len(d) if d else None
                by = (2 * cs.get("FETCH_SIZE", 0) + cs.get("WRITE_SIZE", 0)) * 1024
                w = cs.get("SQ_WAVE_CYCLES", 0)
                pct = [f"{cs.get(c, 0) / w * 100:.0f}" if w else "-" for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")]
                rows.append(((dur or 0) * len(d or []), f"| `{k[0][:80]}` | {k[1]} | {len(d or [])} | " + (f"{dur:.3f}" if dur else "-") + f" | {by / 1e9:.2f} | "
                             + (f"{by / (dur * 1e-3) / 1e12:.2f}" if dur else "-") + " | " + " | ".join(pct) + " |"))
            for _t, line in sorted(rows, reverse=True)[:24]:
                P(line)
        print(open(os.path.join(dst, "train_pmc.md")).read())
    print(open(os.path.join(dst, "att2_pmc.md")).read())
    print(open(os.path.join(dst, "gemm_f16x3_pmc.md")).read())
    print(open(os.path.join(dst, "kt_c4_att3_summary.md")).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
