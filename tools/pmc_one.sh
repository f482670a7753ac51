#!/bin/bash
# SQ / LDS counters of ONE kernel family on tools/gemm_one.py: tools/pmc_one.sh <gemm_one args...>   (out: gpurun_out/r04/pmc_<tag>)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG=${PMC_TAG:-one}
OUT="$ROOT/gpurun_out/r04/pmc_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="python3 $ROOT/tools/gemm_one.py $*"
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d "$OUT/sq1" -- $P > "$OUT/sq1.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/sq2" -- $P > "$OUT/sq2.log" 2>&1
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d "$OUT/mem" -- $P > "$OUT/mem.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("sq1", "sq2", "mem"):
    fs = glob.glob(f"{out}/{sub}/*/*_counter_collection.csv")
    if not fs:
        print(sub, "no counters", open(f"{out}/{sub}.log").read()[-400:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][-60:]
        if "gemm" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(sub, k, {c: f"{sum(v)/len(v):.4g}" for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
find "$OUT" -name "*.csv" -size +2M -delete
