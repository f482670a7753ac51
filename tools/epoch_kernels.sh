#!/bin/bash
# rocprofv3 kernel-trace of a captured small-graph run: tools/epoch_kernels.sh <dataset> <epochs> <out-dir-under-gpurun_out>
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
NAME=${1:-cora_full}; EP=${2:-60}; OUT="$ROOT/gpurun_out/${3:-r04/epoch_kt_$NAME}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp PYTHONPATH="$ROOT${PYTHONPATH:+:$PYTHONPATH}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 -m edgedisentangle_ssl_amd.main --model=DISGAT --sparse --dataset $NAME \
  --fixture $ROOT/tests/golden/data_$NAME.npz --gnn_type AT --att 3 --nhead 8 --nhid 64 --steps 5 --downstream CLS --down_weight 1.0 --finetune \
  --pretrain SupEdge DisEdge DifHead --pre_weight 1 1 1 --pre_edge 1 1 1 --dropout 0.1 --seed 4 --quiet --epochs $EP --capture on > "$OUT/run.log" 2>&1
find "$OUT" -name "*kernel_trace.csv" -delete
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$EP" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); ep = int(sys.argv[2])
tot = sum(int(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over {calls} launches = {tot/1e6/ep:.2f} ms and {calls/ep:.0f} launches per epoch (set-up included)")
for r in rows[:28]:
    print(f"{int(r['TotalDurationNs'])/1e6/ep:8.3f} ms/epoch {int(r['Calls'])/ep:7.1f} calls/epoch {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
PY
