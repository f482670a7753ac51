#!/bin/bash
# rocprofv3 evidence for one round (run on the GPU box through gpurun):  tools/profile_round.sh <out-dir-under-gpurun_out>
#   kernel-trace --stats of bench.py (C4 att 3 AT T_iter, + att 1, att 2, C3/SAGE), separate --pmc FETCH_SIZE and
#   --pmc WRITE_SIZE passes of the headline, SQ / TCC counters of the GEMM kernels on their own shapes (P/Q on the
#   fp32-operand kernel, projection and fuser on the plane-operand kernel) and of the att-2 edge pass / aux scorer.
# PMC passes never combine with sys/hip/hsa tracing (the pool refuses that); the profiled program is python3 itself.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-r05/prof}"
STAGE="${2:-all}"       # all | bench | gemm | att2 | train | sampler: the stages as separate gpurun calls
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1"
run() { # name, rocprof-args..., -- program...
  local name=$1; shift
  echo "[profile] $name"; date +%T
  rocprofv3 "$@" > "$OUT/$name.log" 2>&1 || { echo "[profile] $name FAILED"; tail -5 "$OUT/$name.log"; return 1; }
}
stage() { [ "$STAGE" = all ] || [ "$STAGE" = "$1" ]; }
if stage bench; then
run kt_c4_att3   --kernel-trace --stats --output-format csv -d "$OUT/kt_c4_att3"   -- $B &&
run fetch_c4_att3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_c4_att3" -- $B &&
run write_c4_att3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write_c4_att3" -- $B &&
run kt_c4_att1   --kernel-trace --stats --output-format csv -d "$OUT/kt_c4_att1"   -- $B --att 1 &&
run kt_c4_att2   --kernel-trace --stats --output-format csv -d "$OUT/kt_c4_att2"   -- $B --att 2 &&
run kt_c3_sage   --kernel-trace --stats --output-format csv -d "$OUT/kt_c3_sage"   -- $B --nodes 100000 --edges 2000000 --feat 128 --gnn_type SAGE &&
run kt_c4_fwd    --kernel-trace --stats --output-format csv -d "$OUT/kt_c4_fwd"    -- $B --fwd-only &&
run kt_c4_train  --kernel-trace --stats --output-format csv -d "$OUT/kt_c4_train"  -- python3 $ROOT/tools/train_bench.py --nodes 1000000 --edges 20000000
fi
if stage sampler; then
S="python3 $ROOT/tools/sampler_time.py"
run sampler_kt --kernel-trace --stats --output-format csv -d "$OUT/sampler_kt" -- $S &&
run sampler_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/sampler_fetch" -- $S &&
run sampler_write --pmc WRITE_SIZE --output-format csv -d "$OUT/sampler_write" -- $S
fi
if stage gemm; then
for what in ${GEMM_SHAPES:-pq pq_as proj fuser b2b logits}; do
  tag="gemm_$what"
  if [ "$what" = pq_as ]; then export DISGAT_GEMM_AS=1; what=pq; else unset DISGAT_GEMM_AS; fi
  run ${tag}_sq1 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d "$OUT/${tag}_sq1" -- python3 $ROOT/tools/gemm_one.py $what &&
  run ${tag}_sq2 --pmc SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/${tag}_sq2" -- python3 $ROOT/tools/gemm_one.py $what &&
  run ${tag}_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/${tag}_fetch" -- python3 $ROOT/tools/gemm_one.py $what &&
  run ${tag}_write --pmc WRITE_SIZE --output-format csv -d "$OUT/${tag}_write" -- python3 $ROOT/tools/gemm_one.py $what &&
  run ${tag}_kt --kernel-trace --stats --output-format csv -d "$OUT/${tag}_kt" -- python3 $ROOT/tools/gemm_one.py $what || break
done
unset DISGAT_GEMM_AS
fi
if stage att2; then
# att 2 (the reference's default --att): counters of the edge pass and the aux scorer on their own (tools/kbench.py)
K2="python3 $ROOT/tools/kbench.py --att 2 --what edge aux"
run att2_kt --kernel-trace --stats --output-format csv -d "$OUT/att2_kt" -- $K2 &&
run att2_sq1 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/att2_sq1" -- $K2 &&
run att2_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/att2_fetch" -- $K2 &&
run att2_write --pmc WRITE_SIZE --output-format csv -d "$OUT/att2_write" -- $K2 &&
run att2_sq2 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d "$OUT/att2_sq2" -- $K2
fi
if stage train; then
# the training step's backward kernels (seg_grad_sign, bwd_alpha, the weight-gradient GEMM): counters over one C4 training run
T="python3 $ROOT/tools/train_bench.py --nodes 1000000 --edges 20000000"
run train_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/train_fetch" -- $T &&
run train_write --pmc WRITE_SIZE --output-format csv -d "$OUT/train_write" -- $T &&
run train_sq1 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/train_sq1" -- $T
fi
echo "[profile] done"; date +%T
# keep what travels back small (gpurun merges at most 64 MiB): the per-dispatch traces are not needed (the stats files
# are), and of the per-dispatch counter rows only this library's kernels are
find "$OUT" -name "*kernel_trace.csv" -delete
for f in $(find "$OUT" -name "*counter_collection.csv"); do
  { head -1 "$f"; grep disgat "$f"; } > "$f.tmp" && mv "$f.tmp" "$f"
done
du -sh "$OUT"; du -a "$OUT" | sort -n | tail -5
