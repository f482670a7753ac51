#!/bin/bash
# same-box A/B of one environment switch on bench.py's T_iter:  tools/ab_env.sh VAR "<bench args>" [rounds]
# (interleaved rounds of VAR=0 / VAR=1; prints ms per step and the per-label kernel ms)
VAR=$1; ARGS=$2; R=${3:-2}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
LOG=gpurun_out/r05/ab_$VAR.log
B="python bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2"
for r in $(seq $R); do
  for f in 0 1; do
    echo -n "$VAR=$f $ARGS: " >> $LOG
    env $VAR=$f timeout -k 10 300 $B $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],2), {k:round(v['ms_total']/d['steps'],2) for k,v in d['roofline']['all_kernels'].items()})" >> $LOG 2>&1 || exit 1
  done
done
cat $LOG
