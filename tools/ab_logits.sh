mkdir -p gpurun_out/r05
B="python bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2"
for r in 1 2; do
for f in 0 1; do
  for att in 3 1; do
    echo -n "LOGITS=$f att=$att: " >> gpurun_out/r05/ab_logits.log
    DISGAT_LOGITS=$f timeout -k 10 200 $B --att $att 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],2), {k:round(v['ms_total']/d['steps'],2) for k,v in d['roofline']['all_kernels'].items()})" >> gpurun_out/r05/ab_logits.log 2>&1 || exit 1
  done
done
done
cat gpurun_out/r05/ab_logits.log
