#!/bin/bash
# same-box A/B of the plane-operand GEMM chain (DISGAT_PLANES=1/0) on T_fwd: tools/ab_planes.sh [extra bench flags]
for p in 1 0 1 0; do
  DISGAT_PLANES=$p timeout -k 10 200 python bench.py --fwd-only --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('planes=$p', round(d['ms_per_step'],2), {k:(v['ms_total'],v['launches']) for k,v in d['roofline']['all_kernels'].items()})"
done
