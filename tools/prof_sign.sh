#!/bin/bash
# per-kernel A/B of the sign-recording pair scorer / edge pass under rocprofv3 (run on the GPU box): the plain build ("new") against a
# build with SIGN_FLAGS (default: -DDISGAT_SIGN_NOSTORE=1, the record computed but not stored); profiles/r05/sign_record.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05/sign
for tag in new old; do
  if [ $tag = old ]; then export DISGAT_HIPCC_FLAGS="${SIGN_FLAGS:--DDISGAT_SIGN_NOSTORE=1}"; else unset DISGAT_HIPCC_FLAGS; fi
  python3 -c "import sys; sys.path.insert(0,'$R'); from edgedisentangle_ssl_amd import _lib; _lib.load()" || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05/sign/$tag -- python3 $R/tools/train_bench.py --nodes 1000000 --edges 20000000 --iters 2 > $R/gpurun_out/r05/sign/$tag.log 2>&1 || exit 1
  f=$(find $R/gpurun_out/r05/sign/$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag"; grep -E "aux_att3_kernel|edge_fwd_kernel" $f | cut -c1-200
done
unset DISGAT_HIPCC_FLAGS
python3 -c "import sys; sys.path.insert(0,'$R'); from edgedisentangle_ssl_amd import _lib; _lib.load()"
find $R/gpurun_out/r05/sign -name "*kernel_trace.csv" -delete
