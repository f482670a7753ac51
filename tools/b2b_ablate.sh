#!/bin/bash
# timing ablations + in-kernel phase stamps of proj_fuse_kernel (csrc/gemm_b2b.hip) at the C4 shape: -DBB_DIAG=<mask> builds of
# the library (1 no ring refills, 2 no GEMM 1 MFMAs, 4 no ELU / split arithmetic, 8 no GEMM 2 MFMAs, 16 no Z reloads, 32 stamps).
# The diagnostic build is replaced by a plain one when the script ends, whatever way it ends.
cd "$(dirname "$0")/.."
trap 'DISGAT_HIPCC_FLAGS= python -c "from edgedisentangle_ssl_amd import _lib; _lib.build(force=True)"' EXIT
for d in ${BB_MODES:-32 48 1 4 2 8 10 16}; do
  DISGAT_HIPCC_FLAGS="-DBB_DIAG=$d" python -c "from edgedisentangle_ssl_amd import _lib; _lib.build(force=True)"
  echo -n "BB_DIAG=$d  "
  if [ $((d & 32)) -ne 0 ]; then timeout -k 10 100 python tools/b2b_stamps.py 2>&1 | grep -v amdgpu.ids
  else timeout -k 10 100 python tools/b2b_bench.py --rounds 2 --reps 3 2>&1 | grep "b2b" | grep -v "max |" | sed 's/; chain.*//'; fi
done
