#!/bin/bash
# Same-box A/B of a one-line source variant: tools/ab_pad.sh  (runs on the GPU box; edits only the scratch copy)
set -e
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
SRC=edgedisentangle_ssl_amd/csrc/gemm_split.hip
for pad in 8 16 48; do
  sed -i "s/constexpr int AS_BM = 128, AS_PAD = [0-9]*;/constexpr int AS_BM = 128, AS_PAD = $pad;/" $SRC
  python3 -c "from edgedisentangle_ssl_amd import _lib; _lib.build(force=True)"
  echo "== AS_PAD=$pad"
  python3 tools/gemm_bench.py 2>/dev/null | cut -c 1-46,130-200
done
