#!/bin/bash
# same-box A/B of a compile-time switch:  tools/ab_flags.sh "<-D flags>" "<program + args>" [rounds] [grep pattern]
# _lib rebuilds the library whenever DISGAT_HIPCC_FLAGS differs from what the installed .so was compiled with; the plain
# build is restored at exit.
FLAGS=$1; PROG=$2; R=${3:-2}; PAT=${4:-.}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
trap 'python -c "from edgedisentangle_ssl_amd import _lib; _lib.load()" > /dev/null 2>&1' EXIT
for r in $(seq $R); do
  echo "== plain build"; timeout -k 10 500 python $PROG 2>&1 | grep -v amdgpu.ids | grep -E "$PAT" | tail -4 || exit 1
  echo "== $FLAGS"; DISGAT_HIPCC_FLAGS="$FLAGS" timeout -k 10 500 python $PROG 2>&1 | grep -v amdgpu.ids | grep -E "$PAT" | tail -4 || exit 1
done
