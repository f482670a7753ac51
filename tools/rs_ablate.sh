#!/bin/bash
# ablations of gemm_f16x3_rs_kernel on the P/Q shape (needs a -DRS_DIAG=1 build): tools/rs_ablate.sh
cd "$(dirname "$0")/.."
# the diagnostic build must not stay installed (a later bench would silently run it): rebuild plain on exit
trap 'python -c "from edgedisentangle_ssl_amd import _lib; _lib.build(force=True)"' EXIT
DISGAT_HIPCC_FLAGS="-DRS_DIAG=1" python -c "from edgedisentangle_ssl_amd import _lib; _lib.build(force=True)"
for d in 0 1 2 3 4 5 7; do echo -n "DISGAT_RS_DEBUG=$d  "; DISGAT_RS_DEBUG=$d timeout -k 10 100 python tools/gemm_time.py 1000000 256 2048 2>&1 | grep -v amdgpu.ids; done
