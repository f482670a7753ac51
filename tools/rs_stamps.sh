#!/bin/bash
cd "$(dirname "$0")/.."
# the diagnostic build must not stay installed (a later bench would silently run it): rebuild plain on exit
trap 'python -c "from edgedisentangle_ssl_amd import _lib; _lib.build(force=True)"' EXIT
DISGAT_HIPCC_FLAGS="-DRS_DIAG=1" python -c "from edgedisentangle_ssl_amd import _lib; _lib.build(force=True)"
for d in ${RS_MODES:-32 33 34}; do echo "DISGAT_RS_DEBUG=$d"; DISGAT_RS_DEBUG=$d timeout -k 10 100 python tools/rs_stamps.py 2>&1 | grep -v amdgpu.ids; done
for d in ${RS_TIMES:-0 8}; do echo -n "DISGAT_RS_DEBUG=$d  "; DISGAT_RS_DEBUG=$d timeout -k 10 100 python tools/gemm_time.py 1000000 256 2048 2>&1 | grep -v amdgpu.ids; done
