#!/bin/bash
# same-box A/B of the K <= 256 GEMM kernels: A in LDS (DISGAT_GEMM_AS=1) vs A register-stationary (gemm_rs.hip)
cd "$(dirname "$0")/.."
for sh in "1000000 256 2048" "1000000 256 256 8" "1000000 64 512"; do
  for rep in 1 2; do for as in 1 0; do echo -n "AS=$as "; DISGAT_GEMM_AS=$as timeout -k 10 100 python tools/gemm_time.py $sh 2>&1 | grep -v amdgpu.ids; done; done
done
