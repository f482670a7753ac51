#!/bin/bash
# Same-box A/B of compile-time variants: tools/ab_flags.sh "<cmd>" "<flags A>" "<flags B>" ...   (runs on the GPU box;
# every variant is a full rebuild with DISGAT_HIPCC_FLAGS, the default build is restored at the end)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
cmd="$1"; shift
for flags in "$@"; do
  DISGAT_HIPCC_FLAGS="$flags" python3 -c "from edgedisentangle_ssl_amd import _lib; _lib.build(force=True)"
  echo "== flags: [$flags]"
  eval "$cmd"
done
python3 -c "from edgedisentangle_ssl_amd import _lib; _lib.build(force=True)"
