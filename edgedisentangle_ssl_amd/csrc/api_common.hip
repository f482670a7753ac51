#include "disgat_api.h"

namespace disgat {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace disgat

extern "C" int disgat_abi_version(void) { return 10; }     // == _lib.ABI_VERSION: bumped with every argument-list change
// the -D... flags this library was compiled with (_lib.py: DISGAT_HIPCC_FLAGS; "" for a plain build): a diagnostic build
// (-DRS_DIAG, -DBB_DIAG, -DDISGAT_PL_DIAG: stamps, ablation switches) left in the tree is recognised and rebuilt by
// _lib.load() instead of being benchmarked silently
#ifndef DISGAT_BUILD_FLAGS
#define DISGAT_BUILD_FLAGS ""
#endif
extern "C" const char* disgat_build_flags(void) { return DISGAT_BUILD_FLAGS; }
extern "C" const char* disgat_last_error(void) { return disgat::err_buf(); }
