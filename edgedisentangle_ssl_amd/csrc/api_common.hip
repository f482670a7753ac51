#include "disgat_api.h"

namespace disgat {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace disgat

extern "C" int disgat_abi_version(void) { return 9; }     // == _lib.ABI_VERSION: bumped with every argument-list change
extern "C" const char* disgat_last_error(void) { return disgat::err_buf(); }
