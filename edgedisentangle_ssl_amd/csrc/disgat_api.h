// Host-side helpers shared by the extern "C" launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/disgat_hip.h"

namespace disgat {
char* err_buf();  // thread-local, 512 bytes
inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
  return 0;
}
inline int ilog2_exact(int v) {  // -1 if not a power of two
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
}  // namespace disgat

#define DISGAT_REQUIRE(cond, ...) \
  do {                            \
    if (!(cond)) return disgat::fail(-1, __VA_ARGS__); \
  } while (0)
