#include "common.h"

namespace ovis {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace ovis

extern "C" int ovis_abi_version(void) { return 1; }
extern "C" const char* ovis_last_error(void) { return ovis::err_buf(); }
