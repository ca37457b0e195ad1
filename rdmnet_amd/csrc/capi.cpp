// Library-wide C-ABI plumbing: version and thread-local error text.
#include <cstdarg>
#include <cstdio>

#include "../../include/rdmnet_hip.h"
#include "common.h"

namespace rdm {
namespace {
thread_local char g_error[512] = "";
}
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_error; }
}  // namespace rdm

extern "C" int rdm_abi_version(void) { return RDM_ABI_VERSION; }
extern "C" const char* rdm_last_error(void) { return rdm::last_error(); }
