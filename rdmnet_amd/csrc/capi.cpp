// Library-wide C-ABI plumbing: version and thread-local error text (the library's only state: every mode of a kernel --
// index width of a neighbour table, GroupNorm form -- is an argument of the call that uses it).
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
extern "C" size_t rdm_abi_struct_size(int which) {
  switch (which) {
    case 0: return sizeof(rdm_engine_config);
    case 1: return sizeof(rdm_engine_result);
    case 2: return sizeof(rdm_tensor_view);
    case 3: return sizeof(rdm_kpconv_profile);
    case 4: return sizeof(rdm_data_dict);
    default: return 0;
  }
}
extern "C" const char* rdm_last_error(void) { return rdm::last_error(); }
