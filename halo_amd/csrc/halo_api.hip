// Version / error plumbing of libhalo_hip.so.
#include "halo_common.hpp"

namespace halo {
char *err_buf()
{
    static thread_local char buf[ERR_LEN] = {0};
    return buf;
}
}  // namespace halo

extern "C" int halo_version(void) { return HALO_ABI_VERSION; }
extern "C" const char *halo_last_error(void) { return halo::err_buf(); }
