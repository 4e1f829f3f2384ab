// Library-level entry points of libnerf_amd.so (see include/nerf_amd.h).
#include "common.h"

namespace nerf {
char *error_buffer() {
    static thread_local char buf[256] = {0};
    return buf;
}
}  // namespace nerf

NERF_API int nerf_amd_abi_version(void) { return NERF_AMD_ABI_VERSION; }
NERF_API const char *nerf_amd_last_error(void) { return nerf::error_buffer(); }
