// Error channel and ABI version of libmonopsr_hip.so.
#include "common.h"

namespace mpsr {
char *error_buffer()
{
    static thread_local char buf[kErrorBufferBytes] = {0};
    return buf;
}
}  // namespace mpsr

extern "C" const char *mpsr_last_error(void) { return mpsr::error_buffer(); }
extern "C" int mpsr_abi_version(void) { return 1; }
