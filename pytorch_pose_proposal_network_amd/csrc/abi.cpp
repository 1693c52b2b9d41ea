// libppn: error channel and version (see include/ppn.h).
#include "common.h"

namespace ppn {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace ppn

extern "C" const char* ppn_last_error(void) { return ppn::error_buffer(); }
extern "C" int ppn_version(void) { return 1; }
