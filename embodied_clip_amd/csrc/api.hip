// Library-level entry points of the C-ABI (include/ec_amd.h).
#include "common.h"

extern "C" int ec_version(void) { return 207; }   // 0.2.7 (bump whenever a kernel or a launch plan changes: keys profiles/*traffic*.json)

extern "C" const char* ec_strerror(int code) {
    switch (code) {
        case EC_OK: return "ok";
        case EC_ERR_ARG: return "invalid argument (null pointer or bad enum)";
        case EC_ERR_SHAPE: return "unsupported shape for this kernel";
        case EC_ERR_LAUNCH: return "HIP kernel launch failed";
        case EC_ERR_WORKSPACE: return "workspace too small";
        case EC_ERR_ALLOC: return "host allocation failed";
        case EC_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown error";
    }
}
