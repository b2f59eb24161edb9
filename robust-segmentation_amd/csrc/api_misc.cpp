// Library identification (no device work).
#include "../../include/sea_hip.h"

#ifndef SEA_BUILD_STAMP
#define SEA_BUILD_STAMP "unknown"
#endif

extern "C" int sea_abi_version(void) { return 1; }
extern "C" const char* sea_build_info(void) { return "libsea_hip gfx950 " SEA_BUILD_STAMP; }
