#include "common.h"
PHNET_API int phnet_abi_version(void) { return 1; }
