// Error reporting + ABI version for libsparsefusion_hip.so.
#include "sf_common.h"

thread_local char sf_err_buf[512] = {0};

extern "C" const char* sf_last_error(void) { return sf_err_buf; }
extern "C" int sf_abi_version(void) { return 1; }
extern "C" int sf_operand_is_f16(void) { return SF_OPERAND_F16; }
