// ppp_io.cpp calls ppp_default_params(), which lives in the HIP translation unit; the sanitizer build of the host I/O
// alone gets this zeroing stand-in (the parser tests do not look at parameter defaults).
#include <cstring>
#include "../../include/ppp_hip.h"
extern "C" void ppp_default_params(ppp_params *p) { memset(p, 0, sizeof(*p)); }
