#include "common.h"
extern "C" const char* mu_version_host(void) { return "maskunet_hip 0.1 (gfx950, fp32+fp16 MFMA)"; }
