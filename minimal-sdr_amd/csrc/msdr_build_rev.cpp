// msdr_build_rev.cpp -- the source revision this library was built from (minimal-sdr_amd/Makefile passes it in; recompiled on every make).
#include "../../include/msdr.h"
#ifndef MSDR_BUILD_REV
#define MSDR_BUILD_REV "unknown"
#endif
extern "C" const char *msdr_build_rev(void) { return MSDR_BUILD_REV; }
