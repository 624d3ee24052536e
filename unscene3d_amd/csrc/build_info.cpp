// What this library was built from (usc_build_info): target architecture, compiler, UTC time of the link and a hash over
// every source the objects were compiled from — filled in by the Makefile, rebuilt whenever one of them changes.
#include "../../include/usc3d.h"

#ifndef USC_BUILD_INFO
#define USC_BUILD_INFO "unknown (built without the Makefile)"
#endif

extern "C" const char* usc_build_info(void) { return USC_BUILD_INFO; }
