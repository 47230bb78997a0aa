// bgnn_build_id(): the kernel-source hash this library was built from (include/bgnn.h).  __graft_entry__.build() passes
// -DBGNN_BUILD_ID="<sha256[:16] over csrc/*.hip, *.h>" and recompiles this unit whenever that hash changes.
#include "../../include/bgnn.h"

#ifndef BGNN_BUILD_ID
#define BGNN_BUILD_ID "unknown"
#endif

extern "C" const char *bgnn_build_id(void) { return BGNN_BUILD_ID; }
