/* rpf_engine_testing.h -- test hooks of librpf_engine.so.  NOT part of the drop-in boundary: include/rpf_engine.h is the
 * ABI a reference-side binding uses; what is declared here exists for tests/ only, may change without
 * an rpf_abi_version() bump, and no host code may call it (tests/test_cabi.py checks both headers). */
#ifndef RPF_ENGINE_TESTING_H
#define RPF_ENGINE_TESTING_H

#include "../../include/rpf_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook (tests/test_gpu_fused_abort.py): sabotage fused launches -- after `skip` untouched ones the next
 * `count` (< 0: all) launches fail to assemble.  mode 1: the launch finds a 33rd workgroup on XCD 0 and gives up at
 * once; mode 2: one CU is held by a squatter kernel until the launch has given up (the real failure, seconds);
 * mode 0: disarm.  No effect on an engine that is not on the fused kernel. */
int rpf_debug_fused_fault(rpf_engine* e, int mode, int skip, int count);

#ifdef __cplusplus
}
#endif
#endif
