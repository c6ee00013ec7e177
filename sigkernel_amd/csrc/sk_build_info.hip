// sk_build_info.hip -- what this library was built from, and by what (host code only).  The Makefile passes SK_BUILD_HASH (first 16
// hex digits of the SHA-256 of every source of the library, in sorted order) and SK_BUILD_TOOLCHAIN (`hipcc --version`: HIP and
// clang versions).  The inline-asm kernels rest on what THIS compiler's register allocator does (tools/check_async_hazards.py
// gates the build on it); a library built by another toolchain says so here, and tests/test_abi.py pins the pair.
#include "../../include/sigkernel_amd.h"

#ifndef SK_BUILD_HASH
#define SK_BUILD_HASH "unknown"
#endif
#ifndef SK_BUILD_TOOLCHAIN
#define SK_BUILD_TOOLCHAIN "unknown"
#endif

extern "C" const char *sk_build_info(void) {
    return "sigkernel_amd gfx950; sources " SK_BUILD_HASH "; " SK_BUILD_TOOLCHAIN "; ISA hazard lint passed at build";
}
