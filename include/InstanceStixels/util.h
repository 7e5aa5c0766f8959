/*
 * util.h -- small helpers, counterpart of
 * /root/reference/InstanceStixels/include/InstanceStixels/util.h:22-48.  The reference's
 * CUDA_CHECK_RETURN prints and exit(1)s on a runtime failure (util.h:34-42); IS_CHECK_RETURN
 * keeps that error convention for the C-ABI return codes of the HIP core.
 */
#ifndef INSTANCESTIXELS_AMD_UTIL_H_
#define INSTANCESTIXELS_AMD_UTIL_H_

#include <cstdlib>
#include <iostream>

#include "instance_stixels_core.h"

constexpr int WAVEFRONT_SIZE = 64; /* the reference's WARP_SIZE = 32 has no meaning on CDNA */

#define IS_CHECK_RETURN(value) IsCheckReturnAux(__FILE__, __LINE__, #value, (value))

static inline void IsCheckReturnAux(const char* file, unsigned line, const char* statement,
                                    int rc) {
    if (rc == IS_OK) return;
    std::cerr << statement << " returned " << is_last_error() << "(" << rc << ") at " << file
              << ":" << line << std::endl;
    std::exit(1);
}

static inline int divUp(int total, int grain) { return (total + grain - 1) / grain; }

#endif
