/*
 * configuration.h -- compile-time constants of the stixel library, source-compatible with
 * /root/reference/InstanceStixels/include/InstanceStixels/configuration.h:24-36, but free of
 * CUDA headers so that callers can be plain C++ as well as HIP.
 */
#ifndef INSTANCESTIXELS_AMD_CONFIGURATION_H_
#define INSTANCESTIXELS_AMD_CONFIGURATION_H_

#include <limits>

#include "instance_stixels_core.h"

#define MAX_LOGPROB (std::numeric_limits<float>::infinity()) /* reference: CUDART_INF_F */
constexpr int LOG_LUT_SIZE = 1000000;
constexpr int DOWNSAMPLE_FACTOR = IS_DOWNSAMPLE_FACTOR;
constexpr int MAX_STIXELS_PER_COLUMN = IS_MAX_STIXELS_PER_COLUMN;

typedef float pixel_t;

#endif
