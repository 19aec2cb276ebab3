// Shared helpers for libmcav_depth.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/mcav_depth.h"

#define MCAV_EXPORT extern "C" __attribute__((visibility("default")))

namespace mcav {

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MCAV_OK : MCAV_E_LAUNCH;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Sum over the 64 lanes of a wavefront; every lane gets the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace mcav
