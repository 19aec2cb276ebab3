// Shared helpers for libmcav_depth.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mcav_depth.h"

#define MCAV_EXPORT extern "C" __attribute__((visibility("default")))

namespace mcav {

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MCAV_OK : MCAV_E_LAUNCH;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Tuning and diagnostic switches are COMPILE-TIME.  The shipped library reads no MCAV_* environment variable: every knob below is its
// default.  An experiment build (`make variant NAME=tune FLAGS=-DMCAV_TUNE_ENV` -> ../mcav/libmcav_depth_tune.so, selected with
// MCAV_LIB_PATH) reads them from the environment once; the timing-only forms that return WRONG results (conv_bf16.hip MCAV_PATCH_DIAG,
// conv_igemm.hip MCAV_DIAG) are -D values of such builds and nothing else (VERDICT / ADVICE round 3).
#ifdef MCAV_TUNE_ENV
#define MCAV_KNOB_INT(name, dflt) ([] { const char* e = getenv(name); return e ? atoi(e) : (dflt); }())
#define MCAV_KNOB_FLOAT(name, dflt) ([] { const char* e = getenv(name); return e ? (float)atof(e) : (dflt); }())
#else
#define MCAV_KNOB_INT(name, dflt) (dflt)
#define MCAV_KNOB_FLOAT(name, dflt) (dflt)
#endif

// Sum over the 64 lanes of a wavefront; every lane gets the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------- cross-workgroup hand-off inside ONE launch
// "Every workgroup leaves a partial result; the last one to arrive (a ticket) finishes the job."  What a workgroup writes for ANOTHER to
// read inside the same launch must pass the XCD-local L2: on gfx950 each of the 8 XCDs has its own L2 and only accesses at agent scope (the
// instruction's sc1 bit) go through to / come from the coherence point.  The protocol, every step explicit in the ISA (tests/test_isa_handoff.py
// checks the compiled kernels for it on the CPU, so a compiler or ROCm upgrade that changes the code generation is caught):
//   producer:  handoff_store (global_store ... sc1) of every word it publishes
//              handoff_release(): s_waitcnt vmcnt(0) -- EVERY thread waits until its own stores are acknowledged (a workgroup-scope release
//              fence does NOT do that: the compiler is entitled to -- and round 3's build did -- issue the ticket while the stores were still in
//              flight to another L2 channel; the hand-off then worked by timing only.  ADVICE round 3.)
//              __syncthreads(), then ONE thread takes the ticket: an agent-scope atomic add (performed at the coherence point)
//   finisher:  the thread that drew the last ticket tells its workgroup through LDS + __syncthreads(); the loads that follow are issued after
//              the ticket's value has returned (in-order issue behind the s_waitcnt on the atomic's result) and are handoff_load (sc1).
// An agent-scope release / acquire FENCE pair would be the portable spelling; on this chip it is a write-back + invalidate of the whole
// XCD L2 per workgroup, which took the loss kernel's gather lines with it and doubled its time.
__device__ __forceinline__ void handoff_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void handoff_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void handoff_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float handoff_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double handoff_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void handoff_release() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // compiler ordering (and LDS) ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // ... and this thread's global stores have been acknowledged
}
__device__ __forceinline__ unsigned handoff_ticket(unsigned* t) { return __hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace mcav
