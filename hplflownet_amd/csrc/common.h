// common.h -- shared host-side helpers of libhplbcl.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/hpl_bcl.h"

namespace hpl {

void set_error(const char *fmt, ...);

// out[n] += sum_m X[m*ld + n] (index_ops.hip); `out` is NOT zeroed
void colsum_accumulate(const float *X, int64_t ld, int64_t M, int N, float *out, hipStream_t s);

// strided helpers of the training executor (train_ops.hip)
int zero_cols(float *dst, int64_t ldd, int64_t rows, int cols, hipStream_t s);
int add_cols(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t rows, int cols, hipStream_t s);
int vcopy(const float *src, float *dst, int n, hipStream_t s);

inline hipStream_t to_stream(hplStream s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ inline int64_t imin(int64_t a, int64_t b) { return a < b ? a : b; }
__host__ __device__ inline int64_t imax(int64_t a, int64_t b) { return a > b ? a : b; }

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// hipGetLastError after a launch: launch-configuration errors surface here.
#define HPL_CHECK_LAUNCH(name)                                                         \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            hpl::set_error("%s: kernel launch failed: %s", name, hipGetErrorString(e__)); \
            return HPL_EHIP;                                                           \
        }                                                                              \
    } while (0)

#define HPL_REQUIRE(cond, ...)              \
    do {                                    \
        if (!(cond)) {                      \
            hpl::set_error(__VA_ARGS__);    \
            return HPL_EINVAL;              \
        }                                   \
    } while (0)


// *slot = max(*slot, v) for magnitudes kept as bit patterns (they order like unsigned integers, NaN on top).  Hundreds to thousands of
// workgroups publishing into ONE word serialise on it: look first -- a relaxed device-scope load does not serialise -- and only a
// workgroup that would raise the value pays the atomic.  A stale (smaller) look costs an atomic that changes nothing; the result is
// the same maximum.  (Single forward at N = 8 192: 3.19-3.28 -> 3.07-3.11 ms over the boxes of round 5.)
__device__ __forceinline__ void amax_publish(unsigned *slot, unsigned v) {
    if (v == 0u) return;
    if (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= v) return;
    atomicMax(slot, v);
}
}  // namespace hpl
