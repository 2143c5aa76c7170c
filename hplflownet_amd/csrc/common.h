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

}  // namespace hpl
