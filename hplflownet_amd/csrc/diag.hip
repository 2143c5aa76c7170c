// diag.hip -- libhplbcl_diag.so: measurement helpers that are NOT part of the product library (include/hpl_diag.h).  bench.py
// uses hpl_mfma_probe to quote the matrix-pipe rate the chip sustains at its actual clock next to the datasheet peak.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hpl_diag.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
#define HPL_REQUIRE(cond, ...) do { if (!(cond)) return -1; } while (0)
#define HPL_CHECK_LAUNCH(name) do { if (hipGetLastError() != hipSuccess) return -2; } while (0)
#define HPL_OK 0
static inline hipStream_t to_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// ------------------------------------------------------------------------------------------
// Diagnostic: sustained rate of v_mfma_f32_32x32x2_f32 with no memory traffic (what the chip
// gives at its actual clock under this instruction mix).  Used by tools/ and bench.py to quote
// the measured ceiling next to the 157.3 TFLOP/s datasheet peak.
// ------------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) k_mfma_probe(float *out, int iters) {
    floatx16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)blockIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same number of MFMAs as one dependent chain per wave (one accumulator, as the 32x32-per-wave tiles)
__global__ void __launch_bounds__(256) k_mfma_probe_chain(float *out, int iters) {
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)blockIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
}  // namespace

namespace {
// The same MFMA stream on operands that CHANGE from instruction to instruction (eight pseudo-random values per lane
// and operand, rotated): the matrix pipe's switching activity -- hence power, hence the clock the chip sustains --
// is that of real data, not of the constant operands of k_mfma_probe.  mode 2 adds the LDS fragment traffic of the
// gather-GEMM loop (two ds_read_b32 per MFMA).
template <int MODE>
__global__ void __launch_bounds__(256) k_mfma_probe_data(float *out, int iters, long long *clk) {
    __shared__ float lds[2 * 32 * 130];
    floatx16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u;
        a[i] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
        h = h * 1664525u + 1013904223u;
        b[i] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
    for (int i = threadIdx.x; i < 2 * 32 * 130; i += 256) lds[i] = a[i & 7] * 0.5f + b[(i >> 3) & 7];
    __syncthreads();
    long long c0 = 0, w0 = 0;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); w0 = __builtin_amdgcn_s_memrealtime(); }
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float av = lds[((u * 4 + i) & 31) * 130 + lane + ((it & 1) ? 4160 : 0)];
                    const float bv = lds[((u * 4 + i + 7) & 31) * 130 + 64 + (lane & 31) + ((it & 1) ? 0 : 4160)];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + i) & 7], b[(u * 3 + i) & 7], acc[i], 0, 0, 0);
            }
        }
    }
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_readcyclecounter() - c0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - w0;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
}  // namespace

extern "C" int hpl_mfma_probe_data(float *out, int blocks, int iters, int mode, long long *clk, void *stream) {
    HPL_REQUIRE(out && blocks > 0 && iters > 0 && (mode == 1 || mode == 2), "hpl_mfma_probe_data: bad arguments");
    if (mode == 1) k_mfma_probe_data<1><<<blocks, 256, 0, to_stream(stream)>>>(out, iters, clk);
    else k_mfma_probe_data<2><<<blocks, 256, 0, to_stream(stream)>>>(out, iters, clk);
    HPL_CHECK_LAUNCH("hpl_mfma_probe_data");
    return HPL_OK;
}

extern "C" int hpl_mfma_probe(float *out, int blocks, int iters, void *stream) {
    HPL_REQUIRE(out && blocks > 0 && iters != 0, "hpl_mfma_probe: bad arguments");
    if (iters < 0) k_mfma_probe_chain<<<blocks, 256, 0, to_stream(stream)>>>(out, -iters);
    else k_mfma_probe<<<blocks, 256, 0, to_stream(stream)>>>(out, iters);
    HPL_CHECK_LAUNCH("hpl_mfma_probe");
    return HPL_OK;
}
