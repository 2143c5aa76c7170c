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

// ------------------------------------------------------------------------------------------
// A/B of the splat as SCATTER-ADDS (what models/bilateralNN.py:24-29 `sparse_sum` means literally, and the form BASELINE.json's
// north star names: "LDS-staged ... wavefront-reduced atomicAdd") against the product's CSR segmented reduction (splat_slice.hip
// k_splat: deterministic, no atomics).  Measurement only -- tools/bench_splat_slice.py --atomic; the sums of these kernels depend on
// the order the atomics land in.  out[v, :] += bary[r, n] * norm[v] * feat[n, :] for v = off[r, n]; out is cleared by the call.
//   mode 0: one lane per (point n, remainder r, float4 column), four global_atomic_add_f32 each;
//   mode 1: a workgroup stages its 24 points' <= 96 vertex rows in an LDS open-address table (ds_add_f32), then flushes every
//           occupied slot with one global atomic per element -- entries of the block that share a vertex are combined in LDS.
// ------------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) k_splat_atomic(const float *__restrict__ feat, int64_t ldf, int CV, const float *__restrict__ bary,
                                                      const int32_t *__restrict__ off, int64_t N, const float *__restrict__ norm,
                                                      float *__restrict__ out, int64_t ldo) {
    const int64_t total = 4 * N * CV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t e = i / CV;
        const int cq = (int)(i - e * CV);
        const int64_t n = e >> 2;
        const int r = (int)(e & 3);
        const int32_t v = off[(int64_t)r * N + n];
        if (v < 0) continue;
        const float w = bary[(int64_t)r * N + n] * (norm ? norm[v] : 1.0f);
        const float4 x = *reinterpret_cast<const float4 *>(feat + n * ldf + (int64_t)cq * 4);
        float *dst = out + (int64_t)v * ldo + (int64_t)cq * 4;
        atomicAdd(dst + 0, w * x.x); atomicAdd(dst + 1, w * x.y); atomicAdd(dst + 2, w * x.z); atomicAdd(dst + 3, w * x.w);
    }
}

constexpr int SA_PTS = 24, SA_SLOTS = 128;
__global__ void __launch_bounds__(256) k_splat_atomic_lds(const float *__restrict__ feat, int64_t ldf, int CV, const float *__restrict__ bary,
                                                          const int32_t *__restrict__ off, int64_t N, const float *__restrict__ norm,
                                                          float *__restrict__ out, int64_t ldo) {
    extern __shared__ float sm[];
    int *keys = reinterpret_cast<int *>(sm);                 // [SA_SLOTS] vertex of a slot, -1 = free
    int *eslot = keys + SA_SLOTS;                            // [4 * SA_PTS] slot of every entry of the block
    float *rows = sm + SA_SLOTS + 4 * SA_PTS;                // [SA_SLOTS][4 * CV]
    const int C = 4 * CV;
    for (int64_t p0 = (int64_t)blockIdx.x * SA_PTS; p0 < N; p0 += (int64_t)gridDim.x * SA_PTS) {
        for (int i = threadIdx.x; i < SA_SLOTS; i += 256) keys[i] = -1;
        for (int i = threadIdx.x; i < SA_SLOTS * C; i += 256) rows[i] = 0.f;
        __syncthreads();
        if (threadIdx.x < 4 * SA_PTS) {
            const int64_t n = p0 + (threadIdx.x >> 2);
            const int r = threadIdx.x & 3;
            int slot = -1;
            if (n < N) {
                const int v = off[(int64_t)r * N + n];
                if (v >= 0) {
                    unsigned h = ((unsigned)v * 2654435761u) >> 25;          // 7 bits
                    while (true) {
                        const int prev = atomicCAS(&keys[h], -1, v);
                        if (prev == -1 || prev == v) break;
                        h = (h + 1) & (SA_SLOTS - 1);
                    }
                    slot = (int)h;
                }
            }
            eslot[threadIdx.x] = slot;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 4 * SA_PTS * CV; i += 256) {
            const int e = i / CV, cq = i - e * CV;
            const int slot = eslot[e];
            if (slot < 0) continue;
            const int64_t n = p0 + (e >> 2);
            const int r = e & 3;
            const int v = keys[slot];
            const float w = bary[(int64_t)r * N + n] * (norm ? norm[v] : 1.0f);
            const float4 x = *reinterpret_cast<const float4 *>(feat + n * ldf + (int64_t)cq * 4);
            float *dst = rows + slot * C + cq * 4;
            atomicAdd(dst + 0, w * x.x); atomicAdd(dst + 1, w * x.y); atomicAdd(dst + 2, w * x.z); atomicAdd(dst + 3, w * x.w);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < SA_SLOTS * C; i += 256) {
            const int slot = i / C, c = i - slot * C;
            const int v = keys[slot];
            if (v >= 0) atomicAdd(out + (int64_t)v * ldo + c, rows[i]);
        }
        __syncthreads();
    }
}
}  // namespace

extern "C" int hpl_diag_splat_atomic(const float *feat, int64_t ldf, int C, const float *bary, const int32_t *off, int64_t N,
                                     const float *norm, int64_t H, float *out, int64_t ldo, int mode, void *stream) {
    HPL_REQUIRE(feat && bary && off && out && C > 0 && C % 4 == 0 && ldf % 4 == 0 && ldo >= C && N >= 0 && H >= 0 && (mode == 0 || mode == 1),
                "hpl_diag_splat_atomic: bad arguments");
    if (N == 0 || H == 0) return HPL_OK;
    hipStream_t s = to_stream(stream);
    if (hipMemset2DAsync(out, (size_t)ldo * 4, 0, (size_t)C * 4, (size_t)H, s) != hipSuccess) return -2;
    const int CV = C / 4;
    if (mode == 0) {
        const int64_t blocks = (4 * N * CV + 255) / 256;
        k_splat_atomic<<<(int)(blocks < (1 << 20) ? blocks : (1 << 20)), 256, 0, s>>>(feat, ldf, CV, bary, off, N, norm, out, ldo);
    } else {
        const size_t lds = (size_t)(SA_SLOTS + 4 * SA_PTS + SA_SLOTS * C) * 4;
        HPL_REQUIRE(lds <= 64 * 1024, "hpl_diag_splat_atomic: C too large for the LDS table");
        const int64_t blocks = (N + SA_PTS - 1) / SA_PTS;
        k_splat_atomic_lds<<<(int)(blocks < (1 << 16) ? blocks : (1 << 16)), 256, lds, s>>>(feat, ldf, CV, bary, off, N, norm, out, ldo);
    }
    HPL_CHECK_LAUNCH("hpl_diag_splat_atomic");
    return HPL_OK;
}

// ------------------------------------------------------------------------------------------
// Diagnostic (round 6): what a chain of small DEPENDENT steps costs as kernel launches and as phases of ONE persistent launch
// separated by a grid barrier -- the question behind "one launch for levels 3-6" (DESIGN.md section 9).  A step: workgroup b
// reads `words` floats that workgroup (b + 1) % grid wrote in the step before (a cross-CU, usually cross-XCD dependency, as a
// gather over the previous layer's rows is), adds 1, writes its own block.  mode 0: `steps` launches on the stream; mode 1: one
// launch, a monotonic-counter grid barrier between steps (lane 0: agent release fence -> atomic arrive -> relaxed agent poll ->
// agent acquire fence; MI355X_MICROARCH.md "barrier-counter"); mode 2: the XCD-hierarchical form of the same guide
// ("barrier-xcd": per-XCC arrive counters, one leader per XCC on the top counter, per-XCC release generation).
// ------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ void chain_step(const float *__restrict__ in, float *__restrict__ out, int words, int grid) {
    const int src = ((int)blockIdx.x + 1) % grid;
    for (int i = threadIdx.x; i < words; i += blockDim.x) out[(size_t)blockIdx.x * words + i] = in[(size_t)src * words + i] + 1.0f;
}
__global__ void __launch_bounds__(256) k_chain_step(const float *in, float *out, int words, int grid) { chain_step(in, out, words, grid); }

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

__global__ void __launch_bounds__(256) k_chain_persistent(float *a, float *b, int words, int steps, unsigned *bar, int mode) {
    const int grid = gridDim.x;
    __shared__ unsigned xcc_s;
    if (threadIdx.x == 0) xcc_s = xcc_id();
    __syncthreads();
    const unsigned xcc = xcc_s;
    // bar[0]: top counter; bar[16 * (1 + x)]: arrivals of XCC x; bar[16 * (9 + x)]: release generation of XCC x; bar[16 * 17 + x]: WGs on XCC x
    if (mode == 2 && threadIdx.x == 0) atomicAdd(&bar[16 * 17 + xcc], 1u);          // census (before the first barrier: counted below)
    for (int s = 0; s < steps; ++s) {
        chain_step(s & 1 ? b : a, s & 1 ? a : b, words, grid);
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (mode == 1) {
                atomicAdd(&bar[0], 1u);
                const unsigned want = (unsigned)(s + 1) * (unsigned)grid;
                while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
            } else {
                // step 0 doubles as the census barrier: every workgroup has registered on its XCC before anybody leaves it
                if (s == 0) {
                    atomicAdd(&bar[0], 1u);
                    while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)grid) __builtin_amdgcn_s_sleep(1);
                } else {
                    const unsigned mine = __hip_atomic_load(&bar[16 * 17 + xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned arrived = atomicAdd(&bar[16 * (1 + xcc)], 1u) + 1u;
                    if (arrived == (unsigned)s * mine) {          // last of this XCC: the leader goes to the top counter
                        const unsigned top = atomicAdd(&bar[16 * 18], 1u) + 1u;
                        if (top == (unsigned)s * 8u) {             // last leader (8 XCCs hold workgroups at 256 workgroups): release everybody
                            for (int x = 0; x < 8; ++x) __hip_atomic_store(&bar[16 * (9 + x)], (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    while (__hip_atomic_load(&bar[16 * (9 + xcc)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)s) __builtin_amdgcn_s_sleep(1);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}
}  // namespace

extern "C" int hpl_diag_chain(float *a, float *b, int grid, int words, int steps, unsigned *bar, int mode, void *stream) {
    HPL_REQUIRE(a && b && grid > 0 && grid <= 256 && words > 0 && steps > 0 && (mode == 0 || bar));
    hipStream_t s = to_stream(stream);
    if (mode == 0) {
        for (int i = 0; i < steps; ++i) k_chain_step<<<grid, 256, 0, s>>>(i & 1 ? b : a, i & 1 ? a : b, words, grid);
    } else {
        if (hipMemsetAsync(bar, 0, 4 * 16 * 20, s) != hipSuccess) return -2;
        k_chain_persistent<<<grid, 256, 0, s>>>(a, b, words, steps, bar, mode);
    }
    HPL_CHECK_LAUNCH("hpl_diag_chain");
    return HPL_OK;
}
