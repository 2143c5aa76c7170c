// probe3x.hip -- can ONE wave per SIMD keep the bf16 matrix pipe busy through a split-operand half-step?
//
// Synthetic half-step of a 128 x 256 tile run by 4 waves (1 per SIMD, wave tile 64 x 128 = 2 x 4 MFMA tiles): 48
// v_mfma_f32_32x32x16_bf16 (6 products x 8 tiles), the 18 fragment reads of the NEXT half-step (double-buffered
// registers), NV dummy VALU instructions + NW 8-byte LDS stores + NG global loads standing for the staging work, one
// workgroup barrier.  Prints shader cycles per half-step against the 1 536 of back-to-back MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe3x.hip -o /tmp/probe3x && /tmp/probe3x
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 256;
constexpr int A_STAGE = 3 * 2 * BM * 16, B_STAGE = 3 * 2 * BN * 16;

template <int READS, int BARRIER, int NV, int NW, int NG>
__global__ void __launch_bounds__(256, 1) k_probe(float *out, const float *gsrc, int iters, long long *clk) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * A_STAGE + 3 * B_STAGE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, li = lane & 31, hi = lane >> 5;
    for (int i = t; i < (3 * A_STAGE + 3 * B_STAGE) / 4; i += 256) reinterpret_cast<unsigned *>(smem)[i] = 0x3f803f80u + (i & 7);
    __syncthreads();
    floatx16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    u32x4 fa[2][3][2], fb[2][3][4];
    unsigned a_rofs[2], b_rofs[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) a_rofs[i] = (unsigned)((hi * BM + wm * 64 + i * 32 + li) * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) b_rofs[j] = (unsigned)(3 * A_STAGE + (hi * BN + wn * 128 + j * 32 + li) * 16);
    auto read_frags = [&](int set, int st) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[set][pl][i] = *reinterpret_cast<const u32x4 *>(smem + st * A_STAGE + a_rofs[i] + pl * 2 * BM * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[set][pl][j] = *reinterpret_cast<const u32x4 *>(smem + st * B_STAGE + b_rofs[j] + pl * 2 * BN * 16);
        }
    };
    constexpr int PA[6] = {0, 0, 1, 0, 2, 1}, PB[6] = {0, 1, 0, 2, 0, 1};
    int dummy = lane;
    float4 g[NG > 0 ? NG : 1];
    auto half = [&](int set, int st_next, int it) {
        if (READS) read_frags(set ^ 1, st_next);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][PA[q]][i]),
                                                                        __builtin_bit_cast(bf16x8, fb[set][PB[q]][j]), acc[i][j], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NV; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(dummy) : "v"(k + 1));
#pragma unroll
        for (int k = 0; k < NW; ++k)
            *reinterpret_cast<u32x2 *>(smem + ((st_next + 1) % 3) * A_STAGE + ((t * 8 + k * 2048) % A_STAGE)) = u32x2{(unsigned)dummy, (unsigned)it};
#pragma unroll
        for (int k = 0; k < NG; ++k) g[k] = *reinterpret_cast<const float4 *>(gsrc + ((size_t)(it * 256 + t + k * 4096) & 0xfffff) * 4);
#pragma unroll
        for (int k = 0; k < 48; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, (NV + 47) / 48 + 1, 0);
            if (READS && k < 18) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (NG && k % 6 == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if (NW && k % 12 == 11) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        if (BARRIER) {
            __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
            asm volatile("s_barrier" ::: "memory");
        }
        if (NG) {
#pragma unroll
            for (int k = 0; k < NG; ++k) dummy += (int)g[k].x;
        }
    };
    read_frags(0, 0);
    __syncthreads();
    long long c0 = __builtin_readcyclecounter();
    int st = 0;
    for (int it = 0; it < iters; ++it) {
        half(0, (st + 1) % 3, it);
        half(1, (st + 2) % 3, it);
        st = (st + 2) % 3;
    }
    long long c1 = __builtin_readcyclecounter();
    if (clk && blockIdx.x == 0 && t == 0) clk[0] = c1 - c0;
    float s = (float)dummy;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + t] = s;
}

template <int READS, int BARRIER, int NV, int NW, int NG>
void run(const char *name, float *out, float *gsrc, long long *clk, int blocks, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_probe<READS, BARRIER, NV, NW, NG><<<blocks, 256>>>(out, gsrc, iters, clk);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_probe<READS, BARRIER, NV, NW, NG><<<blocks, 256>>>(out, gsrc, iters, clk);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    long long c = 0;
    hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * iters * 2 * 48 * 32768.0;
    printf("%-44s cycles/half-step %7.1f (ideal 1536: %.3f)   %7.1f TF bf16 = %.3f of 2516.6\n", name, (double)c / (2.0 * iters),
           1536.0 * 2 * iters / (double)c, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 2516.6);
}

int main() {
    float *out, *gsrc;
    long long *clk;
    hipMalloc(&out, 256 * 256 * 4 * 8);
    hipMalloc(&gsrc, 16 << 20);
    hipMemset(gsrc, 0, 16 << 20);
    hipMalloc(&clk, 64);
    const int blocks = 256 * 4, iters = 400;
    run<0, 0, 0, 0, 0>("MFMAs only", out, gsrc, clk, blocks, iters);
    run<1, 0, 0, 0, 0>("+ 18 fragment reads (prefetch)", out, gsrc, clk, blocks, iters);
    run<1, 1, 0, 0, 0>("+ barrier", out, gsrc, clk, blocks, iters);
    run<1, 1, 48, 0, 0>("+ 48 VALU", out, gsrc, clk, blocks, iters);
    run<1, 1, 48, 6, 0>("+ 6 LDS stores", out, gsrc, clk, blocks, iters);
    run<1, 1, 48, 6, 4>("+ 4 global loads", out, gsrc, clk, blocks, iters);
    run<1, 1, 96, 6, 4>("96 VALU, 6 stores, 4 loads", out, gsrc, clk, blocks, iters);
    run<1, 1, 144, 12, 8>("144 VALU, 12 stores, 8 loads", out, gsrc, clk, blocks, iters);
    return 0;
}
