// splat_slice.hip -- the two HBM-bound gathers of the bilateral layers on gfx950.
//
// Both kernels move whole channel-last rows: a row of C floats is read by a group of G
// lanes with one 16-byte load per lane (G = 8..64 chosen so that G*4 >= C when possible),
// so every wave-level load instruction touches 64/G full rows -- coalesced 128-byte
// segments, no LDS needed and no atomics:
//   * splat is a CSR segmented reduction (vertex -> its contributing points), the 64/G
//     lane groups of a wave take alternate contributors and are combined with wave
//     shuffles (DPP/ds_swizzle under the hood); the density normaliser is fused;
//   * slice is a 4-row weighted gather per output point with bias fused.
// Algorithmic bytes: splat 4*C*N + 32*N + 4*(C+1)*H; slice
// 4*C*H + 32*N + 4*C*N.
#include "common.h"

using namespace hpl;

namespace {

template <typename V>
struct vec_ops;
template <>
struct vec_ops<float4> {
    static __device__ __forceinline__ float4 zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ void fma(float4 &a, float w, const float4 &x) {
        a.x = fmaf(w, x.x, a.x); a.y = fmaf(w, x.y, a.y); a.z = fmaf(w, x.z, a.z); a.w = fmaf(w, x.w, a.w);
    }
    static __device__ __forceinline__ float4 shfl_xor_add(float4 a, int o) {
        a.x += __shfl_xor(a.x, o); a.y += __shfl_xor(a.y, o); a.z += __shfl_xor(a.z, o); a.w += __shfl_xor(a.w, o);
        return a;
    }
    static __device__ __forceinline__ float4 scale(float4 a, float s) {
        return make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
    }
    static __device__ __forceinline__ float4 add(float4 a, float4 b) {
        return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
};
template <>
struct vec_ops<float> {
    static __device__ __forceinline__ float zero() { return 0.f; }
    static __device__ __forceinline__ void fma(float &a, float w, const float &x) { a = fmaf(w, x, a); }
    static __device__ __forceinline__ float shfl_xor_add(float a, int o) { return a + __shfl_xor(a, o); }
    static __device__ __forceinline__ float scale(float a, float s) { return a * s; }
    static __device__ __forceinline__ float add(float a, float b) { return a + b; }
};

// V = float4 (vector path) or float (scalar path); CV = number of V columns per row.
template <typename V, int G>
__global__ void __launch_bounds__(256) k_splat(const float *__restrict__ feat, int64_t ldf, int CV,
                                               const int32_t *__restrict__ csr_ptr,
                                               const int32_t *__restrict__ csr_pt,
                                               const float *__restrict__ csr_w,
                                               const float *__restrict__ norm, int64_t H,
                                               float *__restrict__ out, int64_t ldo) {
    constexpr int NG = 64 / G;
    using ops = vec_ops<V>;
    const int lane = threadIdx.x & 63;
    const int g = lane / G, lg = lane % G;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t v = wave0; v < H; v += nwaves) {
        const int32_t b = csr_ptr[v], e = csr_ptr[v + 1];
        const float sc = norm ? norm[v] : 1.0f;
        for (int c0 = 0; c0 < CV; c0 += G) {   // wave-uniform trip count: the shuffles need all lanes
            const int cq = c0 + lg;
            const bool active = cq < CV;
            V acc = ops::zero();
            for (int32_t j = b + g; j < e; j += NG) {
                const int32_t pt = csr_pt[j];
                const float w = csr_w[j];
                if (active) {
                    const V x = *reinterpret_cast<const V *>(feat + (int64_t)pt * ldf + (int64_t)cq * (sizeof(V) / 4));
                    ops::fma(acc, w, x);
                }
            }
#pragma unroll
            for (int o = G; o < 64; o <<= 1) acc = ops::shfl_xor_add(acc, o);
            if (g == 0 && active)
                *reinterpret_cast<V *>(out + v * ldo + (int64_t)cq * (sizeof(V) / 4)) = ops::scale(acc, sc);
        }
    }
}

template <typename V, int G>
__global__ void __launch_bounds__(256) k_slice(const float *__restrict__ Y, int64_t ldy, int CV,
                                               const float *__restrict__ bary,
                                               const int32_t *__restrict__ off, int64_t N,
                                               const float *__restrict__ vscale,
                                               const float *__restrict__ bias, float *__restrict__ out,
                                               int64_t ldo) {
    constexpr int NG = 64 / G;
    using ops = vec_ops<V>;
    const int lane = threadIdx.x & 63;
    const int g = lane / G, lg = lane % G;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t n = wave0 * NG + g; n < N; n += nwaves * NG) {
        int32_t v[4];
        float w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = off[(int64_t)r * N + n];
            w[r] = bary[(int64_t)r * N + n];
            if (vscale && v[r] >= 0) w[r] *= vscale[v[r]];
        }
        for (int cq = lg; cq < CV; cq += G) {
            V acc = ops::zero();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (v[r] >= 0) {
                    const V y = *reinterpret_cast<const V *>(Y + (int64_t)v[r] * ldy + (int64_t)cq * (sizeof(V) / 4));
                    ops::fma(acc, w[r], y);
                }
            }
            if (bias) acc = ops::add(acc, *reinterpret_cast<const V *>(bias + (int64_t)cq * (sizeof(V) / 4)));
            *reinterpret_cast<V *>(out + n * ldo + (int64_t)cq * (sizeof(V) / 4)) = acc;
        }
    }
}

inline int pick_group(int cv) {
    if (cv <= 8) return 8;
    if (cv <= 16) return 16;
    if (cv <= 32) return 32;
    return 64;
}

}  // namespace

extern "C" int hpl_splat(const float *feat, int64_t ldf, int C, const int32_t *csr_ptr,
                         const int32_t *csr_pt, const float *csr_w, const float *norm, int64_t H,
                         float *out, int64_t ldo, hplStream stream) {
    HPL_REQUIRE(feat && csr_ptr && csr_pt && csr_w && out, "hpl_splat: null pointer");
    HPL_REQUIRE(C > 0 && H >= 0 && ldf >= C && ldo >= C, "hpl_splat: bad sizes C=%d H=%lld ldf=%lld ldo=%lld", C,
                (long long)H, (long long)ldf, (long long)ldo);
    if (H == 0) return HPL_OK;
    hipStream_t s = to_stream(stream);
    const bool vec = (C % 4 == 0) && (ldf % 4 == 0) && (ldo % 4 == 0) && aligned16(feat) && aligned16(out);
    const int cv = vec ? C / 4 : C;
    const int G = pick_group(cv);
    const int grid = (int)imin(cdiv(H, 4), 256 * 16);
#define LAUNCH(V, GG) k_splat<V, GG><<<grid, 256, 0, s>>>(feat, ldf, cv, csr_ptr, csr_pt, csr_w, norm, H, out, ldo)
    if (vec) {
        if (G == 8) LAUNCH(float4, 8); else if (G == 16) LAUNCH(float4, 16);
        else if (G == 32) LAUNCH(float4, 32); else LAUNCH(float4, 64);
    } else {
        if (G == 8) LAUNCH(float, 8); else if (G == 16) LAUNCH(float, 16);
        else if (G == 32) LAUNCH(float, 32); else LAUNCH(float, 64);
    }
#undef LAUNCH
    HPL_CHECK_LAUNCH("hpl_splat");
    return HPL_OK;
}

extern "C" int hpl_slice(const float *Y, int64_t ldy, int C, const float *bary, const int32_t *off,
                         int64_t N, const float *vscale, const float *bias, float *out, int64_t ldo,
                         hplStream stream) {
    HPL_REQUIRE(Y && bary && off && out, "hpl_slice: null pointer");
    HPL_REQUIRE(C > 0 && N >= 0 && ldy >= C && ldo >= C, "hpl_slice: bad sizes C=%d N=%lld ldy=%lld ldo=%lld", C,
                (long long)N, (long long)ldy, (long long)ldo);
    if (N == 0) return HPL_OK;
    hipStream_t s = to_stream(stream);
    const bool vec = (C % 4 == 0) && (ldy % 4 == 0) && (ldo % 4 == 0) && aligned16(Y) && aligned16(out) &&
                     (!bias || aligned16(bias));
    const int cv = vec ? C / 4 : C;
    const int G = pick_group(cv);
    const int ng = 64 / G;
    const int grid = (int)imin(cdiv(N, 4 * ng), 256 * 16);
#define LAUNCH(V, GG) k_slice<V, GG><<<grid, 256, 0, s>>>(Y, ldy, cv, bary, off, N, vscale, bias, out, ldo)
    if (vec) {
        if (G == 8) LAUNCH(float4, 8); else if (G == 16) LAUNCH(float4, 16);
        else if (G == 32) LAUNCH(float4, 32); else LAUNCH(float4, 64);
    } else {
        if (G == 8) LAUNCH(float, 8); else if (G == 16) LAUNCH(float, 16);
        else if (G == 32) LAUNCH(float, 32); else LAUNCH(float, 64);
    }
#undef LAUNCH
    HPL_CHECK_LAUNCH("hpl_slice");
    return HPL_OK;
}
