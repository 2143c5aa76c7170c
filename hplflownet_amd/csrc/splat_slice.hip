// splat_slice.hip -- the two HBM-bound gathers of the bilateral layers on gfx950.
//
// Both kernels move whole channel-last rows with one 16-byte word per lane and one lane per
// (row, 16-byte column): consecutive lanes read consecutive words of a gathered row (coalesced
// 64..256-byte segments), no lane idles for row lengths that are not powers of two, no LDS and
// no atomics:
//   * splat is a CSR segmented reduction (vertex -> its contributing points, in CSR order:
//     deterministic) with the density normaliser fused;
//   * slice is a 4-row weighted gather per output point with bias fused, several points per lane
//     in flight.
// Algorithmic bytes: splat 4*C*N + 32*N + 4*(C+1)*H; slice
// 4*C*H + 32*N + 4*C*N.
#include "common.h"

using namespace hpl;

namespace {

inline int64_t round_up_to(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
constexpr int XCD_CHUNK = 64;     // swept 16 / 64 / 256 on the box
// grid size for `blocks` logical blocks and the chunk of the XCD mapping (0 = identity for small grids)
inline int xcd_grid(int64_t blocks, int *chunk) {
    *chunk = blocks >= 8 * XCD_CHUNK ? XCD_CHUNK : 0;
    return (int)(*chunk ? round_up_to(blocks, 8 * XCD_CHUNK) : blocks);
}

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

// Block b runs on XCD b % 8 (private L2).  Chunks of `chunk` consecutive logical blocks go to one
// XCD (neighbouring points share lattice rows), the 8 XCDs work on 8 adjacent chunks.
// The grid is a multiple of 8 * chunk.
__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t chunk) {
    if (chunk == 0) return b;
    const uint32_t q = b >> 3;
    return ((q / chunk) * 8u + (b & 7u)) * chunk + q % chunk;
}

// V = float4 (vector path) or float (scalar path); CV = number of V columns per row.
// One lane per (vertex, V column): a wave covers 64/CV consecutive vertices, every lane walks its
// vertex's contributor list in CSR order (deterministic, no shuffles, no idle lanes when CV is not
// a power of two -- the Down layers have CV = 17).  Lanes of one vertex read the same csr words
// (one L1 broadcast); the feature rows are read as CV consecutive 16-byte words.
// One round of the contributor loop: entries [j, min(j + R, e)) of a vertex, all predicated -- the index / weight words of a round are
// independent loads, then its (up to) R row words are: three dependent loads deep for segments of up to R contributors.  (A scalar tail
// loop over the last 1-3 contributors was two more dependent loads per contributor: the common case on the fine levels, ~3 contributors
// per vertex.)  Absent slots add w = 0 times 0: the sum and its order are those of the plain loop.
template <typename V, int R>
__device__ __forceinline__ void splat_round(V &acc, const float *__restrict__ col, int64_t ldf, const int32_t *__restrict__ csr_pt,
                                            const float *__restrict__ csr_w, int32_t j, int32_t e) {
    using ops = vec_ops<V>;
    int32_t pt[R];
    float w[R];
    V x[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const bool ok = j + u < e;
        pt[u] = ok ? csr_pt[j + u] : -1;
        w[u] = ok ? csr_w[j + u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < R; ++u) x[u] = pt[u] >= 0 ? *reinterpret_cast<const V *>(col + (int64_t)pt[u] * ldf) : ops::zero();
#pragma unroll
    for (int u = 0; u < R; ++u) ops::fma(acc, w[u], x[u]);
}

// V = float4 (vector path) or float (scalar path); CV = number of V columns per row.
// Sparse segments (fine levels: 1.3-3 contributors per vertex): one lane per (vertex, V column) -- a wave covers 64/CV consecutive
// vertices, every lane walks its vertex's contributor list in CSR order (no shuffles, no idle lanes when CV is not a power of two --
// the Down layers have CV = 17).  Lanes of one vertex read the same csr words (one L1 broadcast); the feature rows are read as CV
// consecutive 16-byte words.
// Long segments (level 2 on: 10-21 contributors per vertex, csr_ptr[H] >= 6 H -- decided on the device, the host does not know the
// entry count): G lane groups per vertex (G * CV lanes, 256 / (G * CV) vertices per workgroup pass) take rounds of R / 2 entries in
// turn, group g the rounds g, g + G, ...; the partial sums of groups 1 .. G-1 meet in LDS and are added in group order.  Deterministic
// (a fixed order per vertex, the same as the plain loop for segments of up to R / 2 entries); three times the lanes in flight on the
// levels where a lane's chain of dependent rounds set the time (level 2 of one N = 8 192 cloud: 12.9 -> 10.0 us; in a forward the eight
// splats of levels 3-6 7-12 -> 5-9 us each; launches of 2^19 lanes or more keep one group: splat_launch).
template <typename V, int R>
__global__ void __launch_bounds__(256) k_splat(const float *__restrict__ feat, int64_t ldf, uint32_t CV,
                                               const int32_t *__restrict__ csr_ptr,
                                               const int32_t *__restrict__ csr_pt,
                                               const float *__restrict__ csr_w,
                                               const float *__restrict__ norm, uint32_t total,
                                               float *__restrict__ out, int64_t ldo, int xcd, int accumulate, uint32_t H, uint32_t G) {
    using ops = vec_ops<V>;
    constexpr int VW = sizeof(V) / 4;
    constexpr int RG = R / 2;
    __shared__ V part[256];
    const uint32_t lb = xcd_block(blockIdx.x, (uint32_t)xcd);
    const bool grouped = G > 1 && (uint32_t)csr_ptr[H] >= 6u * H;        // (uniform)
    if (grouped) {
        const uint32_t unit = CV * G, VB = 256u / unit;
        const uint32_t vb = threadIdx.x / unit, rem = threadIdx.x - vb * unit;
        const uint32_t g = rem / CV, cq = rem - g * CV;
        const uint32_t batches = (H + VB - 1) / VB;
        const float *col = feat + (int64_t)cq * VW;
        for (uint32_t bt = lb; bt < batches; bt += gridDim.x) {
            const uint32_t v = bt * VB + vb;
            const bool valid = vb < VB && v < H;
            V acc = ops::zero();
            if (valid) {
                const int32_t b = csr_ptr[v], e = csr_ptr[v + 1];
                // (rounds of RG = R / 2 entries: the lane groups multiply the loads in flight, and this path's extra indices must not
                // cost the sparse path -- the register count of the kernel is that of its larger branch -- its eighth wave per SIMD)
                for (int32_t j = b + (int32_t)g * RG; j < e; j += (int32_t)G * RG) splat_round<V, RG>(acc, col, ldf, csr_pt, csr_w, j, e);
                if (g > 0) part[(vb * (G - 1) + g - 1) * CV + cq] = acc;
            }
            __syncthreads();
            if (valid && g == 0) {
                for (uint32_t gg = 1; gg < G; ++gg) acc = ops::add(acc, part[(vb * (G - 1) + gg - 1) * CV + cq]);
                acc = ops::scale(acc, norm ? norm[v] : 1.0f);
                V *dst = reinterpret_cast<V *>(out + (int64_t)v * ldo + (int64_t)cq * VW);
                *dst = accumulate ? ops::add(*dst, acc) : acc;
            }
            __syncthreads();
        }
        return;
    }
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t idx = lb * 256u + threadIdx.x; idx < total; idx += stride) {
        const uint32_t v = idx / CV, cq = idx - v * CV;
        const int32_t b = csr_ptr[v], e = csr_ptr[v + 1];
        const float sc = norm ? norm[v] : 1.0f;
        const float *col = feat + (int64_t)cq * VW;
        V acc = ops::zero();
        for (int32_t j = b; j < e; j += R) splat_round<V, R>(acc, col, ldf, csr_pt, csr_w, j, e);
        V *dst = reinterpret_cast<V *>(out + (int64_t)v * ldo + (int64_t)cq * VW);
        acc = ops::scale(acc, sc);
        *dst = accumulate ? ops::add(*dst, acc) : acc;
    }
}

// One lane per (point, V column), U items per lane in flight (items idx, idx+S, ...): the index /
// weight words of all U items are loaded first, then all 4*U gathered rows, then the stores.
template <typename V, int U>
__global__ void __launch_bounds__(256) k_slice(const float *__restrict__ Y, int64_t ldy, uint32_t CV,
                                               const float *__restrict__ bary,
                                               const int32_t *__restrict__ off, int64_t N,
                                               const float *__restrict__ vscale,
                                               const float *__restrict__ bias, uint32_t total,
                                               float *__restrict__ out, int64_t ldo, int xcd, int accumulate) {
    using ops = vec_ops<V>;
    constexpr int VW = sizeof(V) / 4;
    const uint32_t S = gridDim.x * 256u;
    const uint32_t lb = xcd_block(blockIdx.x, (uint32_t)xcd);
    for (uint32_t base = lb * 256u + threadIdx.x; base < total; base += S * U) {
        uint32_t n[U], cq[U];
        int32_t v[U][4];
        float w[U][4];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t idx = base + u * S;
            live[u] = idx < total;
            n[u] = live[u] ? idx / CV : 0u;
            cq[u] = live[u] ? idx - n[u] * CV : 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[u][r] = live[u] ? off[(int64_t)r * N + n[u]] : -1;
                w[u][r] = bary[(int64_t)r * N + n[u]];
            }
        }
        if (vscale) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (v[u][r] >= 0) w[u][r] *= vscale[v[u][r]];
        }
        V y[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                y[u][r] = v[u][r] >= 0
                              ? *reinterpret_cast<const V *>(Y + (int64_t)v[u][r] * ldy + (int64_t)cq[u] * VW)
                              : ops::zero();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            V acc = ops::zero();
#pragma unroll
            for (int r = 0; r < 4; ++r) ops::fma(acc, w[u][r], y[u][r]);
            if (bias) acc = ops::add(acc, *reinterpret_cast<const V *>(bias + (int64_t)cq[u] * VW));
            V *dst = reinterpret_cast<V *>(out + (int64_t)n[u] * ldo + (int64_t)cq[u] * VW);
            *dst = accumulate ? ops::add(*dst, acc) : acc;
        }
    }
}


// Patch correlation from per-tap projections (hpl_gather_sum): one lane per (row m, 4 output columns); the K gathered 16-byte words
// of a lane are independent loads (the K indices of a row are read first), the NV lanes of a row read one contiguous run of
// N floats of Z.  Nothing is reused between rows, so there is no LDS stage.
template <int KMAX>
__global__ void __launch_bounds__(256) k_gather_sum(const float *__restrict__ Z, int64_t ldz, const int32_t *__restrict__ nbr,
                                                    int64_t nbr_stride, int64_t M, int K, int NV, int col_step, const float *__restrict__ bias,
                                                    const float *__restrict__ res, int64_t ldres, int64_t res_mod, int act, float slope,
                                                    float *__restrict__ Y, int64_t ldy) {
    const int64_t total = M * NV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / NV;
        const int c = (int)(i - m * NV) * 4;
        int32_t v[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) v[k] = k < K ? nbr[(int64_t)k * nbr_stride + m] : -1;
        float4 x[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            x[k] = v[k] >= 0 ? *reinterpret_cast<const float4 *>(Z + (int64_t)v[k] * ldz + (int64_t)k * col_step + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { acc.x += x[k].x; acc.y += x[k].y; acc.z += x[k].z; acc.w += x[k].w; }
        if (res) {
            const float4 r = *reinterpret_cast<const float4 *>(res + (m % res_mod) * ldres + c);
            acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
        }
        if (bias) {
            const float4 b = *reinterpret_cast<const float4 *>(bias + c);
            acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
        }
        if (act == HPL_ACT_LEAKY) {
            acc.x = acc.x > 0.f ? acc.x : slope * acc.x; acc.y = acc.y > 0.f ? acc.y : slope * acc.y;
            acc.z = acc.z > 0.f ? acc.z : slope * acc.z; acc.w = acc.w > 0.f ? acc.w : slope * acc.w;
        }
        *reinterpret_cast<float4 *>(Y + m * ldy + c) = acc;
    }
}

// inverse of a [K][F*H0] table whose row blocks m / H0 = f are injective maps h -> v (the pc2 table of the patch correlation):
// inv[f][v*K + k] = m for T[k][m] = v >= 0; inv pre-filled with -1 by the caller
__global__ void __launch_bounds__(256) k_table_invert(const int32_t *__restrict__ T, int64_t stride, int K, int64_t M, int64_t H0,
                                                      int64_t H1, int32_t *__restrict__ inv) {
    const int64_t total = (int64_t)K * M;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t k = i / M, m = i - k * M;
        const int32_t v = T[k * stride + m];
        if (v >= 0) inv[(m / H0) * ((int64_t)K * H1) + (int64_t)v * K + k] = (int32_t)m;
    }
}

}  // namespace

namespace {
int splat_launch(const float *feat, int64_t ldf, int C, const int32_t *csr_ptr, const int32_t *csr_pt, const float *csr_w,
                 const float *norm, int64_t H, float *out, int64_t ldo, hplStream stream, int accumulate) {
    HPL_REQUIRE(feat && csr_ptr && csr_pt && csr_w && out, "hpl_splat: null pointer");
    HPL_REQUIRE(C > 0 && H >= 0 && ldf >= C && ldo >= C, "hpl_splat: bad sizes C=%d H=%lld ldf=%lld ldo=%lld", C,
                (long long)H, (long long)ldf, (long long)ldo);
    if (H == 0) return HPL_OK;
    hipStream_t s = to_stream(stream);
    const bool vec = (C % 4 == 0) && (ldf % 4 == 0) && (ldo % 4 == 0) && aligned16(feat) && aligned16(out);
    const int cv = vec ? C / 4 : C;
    HPL_REQUIRE((int64_t)H * cv < (int64_t)1 << 31, "hpl_splat: H*C too large (%lld x %d)", (long long)H, C);
    const uint32_t total = (uint32_t)(H * cv);
    // lane groups per vertex for long segments: the G <= 4 that fills most of the 256 lanes (CV = 17: G = 3, five vertices per pass)
    uint32_t G = 1;
    for (uint32_t g = 2, best = 0; g <= 4; ++g) {
        const uint32_t used = (256u / ((uint32_t)cv * g)) * (uint32_t)cv * g;
        if ((uint32_t)cv * g <= 256u && used >= best) { best = used; G = g; }
    }
    // (measured, tools/bench_splat_slice.py: level 2 of one N = 8 192 cloud 12.9 -> 10.0 us; sixteen such clouds in one launch, 2.5 M lanes,
    // 76 -> 80 us -- a launch that fills the GPU several times over gains nothing from more lanes and pays the two barriers)
    if (total >= (1u << 19)) G = 1;
    // The grid covers the form with MORE workgroups (the host does not know which one the device will take): one pass of 256 / (G CV)
    // vertices per workgroup in the grouped form -- with the plain form's grid every workgroup ran G passes one after the other, G times
    // the chain of dependent loads, and the coarse levels' splats took 17 us instead of 11 --; in the plain form the surplus
    // workgroups find idx >= total and leave.
    int64_t blocks = cdiv((int64_t)total, 256);
    if (G > 1) blocks = imax(blocks, cdiv(H, (int64_t)(256u / ((uint32_t)cv * G))));
    int chunk;
    const int grid = xcd_grid(imin(blocks, 256 * 32), &chunk);
    if (vec) k_splat<float4, 8><<<grid, 256, 0, s>>>(feat, ldf, (uint32_t)cv, csr_ptr, csr_pt, csr_w, norm, total, out, ldo, chunk, accumulate, (uint32_t)H, G);
    else k_splat<float, 8><<<grid, 256, 0, s>>>(feat, ldf, (uint32_t)cv, csr_ptr, csr_pt, csr_w, norm, total, out, ldo, chunk, accumulate, (uint32_t)H, G);
    HPL_CHECK_LAUNCH("hpl_splat");
    return HPL_OK;
}
}  // namespace

extern "C" int hpl_splat(const float *feat, int64_t ldf, int C, const int32_t *csr_ptr, const int32_t *csr_pt,
                         const float *csr_w, const float *norm, int64_t H, float *out, int64_t ldo, hplStream stream) {
    return splat_launch(feat, ldf, C, csr_ptr, csr_pt, csr_w, norm, H, out, ldo, stream, 0);
}
extern "C" int hpl_splat_add(const float *feat, int64_t ldf, int C, const int32_t *csr_ptr, const int32_t *csr_pt,
                             const float *csr_w, const float *norm, int64_t H, float *out, int64_t ldo, hplStream stream) {
    return splat_launch(feat, ldf, C, csr_ptr, csr_pt, csr_w, norm, H, out, ldo, stream, 1);
}

namespace {
int slice_launch(const float *Y, int64_t ldy, int C, const float *bary, const int32_t *off, int64_t N, const float *vscale,
                 const float *bias, float *out, int64_t ldo, hplStream stream, int accumulate) {
    HPL_REQUIRE(Y && bary && off && out, "hpl_slice: null pointer");
    HPL_REQUIRE(C > 0 && N >= 0 && ldy >= C && ldo >= C, "hpl_slice: bad sizes C=%d N=%lld ldy=%lld ldo=%lld", C,
                (long long)N, (long long)ldy, (long long)ldo);
    if (N == 0) return HPL_OK;
    hipStream_t s = to_stream(stream);
    const bool vec = (C % 4 == 0) && (ldy % 4 == 0) && (ldo % 4 == 0) && aligned16(Y) && aligned16(out) &&
                     (!bias || aligned16(bias));
    const int cv = vec ? C / 4 : C;
    HPL_REQUIRE((int64_t)N * cv < (int64_t)1 << 31, "hpl_slice: N*C too large (%lld x %d)", (long long)N, C);
    const uint32_t total = (uint32_t)(N * cv);
    // no grid-stride loop in practice: in-order block dispatch keeps the streams sequential (measured:
    // 4.6 -> 5.6 TB/s on the largest slice)
    int chunk;
    if (vec) {
        constexpr int U = 2;
        const int grid = xcd_grid(cdiv((int64_t)total, 256 * U), &chunk);
        k_slice<float4, U><<<grid, 256, 0, s>>>(Y, ldy, (uint32_t)cv, bary, off, N, vscale, bias, total, out, ldo, chunk, accumulate);
    } else {
        constexpr int U = 4;
        const int grid = xcd_grid(cdiv((int64_t)total, 256 * U), &chunk);
        k_slice<float, U><<<grid, 256, 0, s>>>(Y, ldy, (uint32_t)cv, bary, off, N, vscale, bias, total, out, ldo, chunk, accumulate);
    }
    HPL_CHECK_LAUNCH("hpl_slice");
    return HPL_OK;
}
}  // namespace

extern "C" int hpl_slice(const float *Y, int64_t ldy, int C, const float *bary, const int32_t *off, int64_t N,
                         const float *vscale, const float *bias, float *out, int64_t ldo, hplStream stream) {
    return slice_launch(Y, ldy, C, bary, off, N, vscale, bias, out, ldo, stream, 0);
}
extern "C" int hpl_slice_add(const float *Y, int64_t ldy, int C, const float *bary, const int32_t *off, int64_t N,
                             const float *vscale, const float *bias, float *out, int64_t ldo, hplStream stream) {
    return slice_launch(Y, ldy, C, bary, off, N, vscale, bias, out, ldo, stream, 1);
}


extern "C" int hpl_gather_sum(const float *Z, int64_t ldz, const int32_t *nbr, int64_t nbr_stride, int64_t M, int K, int N,
                              int col_step, const float *bias, const float *res, int64_t ldres, int64_t res_mod, int act, float slope,
                              float *Y, int64_t ldy, hplStream stream) {
    HPL_REQUIRE(Z && nbr && Y, "hpl_gather_sum: null pointer");
    HPL_REQUIRE(M >= 0 && K >= 1 && K <= 15 && N > 0 && N % 4 == 0 && col_step >= 0 && col_step % 4 == 0 && ldz >= (int64_t)(K - 1) * col_step + N &&
                    ldz % 4 == 0 && ldy >= N && ldy % 4 == 0 && aligned16(Z) && aligned16(Y) && (!bias || aligned16(bias)) &&
                    (!res || (aligned16(res) && ldres % 4 == 0 && ldres >= N && res_mod > 0)),
                "hpl_gather_sum: bad sizes / alignment (M=%lld K=%d N=%d ldz=%lld)", (long long)M, K, N, (long long)ldz);
    if (M == 0) return HPL_OK;
    const int nv = N / 4;
    const int grid = (int)imin(cdiv(M * nv, 256), 1 << 20);
    k_gather_sum<15><<<grid, 256, 0, to_stream(stream)>>>(Z, ldz, nbr, nbr_stride, M, K, nv, col_step, bias, res, ldres, res ? res_mod : 1, act,
                                                          slope, Y, ldy);
    HPL_CHECK_LAUNCH("hpl_gather_sum");
    return HPL_OK;
}

extern "C" int hpl_table_invert(const int32_t *T, int64_t stride, int K, int64_t H0, int F, int64_t H1, int32_t *inv, hplStream stream) {
    HPL_REQUIRE(T && inv && K > 0 && F > 0 && H0 > 0 && H1 > 0 && stride >= (int64_t)F * H0, "hpl_table_invert: bad arguments");
    hipStream_t s = to_stream(stream);
    if (hipMemsetAsync(inv, 0xff, (size_t)F * K * H1 * 4, s) != hipSuccess) { set_error("hpl_table_invert: hipMemsetAsync failed"); return HPL_EHIP; }
    const int64_t M = (int64_t)F * H0;
    k_table_invert<<<(int)imin(cdiv((int64_t)K * M, 256), 1 << 16), 256, 0, s>>>(T, stride, K, M, H0, H1, inv);
    HPL_CHECK_LAUNCH("hpl_table_invert");
    return HPL_OK;
}

