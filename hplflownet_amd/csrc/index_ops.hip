// index_ops.hip -- library plumbing and the small integer/element-wise kernels:
// table narrowing, corr-table permutation, the splat CSR build, transpose, column sums,
// LeakyReLU backward.  All HBM-bound; written for 64-wide waves, 256-thread groups.
#include "common.h"
#include "gconv_common.h"

#include <string.h>

namespace hpl {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace hpl

using namespace hpl;

extern "C" int hpl_version(void) { return 100; }
extern "C" const char *hpl_last_error(void) { return hpl::g_err; }

extern "C" int hpl_device_info(int device, int *cu_count, int *wave_size, char *arch, int arch_len) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) {
        set_error("hpl_device_info: %s", hipGetErrorString(e));
        return HPL_ENODEV;
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (wave_size) *wave_size = p.warpSize;
    if (arch && arch_len > 0) {
        strncpy(arch, p.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return HPL_OK;
}

// ---------------------------------------------------------------- narrowing
__global__ void k_narrow(const int64_t *__restrict__ src, int32_t *__restrict__ dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (int32_t)src[i];
}

extern "C" int hpl_index_narrow(const int64_t *src, int32_t *dst, int64_t n, hplStream stream) {
    if (n == 0) return HPL_OK;
    HPL_REQUIRE(src && dst && n > 0, "hpl_index_narrow: null pointer or negative size");
    int grid = (int)imin(cdiv(n, 256), 2048);
    k_narrow<<<grid, 256, 0, to_stream(stream)>>>(src, dst, n);
    HPL_CHECK_LAUNCH("hpl_index_narrow");
    return HPL_OK;
}

// src [F][K][H] -> dst [K][F*H]
template <typename T>
__global__ void k_corr2_permute(const T *__restrict__ src, int32_t *__restrict__ dst, int F, int K,
                                int64_t H) {
    int64_t total = (int64_t)F * K * H;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {   // i indexes dst: (k, f, h)
        int64_t h = i % H;
        int64_t t = i / H;
        int f = (int)(t % F);
        int k = (int)(t / F);
        dst[i] = (int32_t)src[((int64_t)f * K + k) * H + h];
    }
}

extern "C" int hpl_corr2_permute(const int64_t *src, int32_t *dst, int F, int K, int64_t H,
                                 hplStream stream) {
    HPL_REQUIRE(src && dst && F > 0 && K > 0 && H >= 0, "hpl_corr2_permute: bad arguments");
    if (H == 0) return HPL_OK;
    int grid = (int)imin(cdiv((int64_t)F * K * H, 256), 4096);
    k_corr2_permute<int64_t><<<grid, 256, 0, to_stream(stream)>>>(src, dst, F, K, H);
    HPL_CHECK_LAUNCH("hpl_corr2_permute");
    return HPL_OK;
}

extern "C" int hpl_corr2_permute32(const int32_t *src, int32_t *dst, int F, int K, int64_t H,
                                   hplStream stream) {
    HPL_REQUIRE(src && dst && F > 0 && K > 0 && H >= 0, "hpl_corr2_permute32: bad arguments");
    if (H == 0) return HPL_OK;
    int grid = (int)imin(cdiv((int64_t)F * K * H, 256), 4096);
    k_corr2_permute<int32_t><<<grid, 256, 0, to_stream(stream)>>>(src, dst, F, K, H);
    HPL_CHECK_LAUNCH("hpl_corr2_permute32");
    return HPL_OK;
}

namespace hpl {
int exclusive_scan_i32(const int32_t *cnt, int64_t n, int32_t *ptr, int32_t *tmp, hipStream_t s);
}

// ---------------------------------------------------------------- CSR build
__global__ void k_zero_i32(int32_t *p, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = 0;
}

// Entries of one cloud (e = r*N + n over its lattice_offset / barycentric tables) or of a pair of
// clouds laid end to end: entries of cloud 1 follow those of cloud 0, its vertices and points are
// renumbered behind cloud 0's (v + H0, n + N0).
struct Seg2 {
    const int32_t *off0; const float *w0; int64_t ne0, n0;
    const int32_t *off1; const float *w1; int64_t ne1, n1;
    int32_t h0;          // vertices of cloud 0 (shift of cloud 1)
    __device__ __forceinline__ int32_t vertex(int64_t e, int64_t H) const {
        if (e < ne0) { const int32_t v = off0[e]; return (v >= 0 && v < H) ? v : -1; }
        const int32_t v = off1[e - ne0];
        return (v >= 0 && v + h0 < H) ? v + h0 : -1;
    }
    __device__ __forceinline__ int32_t point(int64_t e) const {
        return e < ne0 ? (int32_t)(e % n0) : (int32_t)(n0 + (e - ne0) % n1);
    }
    __device__ __forceinline__ float weight(int64_t e) const { return e < ne0 ? w0[e] : w1[e - ne0]; }
};

__global__ void k_csr_count(const Seg2 sg, int64_t n_entries, int64_t H, int32_t *__restrict__ cnt) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n_entries; i += stride) {
        const int32_t v = sg.vertex(i, H);
        if (v >= 0) atomicAdd(&cnt[v], 1);   // integer atomics: deterministic counts
    }
}

// Exclusive scan of cnt[0..n) into ptr[0..n], ptr[n] = total, in three launches:
// 1024-element block sums -> scan of the (<= 1024) block sums -> per-block scan + offset.
constexpr int SCAN_BLOCK = 1024;   // elements per 256-thread workgroup

__device__ __forceinline__ int block_exclusive_scan_256(int v, int *total) {
    // exclusive scan of one int per thread over a 256-thread workgroup
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) base += (i < w) ? wsum[i] : 0;
    if (total) *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return base + inc - v;
}

__global__ void __launch_bounds__(256) k_scan_sums(const int32_t *__restrict__ in, int64_t n,
                                                   int32_t *__restrict__ block_sums) {
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x * 4;
    int s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += (i0 + j < n) ? in[i0 + j] : 0;
    int total;
    block_exclusive_scan_256(s, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// base: where the running total of the chunks before this one stands (nullptr: 0).  It is the
// previous chunk's out[n], i.e. this chunk's out[0], which is rewritten with the same value.
__global__ void __launch_bounds__(256) k_scan_final(const int32_t *__restrict__ in, int64_t n,
                                                    const int32_t *__restrict__ block_sums, int nb,
                                                    int32_t *out, const int32_t *base) {
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x * 4;
    int v[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = (i0 + j < n) ? in[i0 + j] : 0; s += v[j]; }
    // offset of this block = sum of the block sums before it (<= 1024 of them: summed here, in
    // every block, instead of a separate single-block scan launch)
    __shared__ int boff_s;
    {
        int part = 0;
        for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) part += block_sums[b];
        int tot;
        block_exclusive_scan_256(part, &tot);
        if (threadIdx.x == 0) boff_s = tot + (base ? *base : 0);
        __syncthreads();
    }
    int run = block_exclusive_scan_256(s, nullptr) + boff_s;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (i0 + j < n) out[i0 + j] = run;
        run += v[j];
    }
    if (blockIdx.x == (unsigned)(nb - 1) && threadIdx.x == 0) out[n] = boff_s + block_sums[nb - 1];
}

__global__ void k_csr_fill(const Seg2 sg, int64_t n_entries, int64_t H,
                           const int32_t *__restrict__ ptr, int32_t *__restrict__ cursor,
                           int32_t *__restrict__ ent) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n_entries; i += stride) {
        const int32_t v = sg.vertex(i, H);
        if (v >= 0) {
            int32_t slot = atomicAdd(&cursor[v], 1);
            ent[ptr[v] + slot] = (int32_t)i;
        }
    }
}

// Segments come out of k_csr_fill in atomic (arbitrary) order.  A 16-lane group per vertex ranks
// every entry among its segment (entries are distinct, segments are short: mean 4..15), which
// fixes the summation order of the splat, and emits (point, weight) pairs in that order.
__global__ void __launch_bounds__(256) k_csr_rank(const int32_t *__restrict__ ptr, const int32_t *__restrict__ ent,
                                                  const Seg2 sg, int64_t H,
                                                  int32_t *__restrict__ pt_out, float *__restrict__ w_out) {
    const int64_t v = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int lg = threadIdx.x & 15;
    if (v >= H) return;
    const int32_t b = ptr[v], e = ptr[v + 1];
    for (int32_t i = b + lg; i < e; i += 16) {
        const int32_t x = ent[i];
        int32_t rank = 0;
        for (int32_t j = b; j < e; ++j) rank += (ent[j] < x) ? 1 : 0;
        pt_out[b + rank] = sg.point(x);
        w_out[b + rank] = sg.weight(x);
    }
}

// density normaliser 1 / (sum of weights + 1e-5), summed in segment order (models/bilateralNN.py:183)
__global__ void k_csr_norm(const int32_t *__restrict__ ptr, const float *__restrict__ w, int64_t H,
                           float *__restrict__ norm) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= H) return;
    float s = 0.f;
    for (int32_t i = ptr[v]; i < ptr[v + 1]; ++i) s += w[i];
    norm[v] = 1.0f / (s + 1e-5f);
}

namespace {
int csr_build_impl(const Seg2 &sg, int64_t H, int32_t *csr_ptr, int32_t *csr_pt, float *csr_w, float *norm,
                   int32_t *scratch, hipStream_t s, const char *who) {
    const int64_t ne = sg.ne0 + sg.ne1;
    int gh = (int)imin(cdiv(H + 1, 256), 2048);
    int ge = (int)imin(cdiv(ne, 256), 2048);
    int32_t *cursor = scratch, *ent = scratch + (H + 1), *scan_tmp = ent + ne;
    k_zero_i32<<<gh, 256, 0, s>>>(cursor, H + 1);
    k_csr_count<<<ge, 256, 0, s>>>(sg, ne, H, cursor);
    int rc = exclusive_scan_i32(cursor, H, csr_ptr, scan_tmp, s);
    if (rc != HPL_OK) return rc;
    k_zero_i32<<<gh, 256, 0, s>>>(cursor, H + 1);
    k_csr_fill<<<ge, 256, 0, s>>>(sg, ne, H, csr_ptr, cursor, ent);
    k_csr_rank<<<(int)cdiv(H * 16, 256), 256, 0, s>>>(csr_ptr, ent, sg, H, csr_pt, csr_w);
    k_csr_norm<<<(int)cdiv(H, 256), 256, 0, s>>>(csr_ptr, csr_w, H, norm);
    HPL_CHECK_LAUNCH(who);
    return HPL_OK;
}
}  // namespace

extern "C" int hpl_csr_build(const int32_t *off, const float *bary, int64_t n_entries, int64_t pt_mod,
                             int64_t H, int32_t *csr_ptr, int32_t *csr_pt, float *csr_w, float *norm,
                             int32_t *scratch, hplStream stream) {
    HPL_REQUIRE(off && bary && csr_ptr && csr_pt && csr_w && norm && scratch,
                "hpl_csr_build: null pointer");
    HPL_REQUIRE(n_entries > 0 && pt_mod > 0 && H > 0 && n_entries < (int64_t)INT32_MAX,
                "hpl_csr_build: bad sizes n_entries=%lld H=%lld", (long long)n_entries, (long long)H);
    const Seg2 sg = {off, bary, n_entries, pt_mod, nullptr, nullptr, 0, 1, 0};
    return csr_build_impl(sg, H, csr_ptr, csr_pt, csr_w, norm, scratch, to_stream(stream), "hpl_csr_build");
}

extern "C" int hpl_csr_build_pair(const int32_t *off0, const float *bary0, int64_t N0, int64_t H0,
                                  const int32_t *off1, const float *bary1, int64_t N1, int64_t H1,
                                  int32_t *csr_ptr, int32_t *csr_pt, float *csr_w, float *norm,
                                  int32_t *scratch, hplStream stream) {
    HPL_REQUIRE(off0 && bary0 && off1 && bary1 && csr_ptr && csr_pt && csr_w && norm && scratch,
                "hpl_csr_build_pair: null pointer");
    HPL_REQUIRE(N0 > 0 && N1 > 0 && H0 > 0 && H1 > 0 && 4 * (N0 + N1) < (int64_t)INT32_MAX &&
                    H0 + H1 < (int64_t)INT32_MAX,
                "hpl_csr_build_pair: bad sizes N=(%lld,%lld) H=(%lld,%lld)", (long long)N0, (long long)N1,
                (long long)H0, (long long)H1);
    const Seg2 sg = {off0, bary0, 4 * N0, N0, off1, bary1, 4 * N1, N1, (int32_t)H0};
    return csr_build_impl(sg, H0 + H1, csr_ptr, csr_pt, csr_w, norm, scratch, to_stream(stream),
                          "hpl_csr_build_pair");
}

// ---------------------------------------------------------------- transpose
// dst[m*ldd + c] = src[c*lds + m]; src is (rows_src = C) x (cols_src = M).
__global__ void k_transpose(const float *__restrict__ src, int64_t lds, float *__restrict__ dst,
                            int64_t ldd, int64_t rows, int64_t cols) {
    __shared__ float tile[32][33];
    const int64_t c0 = (int64_t)blockIdx.y * 32, m0 = (int64_t)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
    for (int r = ty; r < 32; r += 8) {
        int64_t c = c0 + r, m = m0 + tx;
        tile[r][tx] = (c < rows && m < cols) ? src[c * lds + m] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int64_t m = m0 + r, c = c0 + tx;
        if (m < cols && c < rows) dst[m * ldd + c] = tile[tx][r];
    }
}

extern "C" int hpl_transpose(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t rows_src,
                             int64_t cols_src, hplStream stream) {
    HPL_REQUIRE(src && dst && rows_src > 0 && cols_src > 0, "hpl_transpose: bad arguments");
    dim3 grid((unsigned)cdiv(cols_src, 32), (unsigned)cdiv(rows_src, 32));
    k_transpose<<<grid, 256, 0, to_stream(stream)>>>(src, lds, dst, ldd, rows_src, cols_src);
    HPL_CHECK_LAUNCH("hpl_transpose");
    return HPL_OK;
}

// ---------------------------------------------------------------- column sums
// out[n] += sum_m X[m*ld + n].  Vector form (N % 4 == 0, 16-byte aligned rows): a 256-thread group covers 256 columns (64 lanes x
// float4) x a slab of rows, 4 row lanes per column group with two rows in flight each, LDS combine, one atomicAdd per (group, column).
// Scalar form: 64 columns per group.  `out` is zeroed by the caller.
__global__ void __launch_bounds__(256) k_colsum4(const float *__restrict__ X, int64_t ld, int64_t M, int N,
                                                 int64_t rows_per_block, float *__restrict__ out) {
    __shared__ float4 part[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int n = (blockIdx.y * 64 + cx) * 4;
    const int64_t m0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t m1 = imin(M, m0 + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), t = s;
    if (n < N) {
        int64_t m = m0 + ry;
        for (; m + 4 < m1; m += 8) {
            const float4 a = *reinterpret_cast<const float4 *>(X + m * ld + n), b = *reinterpret_cast<const float4 *>(X + (m + 4) * ld + n);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
        }
        if (m < m1) {
            const float4 a = *reinterpret_cast<const float4 *>(X + m * ld + n);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    part[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && n < N) {
        const float4 a = part[0][cx], b = part[1][cx], c = part[2][cx], d = part[3][cx];
        atomicAdd(&out[n + 0], (a.x + b.x) + (c.x + d.x));
        atomicAdd(&out[n + 1], (a.y + b.y) + (c.y + d.y));
        atomicAdd(&out[n + 2], (a.z + b.z) + (c.z + d.z));
        atomicAdd(&out[n + 3], (a.w + b.w) + (c.w + d.w));
    }
}

__global__ void __launch_bounds__(256) k_colsum(const float *__restrict__ X, int64_t ld, int64_t M, int N,
                                                int64_t rows_per_block, float *__restrict__ out) {
    __shared__ float part[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int n = blockIdx.y * 64 + cx;
    const int64_t m0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t m1 = imin(M, m0 + rows_per_block);
    float s = 0.f;
    if (n < N)
        for (int64_t m = m0 + ry; m < m1; m += 4) s += X[m * ld + n];
    part[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && n < N) atomicAdd(&out[n], (part[0][cx] + part[1][cx]) + (part[2][cx] + part[3][cx]));
}

__global__ void k_zero_f32(float *p, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

namespace hpl {
void colsum_accumulate(const float *X, int64_t ld, int64_t M, int N, float *out, hipStream_t s) {
    if (M <= 0) return;
    if (N % 4 == 0 && ld % 4 == 0 && aligned16(X)) {
        // slabs of >= 64 rows, ~2 048 workgroups over the matrix (few atomics per column, enough workgroups to hide the loads)
        const int64_t cg = cdiv(N, 256);
        const int64_t rpb = imax(64, cdiv(cdiv(M, imax(1, 2048 / cg)), 8) * 8);
        k_colsum4<<<dim3((unsigned)cdiv(M, rpb), (unsigned)cg), 256, 0, s>>>(X, ld, M, N, rpb, out);
        return;
    }
    const int64_t rpb = 256;
    k_colsum<<<dim3((unsigned)cdiv(M, rpb), (unsigned)cdiv(N, 64)), 256, 0, s>>>(X, ld, M, N, rpb, out);
}
}  // namespace hpl

extern "C" int hpl_colsum(const float *X, int64_t ld, int64_t M, int N, float *out, hplStream stream) {
    HPL_REQUIRE(X && out && M >= 0 && N > 0, "hpl_colsum: bad arguments");
    hipStream_t s = to_stream(stream);
    k_zero_f32<<<(int)cdiv(N, 256), 256, 0, s>>>(out, N);
    colsum_accumulate(X, ld, M, N, out, s);
    HPL_CHECK_LAUNCH("hpl_colsum");
    return HPL_OK;
}

// ---------------------------------------------------------------- LeakyReLU backward
__global__ void k_leaky_bwd(const float *__restrict__ dY, int64_t lddy, const float *__restrict__ Y,
                            int64_t ldy, float slope, float *__restrict__ dX, int64_t lddx, int64_t M,
                            int N) {
    int64_t total = M * N;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        int64_t m = i / N;
        int n = (int)(i - m * N);
        float y = Y[m * ldy + n];
        dX[m * lddx + n] = dY[m * lddy + n] * (y > 0.f ? 1.0f : slope);
    }
}

// four columns per thread (N % 4 == 0, 16-byte aligned rows): 2^tpr_log lanes across a row, 256 >> tpr_log rows per workgroup and
// round, two rows in flight per lane, no division; amax (optional): *amax = max(*amax, largest |dX|) -- the scale the wide data /
// weight gradients that read dX need (hpl_gconv_desc.a_amax), from the values while they are in registers: ONE atomic per workgroup
__global__ void __launch_bounds__(256) k_leaky_bwd4(const float *__restrict__ dY, int64_t lddy, const float *__restrict__ Y, int64_t ldy,
                                                    float slope, float *__restrict__ dX, int64_t lddx, int64_t M, int N4, int tpr_log,
                                                    unsigned *__restrict__ amax) {
    const int tpr = 1 << tpr_log, c0 = threadIdx.x & (tpr - 1), rpb = 256 >> tpr_log;
    const int64_t rstep = (int64_t)gridDim.x * rpb;
    unsigned vmax = 0;
    auto one = [&](float4 g, const float4 y, float *dst) {
        g.x *= y.x > 0.f ? 1.0f : slope;
        g.y *= y.y > 0.f ? 1.0f : slope;
        g.z *= y.z > 0.f ? 1.0f : slope;
        g.w *= y.w > 0.f ? 1.0f : slope;
        *reinterpret_cast<float4 *>(dst) = g;
        vmax = max(max(vmax, __float_as_uint(g.x) & 0x7fffffffu), max(max(__float_as_uint(g.y) & 0x7fffffffu, __float_as_uint(g.z) & 0x7fffffffu), __float_as_uint(g.w) & 0x7fffffffu));
    };
    int64_t m = (int64_t)blockIdx.x * rpb + (threadIdx.x >> tpr_log);
    for (; m + rstep < M; m += 2 * rstep) {
        const int64_t m2 = m + rstep;
        for (int c = c0; c < N4; c += tpr) {
            const float4 y0 = *reinterpret_cast<const float4 *>(Y + m * ldy + c * 4), g0 = *reinterpret_cast<const float4 *>(dY + m * lddy + c * 4);
            const float4 y1 = *reinterpret_cast<const float4 *>(Y + m2 * ldy + c * 4), g1 = *reinterpret_cast<const float4 *>(dY + m2 * lddy + c * 4);
            one(g0, y0, dX + m * lddx + c * 4);
            one(g1, y1, dX + m2 * lddx + c * 4);
        }
    }
    if (m < M)
        for (int c = c0; c < N4; c += tpr)
            one(*reinterpret_cast<const float4 *>(dY + m * lddy + c * 4), *reinterpret_cast<const float4 *>(Y + m * ldy + c * 4), dX + m * lddx + c * 4);
    if (amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = max(vmax, (unsigned)__shfl_xor((int)vmax, o));
        __shared__ unsigned part[4];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = vmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            vmax = max(max(part[0], part[1]), max(part[2], part[3]));
            amax_publish(amax, vmax);
        }
    }
}

extern "C" int hpl_leaky_bwd(const float *dY, int64_t lddy, const float *Y, int64_t ldy, float slope,
                             float *dX, int64_t lddx, int64_t M, int N, hplStream stream) {
    return hpl_leaky_bwd_amax(dY, lddy, Y, ldy, slope, dX, lddx, M, N, nullptr, stream);
}

extern "C" int hpl_leaky_bwd_amax(const float *dY, int64_t lddy, const float *Y, int64_t ldy, float slope,
                                  float *dX, int64_t lddx, int64_t M, int N, float *amax, hplStream stream) {
    HPL_REQUIRE(dY && Y && dX && M >= 0 && N > 0, "hpl_leaky_bwd: bad arguments");
    if (M == 0) return HPL_OK;
    if (N % 4 == 0 && lddy % 4 == 0 && ldy % 4 == 0 && lddx % 4 == 0 && aligned16(dY) && aligned16(Y) && aligned16(dX)) {
        int tpr_log = 0;
        while ((1 << tpr_log) < N / 4 && tpr_log < 8) ++tpr_log;
        const int grid = (int)imax(1, imin(cdiv(M, (int64_t)(256 >> tpr_log) * 2), 2048));
        k_leaky_bwd4<<<grid, 256, 0, to_stream(stream)>>>(dY, lddy, Y, ldy, slope, dX, lddx, M, N / 4, tpr_log, reinterpret_cast<unsigned *>(amax));
        HPL_CHECK_LAUNCH("hpl_leaky_bwd");
        return HPL_OK;
    }
    int grid = (int)imin(cdiv(M * N, 256), 4096);
    k_leaky_bwd<<<grid, 256, 0, to_stream(stream)>>>(dY, lddy, Y, ldy, slope, dX, lddx, M, N);
    HPL_CHECK_LAUNCH("hpl_leaky_bwd");
    if (amax) return hpl_gc::amax_launch(dX, lddx, M, N, amax, to_stream(stream));
    return HPL_OK;
}

// shared with lattice.hip.  tmp: cdiv(n, 1024) + 1 ints.
namespace hpl {
int exclusive_scan_i32(const int32_t *cnt, int64_t n, int32_t *ptr, int32_t *tmp, hipStream_t s) {
    HPL_REQUIRE(n >= 1 && n < (int64_t)INT32_MAX, "exclusive_scan_i32: n=%lld out of range", (long long)n);
    // chunks of 1024 blocks x 1024 elements; a chunk continues from the total the previous one left in
    // ptr[chunk start] (same stream, so it is there)
    constexpr int64_t CHUNK = (int64_t)1024 * SCAN_BLOCK;
    for (int64_t c0 = 0; c0 < n; c0 += CHUNK) {
        const int64_t len = imin(CHUNK, n - c0);
        const int nb = (int)cdiv(len, SCAN_BLOCK);
        k_scan_sums<<<nb, 256, 0, s>>>(cnt + c0, len, tmp);
        k_scan_final<<<nb, 256, 0, s>>>(cnt + c0, len, tmp, nb, ptr + c0, c0 ? ptr + c0 : nullptr);
    }
    HPL_CHECK_LAUNCH("exclusive_scan_i32");
    return HPL_OK;
}
}  // namespace hpl

// ---------------------------------------------------------------- per-tap vertex lists
// list_m / list_row[tap_ptr[f] .. tap_ptr[f+1]) = the vertices m with nbr[f][m] >= 0 (ascending,
// deterministic) and their source rows nbr[f][m].
// Chunks of 1024 vertices per workgroup: count -> scan of the F x chunks counts -> fill.
constexpr int TL_CHUNK = 1024;

__global__ void __launch_bounds__(256) k_taplist_count(const int32_t *__restrict__ nbr, int64_t stride, int64_t M,
                                                       int nb, int32_t *__restrict__ cnt) {
    const int b = blockIdx.x, f = blockIdx.y;
    const int64_t m0 = (int64_t)b * TL_CHUNK + threadIdx.x * 4;
    int c = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) c += (m0 + j < M && nbr[(int64_t)f * stride + m0 + j] >= 0) ? 1 : 0;
    int total;
    block_exclusive_scan_256(c, &total);
    if (threadIdx.x == 0) cnt[f * nb + b] = total;
}

__global__ void __launch_bounds__(256) k_taplist_fill(const int32_t *__restrict__ nbr, int64_t stride, int F,
                                                      int64_t M, int nb, const int32_t *__restrict__ off,
                                                      int32_t *__restrict__ list_m, int32_t *__restrict__ list_row,
                                                      int32_t *__restrict__ tap_ptr) {
    const int b = blockIdx.x, f = blockIdx.y;
    const int64_t m0 = (int64_t)b * TL_CHUNK + threadIdx.x * 4;
    int32_t row[4];
    int c = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        row[j] = (m0 + j < M) ? nbr[(int64_t)f * stride + m0 + j] : -1;
        c += row[j] >= 0 ? 1 : 0;
    }
    int pos = off[f * nb + b] + block_exclusive_scan_256(c, nullptr);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (row[j] >= 0) {
            list_m[pos] = (int32_t)(m0 + j);
            list_row[pos] = row[j];
            ++pos;
        }
    if (b == 0 && threadIdx.x == 0) {
        tap_ptr[f] = off[f * nb];
        if (f == F - 1) tap_ptr[F] = off[F * nb];
    }
}

extern "C" int hpl_tap_lists(const int32_t *nbr, int64_t nbr_stride, int F, int64_t M, int32_t *list_m,
                             int32_t *list_row, int32_t *tap_ptr, int32_t *scratch, hplStream stream) {
    HPL_REQUIRE(nbr && list_m && list_row && tap_ptr && scratch && F >= 1 && F <= 64 && M > 0 && nbr_stride >= M &&
                    (int64_t)F * M < (int64_t)INT32_MAX,
                "hpl_tap_lists: bad arguments (F=%d M=%lld)", F, (long long)M);
    hipStream_t s = to_stream(stream);
    const int nb = (int)cdiv(M, TL_CHUNK);
    int32_t *cnt = scratch, *off = cnt + (int64_t)F * nb, *tmp = off + (int64_t)F * nb + 1;
    k_taplist_count<<<dim3(nb, F), 256, 0, s>>>(nbr, nbr_stride, M, nb, cnt);
    int rc = exclusive_scan_i32(cnt, (int64_t)F * nb, off, tmp, s);
    if (rc != HPL_OK) return rc;
    k_taplist_fill<<<dim3(nb, F), 256, 0, s>>>(nbr, nbr_stride, F, M, nb, off, list_m, list_row, tap_ptr);
    HPL_CHECK_LAUNCH("hpl_tap_lists");
    return HPL_OK;
}

// ---------------------------------------------------------------- table symmetry
// flag[0] &= (nbr[0][m] == m) and (nbr[f][m] = g >= 0  =>  g < M and nbr[F-f][g] == m) for all m, f >= 1
// (holds by construction unless an unchecked key packing aliased, A.2 quirk).
__global__ void k_table_symmetric(const int32_t *__restrict__ nbr, int64_t stride, int F, int64_t M,
                                  int32_t *__restrict__ flag) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (m >= M) return;
    const int32_t g = nbr[(int64_t)f * stride + m];
    bool ok;
    if (f == 0) ok = (g == (int32_t)m);
    else ok = (g < 0) || (g < M && nbr[(int64_t)(F - f) * stride + g] == (int32_t)m);
    if (!ok) *flag = 0;
}

extern "C" int hpl_table_symmetric(const int32_t *nbr, int64_t nbr_stride, int F, int64_t M, int32_t *flag,
                                   hplStream stream) {
    HPL_REQUIRE(nbr && flag && F >= 1 && M > 0 && nbr_stride >= M, "hpl_table_symmetric: bad arguments");
    k_table_symmetric<<<dim3((unsigned)cdiv(M, 256), F), 256, 0, to_stream(stream)>>>(nbr, nbr_stride, F, M, flag);
    HPL_CHECK_LAUNCH("hpl_table_symmetric");
    return HPL_OK;
}
