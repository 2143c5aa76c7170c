// row_order.hip -- the row order of the tap-skipping gather-GEMM launches (hpl_tap_order, hpl_tap_order_keyed).
//
// A launch with a row order processes its M output rows in tiles of consecutive perm[] entries.  Two properties of
// the order matter to the consumer (csrc/gconv.hip):
//   1. rows of one tile should miss the SAME taps (a contraction slice whose taps are absent for the whole tile is
//      skipped): the primary sort key is the F-bit tap-presence mask (its rank in Gray-code order: see k_order_keys);
//   2. rows that are close in the order should gather rows that are close in memory use: inside a mask group the
//      rows follow the Morton code of their lattice key, so that a tile is a spatially compact set of vertices and
//      consecutive tiles of a group gather overlapping neighbour rows (the <= 15 users of a source row meet in one
//      L2 within a few tiles instead of being spread over the launch).
// The sort is a stable LSD radix sort (rocPRIM) on the 64-bit key (mask << 48 | morton48), values = row ids in
// ascending order: the result is deterministic -- tile membership no longer depends on the order in which atomics
// retire, as it did in rounds 1-2 -- and equal keys (the same lattice key in both clouds of a pair) keep ascending
// row ids.  Without keys (a lattice that arrived in the reference's wire format) the order is (mask, row id).
#include "common.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

using namespace hpl;

namespace {

// spread the low 16 bits of x so that bit i lands at bit 3*i
__device__ __forceinline__ unsigned long long spread3(unsigned x) {
    unsigned long long v = x & 0xffffu;
    v = (v | (v << 16)) & 0x0000ff0000ffull;
    v = (v | (v << 8)) & 0x00f00f00f00full;
    v = (v | (v << 4)) & 0x0c30c30c30c3ull;
    v = (v | (v << 2)) & 0x249249249249ull;
    return v;
}

struct KeySrc {
    const int32_t *vk0; int64_t vs0; int64_t H0;      // rows [0, H0): vertex m of cloud 0, key coordinate j at vk0[j*vs0 + m]
    const int32_t *vk1; int64_t vs1;                  // rows [H0, M): vertex m - H0 of cloud 1
};

// per-coordinate minimum of the first three key coordinates over the rows (a lattice key sums to zero, three
// coordinates determine it); mins[3] zero-initialised to INT_MAX by k_order_init
__global__ void k_order_init(int32_t *mins) {
    if (threadIdx.x < 4) mins[threadIdx.x] = 0x7fffffff;
}

__global__ void __launch_bounds__(256) k_order_mins(const KeySrc ks, int64_t M, int32_t *mins) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int v[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
    if (m < M) {
        const bool c0 = m < ks.H0;
        const int32_t *vk = c0 ? ks.vk0 : ks.vk1;
        const int64_t vs = c0 ? ks.vs0 : ks.vs1, i = c0 ? m : m - ks.H0;
#pragma unroll
        for (int j = 0; j < 3; ++j) v[j] = vk[(int64_t)j * vs + i];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int x = v[j];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) x = min(x, __shfl_xor(x, o, 64));
        if ((threadIdx.x & 63) == 0 && x != 0x7fffffff) atomicMin(&mins[j], x);      // integer atomics: deterministic
    }
}

__global__ void __launch_bounds__(256) k_order_keys(const int32_t *__restrict__ nbr, int64_t stride, int F, int64_t M,
                                                    const KeySrc ks, const int32_t *__restrict__ mins,
                                                    unsigned long long *__restrict__ key, int32_t *__restrict__ val) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    unsigned long long mask = 0;
    for (int f = 0; f < F; ++f) mask |= (nbr[(int64_t)f * stride + m] >= 0) ? (1ull << f) : 0ull;
    unsigned long long mort = 0;
    if (ks.vk0) {
        const bool c0 = m < ks.H0;
        const int32_t *vk = c0 ? ks.vk0 : ks.vk1;
        const int64_t vs = c0 ? ks.vs0 : ks.vs1, i = c0 ? m : m - ks.H0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned d = (unsigned)(vk[(int64_t)j * vs + i] - mins[j]);        // >= 0; clamped to 16 bits
            mort |= spread3(d > 0xffffu ? 0xffffu : d) << j;
        }
    }
    // Groups in GRAY-CODE order of their masks (sort key = the mask's position in the reflected Gray sequence): masks of
    // neighbouring groups then differ in one tap, where integer order puts 0111 next to 1000.  Tiles and 32-row blocks that
    // straddle group boundaries (most groups are smaller than a tile) unite fewer taps: on the N=8192 frustum the
    // slices a 128-row tile loads drop by 3 %, the MFMA work of its 32-row blocks by 2 % (tools: DESIGN.md 4.1).
    unsigned long long rank = mask;
    rank ^= rank >> 1; rank ^= rank >> 2; rank ^= rank >> 4; rank ^= rank >> 8;
    key[m] = (rank << 48) | mort;
    val[m] = (int32_t)m;
}

inline int64_t align256(int64_t b) { return (b + 255) / 256 * 256; }

int64_t sort_temp_bytes(int64_t M) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                    (const int32_t *)nullptr, (int32_t *)nullptr, (unsigned)M, 0, 63, (hipStream_t) nullptr);
    return (int64_t)bytes;
}

int order_impl(const int32_t *nbr, int64_t nbr_stride, int F, int64_t M, const KeySrc &ks, int32_t *perm,
               int32_t *scratch, hipStream_t s, const char *who) {
    // scratch: keys in | keys out | values in | 4 mins | rocPRIM temporaries   (hpl_tap_order_scratch_ints)
    char *p = reinterpret_cast<char *>(scratch);
    unsigned long long *kin = reinterpret_cast<unsigned long long *>(p); p += align256(M * 8);
    unsigned long long *kout = reinterpret_cast<unsigned long long *>(p); p += align256(M * 8);
    int32_t *vin = reinterpret_cast<int32_t *>(p); p += align256(M * 4);
    int32_t *mins = reinterpret_cast<int32_t *>(p); p += 256;
    const int grid = (int)cdiv(M, 256);
    if (ks.vk0) {
        k_order_init<<<1, 64, 0, s>>>(mins);
        k_order_mins<<<grid, 256, 0, s>>>(ks, M, mins);
    }
    k_order_keys<<<grid, 256, 0, s>>>(nbr, nbr_stride, F, M, ks, mins, kin, vin);
    size_t tb = (size_t)sort_temp_bytes(M);
    // bits [0, 48) Morton code (all zero without keys: skipped), [48, 48 + F) tap mask
    const unsigned lo = ks.vk0 ? 0u : 48u;
    const hipError_t e = rocprim::radix_sort_pairs(p, tb, kin, kout, vin, perm, (unsigned)M, lo, 48u + (unsigned)F, s);
    if (e != hipSuccess) { set_error("%s: radix sort failed: %s", who, hipGetErrorString(e)); return HPL_EHIP; }
    HPL_CHECK_LAUNCH(who);
    return HPL_OK;
}

}  // namespace

extern "C" int64_t hpl_tap_order_scratch_ints(int64_t M) {
    if (M <= 0) return 0;
    const int64_t bytes = 2 * align256(M * 8) + align256(M * 4) + 256 + align256(sort_temp_bytes(M)) + 256;
    return bytes / 4;
}

extern "C" int hpl_tap_order(const int32_t *nbr, int64_t nbr_stride, int F, int64_t M, int32_t *perm,
                             int32_t *scratch, hplStream stream) {
    HPL_REQUIRE(nbr && perm && scratch && F >= 1 && F <= 15 && M > 0 && M < (int64_t)INT32_MAX,
                "hpl_tap_order: bad arguments (F=%d M=%lld)", F, (long long)M);
    HPL_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 7u) == 0, "hpl_tap_order: scratch must be 8-byte aligned");
    const KeySrc ks = {nullptr, 0, 0, nullptr, 0};
    return order_impl(nbr, nbr_stride, F, M, ks, perm, scratch, to_stream(stream), "hpl_tap_order");
}

extern "C" int hpl_tap_order_keyed(const int32_t *nbr, int64_t nbr_stride, int F, int64_t M, const int32_t *vkeys0,
                                   int64_t vstride0, int64_t H0, const int32_t *vkeys1, int64_t vstride1,
                                   int32_t *perm, int32_t *scratch, hplStream stream) {
    HPL_REQUIRE(nbr && perm && scratch && F >= 1 && F <= 15 && M > 0 && M < (int64_t)INT32_MAX,
                "hpl_tap_order_keyed: bad arguments (F=%d M=%lld)", F, (long long)M);
    HPL_REQUIRE(vkeys0 && vstride0 >= imin(H0, M) && H0 >= 0 && (M <= H0 || (vkeys1 && vstride1 >= M - H0)),
                "hpl_tap_order_keyed: the key arrays do not cover the %lld rows", (long long)M);
    HPL_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 7u) == 0, "hpl_tap_order_keyed: scratch must be 8-byte aligned");
    const KeySrc ks = {vkeys0, vstride0, H0, vkeys1, vstride1};
    return order_impl(nbr, nbr_stride, F, M, ks, perm, scratch, to_stream(stream), "hpl_tap_order_keyed");
}
