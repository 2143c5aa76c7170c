// row_order.hip -- the row order of the tap-skipping gather-GEMM launches (hpl_tap_order).
//
// A launch with a row order processes its M output rows in tiles of consecutive perm[] entries.  Rows of one tile should miss
// the SAME taps (a contraction slice whose taps are absent for the whole tile is skipped): the sort key is the F-bit
// tap-presence mask (its rank in Gray-code order: see k_order_keys), ties by ascending row id.  The sort is a stable LSD radix
// sort (rocPRIM): the result is deterministic.  (A Morton order of the lattice keys inside every mask group -- spatially compact
// tiles -- was built in round 3 and removed in round 5: the L2-miss traffic of the dominant launch did not move, 1.549 vs 1.547 GB,
// profiles/r03b_traffic_*.txt: the <= 15 users of a source row are different TAPS of its consumers, a whole tap's worth of slices
// apart in time.)
#include "common.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

using namespace hpl;

namespace {

__global__ void __launch_bounds__(256) k_order_keys(const int32_t *__restrict__ nbr, int64_t stride, int F, int64_t M,
                                                    unsigned long long *__restrict__ key, int32_t *__restrict__ val) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    unsigned long long mask = 0;
    for (int f = 0; f < F; ++f) mask |= (nbr[(int64_t)f * stride + m] >= 0) ? (1ull << f) : 0ull;
    // Groups in GRAY-CODE order of their masks (sort key = the mask's position in the reflected Gray sequence): masks of
    // neighbouring groups then differ in one tap, where integer order puts 0111 next to 1000.  Tiles and 32-row blocks that
    // straddle group boundaries (most groups are smaller than a tile) unite fewer taps: on the N=8192 frustum the
    // slices a 128-row tile loads drop by 3 %, the MFMA work of its 32-row blocks by 2 % (tools: DESIGN_HISTORY.md 4.1).
    unsigned long long rank = mask;
    rank ^= rank >> 1; rank ^= rank >> 2; rank ^= rank >> 4; rank ^= rank >> 8;
    key[m] = rank << 48;
    val[m] = (int32_t)m;
}

inline int64_t align256(int64_t b) { return (b + 255) / 256 * 256; }

int64_t sort_temp_bytes(int64_t M) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                    (const int32_t *)nullptr, (int32_t *)nullptr, (unsigned)M, 0, 63, (hipStream_t) nullptr);
    return (int64_t)bytes;
}

int order_impl(const int32_t *nbr, int64_t nbr_stride, int F, int64_t M, int32_t *perm,
               int32_t *scratch, hipStream_t s, const char *who) {
    // scratch: keys in | keys out | values in | (256 bytes) | rocPRIM temporaries   (hpl_tap_order_scratch_ints)
    char *p = reinterpret_cast<char *>(scratch);
    unsigned long long *kin = reinterpret_cast<unsigned long long *>(p); p += align256(M * 8);
    unsigned long long *kout = reinterpret_cast<unsigned long long *>(p); p += align256(M * 8);
    int32_t *vin = reinterpret_cast<int32_t *>(p); p += align256(M * 4);
    p += 256;
    const int grid = (int)cdiv(M, 256);
    k_order_keys<<<grid, 256, 0, s>>>(nbr, nbr_stride, F, M, kin, vin);
    size_t tb = (size_t)sort_temp_bytes(M);
    const hipError_t e = rocprim::radix_sort_pairs(p, tb, kin, kout, vin, perm, (unsigned)M, 48u, 48u + (unsigned)F, s);       // bits [48, 48 + F): tap mask
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
    return order_impl(nbr, nbr_stride, F, M, perm, scratch, to_stream(stream), "hpl_tap_order");
}
