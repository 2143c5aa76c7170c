// wgrad3.hip -- the weight gradient of the wide layers on the bf16 matrix pipe with fp32-exact split operands.
//
//   dWt[f*C + c, n] += sum_m A[nbr[f][m], c] * dY[m, n]                                  (as k_wgrad in gconv.hip)
//
// Same arithmetic as gconv3.hip: both fp32 operands are split exactly into three bf16 terms while they are staged into
// LDS, and the six partial products with i + j <= 2 are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 -- fp32-class
// accuracy at 16/6 of the fp32-MFMA rate.  HPL_MATH=f32 keeps the fp32 kernel.
//
// The contraction runs over VERTICES, the index both operands are row-major in (a gathered activation row [c], a
// gradient row [n]); the MFMA wants 8 consecutive contraction elements per lane for a fixed row / column.  gfx950
// reads that transposed: ds_read_b64_tr_b16 hands lane l of a 16-lane group column l of a [4][16] block of 16-bit
// elements (each lane supplies the address of 4 consecutive elements: row l/4, columns 4*(l%4) .. +3; measured with a
// probe on the MI355X).  So the LDS image stays row-major -- the split halves are stored with plain 8-byte writes, no
// software transpose anywhere -- and a fragment is two transpose reads.
//
// Geometry: a workgroup owns a 128 (k: channels of one tap) x 256 (n) tile of dWt and a slab of the tap's vertex list
// (tap mode: exact skipping of absent neighbours, hpl_tap_lists) or of the vertices (dense 1x1 layers); 8 waves as
// 2 x 4, each 64 x 64 (2 x 2 MFMA tiles, 64 accumulator registers); 16 vertices (one MFMA k-step, 24 MFMAs per wave)
// per stage, two LDS stages, the rows of stage s+2 in flight in registers while stage s is multiplied.  Row strides of
// the LDS planes are padded to 16 dwords mod 64 (320 B / 576 B): the four rows a transpose read touches land on
// disjoint banks.  Partial tiles are added to dWt with fp32 atomics (as k_wgrad).
#include "common.h"
#include "gconv_common.h"

#include <stdlib.h>
#include <type_traits>

using namespace hpl;
using namespace hpl_gc;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

// (x0, x1) -> packed bf16 pairs hi / mid / lo with x = hi + mid + lo exactly (gconv3.hip)
__device__ __forceinline__ void split2(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    const float2_t v = {x0, x1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    const float2_t rv = {r0, r1};
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(rv, bf16x2));
    const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    const float2_t sv = {s0, s1};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(sv, bf16x2));
}

// PL = 2 (round 5): (x0, x1) * s -> packed fp16 pairs hi / lo (gconv3.hip split2h)
__device__ __forceinline__ void split2h(float x0, float x1, float s, unsigned &h, unsigned &l) {
    const float2_t v = {x0 * s, x1 * s};
    const f16x2 hh = __builtin_convertvector(v, f16x2);
    h = __builtin_bit_cast(unsigned, hh);
    const float2_t hf = __builtin_convertvector(hh, float2_t);
    const float2_t r = {v.x - hf.x, v.y - hf.y};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

constexpr int W3_BKR = 128, W3_BN = 256, W3_BMS = 16, W3_NT = 512;
constexpr int W3_SA = 320, W3_SB = 576;                    // bytes per LDS row: 128 / 256 x 16 bit + pad (16 dwords mod 64)

// PL = 3: bf16 triples; PL = 2: fp16 pairs of both operands, each scaled by the power of two of its largest magnitude
template <bool TAP, int PL>
__global__ void __launch_bounds__(W3_NT) k_wgrad3(const WParams p) {
    constexpr int BKR = W3_BKR, BN = W3_BN, BMS = W3_BMS, SA = W3_SA, SB = W3_SB;
    constexpr int W3_A_STAGE = PL * W3_BMS * W3_SA;            // [plane][m][c]
    constexpr int W3_B_STAGE = PL * W3_BMS * W3_SB;            // [plane][m][n]
    constexpr int W3_STAGE = W3_A_STAGE + W3_B_STAGE;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * W3_STAGE];
    float s_a = 1.f, s_d = 1.f;
    if constexpr (PL == 2) { s_a = split_scale(p.a_amax[0]); s_d = split_scale(p.dy_amax[0]); }

    const int tile_k = blockIdx.x / p.tiles_n, tile_n = blockIdx.x % p.tiles_n;
    const int n0 = tile_n * BN;
    const int tap = TAP ? tile_k / p.c_tiles : 0;
    const int c0_tile = TAP ? (tile_k - tap * p.c_tiles) * BKR : tile_k * BKR;
    const int k0 = tap * p.C + c0_tile;
    const int32_t *vm = TAP ? p.tap_m + p.tap_ptr[tap] : nullptr;
    const int32_t *vrow = TAP ? p.tap_row + p.tap_ptr[tap] : nullptr;
    const int64_t m_total = TAP ? (int64_t)(p.tap_ptr[tap + 1] - p.tap_ptr[tap]) : p.M;
    const int64_t per = TAP ? ((m_total + gridDim.y - 1) / gridDim.y + BMS - 1) / BMS * BMS : p.m_per_split;
    const int64_t mb = (int64_t)blockIdx.y * per;
    const int64_t me = imin(m_total, mb + per);
    if (mb >= me) return;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, hi = lane >> 5;

    // staging roles: one float4 of a gathered row (32 threads per row), two float4 of gradient rows (64 threads per row).
    // Every load goes through a buffer descriptor with an out-of-range offset for what does not exist (rows past the slab,
    // channels past C, columns past N: zeros, no branches), as inline asm: hipcc does not count an asm load in its s_waitcnt
    // bookkeeping, so the rows of stage s+2 stay in flight across the barriers; completion is counted by hand (loads
    // complete in order, every step issues the same 3 index + 3 row loads) and `pin` orders the compiler's reads behind
    // the wait (the scheme of gconv3.hip).
    constexpr unsigned OOB = 0x80000000u;
    const int ar = t >> 5, br = t >> 6;
    const unsigned a_cofs = (c0_tile + (t & 31) * 4 < p.C) ? (unsigned)(c0_tile + (t & 31) * 4) * 4u : OOB;      // (C % 4 == 0)
    const unsigned b_cofs = (n0 + (t & 63) * 4 < p.N) ? (unsigned)(n0 + (t & 63) * 4) * 4u : OOB;                // (N % 4 == 0)
    const unsigned lda_b = (unsigned)p.lda * 4u, lddy_b = (unsigned)p.lddy * 4u;
    const int32x4_t rs_a = make_rsrc(p.A, 0x7fffffff), rs_b = make_rsrc(p.dY, 0x7fffffff);
    const int32x4_t rs_row = make_rsrc(TAP ? (const void *)(vrow + mb) : (const void *)p.A, TAP ? (int)((me - mb) * 4) : 0);
    const int32x4_t rs_m = make_rsrc(TAP ? (const void *)(vm + mb) : (const void *)p.A, TAP ? (int)((me - mb) * 4) : 0);
    const int slab = (int)(me - mb);

    struct Idx { int row, m0, m1; };
    struct Rows { float4_t a, b0, b1; };
    // index loads of stage st (tap mode): list entries past the slab read as 0 (out of the descriptor's range) and are
    // recognised by position in issue_rows
    auto issue_idx = [&](Idx &ix, int st) {
        if (!TAP) return;
        const int32x4_t r1 = rs_row, r2 = rs_m;
        const unsigned oa = (unsigned)(st * BMS + ar) * 4u, ob0 = (unsigned)(st * BMS + br) * 4u, ob1 = ob0 + 32u;
        asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(ix.row) : "v"(oa), "s"(r1) : "memory");
        asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(ix.m0) : "v"(ob0), "s"(r2) : "memory");
        asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(ix.m1) : "v"(ob1), "s"(r2) : "memory");
    };
    auto pin_idx = [&](Idx &ix) {
        if (TAP) asm volatile("" : "+v"(ix.row), "+v"(ix.m0), "+v"(ix.m1));
    };
    auto issue_rows = [&](Rows &r, const Idx &ix, int st) {
        const int ja = st * BMS + ar, jb = st * BMS + br;
        const int row = TAP ? ix.row : (int)mb + ja, m0 = TAP ? ix.m0 : (int)mb + jb, m1 = TAP ? ix.m1 : (int)mb + jb + 8;
        const unsigned oa = (ja < slab && a_cofs != OOB && row >= 0) ? (unsigned)row * lda_b + a_cofs : OOB;
        const unsigned o0 = (jb < slab && b_cofs != OOB) ? (unsigned)m0 * lddy_b + b_cofs : OOB;
        const unsigned o1 = (jb + 8 < slab && b_cofs != OOB) ? (unsigned)m1 * lddy_b + b_cofs : OOB;
        const int32x4_t r1 = rs_a, r2 = rs_b;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r.a) : "v"(oa), "s"(r1) : "memory");
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r.b0) : "v"(o0), "s"(r2) : "memory");
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r.b1) : "v"(o1), "s"(r2) : "memory");
    };
    auto pin_rows = [&](Rows &r) { asm volatile("" : "+v"(r.a), "+v"(r.b0), "+v"(r.b1)); };
    auto wait_vm = [](auto n_tag) {                // vmcnt <= N (loads complete in order)
        constexpr int N = decltype(n_tag)::value;
        __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4) | (0xF << 8));
    };
    auto store_lds = [&](int st, const Rows &r) {
        unsigned char *sa = smem + st * W3_STAGE + ar * SA + (t & 31) * 8;
        unsigned h0, m0, l0, h1, m1, l1;
        if constexpr (PL == 2) {
            split2h(r.a.x, r.a.y, s_a, h0, l0);
            split2h(r.a.z, r.a.w, s_a, h1, l1);
            *reinterpret_cast<u32x2 *>(sa) = u32x2{h0, h1};
            *reinterpret_cast<u32x2 *>(sa + BMS * SA) = u32x2{l0, l1};
        } else {
            split2(r.a.x, r.a.y, h0, m0, l0);
            split2(r.a.z, r.a.w, h1, m1, l1);
            *reinterpret_cast<u32x2 *>(sa) = u32x2{h0, h1};
            *reinterpret_cast<u32x2 *>(sa + BMS * SA) = u32x2{m0, m1};
            *reinterpret_cast<u32x2 *>(sa + 2 * BMS * SA) = u32x2{l0, l1};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4_t &v = i ? r.b1 : r.b0;
            unsigned char *sb = smem + st * W3_STAGE + W3_A_STAGE + (br + 8 * i) * SB + (t & 63) * 8;
            if constexpr (PL == 2) {
                split2h(v.x, v.y, s_d, h0, l0);
                split2h(v.z, v.w, s_d, h1, l1);
                *reinterpret_cast<u32x2 *>(sb) = u32x2{h0, h1};
                *reinterpret_cast<u32x2 *>(sb + BMS * SB) = u32x2{l0, l1};
            } else {
                split2(v.x, v.y, h0, m0, l0);
                split2(v.z, v.w, h1, m1, l1);
                *reinterpret_cast<u32x2 *>(sb) = u32x2{h0, h1};
                *reinterpret_cast<u32x2 *>(sb + BMS * SB) = u32x2{m0, m1};
                *reinterpret_cast<u32x2 *>(sb + 2 * BMS * SB) = u32x2{l0, l1};
            }
        }
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transpose-read addresses: lane -> (vertex row inside the k-step, first of its 4 consecutive columns).  The 16-lane
    // group g = lane / 16 serves columns 16 * (g & 1) .. +15 of a 32-wide MFMA operand tile and vertices 8 * (g >> 1) .. +7
    // (two reads of 4); lane i of the group addresses row i / 4, columns 4 * (i % 4) .. +3 of that [4][16] block.
    const int g16 = lane >> 4, i16 = lane & 15;
    const int fr_row = (g16 >> 1) * 8 + (i16 >> 2);
    const int fr_col = (g16 & 1) * 16 + (i16 & 3) * 4;
    const unsigned a_fofs = (unsigned)(fr_row * SA + (wm * 64 + fr_col) * 2);
    const unsigned b_fofs = (unsigned)(W3_A_STAGE + fr_row * SB + (wn * 64 + fr_col) * 2);

    auto frag = [&](const unsigned char *base) {
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(base));
        return lo;
    };
    auto multiply = [&](int st) {
        const unsigned char *sa = smem + st * W3_STAGE + a_fofs;
        const unsigned char *sb = smem + st * W3_STAGE + b_fofs;
        s16x8 af[PL][2], bf[PL][2];
#pragma unroll
        for (int pl = 0; pl < PL; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const s16x4 a0 = frag(sa + pl * BMS * SA + i * 64), a1 = frag(sa + pl * BMS * SA + i * 64 + 4 * SA);
                const s16x4 b0 = frag(sb + pl * BMS * SB + i * 64), b1 = frag(sb + pl * BMS * SB + i * 64 + 4 * SB);
                af[pl][i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                bf[pl][i] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        constexpr int NQ = PL == 3 ? 6 : 3;
        constexpr int PA[6] = {0, 0, 1, 0, PL == 3 ? 2 : 0, 1};
        constexpr int PB[6] = {0, 1, 0, PL == 3 ? 2 : 0, 0, 1};
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (PL == 3)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[PA[q]][i]),
                                                                            __builtin_bit_cast(bf16x8, bf[PB[q]][j]),
                                                                            acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[PA[q]][i]),
                                                                           __builtin_bit_cast(f16x8, bf[PB[q]][j]),
                                                                           acc[i][j], 0, 0, 0);
                }
    };

    // ROUND 5: the wait in front of the split + store is a FULL drain (vmcnt(0)), not "all but the newest three loads".  With the
    // counted wait the kernel was correct alone and WRONG beside memory-bound kernels of another stream (weight gradients on the
    // side stream of the training step; reproduced in isolation against a stream of elementwise kernels: relative error 2-4,
    // tools/.. tests/test_gpu_wgrad3.py::test_concurrent_streams): whenever the rows of stage st+2 were still in flight while
    // stage st+1 was split and stored, results went wrong -- also with the rows of st+1 known to have landed (a drain at the top
    // of the step did not help, a drain here does), so the in-order-completion argument below is not the whole story on this
    // chip.  Cost of the drain: < 1 % (the 24 MFMAs of a step cover the latency of the loads issued in front of them).
    // Step st multiplies LDS stage st & 1.  In flight at its start, oldest first: the indices of st+2, the rows of st+1.
    //   1. issue the index loads of st+3                        2. wait for the indices of st+2 (the oldest 3), issue its rows
    //   3. multiply                                             4. wait for the rows of st+1, split + store them into the other stage
    // Every step issues all six loads (out of range past the slab), so the counts never change.
    using N0 = std::integral_constant<int, 0>;
    using N6 = std::integral_constant<int, TAP ? 6 : 3>;
    const int nsteps = (slab + BMS - 1) / BMS;
    Idx ix0 = {0, 0, 0}, ix1 = {0, 0, 0};
    Rows r0, r1;
    issue_idx(ix0, 0);
    issue_idx(ix1, 1);
    wait_vm(std::integral_constant<int, TAP ? 3 : 0>{});
    pin_idx(ix0);
    issue_rows(r0, ix0, 0);
    issue_idx(ix0, 2);                         // in flight: idx(1), rows(0), idx(2)
    wait_vm(N6{});
    pin_idx(ix1);
    issue_rows(r1, ix1, 1);                    // rows(0), idx(2), rows(1)
    wait_vm(N6{});
    pin_rows(r0);
    store_lds(0, r0);
    __syncthreads();                           // in flight: idx(2), rows(1)
    auto body = [&](int st, int cur, Idx &ix_next, Idx &ix_cur, Rows &fill, Rows &ready) {
        issue_idx(ix_next, st + 3);            // idx(st+2), rows(st+1), idx(st+3)
        wait_vm(N6{});
        pin_idx(ix_cur);
        issue_rows(fill, ix_cur, st + 2);      // rows(st+1), idx(st+3), rows(st+2)
        multiply(cur);
        wait_vm(N0{});
        pin_rows(ready);
        store_lds(cur ^ 1, ready);
        // the split of the next stage in the shadows of this stage's MFMAs (one basic block: nothing here is conditional)
#pragma unroll
        for (int k = 0; k < (PL == 3 ? 24 : 12); ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, PL == 3 ? 4 : 6, 0);      // VALU
            if (PL == 2 || k % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // 1 DS write
        }
        __syncthreads();
    };
    // (index sets alternate with the steps: ix0 holds the indices of even stages; row sets: r1 holds odd stages)
    for (int st = 0; st < nsteps; st += 2) {
        body(st, 0, ix1, ix0, r0, r1);
        if (st + 1 < nsteps) body(st + 1, 1, ix0, ix1, r1, r0);
    }
    wait_vm(N0{});

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Buffer atomics with
    // 32-bit byte offsets (the image is far below 2 GB: checked by launch_wgrad3), out of range for rows past C / columns past N
    const float u_a = split_unscale(s_a), u_d = split_unscale(s_d);      // (1 for PL = 3)
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)p.dWt, (short)0, 0x7fffffff, 0x00020000);
    const unsigned ldw_b = (unsigned)p.ldw * 4u;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + li;
            const unsigned nb = n < p.N ? (unsigned)n * 4u : OOB;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kr = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const unsigned o = (c0_tile + kr < p.C) ? __umul24((unsigned)(k0 + kr), ldw_b) + nb : OOB;
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(PL == 2 ? acc[i][j][r] * u_a * u_d : acc[i][j][r], rs_w, (int)o, 0, 0);
            }
        }
}

}  // namespace

bool hpl_gc::launch_wgrad3(WParams &p, bool tap, int64_t m_len, hipStream_t s) {
    // qualifying launches: the wide layers (tap mode stencils, dense 1x1 convs) with enough vertices to fill the chip
    static const int on = getenv("HPL_WGRAD3") ? atoi(getenv("HPL_WGRAD3")) : 1;
    constexpr int min_rows = 8192;
    if (!on || !split3_enabled()) return false;
    if (!(tap || (p.F == 1 && !p.nbr))) return false;
    if (p.N < 256 || p.N % 4 != 0 || p.C < 128 || p.C % 4 != 0 || p.M < min_rows) return false;
    // (32-bit byte offsets in the buffer loads)
    if (p.rows_a <= 0 || p.rows_a * p.lda * 4 >= (int64_t)0x7fffffff || p.M * p.lddy * 4 >= (int64_t)0x7fffffff) return false;
    if ((int64_t)p.K * p.ldw * 4 >= (int64_t)0x7fffffff || p.K >= (1 << 24) || p.ldw * 4 >= (1 << 24)) return false;
    p.c_tiles = (int)cdiv(p.C, W3_BKR);
    p.tiles_n = (int)cdiv(p.N, W3_BN);
    const int tiles = (tap ? p.F : 1) * p.c_tiles * p.tiles_n;
    // slabs of the vertex loop: one workgroup per CU at a time; ~8 workgroups per CU over the launch even out the unequal
    // tap lists, each with >= 512 vertices (the atomic epilogue of a 128 x 256 tile costs about 100 vertices' worth)
    constexpr int force = 0;
    const int64_t len = tap ? imax(1, m_len / 2) : m_len;
    int64_t splits = imax(1, imin(cdiv(2048, tiles), cdiv(len, 512)));
    // dense layers: equal slabs, so one workgroup per CU in ONE round is the best cut (measured: 25 841 x 1024 x 1024 in 7
    // slabs = 224 workgroups 0.28 ms, 14 slabs 0.30 ms, 50 slabs 0.37 ms -- every slab pays an atomic epilogue of 32 K adds)
    if (!tap) splits = imax(1, imin(256 / imax(1, tiles), cdiv(len, 256)));
    if (force > 0) splits = force;
    p.m_per_split = cdiv(cdiv(m_len, splits), W3_BMS) * W3_BMS;
    if (!tap) splits = cdiv(m_len, p.m_per_split);
    const dim3 grid((unsigned)tiles, (unsigned)splits);
    // fp16 pairs when both largest magnitudes were given (and the mode allows), else the exact bf16 triples
    const bool pairs = split_planes() == 2 && p.a_amax && p.dy_amax;
    if (pairs) {
        if (tap) k_wgrad3<true, 2><<<grid, W3_NT, 0, s>>>(p);
        else k_wgrad3<false, 2><<<grid, W3_NT, 0, s>>>(p);
    } else {
        if (tap) k_wgrad3<true, 3><<<grid, W3_NT, 0, s>>>(p);
        else k_wgrad3<false, 3><<<grid, W3_NT, 0, s>>>(p);
    }
    return true;
}
