// gconv3.hip -- the wide gather-GEMM launches on the bf16 matrix pipe with fp32-exact split operands.
//
//   Y[m, n] = act(bias[n] + res + sum_{f<F} sum_{c<C} A[nbr[f][m], c] * Wt[f*C + c, n])        (as gconv.hip)
//
// gfx950 multiplies fp32 operands on the matrix cores at 1/16 of the bf16 rate (v_mfma_f32_32x32x2_f32: 64 cycles
// for k = 2; v_mfma_f32_32x32x16_bf16: 32 cycles for k = 16).  An fp32 number is EXACTLY the sum of three bf16
// numbers (8 + 8 + 8 significand bits: hi = rne(x), mid = rne(x - hi), lo = x - hi - mid, every step exact), so
//      a * b = sum_{i,j} a_i * b_j,      a = a_0 + a_1 + a_2,  b = b_0 + b_1 + b_2,
// and each partial product of two bf16 values is exact in fp32.  This kernel accumulates the six partial products
// with i + j <= 2 in fp32 on the bf16 MFMA; the three it drops are bounded by 2^-25 |a b| together (|a_i| <= 2^-9i |a|
// with round-to-nearest splits) -- below the 2^-24 rounding of an fp32 fused multiply-add.  6 MFMAs of 32 cycles per
// k = 16 against 8 of 64: 16/6 = 2.67x the fp32-MFMA rate at fp32-class accuracy (tests/test_gpu_split3.py measures
// both paths against float64).  Activations stay fp32 in HBM and are split while they are staged into LDS; the
// weights are split once per model into three planes (hpl_weight_split3).
//
// Geometry: 128 x (64*WGN) output tile, 2 x WGN waves, each wave 64 x 64 (2 x 2 MFMA tiles of 32 x 32, 64 accumulator
// registers).  The contraction runs over the tile's list of needed 32-wide slices (as gconv.hip: slices whose taps are
// absent for the whole tile are skipped; a wave whose 64 rows lack the taps of a slice skips its MFMAs) in HALF-slices of
// 16 -- one MFMA k-step, 24 MFMAs per wave.  Gathered rows: fp32 from HBM as full 128-byte lines (inline-asm buffer loads,
// completion counted by hand), NB - 1 register sets, split while they are staged into a ring of three LDS stages
// [plane][k-block][row][8 bf16] (row slots XOR-swizzled).  Weights: LDS-direct loads of the split image (the three planes
// are stored [k/8][n][8] in HBM, i.e. already in MFMA B-fragment order) into a ring of NB stages, NB - 1 half-steps ahead.
//
// WGN = 4 (128 x 256, 8 waves, one workgroup per CU: the wide forward launches) runs a PING-PONG schedule: a half-step is a
// memory phase (its 12 fragment reads) and a compute phase (its 24 MFMAs with everything else of the half-step -- weight
// loads, index reads + gathered loads, split + store -- interleaved in one basic block), each closed by a workgroup barrier;
// the second wave row enters one barrier late, so on every SIMD one wave computes while the other reads (see PP below).
// WGN = 2 (128 x 128, 4 waves, two workgroups per CU: N that would pad a 256-wide tile by > 8 %, i.e. the data gradients)
// keeps one barrier per half-step, MFMAs and the rest likewise in one block.
// Mid-size stencils (too few tiles for 256 CUs) are split over K: shares of the slice range -> partial tiles -> k_gconv_finish.
#include "common.h"
#include "gconv_common.h"

#include <stdlib.h>
#include <string>
#include <type_traits>

using namespace hpl;
using namespace hpl_gc;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// (Diagnostic builds of rounds 3-4 -- per-phase cycle stamps, per-workgroup residence records, the one-barrier form of the 8-wave
// tile -- produced profiles/r03s_phase_probe.txt, r03s_variants_ab.txt, r04b_split3_pp3_ab.txt, r04h_tile_balance.txt and were removed
// in round 5; DESIGN_HISTORY.md 4.1 / 4.8 quote their results.)

namespace {

// (x0, x1) -> packed bf16 pairs hi / mid / lo with x = hi + mid + lo exactly (round-to-nearest-even at each level)
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

// PL = 2 (round 5): (x0, x1) * s -> packed fp16 pairs hi / lo, hi = rne(x s), lo = rne(x s - hi): x s = hi + lo up to 2^-22 |x s|
// (s: the power of two that puts the matrix's largest magnitude into [2^14, 2^15), gconv_common.h split_scale; the residual
// x s - hi is exact in fp32, so lo is ONE rounding).  Three partial products hi*hi + hi*lo + lo*hi on the fp16 MFMA.
__device__ __forceinline__ void split2h(float x0, float x1, float s, unsigned &h, unsigned &l) {
    const float2_t v = {x0 * s, x1 * s};
    const f16x2 hh = __builtin_convertvector(v, f16x2);
    h = __builtin_bit_cast(unsigned, hh);
    const float2_t hf = __builtin_convertvector(hh, float2_t);
    const float2_t r = {v.x - hf.x, v.y - hf.y};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

// Second pass of the range guard (hpl_gconv_desc.a_guard): the residual of the first split, r = (x s - hi) - lo (both subtractions
// exact in fp32), times 2^24, again as an fp16 pair.  |x s - hi| <= 2^3 and |r| <= 2^-9 for |x s| < 2^15, so r 2^24 <= 2^15 fits fp16;
// for an element too small for a normal lo (|x s| < 2^-3) r is what the first pass lost: it is carried here to 2^-24 of ITS size.
__device__ __forceinline__ void split2h_resid(float x0, float x1, float s, unsigned &h, unsigned &l) {
    const float2_t v = {x0 * s, x1 * s};
    const f16x2 hh = __builtin_convertvector(v, f16x2);
    const float2_t hf = __builtin_convertvector(hh, float2_t);
    const float2_t r = {v.x - hf.x, v.y - hf.y};
    const f16x2 ll = __builtin_convertvector(r, f16x2);
    const float2_t lf = __builtin_convertvector(ll, float2_t);
    constexpr float K = (float)(1 << RESID_SHIFT);
    const float2_t r2 = {(r.x - lf.x) * K, (r.y - lf.y) * K};
    const f16x2 h2 = __builtin_convertvector(r2, f16x2);
    h = __builtin_bit_cast(unsigned, h2);
    const float2_t h2f = __builtin_convertvector(h2, float2_t);
    const float2_t q = {r2.x - h2f.x, r2.y - h2f.y};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(q, f16x2));
}

constexpr int BM3 = 128;

// NB = stages of the weight-fragment ring (3 or 4); the gathered rows use NB - 1 register sets: with NB = 4 every load has one
// more half-step to land (the end-of-half-step wait then leaves the loads of TWO half-steps in flight)
// GUARD (PL = 2): the SECOND launch of a guarded operand (hpl_gconv_desc.a_guard).  The first launch (GUARD = false) is the round-5
// kernel; if the operand's guard trips (a row 2^18 below the matrix's largest magnitude) its epilogue stores the sums WITHOUT the
// activation (and publishes no magnitudes, writes no second destination).  The second launch leaves at once unless the guard trips;
// then it runs the same slice lists on the residuals of the first split, (a s - hi - lo) 2^24 as fp16 pairs, adds what the first
// launch stored (the host passes res = Y, no bias) and finishes: activation, second destination, magnitudes.  Keeping the second
// pass out of the first kernel keeps that kernel's code (15.5 k instructions) and registers what they were: with both passes in one
// kernel (24.4 k instructions) the unguarded launches ran 4-10 % slower (profiles/r06b_split3_guard_in_kernel.txt).
template <int WGN, int F_LDS, int NB, int PL, bool GUARD = false>
__device__ __forceinline__ void gconv3_body(const GParams &p, const unsigned block) {
    static_assert(PL == 2 || PL == 3, "operand planes: 2 (fp16 pairs) or 3 (bf16 triples)");
    static_assert(!GUARD || PL == 2, "the range guard belongs to the fp16-pair form");
    bool tripped = false;
    if constexpr (PL == 2) tripped = guard_tripped(p.a_amax, p.a_guard);      // (uniform)
    if constexpr (GUARD) { if (!tripped) return; }
    constexpr int BM = BM3, BN = 64 * WGN, NT = 128 * WGN;
    // Gathered loads: a thread's pass p fetches float4 column (t & 3) of HALF p & 1 of tile row t / 4 + (p / 2) * ROWS_PP: four
    // lanes = the 64 bytes a row contributes to a half-slice.  (Eight lanes per full 128-byte line would give every thread one
    // float4 of each half too, but WHICH pass holds which half would depend on the lane: the half-step's stores then select
    // per lane -- 8 v_cndmask + 4 v_mov_b64 per slice beside the MFMAs, where every VALU instruction costs 5-10 cycles.)
    constexpr int ROWS_PP = NT / 4;                 // rows covered by one gathered load instruction of the workgroup
    constexpr int A_PASSES = 2 * BM / ROWS_PP;      // float4 per thread and slice: 4 (WGN = 2) or 2 (WGN = 4)
    constexpr int HALF_PASSES = A_PASSES / 2;
    constexpr int A_STAGE = PL * 2 * BM * 16;       // bytes: [plane][kb 0..1][row][8 x 16 bit]
    constexpr int B_STAGE = PL * 2 * BN * 16;       //        [plane][kb 0..1][n][8 x 16 bit]
    constexpr int NA = 3;                            // stages of the gathered-row ring
    static_assert(NB == 3 || NB == 4, "weight ring of 3 or 4 stages");
    constexpr int ASETS = NB - 1;                    // register sets of gathered rows in flight
    constexpr int B_BASE = NA * A_STAGE;             // LDS: A ring | B ring | indices
    constexpr int B_CHUNKS_PER_WAVE = PL;           // 2 * PL * WGN chunks of 1 KiB per half-step over 2 * WGN waves
    constexpr int KLIST = 1024;
    // (a 3-stage weight ring has its next half-step still in flight; the 4-wave tile runs two workgroups per CU in 128 registers)
    // PP: the two wave rows of the 8-wave tile (waves 0-3 / 4-7: one wave of each on every SIMD) run half a half-step apart.
    // A half-step is a MEMORY phase (the 12 fragment reads of this half-step -- what has to wait for the barrier) and a
    // COMPUTE phase (its 24 MFMAs; weight loads, index reads + gathered loads, split + store of the rows staged for later are
    // interleaved with them), each closed by a workgroup barrier; the second wave row enters one barrier late, so while one
    // wave of a SIMD issues its MFMAs the other waits for its fragments -- without the offset both waves of a SIMD reach
    // the same phase together and the matrix pipe idles through every memory phase (ablation of the one-barrier form: 39 %
    // of the dense launch's time was not MFMA issue).  Hazards: an A stage is written two half-steps and a weight stage three
    // half-steps before it is read, i.e. >= 3 barriers before the first reader of either row; a stage's last reader (the
    // late row, one barrier behind) is still >= 1 barrier ahead of its next writer.
    constexpr bool PP = WGN == 4;
    static_assert(A_PASSES == 2 || A_PASSES == 4, "two halves of a slice per thread");
    // ONE LDS array (a second __shared__ object makes hipcc drain vmcnt before every ds_read of an LDS-DMA pipeline)
    __shared__ __attribute__((aligned(16))) unsigned char smem[NA * A_STAGE + NB * B_STAGE + ((F_LDS + 1) * BM + BM + 8) * 4 + KLIST * 2];
    int *Is = reinterpret_cast<int *>(smem + NA * A_STAGE + NB * B_STAGE);
    int *Vs = Is + (F_LDS + 1) * BM;               // (the index table has one more tap than F_LDS: a row of absent entries)
    int *tapmask_s = Vs + BM;                       // [0] taps of the tile, [1] slices needed, [2..5] taps of its 32-row blocks
    unsigned short *Ks = reinterpret_cast<unsigned short *>(tapmask_s + 8);

    int tile_m, tile_n;
    tile_coords(p, tile_m, tile_n, block);
    if (tile_m < 0) return;
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 31, hi = lane >> 5;

    const bool probe = p.clock_probe && (block & 63) == 0 && t == 0;
    long long probe_c = 0, probe_w = 0;
    if (probe) { probe_c = (long long)__builtin_readcyclecounter(); probe_w = (long long)__builtin_amdgcn_s_memrealtime(); }

    // ---- tile prologue: output rows, source rows of every (tap, tile row), tap masks
    if (p.tile_idx && p.tile_bm == BM) {
        const int32_t *ti = p.tile_idx + (int64_t)tile_m * p.F * BM;
        if (t < 8) tapmask_s[t] = p.tile_mask[(int64_t)tile_m * 8 + t];
        for (int r = t; r < BM; r += NT) {
            const int64_t m = m0 + r;
            Vs[r] = (m < p.M) ? (p.row_perm ? p.row_perm[m] : (int)m) : -1;
        }
        // (the table holds the BYTE offset of a source row, 0x80000000 for an absent one -- adding a column offset keeps that
        // beyond the buffer's range, so a gathered load needs neither a multiply nor a validity select; a_bytes < 2^31 by
        // contract.  Taps F .. F_LDS are all absent: a slice that straddles the end of the last tap reads them.)
        for (int i = t; i < (F_LDS + 1) * BM; i += NT) {
            const int row = (i < p.F * BM) ? ti[i] : -1;
            Is[i] = row >= 0 ? row * (int)(p.lda * 4) : (int)0x80000000;
        }
    } else {
        if (t < 8) tapmask_s[t] = 0;
        for (int r = t; r < BM; r += NT) {
            const int64_t m = m0 + r;
            Vs[r] = (m < p.M) ? (p.row_perm ? p.row_perm[m] : (int)m) : -1;
        }
        __syncthreads();
        int mybits = 0;
        for (int i = t; i < F_LDS * BM; i += NT) {       // NT % BM == 0: a thread keeps its tile row
            const int f = i / BM, r = i - f * BM;
            const int v = Vs[r];
            int row = -1;
            if (v >= 0 && f < p.F) row = p.nbr ? p.nbr[(int64_t)f * p.nbr_stride + v] : (int)((int64_t)f * p.reg_stride + v);
            Is[i] = row >= 0 ? row * (int)(p.lda * 4) : (int)0x80000000;
            mybits |= (row >= 0) ? (1 << f) : 0;
        }
        for (int i = F_LDS * BM + t; i < (F_LDS + 1) * BM; i += NT) Is[i] = (int)0x80000000;
        if (mybits) {
            atomicOr(tapmask_s, mybits);
            atomicOr(tapmask_s + 2 + ((t % BM) >> 5), mybits);
        }
    }
    __syncthreads();
    const int tapmask = __builtin_amdgcn_readfirstlane(tapmask_s[0]);
    const bool blockskip = p.C >= BK;
    int bmask[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) bmask[i] = __builtin_amdgcn_readfirstlane(tapmask_s[2 + wm * 2 + i]);
    // list of the slices this tile needs (entry: bits 0..9 slice, 10..13 first tap, 14 "also the next tap")
    const int nk = (p.K + BK - 1) / BK;
    if (wave == 0) {
        int count = 0;
        for (int base = 0; base < nk; base += 64) {
            const int kt = base + lane;
            bool need = false;
            int f_lo = 0;
            if (kt < nk) {
                f_lo = (kt * BK) / p.C;
                const int f_hi = min((kt * BK + BK - 1) / p.C, p.F - 1);
                int bits = 0;
                for (int f = f_lo; f <= f_hi; ++f) bits |= 1 << f;
                need = (tapmask & bits) != 0;
            }
            const unsigned long long bal = __ballot(need);
            if (need) {
                int e = kt;
                e |= (f_lo << 10) | ((blockskip && (kt * BK + BK - 1) / p.C > f_lo && f_lo + 1 < p.F) ? (1 << 14) : 0);
                Ks[count + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)e;
            }
            count += __popcll(bal);
        }
        if (lane == 0) tapmask_s[1] = count;
    }
    __syncthreads();
    int nsl = __builtin_amdgcn_readfirstlane(tapmask_s[1]);
    // split-K (mid-size stencils: too few tiles for 256 CUs): this workgroup takes the list entries whose slice INDEX lies in
    // its share of [0, nk) -- cut points that do not depend on the tile, so a row's partial sums cover the same k ranges
    // wherever the row order puts it -- and writes a partial tile (k_gconv_finish adds the shares in split order)
    int split = 0;
    if (p.splits > 1) {
        const int ngrid = p.row_perm && p.col_share > 0 ? p.tiles_n * p.col_share * p.col_rows : p.tiles_m * p.tiles_n;
        split = (int)block / ngrid;
        const int k_lo = (int)((int64_t)nk * split / p.splits), k_hi = (int)((int64_t)nk * (split + 1) / p.splits);
        auto below = [&](int kt) {           // list entries with slice index < kt (the list is ascending)
            int a = 0, b = nsl;
            while (a < b) {
                const int mid = (a + b) >> 1;
                if ((int)(Ks[mid] & 1023) < kt) a = mid + 1; else b = mid;
            }
            return a;
        };
        const int lo = below(k_lo), hi_e = below(k_hi);
        __syncthreads();
        if (lo > 0) {                        // compact this share to the front of the list
            unsigned short mine[KLIST / (128 * WGN) + 1];
            int cnt = 0;
            for (int i = t; i < hi_e - lo; i += NT) mine[cnt++] = Ks[lo + i];
            __syncthreads();
            cnt = 0;
            for (int i = t; i < hi_e - lo; i += NT) Ks[i] = mine[cnt++];
        }
        __syncthreads();
        nsl = hi_e - lo;
    }

    // ---- staging state
    constexpr unsigned OOB = 0x80000000u;
    const int32x4_t rsrc_a = make_rsrc(p.A, (int)p.a_bytes);
    const int k4 = t & 3, arow0 = t >> 2;
    // weight planes: rows of the image that exist = p.w_bytes / (ldw * 4) (a multiple of 8 by contract)
    const unsigned w3_bytes = (unsigned)(p.w_bytes / 2);                  // (rows / 8) * ldw * 16 bytes per plane
    __amdgpu_buffer_rsrc_t rsrc_b[PL];
#pragma unroll
    for (int pl = 0; pl < PL; ++pl)
        rsrc_b[pl] = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char *>(reinterpret_cast<const unsigned char *>(p.Wt3) + (int64_t)pl * p.w3_plane_stride),
            (short)0, (int)w3_bytes, 0x00020000);
    const unsigned ldw16 = (unsigned)p.ldw * 16u;

    // Gathered rows: buffer_load_dwordx4 through inline asm -- hipcc does not count an asm load in its s_waitcnt
    // bookkeeping, so the loads of slice s+2 stay in flight across the half-step barriers instead of being drained at
    // the first use of ANY load result inside the loop (its scoreboard merges conservatively over the back edge: the
    // builtin form waited vmcnt(0) before every LDS store).  Completion is counted by hand: the end-of-half-step wait
    // leaves only that half-step's own loads in flight (loads complete in order), so a register set is complete one
    // half-step after its loads were issued; `pin` then orders the compiler's reads behind that wait.
    float4_t ra[ASETS][A_PASSES];
    // load_a_rows: (f, c) of this thread's float4 columns in slice kt and the LDS reads of their source rows;
    // load_a_issue: the loads.  Two calls, so that the LDS round trip of the indices sits behind the first MFMAs of a
    // half-step instead of in front of them.
    int a_rows[A_PASSES], a_c[A_PASSES];
    auto load_a_rows = [&](int e) {               // e: slice-list entry (slice | first tap << 10): no running state, no loop
        const int f0 = (e >> 10) & 15;
        const int c0 = (e & 1023) * BK - f0 * p.C;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            int c = c0 + ((i & 1) * 4 + k4) * 4, f = f0;      // (pass i: half i & 1)
            const bool wrap = c >= p.C;                // (C >= 32: a slice touches at most two taps)
            c = wrap ? c - p.C : c;
            f += wrap ? 1 : 0;                         // (<= F <= F_LDS: tap F of the table is all absent)
            a_rows[i] = Is[f * BM + arow0 + (i >> 1) * ROWS_PP];
            a_c[i] = c;
        }
    };
    auto load_a_issue = [&](auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        const int32x4_t rs = rsrc_a;                  // (asm operands inside a generic lambda must be its own locals)
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            float4_t &dst = ra[SET][i];
            // (an absent row's entry is 0x80000000: with the column offset still beyond the descriptor's range -> zeros)
            const unsigned o = (unsigned)a_rows[i] + (unsigned)a_c[i] * 4u;
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(o), "s"(rs) : "memory");
        }
    };
    auto load_a = [&](auto set_tag, int e) {
        load_a_rows(e);
        load_a_issue(set_tag);
    };
    auto pin = [&](auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            float4_t &x = ra[SET][i];
            asm volatile("" : "+v"(x));
        }
    };
    // half h of register set `set`: split + store into stage `st` (a thread holds float4 column (t & 3) of half h in its
    // passes 2 * j + h)
    // LDS slot of tile row r in k-block kb of a stage: r ^ swz(r, kb), swz = 2 * (r / 32) ^ 4 * kb (bits 1..2 of the
    // row).  Without it the 16 lanes of a store group differ only in address bits that do not reach the bank index
    // (k-blocks are 2 KiB apart, the two row sets of a thread 32 or 64 rows): 4-way conflicts, half of the LDS cycles
    // (SQ_LDS_BANK_CONFLICT).  The fragment reads stay conflict-free: the XOR permutes rows inside aligned groups of 8.
    auto a_slot = [](int row, int kb) { return row ^ ((((row >> 5) & 3) << 1) ^ (kb << 2)); };
    const int kb_w = (t >> 1) & 1;
    // PL = 2: the scales of the two operands (powers of two from the matrices' largest magnitudes, device scalars)
    float s_a = 1.f, s_out = 1.f, s_out2 = 1.f;
    if constexpr (PL == 2) {
        s_a = split_scale(p.a_amax[0]);
        s_out = split_unscale(s_a);
        if constexpr (GUARD) s_out *= 1.0f / (float)(1 << RESID_SHIFT);
        s_out2 = split_unscale(split_scale(p.w_amax[0]));
    }
    auto store_a = [&](auto set_tag, int h, int st, int j) {
        constexpr int SET = decltype(set_tag)::value;
        const float4_t v = h ? ra[SET][2 * j + 1] : ra[SET][2 * j];      // (h is a literal at every call site)
        const int row = arow0 + j * ROWS_PP;
        unsigned char *base = smem + st * A_STAGE + (kb_w * BM + a_slot(row, kb_w)) * 16 + (t & 1) * 8;
        if constexpr (PL == 2) {
            unsigned h0, l0, h1, l1;
            if constexpr (GUARD) {
                split2h_resid(v.x, v.y, s_a, h0, l0);
                split2h_resid(v.z, v.w, s_a, h1, l1);
            } else {
                split2h(v.x, v.y, s_a, h0, l0);
                split2h(v.z, v.w, s_a, h1, l1);
            }
            *reinterpret_cast<u32x2 *>(base) = u32x2{h0, h1};
            *reinterpret_cast<u32x2 *>(base + 2 * BM * 16) = u32x2{l0, l1};
        } else {
            unsigned h0, m0_, l0, h1, m1, l1;
            split2(v.x, v.y, h0, m0_, l0);
            split2(v.z, v.w, h1, m1, l1);
            *reinterpret_cast<u32x2 *>(base) = u32x2{h0, h1};
            *reinterpret_cast<u32x2 *>(base + 2 * BM * 16) = u32x2{m0_, m1};
            *reinterpret_cast<u32x2 *>(base + 4 * BM * 16) = u32x2{l0, l1};
        }
    };
    const unsigned b_colofs = ((unsigned)(n0 + wn * 64 + lane) < (unsigned)p.ldw) ? (unsigned)(n0 + wn * 64 + lane) * 16u : OOB;
    // weight fragments of half h of slice kt straight into stage st: wave (wm, wn) fetches k-block wm of the half for
    // the 64 columns of column block wn, one 1 KiB LDS-direct load per plane
    auto load_b = [&](int kt, int h, int st) {
        const unsigned kbg = (unsigned)(kt * (BK / 8) + h * 2 + wm);
        const unsigned off = kbg * ldw16 + b_colofs;      // (b_colofs = 0x80000000 for a column past the image: stays out of range)
#pragma unroll
        for (int pl = 0; pl < PL; ++pl)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc_b[pl],
                (__attribute__((address_space(3))) void *)(smem + B_BASE + st * B_STAGE + ((pl * 2 + wm) * BN + wn * 64) * 16),
                16, (int)off, 0, 0, 0);
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    unsigned a_rofs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + i * 32 + li;
        a_rofs[i] = (unsigned)((hi * BM + (row ^ ((((row >> 5) & 3) << 1) ^ (hi << 2)))) * 16);
    }
    const unsigned b_rofs = (unsigned)(B_BASE + (hi * BN + wn * 64 + li) * 16);

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using T = std::true_type;
    using Fl = std::false_type;
    auto wait_vm_lgkm0 = [](auto n_tag) {          // vmcnt <= N (loads complete in order), lgkmcnt = 0
        constexpr int N = decltype(n_tag)::value;
        __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4) | (0x0 << 8));
    };

    using S2 = std::integral_constant<int, 2>;
    if (nsl > 0) {
        // the slice list is read through a register: lane l holds entry ks_cb + l (refilled every 32 slices), an entry is a
        // v_readlane -- no LDS round trip (+ the in-order wait behind whatever else is queued there) in front of the half-steps
        int ks_cb = 0;
        int ks_reg = (int)Ks[min(lane, KLIST - 1)];
        auto ks_at = [&](int x) { return __builtin_amdgcn_readlane(ks_reg, x - ks_cb); };
        auto ks_advance = [&](int sl) {
            if (sl - ks_cb >= 32) { ks_cb += 32; ks_reg = (int)Ks[min(ks_cb + lane, KLIST - 1)]; }
        };
        auto kt_at = [&](int sl) { return ks_at(sl) & 1023; };
        // ---- fill: weight fragments of the first NB - 1 half-steps, slice 0 -> A stages 0 and 1, slices 1 .. ASETS-1 -> their
        // register sets; everything has landed before the first barrier (the first half-step stages slice 1 at once)
#pragma unroll
        for (int g = 0; g < NB - 1; ++g)
            if (g / 2 < nsl) load_b(kt_at(g / 2), g & 1, g);
        load_a(S0{}, ks_at(0));
        wait_vm_lgkm0(S0{});
        pin(S0{});
#pragma unroll
        for (int j = 0; j < HALF_PASSES; ++j) { store_a(S0{}, 0, 0, j); store_a(S0{}, 1, 1, j); }
        if (nsl > 1) load_a(S1{}, ks_at(1));
        if constexpr (ASETS == 3) { if (nsl > 2) load_a(S2{}, ks_at(2)); }
        wait_vm_lgkm0(S0{});
        asm volatile("s_barrier" ::: "memory");

        if (PP && wm == 1) asm volatile("s_barrier" ::: "memory");      // the second wave row runs one phase behind

        int sta = 0, stb = 0;                        // stages of the half-step being multiplied (A ring, B ring)
        // One half-step g = 2*s + h: multiply (A stage sta, B stage stb; slice entry e_cur);
        //   B: LDS-direct loads of the weight fragments of half-step g + NB - 1 (slice kt_b, half hb) -> B stage (stb + NB - 1) % NB
        //   W: split + store half h of slice s + 1 (register set SETW) -> A stage (sta + 2) % 3
        //   L: gathered loads of slice s + ASETS (kt_l) -> register set SETL (h == 0 only)
        // The wait at its end leaves the loads of the last NB - 2 half-steps in flight (`inflight`, counted by the caller).
        auto halfstep = [&](int e_cur, int h, auto b_tag, int kt_b, int hb_b, auto w_tag, auto setw_tag, auto l_tag, auto setl_tag,
                            int kt_l, auto inflight_tag) {
            constexpr bool B = decltype(b_tag)::value, W = decltype(w_tag)::value, L = decltype(l_tag)::value;
            const int sta2 = sta >= 1 ? sta - 1 : 2;                      // (sta + 2) % 3
            const int stb2 = stb >= 1 ? stb - 1 : NB - 1;                  // (stb + NB - 1) % NB
            const unsigned char *sa = smem + sta * A_STAGE;
            const unsigned char *sb = smem + stb * B_STAGE;
            bool need[2];
            {
                const int f_lo = (e_cur >> 10) & 15, two = (e_cur >> 14) & 1;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    need[i] = !blockskip || (((bmask[i] >> f_lo) | (two ? (bmask[i] >> (f_lo + 1)) : 0)) & 1);
            }
            u32x4 af[PL][2], bf[PL][2];
            // partial products a_i * b_j, i + j <= PL - 1: 6 of the 9 (bf16 triples) / 3 of the 4 (fp16 pairs)
            constexpr int NQ = PL == 3 ? 6 : 3;
            constexpr int PA[6] = {0, 0, 1, 0, PL == 3 ? 2 : 0, 1};
            constexpr int PB[6] = {0, 1, 0, PL == 3 ? 2 : 0, 0, 1};
            // (the operands SWAPPED: the accumulators hold the 32 x 32 block TRANSPOSED -- lane l & 31 is the output row, register r the
            // column (r & 3) + 8 (r >> 2) + 4 (l >> 5) -- so that four registers are four consecutive columns of a row: the epilogue
            // moves 16 bytes per lane and instruction; both fragments have the same lane layout, the swap costs nothing)
            auto mfma = [&](const u32x4 &a, const u32x4 &b, floatx16 &c) {
                if constexpr (PL == 3)
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
                else
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, b), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
            };
            if constexpr (PP) {
                // ---- memory phase: only what has to wait for the barrier -- the twelve fragment reads of this half-step.
#pragma unroll
                for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        af[pl][i] = *reinterpret_cast<const u32x4 *>(sa + a_rofs[i] + pl * 2 * BM * 16);
                        bf[pl][i] = *reinterpret_cast<const u32x4 *>(sb + b_rofs + pl * 2 * BN * 16 + i * 32 * 16);
                    }
                // (PL = 2, round 5: split + store moved up here, where the wave waits for the barrier anyway: 411 -> 434 us on bcn1_'s
                // first pass, 273 -> 291 us on bcn2_'s -- it stays in the compute phase)
                // the loads of the compute phase before last have landed (in flight: the last compute phase's); fragments here,
                // the LDS stores of the last compute phase done
                wait_vm_lgkm0(inflight_tag);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                // ---- compute phase: the 24 MFMAs of the wave's 64 rows and, in their shadow, everything else of the half-step
                // that does not depend on the barrier: the weight loads three half-steps ahead, index reads + gathered loads,
                // split + store of the rows staged for the half-step after next.  One basic block, so that hipcc can interleave
                // (the 32-row block skip of the non-ping-pong path would cut it into four); a wave whose 64 rows have no tap in the
                // slice skips the MFMAs altogether.
                // Measured (cycles per half-step and wave, -DHPL_PHASE_PROBE=1, profiles/r03s_phase_probe.txt): memory phase 450-480,
                // compute phase 975-1015 -- the interleaved groups (weight loads / gathered loads / split + store) add 50-110 each
                // to the 815-835 of the bare MFMAs: an instruction beside the MFMAs costs 5-10 cycles of issue, nothing is free.
                // With all three groups in the memory phase instead: 1190 / 790 (before the instruction trims); with the gathered
                // loads and the stores there 617 us against 597 us for the launch: this split is the fastest of the four tried.
                // Round 5, pair form, timing-only builds without one group each (bcn1_'s first pass, 403 us): no weight loads 378, no
                // index reads + gathered loads 375, no split + store 366, none of the three 320 us -- against 131 us of matrix-pipe time.
                // The staging is a fifth of the launch; the rest is the tile structure itself (rounds of unequal tiles, wave rows
                // without a tap of the slice, prologue / epilogue, the phases' barriers).
                auto others = [&]() {
                    if constexpr (B) load_b(kt_b, hb_b, stb2);
                    // (the row indices of these loads were read from LDS one slice ago -- no wait on the LDS queue, which the
                    // other wave row's fragment reads fill, in front of the MFMAs behind this point; kt_l = the NEXT slice's entry)
                    if constexpr (L) { load_a_issue(setl_tag); load_a_rows(kt_l); }
                    if constexpr (W) {
                        if (h == 0) pin(setw_tag);
#pragma unroll
                        for (int j = 0; j < HALF_PASSES; ++j) store_a(setw_tag, h, sta2, j);
                    }
                };
                if (need[0] || need[1]) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) mfma(af[PA[q]][i], bf[PB[q]][j], acc[i][j]);
                    others();
                    // interleave: one MFMA, then a few of the other instructions
#pragma unroll
                    for (int k = 0; k < 4 * NQ; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);      // 4 VALU / SALU
                        if (k % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read
                        if (k % 4 == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
                        if (k % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // 1 DS write
                    }
                } else {
                    others();
                }
                // (own LDS stores: waited for at the end of the next memory phase, one barrier before anybody reads them)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                sta = sta == 2 ? 0 : sta + 1;
                stb = stb == NB - 1 ? 0 : stb + 1;
                return;
            }
            if (need[0] || need[1]) {
#pragma unroll
                for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        af[pl][i] = *reinterpret_cast<const u32x4 *>(sa + a_rofs[i] + pl * 2 * BM * 16);
                        bf[pl][i] = *reinterpret_cast<const u32x4 *>(sb + b_rofs + pl * 2 * BN * 16 + i * 32 * 16);
                    }
            }
            if constexpr (B) load_b(kt_b, hb_b, stb2);
            if constexpr (L) load_a_rows(kt_l);
            if constexpr (W) { if (h == 0) pin(setw_tag); }
            // products a_i * b_j with i + j <= 2 in the order their fragments arrive from LDS (planes are read hi, mid, lo: the
            // hi x hi product starts after the first four reads instead of after all twelve)
            // The 24 MFMAs of the wave's 64 rows and the rest of the half-step (gathered loads, split + store) in ONE basic
            // block, so that hipcc can put the other instructions into the MFMAs' shadows; a wave whose 64 rows lack the taps of
            // the slice skips the MFMAs.  (A path per 32-row block cuts the block in three -- no interleave; an if / else over
            // "both blocks" / "one block" made hipcc keep two homes for the 64 accumulator registers and copy them every half-step.)
            auto rest = [&]() {
                if constexpr (L) load_a_issue(setl_tag);
                if constexpr (W) {
#pragma unroll
                    for (int j = 0; j < HALF_PASSES; ++j) store_a(setw_tag, h, sta2, j);
                }
            };
            if (need[0] || need[1]) {
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) mfma(af[PA[q]][i], bf[PB[q]][j], acc[i][j]);
                rest();
#pragma unroll
                for (int k = 0; k < 4 * NQ; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);      // 4 VALU / SALU
                    if (k % 4 == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read
                    if (k % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // 1 DS write
                }
            } else {
                rest();
            }
            // everything older than the last NB - 2 half-steps' loads has landed (loads complete in order); own LDS stores done
            wait_vm_lgkm0(inflight_tag);
            asm volatile("s_barrier" ::: "memory");
            sta = sta == 2 ? 0 : sta + 1;
            stb = stb == NB - 1 ? 0 : stb + 1;
        };
        // one slice (two half-steps) at position U of the unrolled steady loop: register sets by position
        constexpr int NLB = B_CHUNKS_PER_WAVE, NLA = A_PASSES;
        auto slice_steady = [&](int sl, auto u_tag) {
            constexpr int U = decltype(u_tag)::value;
            using SW = std::integral_constant<int, (U + 1) % ASETS>;
            using SL = std::integral_constant<int, U % ASETS>;
            ks_advance(sl);
            const int e = ks_at(sl);
            const int k1 = kt_at(sl + 1), kl = ks_at(sl + ASETS);
            if constexpr (PP) {
                // (the wait closes the memory phase: in flight = what the compute phase of the half-step before issued)
                // (round 5, fp16 pairs: half the MFMAs per half-step = half the time a load has to land.  Measured: a weight ring of FIVE
                // stages, four half-steps ahead, with the wait leaving the loads of the last two compute phases in flight -- five barrier
                // intervals per load instead of three: 387 vs 388 us on bcn1_'s first pass.  Load latency is not what the pair form waits for.)
                halfstep(e, 0, T{}, k1, 1, T{}, SW{}, T{}, SL{}, ks_at(sl + ASETS + 1), std::integral_constant<int, NLB>{});
                halfstep(e, 1, T{}, kt_at(sl + 2), 0, T{}, SW{}, Fl{}, SL{}, 0, std::integral_constant<int, NLB + NLA>{});
            } else
            if constexpr (NB == 3) {
                halfstep(e, 0, T{}, k1, 0, T{}, SW{}, T{}, SL{}, kl, std::integral_constant<int, NLB + NLA>{});
                halfstep(e, 1, T{}, k1, 1, T{}, SW{}, Fl{}, SL{}, 0, std::integral_constant<int, NLB>{});
            } else {
                // weight fragments three half-steps ahead: (g + 3) = slice s+1 half 1, then slice s+2 half 0
                halfstep(e, 0, T{}, k1, 1, T{}, SW{}, T{}, SL{}, kl, std::integral_constant<int, 2 * NLB + NLA>{});
                halfstep(e, 1, T{}, kt_at(sl + 2), 0, T{}, SW{}, Fl{}, SL{}, 0, std::integral_constant<int, 2 * NLB + NLA>{});
            }
        };
        // the same with its loads / stagings switched off as the list runs out; waits drain everything (tail only)
        auto slice_tail = [&](int sl, auto u_tag) {
            constexpr int U = decltype(u_tag)::value;
            using SW = std::integral_constant<int, (U + 1) % ASETS>;
            using SL = std::integral_constant<int, U % ASETS>;
            using Z = std::integral_constant<int, 0>;
            ks_advance(sl);
            const int e = ks_at(sl);
            const bool w = sl + 1 < nsl, l = sl + ASETS < nsl;
            const int kl = l ? ks_at(sl + ASETS + (PP ? 1 : 0)) : 0;      // (ping-pong: indices one slice ahead)
            // half 0
            {
                const int gb = 2 * sl + NB - 1;                 // half-step whose weight fragments are fetched now
                const bool b = gb / 2 < nsl;
                const int kb = b ? kt_at(gb / 2) : 0;
                if (b && w && l) halfstep(e, 0, T{}, kb, gb & 1, T{}, SW{}, T{}, SL{}, kl, Z{});
                else if (b && w) halfstep(e, 0, T{}, kb, gb & 1, T{}, SW{}, Fl{}, SL{}, 0, Z{});
                else if (w) halfstep(e, 0, Fl{}, 0, 0, T{}, SW{}, Fl{}, SL{}, 0, Z{});
                else if (b) halfstep(e, 0, T{}, kb, gb & 1, Fl{}, SW{}, Fl{}, SL{}, 0, Z{});
                else halfstep(e, 0, Fl{}, 0, 0, Fl{}, SW{}, Fl{}, SL{}, 0, Z{});
            }
            {
                const int gb = 2 * sl + 1 + NB - 1;
                const bool b = gb / 2 < nsl;
                const int kb = b ? kt_at(gb / 2) : 0;
                if (b && w) halfstep(e, 1, T{}, kb, gb & 1, T{}, SW{}, Fl{}, SL{}, 0, Z{});
                else if (w) halfstep(e, 1, Fl{}, 0, 0, T{}, SW{}, Fl{}, SL{}, 0, Z{});
                else if (b) halfstep(e, 1, T{}, kb, gb & 1, Fl{}, SW{}, Fl{}, SL{}, 0, Z{});
                else halfstep(e, 1, Fl{}, 0, 0, Fl{}, SW{}, Fl{}, SL{}, 0, Z{});
            }
        };
        if constexpr (PP) load_a_rows(ks_at(ASETS));      // (row indices of the first gathered loads issued inside the loop)
        int sl = 0;
        // steady state: every load / staging of the ASETS slices of an iteration exists -- no conditions inside
        for (; sl + 2 * ASETS < nsl; sl += ASETS) {
            slice_steady(sl, S0{});
            slice_steady(sl + 1, S1{});
            if constexpr (ASETS == 3) slice_steady(sl + 2, S2{});
        }
        for (; sl < nsl; sl += ASETS) {
            slice_tail(sl, S0{});
            if (sl + 1 < nsl) slice_tail(sl + 1, S1{});
            if constexpr (ASETS == 3) { if (sl + 2 < nsl) slice_tail(sl + 2, S2{}); }
        }
        if (PP && wm == 0) asm volatile("s_barrier" ::: "memory");      // (the second wave row's last compute phase)
    }
    if constexpr (PL == 2) {        // undo the operand scales: two exact multiplications by powers of two
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] * s_out * s_out2;
    }
    if constexpr (GUARD) {
        if (p.guard_trips && t == 0 && tile_m == 0 && tile_n == 0 && split == 0) atomicAdd(p.guard_trips, 1);
    }
    const bool defer = !GUARD && tripped;       // the second launch finishes this tile: store the sums, nothing else

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // Measured: with 64-bit addresses, an LDS read of the output row and (second tap-group pass) a load -> add -> store chain
    // per element, the epilogue took 30 k of a tile's 355 k cycles -- 4 700 VALU instructions.  Fast path (every operand below
    // 2 GB, the residual not wrapped): per 32 x 32 block the 16 output rows of a lane by four 16-byte LDS reads, 32-bit byte
    // offsets (v_mul_u32_u24), buffer loads / stores whose offset is out of range for rows past M, columns past N and rows
    // past rows2 -- no branches; all residual loads of a block before its stores.
    typedef int int32x4v __attribute__((ext_vector_type(4)));
    const bool fast_epi = p.epi_fast != 0;
    if (fast_epi) {
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
            p.splits > 1 ? (void *)(p.partial + (int64_t)split * p.M * p.N) : (void *)p.Y, (short)0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
            p.res ? (void *)const_cast<float *>(p.res) : (void *)p.Y, (short)0, p.res ? 0x7fffffff : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_y2 = __builtin_amdgcn_make_buffer_rsrc(
            p.Y2 ? (void *)p.Y2 : (void *)p.Y, (short)0, (p.Y2 && p.splits <= 1) ? 0x7fffffff : 0, 0x00020000);   // (split-K partials never go to Y2: k_gconv_finish writes it)
        const unsigned ldy_b = (unsigned)(p.splits > 1 ? p.N : p.ldy) * 4u, ldr_b = (unsigned)p.ldres * 4u, ldy2_b = (unsigned)p.ldy2 * 4u;
        const bool plain = p.splits <= 1;
        unsigned ymax = 0;                                   // largest |y| this lane stores (p.y_amax)
        unsigned gmin = 0xffffffffu;                         // smallest non-zero row maximum over the wave's 64 columns (p.y_guard)
        const bool want_guard = p.y_guard && p.y_amax && plain && !defer;      // (uniform)
        const bool res_wrap = p.res && p.res_mod < p.M;      // (uniform)
        const float res_inv = res_wrap ? 1.0f / (float)p.res_mod : 0.f;
        // Round 6.  (a) profiles/r06r_tile_phase_probe.txt: the epilogue took 19.4 k cycles of EVERY tile (19 % of a dense one) -- each
        // 32 x 32 block loaded its residual values (at an out-of-range offset when there is none) and waited for them behind the stores
        // of the block before: four store -> acknowledge -> load round trips per tile.  Without a residual (uniform) there are no loads
        // now, with one all of them are issued before the first store; the second destination is stored only if there is one: 13.8 k.
        // (b) The accumulators are transposed (see mfma above): lane li holds output row wm * 64 + i * 32 + li, its registers 4 q .. 4 q + 3
        // the columns 8 q + 4 hi .. + 3 of block (i, j) -- 16 accesses of 16 bytes per lane instead of 64 (+ 64) of 4.
        const bool has_res = p.res != nullptr && plain;     // (uniform; split-K partials are stored raw)
        const bool has_y2 = p.Y2 != nullptr && plain && !defer;      // (uniform)
        int mrow[2];
        unsigned yrow[2], rrow[2], y2row[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = Vs[wm * 64 + i * 32 + li];
            mrow[i] = m;
            int rr = m;
            if (res_wrap) {          // residual row = m % res_mod (the correlation layer: rows f * H + v add row v): m < 2^24 is exact in
                // fp32, the quotient by reciprocal is off by at most one
                const int q = (int)((float)rr * res_inv);
                rr -= (int)__umul24((unsigned)q, (unsigned)p.res_mod);
                rr = rr < 0 ? rr + (int)p.res_mod : (rr >= (int)p.res_mod ? rr - (int)p.res_mod : rr);
            }
            yrow[i] = __umul24((unsigned)m, ldy_b);
            rrow[i] = __umul24((unsigned)rr, ldr_b);
            y2row[i] = __umul24((unsigned)m, ldy2_b);
        }
        // byte offset of columns n .. n + 3 of a row (OOB: row past M, column past N -- N % 4 == 0 on this path)
        auto at = [&](int m, unsigned row_b, int n) { return (m >= 0 && n < p.N) ? row_b + (unsigned)n * 4u : OOB; };
        auto col_of = [&](int j, int q) { return n0 + wn * 64 + j * 32 + 8 * q + 4 * hi; };
        u32x4 rv[2][2][4];
        if (has_res) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        rv[i][j][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, (int)at(mrow[i], rrow[i], col_of(j, q)), 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned rowmax = 0;                             // largest |y| of this lane's row over its 32 columns of the wave's 64
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = col_of(j, q);
                    float4_t bs = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias && plain && n < p.N) bs = *reinterpret_cast<const float4_t *>(p.bias + n);
                    float v[4] = {acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    const float bb[4] = {bs.x, bs.y, bs.z, bs.w};
                    const unsigned rr4[4] = {rv[i][j][q].x, rv[i][j][q].y, rv[i][j][q].z, rv[i][j][q].w};
                    u32x4 o;
                    unsigned ob[4];
                    const bool live = mrow[i] >= 0 && n < p.N;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float x = v[c];
                        if (plain) {
                            x = x + bb[c];
                            if (has_res) x += __builtin_bit_cast(float, rr4[c]);
                            if (p.act == HPL_ACT_LEAKY && !defer) x = x > 0.f ? x : p.slope * x;
                        }
                        ob[c] = __builtin_bit_cast(unsigned, x);
                        const unsigned av = live ? (ob[c] & 0x7fffffffu) : 0u;
                        ymax = max(ymax, av);
                        rowmax = max(rowmax, av);
                    }
                    o.x = ob[0]; o.y = ob[1]; o.z = ob[2]; o.w = ob[3];
                    __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, (int)at(mrow[i], yrow[i], n), 0, 0);
                    if (has_y2) __builtin_amdgcn_raw_buffer_store_b128(o, rs_y2, (int)at(mrow[i] < (int)p.rows2 ? mrow[i] : -1, y2row[i], n), 0, 0);
                }
            if (want_guard) {            // the row's maximum over the wave's 64 columns: this lane's 32 and those of lane ^ 32
                rowmax = max(rowmax, (unsigned)__shfl_xor((int)rowmax, 32));
                gmin = rowmax ? min(gmin, rowmax) : gmin;
            }
        }
        if (p.y_amax && plain && !defer) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ymax = max(ymax, (unsigned)__shfl_xor((int)ymax, o));
            unsigned inv = 0u;
            if (want_guard) {           // ~bits order the other way round: the largest ~maximum is the smallest row maximum
                inv = gmin == 0xffffffffu ? 0u : ~gmin;          // (a lane: its two rows over the wave's 64 columns)
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) inv = max(inv, (unsigned)__shfl_xor((int)inv, o));
            }
            // ONE publisher per workgroup (round 6; was one per wave): the tiles of a one-round launch end together, so looking
            // before the atomic does not thin them out -- 2 048 waves queued on the two words of a slot at the end of a dense
            // 256-tile launch.  The stages of the contraction loop are dead here; the barrier in front keeps the waves that
            // still read fragments of the last half-step apart from the scratch words.
            __syncthreads();
            unsigned *red = reinterpret_cast<unsigned *>(smem);
            if (lane == 0) { red[2 * wave] = ymax; red[2 * wave + 1] = inv; }
            __syncthreads();
            if (t == 0) {
#pragma unroll
                for (int w = 1; w < NT / 64; ++w) { ymax = max(ymax, red[2 * w]); inv = max(inv, red[2 * w + 1]); }
                amax_publish(reinterpret_cast<unsigned *>(p.y_amax), ymax);
                if (want_guard) amax_publish(p.y_guard, inv);
            }
        }
    } else
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t m = Vs[wm * 64 + i * 32 + li];          // (transposed accumulators: the lane is the row, the register the column)
            if (m < 0) continue;
            const int res_mod = (int)p.res_mod;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 64 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (n >= p.N) continue;
                const float bsv = p.bias ? p.bias[n] : 0.f;
                if (p.splits > 1) {          // raw partial sum
                    p.partial[((int64_t)split * p.M + m) * p.N + n] = acc[i][j][r];
                    continue;
                }
                float v = acc[i][j][r] + bsv;
                if (p.res) v += p.res[(int64_t)((int)m < res_mod ? (int)m : (int)m % res_mod) * p.ldres + n];
                if (p.act == HPL_ACT_LEAKY && !defer) v = v > 0.f ? v : p.slope * v;
                p.Y[m * p.ldy + n] = v;
                if (p.Y2 && m < p.rows2 && !defer) p.Y2[m * p.ldy2 + n] = v;
            }
        }
    if (probe) {
        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe),
                  (unsigned long long)((long long)__builtin_readcyclecounter() - probe_c));
        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe) + 1,
                  (unsigned long long)((long long)__builtin_amdgcn_s_memrealtime() - probe_w));
    }
}

// The guard's second launch (GUARD = true) has the grid of the first and almost always leaves at once.  What it costs the pipelined
// loop then does not depend on its size (a persistent form of 8 / 32 / 128 / 1024 workgroups: 438-462 / 444-457 / 444-455 / 432-459
// pairs/s, without the guard 474-478: profiles/r06j_guard_wgs_ab.txt; the persistent form needed 253-256 registers and is gone) -- it
// is the extra dependent launch behind every wide one (second launches without the guard words: -4 %, the words without the
// launches: -1 %: profiles/r06k_guard_parts_ab.txt): in the loop such a launch takes 14.5 us against 4.8 us alone
// (profiles/r06m_queue_gaps.txt) -- this kernel needs a CU of its own (100 KB of LDS, 8 waves of 221 registers), which other pairs'
// wide tiles hold; a one-wave empty kernel behind EVERY op of the forward costs 1-2 % (profiles/r06l_dummy_launch_ab.txt).  The
// second pass as a second BODY inside the first launch, tile by tile (no launch at all), makes the first body 7-15 % slower (37-42 k
// instructions: profiles/r06n_guard_second_body_ab.txt), like the two contraction loops of profiles/r06b_*: the second launch stays.
template <int WGN, int F_LDS, int PL, int NB = 3, bool GUARD = false>
__global__ void __launch_bounds__(128 * WGN, 2) k_gconv3(const GParams p) {
    gconv3_body<WGN, F_LDS, NB, PL, GUARD>(p, blockIdx.x);
}

// the 8-wave tile (128 x 256, ping-pong wave rows): one workgroup per CU
template <int F_LDS, int NB, int PL, bool GUARD = false>
__global__ void __launch_bounds__(512, 2) k_gconv3w(const GParams p) {
    gconv3_body<4, F_LDS, NB, PL, GUARD>(p, blockIdx.x);
}

// Wt [k_rows][ldw] fp32 -> three bf16 planes [k_rows/8][ldw][8]
__global__ void __launch_bounds__(256) k_weight_split3(const float *__restrict__ Wt, int64_t k_rows, int64_t ldw,
                                                        unsigned char *__restrict__ dst, int64_t plane_stride) {
    // one thread per (kb, n): 8 strided reads (coalesced across n), one 16-byte store per plane
    const int64_t total = (k_rows / 8) * ldw;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int64_t kb = i / ldw, n = i - kb * ldw;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = Wt[(kb * 8 + j) * ldw + n];
        u32x4 h, m, l;
        unsigned a, b, c;
        split2(x[0], x[1], a, b, c); h.x = a; m.x = b; l.x = c;
        split2(x[2], x[3], a, b, c); h.y = a; m.y = b; l.y = c;
        split2(x[4], x[5], a, b, c); h.z = a; m.z = b; l.z = c;
        split2(x[6], x[7], a, b, c); h.w = a; m.w = b; l.w = c;
        *reinterpret_cast<u32x4 *>(dst + i * 16) = h;
        *reinterpret_cast<u32x4 *>(dst + plane_stride + i * 16) = m;
        *reinterpret_cast<u32x4 *>(dst + 2 * plane_stride + i * 16) = l;
    }
}

// largest magnitude of a [rows][cols] block: the bits of |x| order like unsigned integers (NaN above everything: it survives)
__device__ __forceinline__ void amax_reduce(unsigned v, unsigned *slot) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
    __shared__ unsigned part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        v = max(max(part[0], part[1]), max(part[2], part[3]));
        amax_publish(slot, v);
    }
}

// a block covers 256 >> tpr_log rows at a time, 2^tpr_log lanes across a row's columns (V = float4 / float): no division in the loops
template <typename V>
__global__ void __launch_bounds__(256) k_amax(const float *__restrict__ X, int64_t ld, int64_t rows, int colsv, int tpr_log, unsigned *__restrict__ slot) {
    constexpr int VW = sizeof(V) / 4;
    const int tpr = 1 << tpr_log, c0 = threadIdx.x & (tpr - 1), rpb = 256 >> tpr_log;
    unsigned v = 0;
    auto fold = [&](const V &x) {
        const unsigned *e = reinterpret_cast<const unsigned *>(&x);
#pragma unroll
        for (int u = 0; u < VW; ++u) v = max(v, e[u] & 0x7fffffffu);
    };
    const int64_t rstep = (int64_t)gridDim.x * rpb;
    int64_t r = (int64_t)blockIdx.x * rpb + (threadIdx.x >> tpr_log);
    // four rows in flight per lane (the matrix is read once: latency, not bandwidth, sets the pace of a plain loop)
    for (; r + 3 * rstep < rows; r += 4 * rstep) {
        const float *p0 = X + r * ld, *p1 = p0 + rstep * ld, *p2 = p1 + rstep * ld, *p3 = p2 + rstep * ld;
        for (int c = c0; c < colsv; c += tpr) {
            const V a = *reinterpret_cast<const V *>(p0 + (int64_t)c * VW), b = *reinterpret_cast<const V *>(p1 + (int64_t)c * VW);
            const V d = *reinterpret_cast<const V *>(p2 + (int64_t)c * VW), f = *reinterpret_cast<const V *>(p3 + (int64_t)c * VW);
            fold(a); fold(b); fold(d); fold(f);
        }
    }
    for (; r < rows; r += rstep) {
        const float *p0 = X + r * ld;
        for (int c = c0; c < colsv; c += tpr) fold(*reinterpret_cast<const V *>(p0 + (int64_t)c * VW));
    }
    amax_reduce(v, slot);
}

// the same with the range-guard word (hpl_amax_rows): 2^tpr_log <= 64 lanes across a row, so a row's maximum is a shuffle reduction;
// *guard = max over the rows with a non-zero entry of ~bits(row maximum)
template <typename V>
__global__ void __launch_bounds__(256) k_amax_rows(const float *__restrict__ X, int64_t ld, int64_t rows, int colsv, int tpr_log,
                                                   unsigned *__restrict__ slot, unsigned *__restrict__ guard) {
    constexpr int VW = sizeof(V) / 4;
    const int tpr = 1 << tpr_log, c0 = threadIdx.x & (tpr - 1), rpb = 256 >> tpr_log;
    unsigned v = 0, ginv = 0;
    auto fold = [&](unsigned &m, const V &x) {
        const unsigned *e = reinterpret_cast<const unsigned *>(&x);
#pragma unroll
        for (int u = 0; u < VW; ++u) m = max(m, e[u] & 0x7fffffffu);
    };
    auto row_done = [&](unsigned m) {            // m: this lane's share of a row
        for (int o = tpr >> 1; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        v = max(v, m);
        ginv = m ? max(ginv, ~m) : ginv;
    };
    const int64_t rstep = (int64_t)gridDim.x * rpb;
    // four rows of a lane group in flight; a group short of four rows re-reads its first one (max and guard are idempotent), so the
    // tail of the matrix is one round of loads as well (a single-row tail loop took a 25 841-row matrix in 1 + 3 dependent rounds:
    // 19.7 against 12.4 us for hpl_amax's pass over the same bytes).  Uniform trip counts inside a row's lane group.
    for (int64_t r = (int64_t)blockIdx.x * rpb + (threadIdx.x >> tpr_log); r < rows; r += 4 * rstep) {
        const int64_t r1 = r + rstep < rows ? r + rstep : r, r2 = r + 2 * rstep < rows ? r + 2 * rstep : r, r3 = r + 3 * rstep < rows ? r + 3 * rstep : r;
        const float *p0 = X + r * ld, *p1 = X + r1 * ld, *p2 = X + r2 * ld, *p3 = X + r3 * ld;
        unsigned m0 = 0, m1 = 0, m2 = 0, m3 = 0;
        for (int c = c0; c < colsv; c += tpr) {
            const V a = *reinterpret_cast<const V *>(p0 + (int64_t)c * VW), b = *reinterpret_cast<const V *>(p1 + (int64_t)c * VW);
            const V d = *reinterpret_cast<const V *>(p2 + (int64_t)c * VW), f = *reinterpret_cast<const V *>(p3 + (int64_t)c * VW);
            fold(m0, a); fold(m1, b); fold(m2, d); fold(m3, f);
        }
        row_done(m0); row_done(m1); row_done(m2); row_done(m3);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ginv = max(ginv, (unsigned)__shfl_xor((int)ginv, o));
    __shared__ unsigned gpart[4];
    if ((threadIdx.x & 63) == 0) gpart[threadIdx.x >> 6] = ginv;
    amax_reduce(v, slot);                       // (its barrier orders gpart too)
    if (threadIdx.x == 0) amax_publish(guard, max(max(gpart[0], gpart[1]), max(gpart[2], gpart[3])));
}

__global__ void k_amax_clear(const hpl_split3_job *__restrict__ jobs, int njobs) {
    for (int j = threadIdx.x; j < njobs; j += blockDim.x)
        if (jobs[j].planes == 2 && jobs[j].amax) *jobs[j].amax = 0.f;
}

// the images of a split batch (planes == 2 jobs): blockIdx.y = job
__global__ void __launch_bounds__(256) k_amax_batch(const hpl_split3_job *__restrict__ jobs) {
    const hpl_split3_job jb = jobs[blockIdx.y];
    unsigned v = 0;
    if (jb.planes == 2) {
        const int64_t total4 = jb.k_rows * jb.ldw / 4;      // (ldw % 4 == 0: the image is contiguous)
        const uint4 *X = reinterpret_cast<const uint4 *>(jb.Wt);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
            const uint4 x = X[i];
            v = max(max(v, x.x & 0x7fffffffu), max(max(x.y & 0x7fffffffu, x.z & 0x7fffffffu), x.w & 0x7fffffffu));
        }
    }
    amax_reduce(v, reinterpret_cast<unsigned *>(jb.amax ? jb.amax : const_cast<float *>(jb.Wt)));      // (v == 0 for other jobs: no store)
}

// Wt [k_rows][ldw] fp32 -> two fp16 planes [k_rows/8][ldw][8] of Wt * split_scale(*amax)
__device__ __forceinline__ void split2h_store(const float *x, float sc, unsigned char *dst, int64_t plane_stride, int64_t i) {
    u32x4 h, l;
    unsigned a, b;
    split2h(x[0], x[1], sc, a, b); h.x = a; l.x = b;
    split2h(x[2], x[3], sc, a, b); h.y = a; l.y = b;
    split2h(x[4], x[5], sc, a, b); h.z = a; l.z = b;
    split2h(x[6], x[7], sc, a, b); h.w = a; l.w = b;
    *reinterpret_cast<u32x4 *>(dst + i * 16) = h;
    *reinterpret_cast<u32x4 *>(dst + plane_stride + i * 16) = l;
}

__global__ void __launch_bounds__(256) k_weight_split2h(const float *__restrict__ Wt, int64_t k_rows, int64_t ldw,
                                                         unsigned char *__restrict__ dst, int64_t plane_stride,
                                                         const float *__restrict__ amax) {
    const float sc = split_scale(amax[0]);
    const int64_t total = (k_rows / 8) * ldw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t kb = i / ldw, n = i - kb * ldw;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = Wt[(kb * 8 + j) * ldw + n];
        split2h_store(x, sc, dst, plane_stride, i);
    }
}

}  // namespace

// many images in one launch (a training step re-splits ~20 images after every optimiser step): blockIdx.y = job
__global__ void __launch_bounds__(256) k_weight_split3_batch(const hpl_split3_job *__restrict__ jobs) {
    const hpl_split3_job jb = jobs[blockIdx.y];
    const float *Wt = jb.Wt;
    unsigned char *dst = reinterpret_cast<unsigned char *>(jb.dst);
    const int64_t ldw = jb.ldw, total = (jb.k_rows / 8) * ldw;
    const float sc = jb.planes == 2 ? split_scale(jb.amax[0]) : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t kb = i / ldw, n = i - kb * ldw;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = Wt[(kb * 8 + j) * ldw + n];
        if (jb.planes == 2) { split2h_store(x, sc, dst, jb.plane_stride, i); continue; }
        u32x4 h, m, l;
        unsigned a, b, c;
        split2(x[0], x[1], a, b, c); h.x = a; m.x = b; l.x = c;
        split2(x[2], x[3], a, b, c); h.y = a; m.y = b; l.y = c;
        split2(x[4], x[5], a, b, c); h.z = a; m.z = b; l.z = c;
        split2(x[6], x[7], a, b, c); h.w = a; m.w = b; l.w = c;
        *reinterpret_cast<u32x4 *>(dst + i * 16) = h;
        *reinterpret_cast<u32x4 *>(dst + jb.plane_stride + i * 16) = m;
        *reinterpret_cast<u32x4 *>(dst + 2 * jb.plane_stride + i * 16) = l;
    }
}

extern "C" int hpl_weight_split3_batch(const hpl_split3_job *jobs, int njobs, int64_t max_elems, int any_pairs, hplStream stream) {
    HPL_REQUIRE(jobs && njobs > 0 && njobs <= 65535 && max_elems > 0, "hpl_weight_split3_batch: bad arguments");
    if (any_pairs) {          // fp16-pair jobs: their largest magnitudes first
        k_amax_clear<<<1, 256, 0, to_stream(stream)>>>(jobs, njobs);
        dim3 ga((unsigned)imin(cdiv(max_elems / 4, 256), 256), (unsigned)njobs);
        k_amax_batch<<<ga, 256, 0, to_stream(stream)>>>(jobs);
    }
    dim3 grid((unsigned)imin(cdiv(max_elems / 8, 256), 2048), (unsigned)njobs);
    k_weight_split3_batch<<<grid, 256, 0, to_stream(stream)>>>(jobs);
    HPL_CHECK_LAUNCH("hpl_weight_split3_batch");
    return HPL_OK;
}

// *slot (cleared by the caller) = max(*slot, largest magnitude of the block)
int hpl_gc::amax_launch(const float *X, int64_t ld, int64_t rows, int cols, float *slot, hipStream_t s, unsigned *guard) {
    if (rows <= 0) return HPL_OK;
    const bool vec = cols % 4 == 0 && ld % 4 == 0 && aligned16(X);
    int cv = vec ? cols / 4 : cols;
    int64_t nrows = rows;
    if (guard) {            // row maxima: the rows stay rows, <= 64 lanes across one
        int tl = 0;
        while ((1 << tl) < cv && tl < 6) ++tl;
        const int rpb = 256 >> tl;
        // (grid cap, 9 reductions of a forward: 256: 145 us, 512: 127, 1024: 124, 2048: 161, 4096: 33-47 us for each of the three large ones -- a
        // one-round grid ends all at once, on the slot's two atomics)
        const int grid = (int)imax(1, imin(cdiv(nrows, (int64_t)rpb * 4), 1024));
        if (vec) k_amax_rows<float4><<<grid, 256, 0, s>>>(X, ld, nrows, cv, tl, reinterpret_cast<unsigned *>(slot), guard);
        else k_amax_rows<float><<<grid, 256, 0, s>>>(X, ld, nrows, cv, tl, reinterpret_cast<unsigned *>(slot), guard);
        HPL_CHECK_LAUNCH("hpl_amax_rows");
        return HPL_OK;
    }
    if (ld == cols && (int64_t)rows * cv >= 256) {          // contiguous: any row length will do -- 256 lanes across a "row"
        const int64_t total = rows * cv;
        int w = 256;
        while (total % w) w >>= 1;
        if (w >= 64) { cv = w; nrows = total / w; ld = (int64_t)w * (vec ? 4 : 1); }
    }
    int tpr_log = 0;
    while ((1 << tpr_log) < cv && tpr_log < 8) ++tpr_log;
    const int rpb = 256 >> tpr_log;
    const int grid = (int)imax(1, imin(cdiv(nrows, (int64_t)rpb * 4), 1024));      // (one atomic per workgroup: few, long-running workgroups)
    if (vec) k_amax<float4><<<grid, 256, 0, s>>>(X, ld, nrows, cv, tpr_log, reinterpret_cast<unsigned *>(slot));
    else k_amax<float><<<grid, 256, 0, s>>>(X, ld, nrows, cv, tpr_log, reinterpret_cast<unsigned *>(slot));
    HPL_CHECK_LAUNCH("hpl_amax");
    return HPL_OK;
}

extern "C" int hpl_amax(const float *X, int64_t ld, int64_t rows, int32_t cols, float *slot, hplStream stream) {
    HPL_REQUIRE(X && slot && rows >= 0 && cols > 0 && ld >= cols, "hpl_amax: bad arguments (rows=%lld cols=%d ld=%lld)", (long long)rows, cols, (long long)ld);
    hipStream_t s = to_stream(stream);
    if (hipMemsetAsync(slot, 0, 4, s) != hipSuccess) { set_error("hpl_amax: hipMemsetAsync failed"); return HPL_EHIP; }
    return amax_launch(X, ld, rows, cols, slot, s);
}

extern "C" int hpl_amax_rows(const float *X, int64_t ld, int64_t rows, int32_t cols, float *slot, uint32_t *guard, hplStream stream) {
    HPL_REQUIRE(X && slot && guard && rows >= 0 && cols > 0 && ld >= cols, "hpl_amax_rows: bad arguments (rows=%lld cols=%d ld=%lld)", (long long)rows, cols, (long long)ld);
    hipStream_t s = to_stream(stream);
    if (hipMemsetAsync(slot, 0, 4, s) != hipSuccess || hipMemsetAsync(guard, 0, 4, s) != hipSuccess) { set_error("hpl_amax_rows: hipMemsetAsync failed"); return HPL_EHIP; }
    return amax_launch(X, ld, rows, cols, slot, s, guard);
}

extern "C" int hpl_weight_split2h(const float *Wt, int64_t k_rows, int64_t ldw, void *dst, int64_t plane_stride, float *amax,
                                  hplStream stream) {
    HPL_REQUIRE(Wt && dst && amax && k_rows > 0 && k_rows % 8 == 0 && ldw > 0 && plane_stride >= k_rows * ldw * 2 &&
                    plane_stride % 16 == 0 && aligned16(dst),
                "hpl_weight_split2h: bad arguments (k_rows=%lld ldw=%lld)", (long long)k_rows, (long long)ldw);
    const int rc = hpl_amax(Wt, ldw, k_rows, (int32_t)ldw, amax, stream);
    if (rc) return rc;
    const int grid = (int)imin(cdiv(k_rows / 8 * ldw, 256), 16384);
    k_weight_split2h<<<grid, 256, 0, to_stream(stream)>>>(Wt, k_rows, ldw, reinterpret_cast<unsigned char *>(dst), plane_stride, amax);
    HPL_CHECK_LAUNCH("hpl_weight_split2h");
    return HPL_OK;
}

extern "C" int hpl_weight_split3(const float *Wt, int64_t k_rows, int64_t ldw, void *dst, int64_t plane_stride,
                                 hplStream stream) {
    HPL_REQUIRE(Wt && dst && k_rows > 0 && k_rows % 8 == 0 && ldw > 0 && plane_stride >= k_rows * ldw * 2 &&
                    plane_stride % 16 == 0 && aligned16(dst),
                "hpl_weight_split3: bad arguments (k_rows=%lld ldw=%lld)", (long long)k_rows, (long long)ldw);
    const int grid = (int)imin(cdiv(k_rows / 8 * ldw, 256), 16384);
    k_weight_split3<<<grid, 256, 0, to_stream(stream)>>>(Wt, k_rows, ldw, reinterpret_cast<unsigned char *>(dst), plane_stride);
    HPL_CHECK_LAUNCH("hpl_weight_split3");
    return HPL_OK;
}

// HPL_MATH=f32 keeps every launch on the fp32 MFMA (A/B runs, parity baselines)
bool hpl_gc::split3_enabled() { return split_planes() != 0; }

// HPL_MATH: f32 = fp32 MFMA everywhere; bf16x3 = the exact bf16 triples of rounds 3-4; default = fp16 pairs
int hpl_gc::split_planes() {
    static const int planes = [] {
        const char *m = getenv("HPL_MATH");
        if (m && std::string(m) == "f32") return 0;
        if (m && std::string(m) == "bf16x3") return 3;
        return 2;
    }();
    return planes;
}

bool hpl_gc::launch_split3(GParams &p, hipStream_t s) {
    // qualifying launches: row-ordered stencil passes (or dense GEMMs) of wide layers, no scatter / split-K
    if (!split3_enabled() || p.scat || p.C < 32 || p.K > 32768) return false;
    if (p.F > 15 || (p.w_bytes / (p.ldw * 4)) % 8 != 0) return false;
    constexpr int min_rows = 8192;
    // single-pass stencils of the mid-size levels (bcn3_: 9 433 rows x 15 taps x 388 channels) run faster on the fp32 kernel's
    // 64 x 64 tiles (0.31 vs 0.45 ms: a 128-row tile unites many more tap masks); dense launches gain from 8 192 rows on
    constexpr int min_rows_stencil = 16384;
    // ... unless the launch still fills most of the GPU with 128 x 256 tiles in ONE round (small clouds: at N = 2 048 points the
    // two wide Up convs are 51 x 4 and 68 x 2 tiles -- on the fp32 kernel they were 1.18 + 0.59 ms of a 3.7 ms forward)
    constexpr int fill_tiles = 128;
    constexpr int floor_rows = 1024;
    // (round 5: the narrow 15-tap stencils of levels 0-2 on the 128 x 128 tile of the pair form, N = 64: 52 -> 59, 79 -> 67, 34 -> 43 us
    // with their reduction -- no gain; N = 128, C = 132: 183 -> 100 us at level 1, a shape the model does not have)
    if (p.N < 256 || p.M < floor_rows) return false;
    const int64_t tiles256 = cdiv(p.M, BM3) * cdiv(p.N, 256);
    const bool fills = tiles256 >= fill_tiles;
    if (p.F == 1 && p.M < min_rows && !fills) return false;
    // Stencils below min_rows_stencil rows that leave most CUs without a tile (bcn3_: 74 row tiles x 1 column tile): they run here
    // only split over K into enough workgroups for one round (partial tiles in the caller's workspace), else on the fp32 kernel
    int splitk = 1;
    if (p.F > 1 && p.M < min_rows_stencil && !fills) {
        constexpr int mid_split = 1;
        constexpr int mid_min_rows = 1024;      // (round 4: from 8 192; N = 2 048 clouds +8 %, N = 8 192 unchanged)
        if (p.M < mid_min_rows) return false;
        const int64_t tiles = tiles256;
        const int nk_all = (p.K + BK - 1) / BK;
        splitk = (int)imin(imin(8, 256 / imax(1, tiles)), nk_all / 16);
        const int sets = (p.planes == 2 && p.a_guard) ? 2 : 1;       // (a guarded launch may add a second set of partial tiles)
        if (!mid_split || !p.ws || p.scat || splitk < 2 || (int64_t)sets * splitk * p.M * p.N * 4 > p.ws_bytes || p.N % 256 > 0) return false;
    }
    p.tiles_m = (int)cdiv(p.M, BM3);
    // 128 x 256 tiles (8 waves, one workgroup per CU) where N allows: 5-12 % faster than 128 x 128 on every wide launch of
    // the model (profiles/r03c_split3_kernel_ab.txt) although they leave fewer tiles per CU
    constexpr int wide = 256;
    // any N: columns past N are neither loaded (the image's row length bounds the loads) nor stored; the wider tile unless
    // its padding costs more than it gains (N = 580, the data gradient of bcn1_: 3 x 256 = 768 vs 5 x 128 = 640 columns)
    constexpr int pct = 108;      // (round 5, fp16 pairs: 135 -- the 256-wide tile for the data gradients too -- 539 -> 600 us, 332 -> 380 us)
    const bool bn256 = splitk > 1 || (wide == 256 && cdiv(p.N, 256) * 256 * 100 <= cdiv(p.N, 128) * 128 * pct);
    // (round 6: the 128-wide tile for dense launches whose 128 x 256 tiles cover at most half of the CUs -- the last 1x1 convs on the
    // points, 64 x 2 tiles; bcn3_'s dense convs, 74 x 1 -- i.e. twice the workgroups with half of the matrix-pipe work each: single
    // forward 3.12-3.24 -> 3.09-3.12 ms, pipelined rate 443-456 -> 440-451 pairs/s in one call, profiles/r06z_narrow_fill_ab.txt.
    // The gathered rows are staged twice; in the loop other pairs' launches fill the idle CUs anyway.  Not kept.)
    const int BN = bn256 ? 256 : 128;
    p.tiles_n = (int)cdiv(p.N, BN);
    if (p.tile_bm != BM3) p.tile_idx = nullptr;
    p.splits = 1; p.partial = nullptr;
    int grid = p.tiles_m * p.tiles_n;
    p.col_share = 0; p.col_rows = 0;
    if (splitk > 1) {
        p.splits = splitk;
        p.partial = p.ws;
    }
    if (p.row_perm) {
        int g = 8, b = p.tiles_n % 8;
        while (b) { const int tt = g % b; g = b; b = tt; }
        p.col_share = 8 / g;
        p.col_rows = p.col_share == 1 ? p.tiles_m : (int)cdiv(p.tiles_m, COL_CHUNK * p.col_share) * COL_CHUNK;
        grid = p.tiles_n * p.col_share * p.col_rows;
        // Round 6: ROW-major XCD order (gconv_common.h tile_coords, col_share < 0) -- the column tiles of a tile-row resident on one
        // XCD together, so that its gathered rows are fetched into that L2 once instead of once per column tile.  A/B in one call
        // (profiles/r06a_xcd_order_ab.txt, column-major -> row-major): forward passes 380 -> 378, 291 -> 287, 257 -> 255, 218 -> 215 us;
        // the data gradients (five / three 128-wide column tiles) 527 -> 432 us (bcn1_), 322 -> 305 us (bcn2_).
        constexpr int xcd_row_major = 1;
        if (xcd_row_major && p.splits <= 1 && p.tiles_n > 1 && p.tiles_n <= 8) {
            p.col_share = -1;
            p.col_rows = (int)cdiv(p.tiles_m, 8);
            grid = 8 * p.col_rows * p.tiles_n;
        }
    }
    grid *= p.splits;
    // the epilogue's 32-bit buffer addressing: every destination / residual below 2 GB, rows below 2^24, the residual not wrapped
    {
        const int64_t lim = (int64_t)0x7fffffff;
        const int64_t rows_y = p.M;
        p.epi_fast = (rows_y < (1 << 24) && rows_y * (p.splits > 1 ? p.N : p.ldy) * 4 < lim && p.ldy * 4 < (1 << 24) &&
                      (!p.res || (p.res_mod > 0 && p.res_mod < (1 << 24) && imin(p.res_mod, rows_y) * p.ldres * 4 < lim && p.ldres * 4 < (1 << 24))) &&
                      (!p.Y2 || (p.rows2 * p.ldy2 * 4 < lim && p.ldy2 * 4 < (1 << 24))) &&
                      // ... and 16 bytes per access: four consecutive columns of a row
                      p.N % 4 == 0 && (p.splits > 1 ? aligned16(p.partial) : (p.ldy % 4 == 0 && aligned16(p.Y))) && (!p.bias || aligned16(p.bias)) &&
                      (!p.res || (p.ldres % 4 == 0 && aligned16(p.res))) && (!p.Y2 || (p.ldy2 % 4 == 0 && aligned16(p.Y2)))) ? 1 : 0;
        static const int epi = getenv("HPL_SPLIT3_EPILOGUE") ? atoi(getenv("HPL_SPLIT3_EPILOGUE")) : 1;
        if (!epi) p.epi_fast = 0;
    }
    p.y_amax_done = (p.y_amax && p.epi_fast && p.splits <= 1) ? 1 : 0;      // (else hpl_gconv_forward reduces Y afterwards)
    // instances by the taps whose indices a tile stages in LDS: 1 (dense GEMMs), <= 8 (tap-group passes), <= 15
    // (a three-stage weight ring for the 256-wide tile was A/B'd in round 3 and lost: four stages stay)
    if (p.planes == 2) {
        if (bn256) {
            if (p.F == 1) k_gconv3w<1, 4, 2><<<grid, 512, 0, s>>>(p);
            else if (p.F <= 8) k_gconv3w<8, 4, 2><<<grid, 512, 0, s>>>(p);
            else k_gconv3w<15, 4, 2><<<grid, 512, 0, s>>>(p);
        } else {
            if (p.F == 1) k_gconv3<2, 1, 2><<<grid, 256, 0, s>>>(p);
            else if (p.F <= 8) k_gconv3<2, 8, 2><<<grid, 256, 0, s>>>(p);
            else k_gconv3<2, 15, 2><<<grid, 256, 0, s>>>(p);
        }
        if (p.a_guard) {
            // the guard's second launch: the same tiles on the residuals; it adds what the first launch stored and finishes.
            // (It leaves at once when the operand has no quiet row: an empty launch of this grid, 2-3 us.)
            GParams q = p;
            const int ggrid = grid;
            if (p.splits > 1) { q.partial = p.partial + (int64_t)p.splits * p.M * p.N; p.guard_partials = 1; }      // a second set of partial tiles (k_gconv_finish adds both)
            else { q.res = p.Y; q.ldres = p.ldy; q.res_mod = p.M; q.bias = nullptr; }
            if (bn256) {
                if (p.F == 1) k_gconv3w<1, 4, 2, true><<<ggrid, 512, 0, s>>>(q);
                else if (p.F <= 8) k_gconv3w<8, 4, 2, true><<<ggrid, 512, 0, s>>>(q);
                else k_gconv3w<15, 4, 2, true><<<ggrid, 512, 0, s>>>(q);
            } else {
                if (p.F == 1) k_gconv3<2, 1, 2, 3, true><<<ggrid, 256, 0, s>>>(q);
                else if (p.F <= 8) k_gconv3<2, 8, 2, 3, true><<<ggrid, 256, 0, s>>>(q);
                else k_gconv3<2, 15, 2, 3, true><<<ggrid, 256, 0, s>>>(q);
            }
        }
    } else if (bn256) {
        if (p.F == 1) k_gconv3w<1, 4, 3><<<grid, 512, 0, s>>>(p);
        else if (p.F <= 8) k_gconv3w<8, 4, 3><<<grid, 512, 0, s>>>(p);
        else k_gconv3w<15, 4, 3><<<grid, 512, 0, s>>>(p);
    } else {
        if (p.F == 1) k_gconv3<2, 1, 3><<<grid, 256, 0, s>>>(p);
        else if (p.F <= 8) k_gconv3<2, 8, 3><<<grid, 256, 0, s>>>(p);
        else k_gconv3<2, 15, 3><<<grid, 256, 0, s>>>(p);
    }
    return true;
}
