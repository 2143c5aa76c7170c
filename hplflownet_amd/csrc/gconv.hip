// gconv.hip -- the per-vertex dense contraction of the bilateral layers as a gather-GEMM on
// the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD).
//
//   Y[m, n] = act(bias[n] + res[m % res_mod, n] + sum_{f<F} sum_{c<C} A[nbr[f][m], c] * Wt[f*C + c, n])
//
// The reference materialises the gathered operand (up to 899 MB for bcn1_,
// models/bilateralNN.py:215-217) and hands it to Conv2d; here a workgroup owns a BM x BN
// output tile, walks the flat contraction index k = f*C + c in steps of 32 and gathers each
// 32-wide slice of its BM neighbour rows straight from the channel-last activation matrix
// (one 128-byte line per row per step) into LDS, stored k-major so that both MFMA operands
// are read with conflict-free ds_read_b32 (lane i of a half-wave reads element i of LDS row
// k).  Global->register loads of step t+1 are issued before the MFMAs of step t (two LDS
// buffers, one barrier per step); each f32 MFMA occupies its SIMD for 64 cycles, so one
// step is 4096 MFMA cycles per wave at 2x2 register tiling -- the loads have that long to
// land.  A missing neighbour (-1) contributes zeros, no branch in the MFMA loop.
//
// Roofline: MFMA fp32 (157.3 TFLOP/s); flops = 2*M*F*C*N.
#include "common.h"
#include "gconv_common.h"
#include <cstdlib>
#include <cmath>
#include <string>

#include <stdlib.h>
#include <type_traits>

using namespace hpl;
using namespace hpl_gc;



namespace {
// (s_setprio(3) over the tile prologue was measured in round 2, profiles/r02w_prologue_prio.txt: the prologue shrinks 21 k -> 15 k
// cycles and the co-resident workgroup's loop slows by the same amount -- zero-sum, removed.)

// COMPACT: the LDS budget of a third workgroup per CU (160 KiB / 3 = 54 613 B): A rows padded by 1 instead of 2 floats
// (the transposing store stays conflict-free: bank = 4*kq + r over kq < 8, r < 4), slice list of 512 entries (K <= 16 384).
template <int BM, int BN, int WGM, int WGN, bool AVEC, int F_LDS, bool COMPACT = false>
__global__ void __launch_bounds__(64 * WGM * WGN, COMPACT ? 6 : 1) k_gconv(const GParams p) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int LDA_S = COMPACT ? BM + 1 : BM + 2;   // +2: the transposing ds_write_b32 of 8 lanes x 4 k stay <= 2-way
    constexpr int LDB_S = BN;
    constexpr int A_ROWS_PER_PASS = NT / 8;              // 8 float4 per gathered row slice
    constexpr int A_PASSES = BM / A_ROWS_PER_PASS;
    constexpr int B_F4_PER_ROW = BN / 4;
    constexpr int B_ROWS_PER_PASS = NT / B_F4_PER_ROW;
    constexpr int B_PASSES = BK / B_ROWS_PER_PASS;
    static_assert(A_PASSES >= 1 && B_PASSES >= 1, "tile too small for the thread count");

    // one LDS array (A ring | B ring | neighbour indices of this tile, [F][BM] ints)
    // + output row of every tile row [BM] + the tile's tap mask [1])
    // + list of the contraction slices this tile needs [KLIST ushort])
    constexpr int KLIST = COMPACT ? 512 : 1024;
    static_assert(NT % BM == 0 && BM / 32 <= 4, "a thread stages indices of one 32-row block only");
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * LDA_S + 2 * BK * LDB_S + F_LDS * BM + BM + 8 + KLIST / 2];
    float *As = smem;
    float *Bs = smem + 2 * BK * LDA_S;
    int *Is = reinterpret_cast<int *>(smem + 2 * BK * LDA_S + 2 * BK * LDB_S);
    int *Vs = Is + F_LDS * BM;          // vertex (output row) of tile row r, -1 past M
    int *tapmask_s = Vs + BM;          // [0] tap mask, [1] number of needed slices, [2..5] masks of the 32-row blocks
    unsigned short *Ks = reinterpret_cast<unsigned short *>(tapmask_s + 8);

    int tile_m, tile_n;
    tile_coords(p, tile_m, tile_n);
    if (tile_m < 0) return;             // surplus workgroup of a rounded-up grid (uniform: no barrier passed yet)
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // clock probe: every 64th workgroup adds its residence in shader cycles and in 100 MHz wall ticks
    const bool probe = p.clock_probe && (blockIdx.x & 63) == 0 && t == 0;
    long long probe_c = 0, probe_w = 0;
    if (probe) {
        probe_c = (long long)__builtin_readcyclecounter();
        probe_w = (long long)__builtin_amdgcn_s_memrealtime();
    }
    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 31, hi = lane >> 5;

    // ---- A staging state: this thread always loads float4 column kq of the slice
    const int kq = t & 7;
    const int arow0 = t >> 3;
    int f_t = 0, c_t = 0;              // (f, c) of flat index k0 + kq*4, set per step
    // ---- B staging state
    const int bn4 = t % B_F4_PER_ROW;
    const int brow0 = t / B_F4_PER_ROW;

    // gathered rows (A): two register sets -- the global loads of slice t+2 are issued while slice t is
    // multiplied and slice t+1 (loaded one step earlier) is stored to LDS
    float4 ra[2][A_PASSES];

    // Stage the tile's source-row indices once: every later step reads them from LDS, so the
    // gather of step t+1 is a burst of independent loads (no global index -> address chain).
    // Rows past M, missing neighbours (-1) and taps past F all become -1.
    // With a row permutation (vertices sorted by tap mask, hpl_tap_order) tile row r is vertex
    // row_perm[m0 + r]: rows of one tile then miss the same taps, and a whole 32-wide slice whose
    // taps are absent for all BM rows is skipped (no loads, no MFMAs, no barrier).
    if (p.tile_idx) {
        // Indices and tap masks of this tile were computed once per lattice (hpl_tile_index): three independent,
        // coalesced loads and ONE barrier instead of the dependent chain row_perm -> nbr -> masks.
        const int32_t *ti = p.tile_idx + (int64_t)tile_m * p.F * BM;
        if (t < 8) tapmask_s[t] = p.tile_mask[(int64_t)tile_m * 8 + t];
        for (int r = t; r < BM; r += NT) {
            const int64_t m = m0 + r;
            Vs[r] = (m < p.M) ? (p.row_perm ? p.row_perm[m] : (int)m) : -1;
        }
        for (int i = t; i < F_LDS * BM; i += NT) Is[i] = (i < p.F * BM) ? ti[i] : -1;
    } else {
    if (t < 8) tapmask_s[t] = 0;
    for (int r = t; r < BM; r += NT) {
        const int64_t m = m0 + r;
        Vs[r] = (m < p.M) ? (p.row_perm ? p.row_perm[m] : (int)m) : -1;
    }
    __syncthreads();
    {
        int mybits = 0;
        for (int i = t; i < F_LDS * BM; i += NT) {
            const int f = i / BM, r = i - f * BM;
            const int v = Vs[r];
            const int row = (v >= 0) ? (int)src_row(p, f, v) : -1;
            Is[i] = row;
            mybits |= (row >= 0) ? (1 << f) : 0;
        }
        if (mybits) {
            atomicOr(tapmask_s, mybits);
            atomicOr(tapmask_s + 2 + ((t % BM) >> 5), mybits);      // NT % BM == 0: r = t % BM for every i of this thread
        }
    }
    }
    __syncthreads();
    const int tapmask = __builtin_amdgcn_readfirstlane(*tapmask_s);
    // A 32-row block (one MFMA tile of a wave) whose rows all miss the taps of a slice skips the MFMAs
    // of that slice (the loads, stores and the barrier stay).  Needs <= 2 taps per slice (C >= 32) so
    // that the list entry can carry them: bits 0..9 slice, 10..13 first tap, 14 "also the next tap".
    const bool blockskip = p.C >= BK;
    int bmask[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) bmask[i] = __builtin_amdgcn_readfirstlane(tapmask_s[2 + wm * TM + i]);
    // Buffer descriptors (wave-uniform, built from kernel arguments): 32-bit byte offsets keep the
    // per-load address arithmetic to a multiply-add, and an out-of-range offset returns zeros in
    // hardware -- that is how absent neighbours (-1), taps past F and columns past ldw read as 0.
    constexpr unsigned OOB = 0x80000000u;   // > any extent (< 2 GiB, checked on the host), no 32-bit wrap
    const int32x4_t rsrc_a = make_rsrc(p.A, (int)p.a_bytes);
    const int32x4_t rsrc_b = make_rsrc(p.Wt, (int)p.w_bytes);
    const unsigned lda_b = (unsigned)p.lda * 4u, ldw_b = (unsigned)p.ldw * 4u;
    const bool bvalid = (n0 + bn4 * 4) < p.ldw;
    const unsigned boff0 = bvalid ? (unsigned)brow0 * ldw_b + (unsigned)(n0 + bn4 * 4) * 4u : OOB;

    // (f0, c0) of the first element of the slice being loaded; advanced incrementally (wave-uniform)
    int f0_u = 0, c0_u = 0, k_u = 0;
    // The prefetch of the next slice is cut into small pieces that are dropped between the MFMA
    // groups of the current slice (each 4-MFMA group shadows ~256 cycles of other issue):
    //   load_begin: (f, c) bookkeeping + LDS reads of the 4 row indices
    //   load_a(i):  one gathered row slice (buffer_load_dwordx4)    load_b(i): one weight row slice
    int rows_n[A_PASSES];
    int k0_n = 0;
    auto load_begin = [&](int k0) {
        k0_n = k0;
        c0_u += k0 - k_u;       // (f, c) of this thread's float4 column in the slice starting at k0 >= k_u
        k_u = k0;
        while (c0_u >= p.C) { c0_u -= p.C; ++f0_u; }
        c_t = c0_u + kq * 4;
        f_t = f0_u;
        while (c_t >= p.C) { c_t -= p.C; ++f_t; }
        if (AVEC) {
            const int fi = min(f_t, F_LDS - 1);
#pragma unroll
            for (int i = 0; i < A_PASSES; ++i) rows_n[i] = Is[fi * BM + arow0 + i * A_ROWS_PER_PASS];
        }
    };
    auto load_a = [&](int set, int i) {
        if (AVEC) {
            const bool ok = (f_t < p.F) && (rows_n[i] >= 0);
            const unsigned off = ok ? (unsigned)rows_n[i] * lda_b + (unsigned)c_t * 4u : OOB;
            const float4_t v = buffer_load_f32x4(rsrc_a, (int)off, 0, 0);
            ra[set][i] = make_float4(v.x, v.y, v.z, v.w);
        } else {   // generic path: any C / alignment, element by element
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0_n + kq * 4 + j;
                const int f = k / p.C, c = k - f * p.C;
                const int row = (f < p.F) ? Is[min(f, F_LDS - 1) * BM + arow0 + i * A_ROWS_PER_PASS] : -1;
                e[j] = (row >= 0) ? p.A[(int64_t)row * p.lda + c] : 0.f;
            }
            ra[set][i] = make_float4(e[0], e[1], e[2], e[3]);
        }
    };
    // Weight rows go straight into LDS (buffer_load_dwordx4 ... lds): the 64 x 16 bytes of a wave are
    // whole rows of the k-major B tile, contiguous in LDS in lane order -- no staging registers, no
    // ds_write for B (measured: 3 % on the big launches).  The same for the gathered rows needs a
    // row-major, XOR-swizzled A tile whose fragment reads are 2-way bank conflicts: measured 4 %
    // *slower*, so A keeps the register path and the transposing store.
    const __amdgpu_buffer_rsrc_t lrsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.Wt), (short)0,
                                                                           (int)p.w_bytes, 0x00020000);
    (void)rsrc_b;
    const int wave_brow0 = (wave * 64) / B_F4_PER_ROW;         // first tile row of this wave's 64 lanes
    auto load_b_lds = [&](int buf, int k0, int i) {
        const unsigned off = bvalid ? boff0 + (unsigned)(k0 + i * B_ROWS_PER_PASS) * ldw_b : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            lrsrc_b,
            (__attribute__((address_space(3))) void *)(Bs + buf * BK * LDB_S + (wave_brow0 + i * B_ROWS_PER_PASS) * LDB_S),
            16, (int)off, 0, 0, 0);
    };
    auto load_regs = [&](int set, int k0) {
        load_begin(k0);
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) load_a(set, i);
    };

    auto store_a = [&](int set, int buf, int i) {
        float *a = As + buf * BK * LDA_S;
        const int r = arow0 + i * A_ROWS_PER_PASS;
        a[(kq * 4 + 0) * LDA_S + r] = ra[set][i].x;
        a[(kq * 4 + 1) * LDA_S + r] = ra[set][i].y;
        a[(kq * 4 + 2) * LDA_S + r] = ra[set][i].z;
        a[(kq * 4 + 3) * LDA_S + r] = ra[set][i].w;
    };
    auto store_lds = [&](int set, int buf) {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) store_a(set, buf, i);
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    // Slice kt covers taps f_lo..f_hi (at most two when C >= 32); it is needed iff one of them is
    // present for some row of the tile.  Wave 0 compacts the needed slices into Ks (ballot prefix).
    if (wave == 0) {
        int count = 0;
        for (int base = 0; base < nk; base += 64) {
            const int kt = base + lane;
            bool need = false;
            if (kt < nk) {
                const int f_lo = (kt * BK) / p.C;
                const int f_hi = min((kt * BK + BK - 1) / p.C, p.F - 1);
                int bits = 0;
                for (int f = f_lo; f <= f_hi; ++f) bits |= 1 << f;
                need = (tapmask & bits) != 0;
            }
            const unsigned long long bal = __ballot(need);
            if (need) {
                int e = kt;
                if (blockskip) {
                    const int f_lo = (kt * BK) / p.C;
                    e |= (f_lo << 10) | (((kt * BK + BK - 1) / p.C > f_lo && f_lo + 1 < p.F) ? (1 << 14) : 0);
                }
                Ks[count + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)e;
            }
            count += __popcll(bal);
        }
        if (lane == 0) tapmask_s[1] = count;
    }
    __syncthreads();
    const int nlist = __builtin_amdgcn_readfirstlane(tapmask_s[1]);
    // split-K (small M): this workgroup handles slices [lo, hi) of the list and writes a partial tile
    const int split = p.splits > 1 ? blockIdx.x / (p.tiles_m * p.tiles_n) : 0;    // (a rounded-up grid has splits == 1)
    // The cut points are slice INDICES (split s owns slices s*nk/splits .. (s+1)*nk/splits of the full K range, whatever
    // this tile's list keeps of them): a row's partial sums then cover the same k ranges in every tile it can land in,
    // so the result does not depend on the row order (which the lattice build fills with atomics).
    int lo = 0, hi_i = nlist;
    if (p.splits > 1) {
        const int k_lo = (int)((int64_t)nk * split / p.splits), k_hi = (int)((int64_t)nk * (split + 1) / p.splits);
        auto below = [&](int kt) {           // list entries with slice index < kt (the list is ascending)
            int a = 0, b = nlist;
            while (a < b) {
                const int mid = (a + b) >> 1;
                if ((int)(Ks[mid] & 1023) < kt) a = mid + 1; else b = mid;
            }
            return a;
        };
        lo = below(k_lo);
        hi_i = below(k_hi);
    }
    const int nsl = hi_i - lo;        // slices of this workgroup: list entries lo .. hi_i-1
    if (nsl > 0) {
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) load_b_lds(0, (int)(Ks[lo] & 1023) * BK, i);
        load_regs(0, (int)(Ks[lo] & 1023) * BK);
        store_lds(0, 0);
        if (nsl > 1) load_regs(1, (int)(Ks[lo + 1] & 1023) * BK);      // slice 1 stays in flight in set 1
    }
    // the LDS-direct weight rows and the staged slice have landed; the register loads of slice 1 (issued last,
    // loads complete in order) stay in flight across the barrier
    if (nsl > 1) {
        constexpr int inflight = A_PASSES;
        __builtin_amdgcn_s_waitcnt((inflight & 0xF) | ((inflight >> 4) << 14) | (0x7 << 4) | (0x0 << 8));
    } else {
        __builtin_amdgcn_s_waitcnt(0);
    }
    __syncthreads();
    int cur = 0;
    // One contraction step t (compile-time flags).  P = t & 1.  LOAD: issue the loads of slice t+2
    // (kt_load) into register set P; STORE: stage slice t+1 (register set 1-P, loaded during step t-1)
    // into the other LDS buffer.
    auto step = [&](int e_cur, int kt_load, int kt_b, auto load_tag, auto store_tag, auto parity_tag) {
        constexpr bool do_load = decltype(load_tag)::value;
        constexpr bool do_store = decltype(store_tag)::value;
        constexpr int P = decltype(parity_tag)::value;
        // the prefetch / staging pieces of this step, one call per k-pair
        auto pieces = [&](int kk) {
            // weights of slice t+1 (kt_b) straight into the free LDS buffer, first in the queue so that
            // the end-of-step wait can leave the register loads of slice t+2 in flight
            if constexpr (do_store) {
                if (kk >= 1 && kk - 1 < B_PASSES) load_b_lds(cur ^ 1, kt_b * BK, kk - 1);
            }
            if constexpr (do_load) {
                if (kk == 0) load_begin(kt_load * BK);
                if (kk >= 6 && kk - 6 < A_PASSES) load_a(P, kk - 6);
            }
            if constexpr (do_store) {
                if (kk >= 10 && kk - 10 < A_PASSES) store_a(1 - P, cur ^ 1, kk - 10);
            }
            if (kk == BK / 2 - 1) {
                constexpr int inflight = do_load ? A_PASSES : 0;      // vmcnt <= inflight: the LDS-direct rows landed
                __builtin_amdgcn_s_waitcnt((inflight & 0xF) | ((inflight >> 4) << 14) | (0x7 << 4) | (0xF << 8));
            }
        };
        bool need[TM], need_any = false;
        {
            const int f_lo = (e_cur >> 10) & 15, two = (e_cur >> 14) & 1;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                need[i] = !blockskip || (((bmask[i] >> f_lo) | (two ? (bmask[i] >> (f_lo + 1)) : 0)) & 1);
                need_any |= need[i];
            }
        }
        if (!need_any) {          // wave-uniform: none of this wave's row blocks has a tap of this slice
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) pieces(kk);
            __syncthreads();
            cur ^= 1;
            return;
        }
        const float *a = As + cur * BK * LDA_S + wm * WTM + li;
        const float *b = Bs + cur * BK * LDB_S + wn * WTN + li;
        // fragments of k-pair kk+1 are fetched from LDS while the MFMAs of kk run
        float av[2][TM], bv[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) av[0][i] = a[hi * LDA_S + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[0][j] = b[hi * LDB_S + j * 32];
        // Everything that is not an MFMA is placed in the shadow of the 64-cycle MFMAs of this
        // step: the address arithmetic + global loads of step t+1 right after the first k-pair,
        // the LDS stores of those loads during the second half (the loads have had ~2000 cycles).
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            if (kk + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) av[(kk + 1) & 1][i] = a[((kk + 1) * 2 + hi) * LDA_S + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[(kk + 1) & 1][j] = b[((kk + 1) * 2 + hi) * LDB_S + j * 32];
            }
            // keep the ds_reads of kk+1 ahead of the MFMAs of kk (hipcc otherwise sinks them below
            // the MFMAs and waits lgkmcnt(0) right after: ~50 idle matrix-pipe cycles per k-pair)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (TM > 1 && !need[i]) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1][i], bv[kk & 1][j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            pieces(kk);
        }
        __syncthreads();
        cur ^= 1;
    };
    {
        using T = std::true_type;
        using F = std::false_type;
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        int t = 0;
        while (t + 2 < nsl) {                       // steps that both load (slice t+2) and store (slice t+1)
            step((int)Ks[lo + t], (int)(Ks[lo + t + 2] & 1023), (int)(Ks[lo + t + 1] & 1023), T{}, T{}, P0{});
            ++t;
            if (t + 2 >= nsl) break;
            step((int)Ks[lo + t], (int)(Ks[lo + t + 2] & 1023), (int)(Ks[lo + t + 1] & 1023), T{}, T{}, P1{});
            ++t;
        }
        if (t + 1 < nsl) {                          // last but one: only stage the last slice
            if (t & 1) step((int)Ks[lo + t], -1, (int)(Ks[lo + t + 1] & 1023), F{}, T{}, P1{});
            else step((int)Ks[lo + t], -1, (int)(Ks[lo + t + 1] & 1023), F{}, T{}, P0{});
            ++t;
        }
        if (t < nsl) step((int)Ks[lo + t], -1, -1, F{}, F{}, P0{});
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // Fast form (gconv3.hip's: every operand below 2 GB, residual not wrapped, no scatter): output rows by 16-byte LDS reads,
    // 32-bit byte offsets, out-of-range offsets instead of branches, the residual loads of a block before its stores.  The
    // generic form below pays a 64-bit modulo and a load -> add -> store chain per element.
    if (p.epi_fast && !p.scat) {
        typedef int int32x4v __attribute__((ext_vector_type(4)));
        constexpr unsigned OOB_E = 0x80000000u;
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
            p.splits > 1 ? (void *)(p.partial + (int64_t)split * p.M * p.N) : (void *)p.Y, (short)0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
            p.res ? (void *)const_cast<float *>(p.res) : (void *)p.Y, (short)0, p.res ? 0x7fffffff : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_y2 = __builtin_amdgcn_make_buffer_rsrc(
            p.Y2 ? (void *)p.Y2 : (void *)p.Y, (short)0, (p.Y2 && p.splits <= 1) ? 0x7fffffff : 0, 0x00020000);   // (split-K partials never go to Y2: k_gconv_finish writes it)
        const unsigned ldy_b = (unsigned)(p.splits > 1 ? p.N : p.ldy) * 4u, ldr_b = (unsigned)p.ldres * 4u, ldy2_b = (unsigned)p.ldy2 * 4u;
        const bool plain = p.splits <= 1;
        const bool res_wrap = p.res && p.res_mod < p.M;      // (uniform)
        const float res_inv = res_wrap ? 1.0f / (float)p.res_mod : 0.f;
        const bool has_res = p.res != nullptr && plain, has_y2 = p.Y2 != nullptr && plain;      // (uniform)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 32 + li;
                const unsigned nb = n < p.N ? (unsigned)n * 4u : OOB_E;      // (a column past N keeps every sum below out of range)
                const float bsv = (p.bias && plain && n < p.N) ? p.bias[n] : 0.f;
                int mrow[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int32x4v mv = *reinterpret_cast<const int32x4v *>(Vs + wm * WTM + i * 32 + 8 * q + 4 * hi);
                    mrow[4 * q + 0] = mv.x; mrow[4 * q + 1] = mv.y; mrow[4 * q + 2] = mv.z; mrow[4 * q + 3] = mv.w;
                }
                // (round 6: a launch without a residual issues no residual loads -- it used to load 16 out-of-range words and wait for them
                // in front of its stores, a memory round trip per block for nothing -- and one without a second destination no second
                // stores: csrc/gconv3.hip's epilogue, profiles/r06r_tile_phase_probe.txt)
                float rv[16];
                if (has_res) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int rr = mrow[r];
                        if (res_wrap) {          // residual row = m % res_mod (the correlation layer: rows f * H + v add row v): m < 2^24 is
                            // exact in fp32, the quotient by reciprocal is off by at most one
                            const int q = (int)((float)rr * res_inv);
                            rr -= (int)__umul24((unsigned)q, (unsigned)p.res_mod);
                            rr = rr < 0 ? rr + (int)p.res_mod : (rr >= (int)p.res_mod ? rr - (int)p.res_mod : rr);
                        }
                        const unsigned ro = mrow[r] >= 0 ? __umul24((unsigned)rr, ldr_b) + nb : OOB_E;
                        rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, (int)ro, 0, 0));
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r];
                    if (plain) {
                        v = v + bsv;
                        if (has_res) v += rv[r];
                        if (p.act == HPL_ACT_LEAKY) v = v > 0.f ? v : p.slope * v;
                    }
                    const unsigned yo = mrow[r] >= 0 ? __umul24((unsigned)mrow[r], ldy_b) + nb : OOB_E;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_y, (int)yo, 0, 0);
                    if (has_y2) {
                        const unsigned y2o = (mrow[r] >= 0 && mrow[r] < (int)p.rows2) ? __umul24((unsigned)mrow[r], ldy2_b) + nb : OOB_E;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_y2, (int)y2o, 0, 0);
                    }
                }
            }
    } else
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WTN + j * 32 + li;
            if (n >= p.N) continue;
            const float bsv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = Vs[wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi];   // output row
                if (m < 0) continue;
                if (p.splits > 1) {   // raw partial sum; k_gconv_finish adds them in split order
                    p.partial[((int64_t)split * p.M + m) * p.N + n] = acc[i][j][r];
                    continue;
                }
                float v = acc[i][j][r] + bsv;
                if (p.res) v += p.res[(m % p.res_mod) * p.ldres + n];
                if (p.act == HPL_ACT_LEAKY) v = v > 0.f ? v : p.slope * v;
                if (p.scat) {
                    const int k = n / p.scat_c, c = n - k * p.scat_c;
                    const int32_t tgt = p.scat[(int64_t)k * p.scat_stride + m];
                    if (tgt >= 0) atomicAdd(p.Y + (int64_t)tgt * p.ldy + c, v);
                } else {
                    p.Y[m * p.ldy + n] = v;
                    if (p.Y2 && m < p.rows2) p.Y2[m * p.ldy2 + n] = v;
                }
            }
        }
    if (probe) {
        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe),
                  (unsigned long long)((long long)__builtin_readcyclecounter() - probe_c));
        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe) + 1,
                  (unsigned long long)((long long)__builtin_amdgcn_s_memrealtime() - probe_w));
    }
}

// split-K epilogue: Y = act(bias + res + sum_s partial[s]) in fixed split order
// (a split-K launch of the pair form whose range guard tripped left a second set of partial tiles -- the residual pass -- behind the first)
__global__ void k_gconv_finish(const GParams p) {
    const int nsp = (p.guard_partials && guard_tripped(p.a_amax, p.a_guard)) ? 2 * p.splits : p.splits;
    const int64_t total = p.M * p.N;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int64_t m = i / p.N;
        const int n = (int)(i - m * p.N);
        float acc = 0.f;
        for (int sidx = 0; sidx < nsp; ++sidx) acc += p.partial[(int64_t)sidx * total + i];
        float v = acc + (p.bias ? p.bias[n] : 0.f);
        if (p.res) v += p.res[(m % p.res_mod) * p.ldres + n];
        if (p.act == HPL_ACT_LEAKY) v = v > 0.f ? v : p.slope * v;
        p.Y[m * p.ldy + n] = v;
        if (p.Y2 && m < p.rows2) p.Y2[m * p.ldy2 + n] = v;
    }
}

// the same, four columns per thread, 32-bit index arithmetic (N % 4 == 0, every row 16-byte aligned, M * N < 2^31): the additions in
// the same order, so the same bits
__global__ void __launch_bounds__(256) k_gconv_finish4(const GParams p) {
    const unsigned N4 = (unsigned)p.N / 4u, total4 = (unsigned)p.M * N4;
    const int64_t total = p.M * p.N;
    const unsigned res_mod = (unsigned)p.res_mod;
    const int nsp = (p.guard_partials && guard_tripped(p.a_amax, p.a_guard)) ? 2 * p.splits : p.splits;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total4; i += gridDim.x * 256u) {
        const unsigned m = i / N4, n = (i - m * N4) * 4u;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sidx = 0; sidx < nsp; ++sidx) {
            const float4 v = *reinterpret_cast<const float4 *>(p.partial + (int64_t)sidx * total + (int64_t)i * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (p.bias) {
            const float4 b = *reinterpret_cast<const float4 *>(p.bias + n);
            acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
        }
        if (p.res) {
            const unsigned mr = m < res_mod ? m : m % res_mod;
            const float4 r = *reinterpret_cast<const float4 *>(p.res + (int64_t)mr * p.ldres + n);
            acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
        }
        if (p.act == HPL_ACT_LEAKY) {
            acc.x = acc.x > 0.f ? acc.x : p.slope * acc.x;
            acc.y = acc.y > 0.f ? acc.y : p.slope * acc.y;
            acc.z = acc.z > 0.f ? acc.z : p.slope * acc.z;
            acc.w = acc.w > 0.f ? acc.w : p.slope * acc.w;
        }
        *reinterpret_cast<float4 *>(p.Y + (int64_t)m * p.ldy + n) = acc;
        if (p.Y2 && m < (unsigned)p.rows2) *reinterpret_cast<float4 *>(p.Y2 + (int64_t)m * p.ldy2 + n) = acc;
    }
}

// one thread per output element; sequential fmaf chain in k order (what one MFMA lane does)
__global__ void k_gconv_naive(const GParams p) {
    const int64_t total = p.M * p.N;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int64_t m = i / p.N;
        const int n = (int)(i - m * p.N);
        float acc = 0.f;
        for (int f = 0; f < p.F; ++f) {
            const int64_t row = src_row(p, f, m);
            if (row < 0) continue;
            const float *a = p.A + row * p.lda;
            const float *w = p.Wt + (int64_t)f * p.C * p.ldw + n;
            for (int c = 0; c < p.C; ++c) acc = fmaf(a[c], w[(int64_t)c * p.ldw], acc);
        }
        float v = acc + (p.bias ? p.bias[n] : 0.f);
        if (p.res) v += p.res[(m % p.res_mod) * p.ldres + n];
        if (p.act == HPL_ACT_LEAKY) v = v > 0.f ? v : p.slope * v;
        if (p.scat) {
            const int k = n / p.scat_c, c = n - k * p.scat_c;
            const int32_t tgt = p.scat[(int64_t)k * p.scat_stride + m];
            if (tgt >= 0) atomicAdd(p.Y + (int64_t)tgt * p.ldy + c, v);
        } else {
            p.Y[m * p.ldy + n] = v;
            if (p.Y2 && m < p.rows2) p.Y2[m * p.ldy2 + n] = v;
        }
    }
}

}  // namespace

int hpl_gc::fill_params(const hpl_gconv_desc *d, GParams &p, const char *who) {
    HPL_REQUIRE(d, "%s: null descriptor", who);
    HPL_REQUIRE(d->A && d->Wt && d->Y, "%s: null A / Wt / Y", who);
    HPL_REQUIRE(d->M >= 0 && d->C > 0 && d->F > 0 && d->N > 0, "%s: bad sizes M=%lld C=%d F=%d N=%d", who,
                (long long)d->M, d->C, d->F, d->N);
    HPL_REQUIRE(d->F <= 15, "%s: F=%d taps (radius > 1) not supported by the LDS-staged index table", who, d->F);
    HPL_REQUIRE(d->rows_a > 0 && d->rows_a < (int64_t)INT32_MAX, "%s: rows_a out of range", who);
    HPL_REQUIRE((int64_t)d->F * d->C <= 32768, "%s: contraction length F*C = %lld > 32768 (slice list in LDS)", who,
                (long long)d->F * d->C);
    HPL_REQUIRE(d->lda >= d->C, "%s: lda %lld < C %d", who, (long long)d->lda, d->C);
    HPL_REQUIRE(d->ldw >= d->N && d->ldw % 4 == 0 && aligned16(d->Wt), "%s: ldw must be a multiple of 4 >= N and Wt 16-byte aligned", who);
    HPL_REQUIRE(d->nbr || d->F == 1 || d->reg_stride > 0, "%s: F > 1 needs a neighbour table or reg_stride", who);
    HPL_REQUIRE(!d->res || (d->res_mod > 0 && d->ldres >= d->N), "%s: bad residual description", who);
    HPL_REQUIRE(!d->scat || d->scat_c > 0, "%s: bad scatter description", who);
    HPL_REQUIRE(d->scat || d->ldy >= d->N, "%s: ldy %lld < N %d", who, (long long)d->ldy, d->N);
    HPL_REQUIRE(d->act == HPL_ACT_NONE || d->act == HPL_ACT_LEAKY, "%s: unknown activation %d", who, d->act);
    p.A = d->A; p.lda = d->lda; p.rows_a = d->rows_a;
    p.nbr = d->nbr; p.nbr_stride = d->nbr_stride; p.reg_stride = d->reg_stride;
    p.M = d->M; p.C = d->C; p.F = d->F; p.K = d->F * d->C;
    p.Wt = d->Wt; p.ldw = d->ldw; p.N = d->N; p.act = d->act; p.slope = d->slope;
    p.bias = d->bias; p.res = d->res; p.ldres = d->ldres; p.res_mod = d->res_mod;
    p.Y = d->Y; p.ldy = d->ldy;
    p.Y2 = d->Y2; p.ldy2 = d->ldy2; p.rows2 = d->Y2 ? d->rows2 : 0;
    HPL_REQUIRE(!d->Y2 || (!d->scat && d->ldy2 >= d->N && d->rows2 >= 0), "%s: bad second destination", who);
    p.scat = d->scat; p.scat_stride = d->scat_stride; p.scat_c = d->scat_c;
    p.row_perm = d->row_perm;
    p.tile_idx = (d->tile_idx && d->tile_mask && d->row_perm) ? d->tile_idx : nullptr;
    p.tile_mask = d->tile_mask;
    p.tile_bm = d->tile_bm;
    p.clock_probe = reinterpret_cast<long long *>(d->clock_probe);
    p.perm_chunk = 4;      // measured on bcn1_/bcn2_ (64-row tiles): 1: 2.61/1.33 ms, 2: 2.73/1.31, 4: 2.62/1.26, 8: 2.66/1.26, 16: 2.96/1.27
    p.ws = d->ws; p.ws_bytes = d->ws ? d->ws_bytes : 0; p.splits = 1; p.partial = nullptr;
    p.Wt3 = d->Wt3; p.w3_plane_stride = d->wt3_plane_stride;
    p.planes = d->wt3_planes == 2 ? 2 : 3; p.a_amax = d->a_amax; p.w_amax = d->w_amax;
    HPL_REQUIRE(!d->Wt3 || d->wt3_planes == 0 || d->wt3_planes == 2 || d->wt3_planes == 3, "%s: wt3_planes = %d", who, d->wt3_planes);
    if (p.planes == 2 && !(p.a_amax && p.w_amax)) p.Wt3 = nullptr;      // fp16 pairs need both scales: the launch stays on the fp32 MFMA
    p.y_amax = d->y_amax; p.y_amax_done = 0;
    p.a_guard = d->a_guard; p.y_guard = d->y_guard; p.guard_trips = d->guard_trips; p.guard_partials = 0;
    HPL_REQUIRE(!(d->y_amax && d->scat), "%s: y_amax with a scatter epilogue", who);
    p.col_share = 0; p.col_rows = 0;
    p.tiles_m = p.tiles_n = 0;
    p.a_bytes = ((d->rows_a - 1) * d->lda + d->C) * 4;
    const int64_t w_rows = d->w_rows ? (int64_t)d->w_rows : cdiv(p.K, 32) * 32;
    HPL_REQUIRE(w_rows >= p.K, "%s: w_rows=%lld < F*C=%d", who, (long long)w_rows, p.K);
    p.w_bytes = imin(w_rows, cdiv(p.K, 32) * 32) * d->ldw * 4;
    HPL_REQUIRE(p.a_bytes < (int64_t)INT32_MAX && p.w_bytes < (int64_t)INT32_MAX,
                "%s: A or Wt spans >= 2 GiB (32-bit buffer offsets)", who);
    return HPL_OK;
}

namespace {
template <int BM, int BN, int WGM, int WGN>
void launch_cfg(GParams &p, bool avec, hipStream_t s) {
    p.tiles_m = (int)cdiv(p.M, BM);
    p.tiles_n = (int)cdiv(p.N, BN);
    if (p.tile_bm != BM || p.F == 1) p.tile_idx = nullptr;      // the table was cut for another tile height
    // Small M: too few tiles to fill 256 CUs and a long serial slice loop (one exposed memory
    // latency per slice).  Split the slice list over up to 16 workgroups per tile, partial tiles
    // go to the caller's workspace and are summed in fixed order by k_gconv_finish.
    p.splits = 1;
    const int tiles = p.tiles_m * p.tiles_n;
    const int nk = (p.K + BK - 1) / BK;
    if (p.ws && !p.scat && tiles <= 128 && nk >= 8) {
        int sp = (int)imin(imin(16, nk / 4), imax(1, 512 / tiles));
        while (sp > 1 && (int64_t)sp * p.M * p.N * 4 > p.ws_bytes) --sp;
        p.splits = sp;
        p.partial = p.ws;
    } else if (p.ws && !p.scat && nk >= 32) {
        // Mid-size launches: a few hundred tiles on 256 CUs leave the chip unevenly loaded (544 tiles of a level-0 conv
        // on dense data = 2.1 per CU: the CUs holding three set the time, 71 %).  Splitting K multiplies the work items;
        // pick the split count with the best (load balance) x (1 - cost of writing and re-reading the partial tiles).
        constexpr int mid = 1;
        constexpr int WG_PER_CU = 16 / (WGM * WGN);                    // four waves per SIMD fill a CU
        const double slots = 256.0 * WG_PER_CU;
        if (mid && tiles < 6 * slots) {
            double best = 0.0;
            int best_sp = 1;
            for (int sp = 1; sp <= 16 && sp * 8 <= nk; ++sp) {
                if (sp > 1 && (int64_t)sp * p.M * p.N * 4 > p.ws_bytes) break;
                const double rounds = tiles * sp / slots;
                const double eff = rounds / std::ceil(rounds);
                // partial tiles: 8 B per element and split through HBM (~4 TB/s) against 2*K flop per element (~100 TF)
                const double cost = sp == 1 ? 0.0 : 100.0 * sp / (double)p.K;
                const double score = eff * (1.0 - cost);
                if (score > best * 1.03) { best = score; best_sp = sp; }      // (a larger split must pay for itself)
            }
            if (best_sp > 1) { p.splits = best_sp; p.partial = p.ws; }
        }
    }
    int grid = tiles * p.splits;
    {   // the epilogue's 32-bit buffer addressing (see k_gconv): every destination / residual below 2 GB, rows below 2^24
        const int64_t lim = (int64_t)0x7fffffff;
        static const int epi = getenv("HPL_GCONV_EPILOGUE") ? atoi(getenv("HPL_GCONV_EPILOGUE")) : 1;
        p.epi_fast = (epi && !p.scat && p.M < (1 << 24) && p.M * (p.splits > 1 ? p.N : p.ldy) * 4 < lim && p.ldy * 4 < (1 << 24) &&
                      (!p.res || (p.res_mod > 0 && p.res_mod < (1 << 24) && imin(p.res_mod, p.M) * p.ldres * 4 < lim && p.ldres * 4 < (1 << 24))) &&
                      (!p.Y2 || (p.rows2 * p.ldy2 * 4 < lim && p.ldy2 * 4 < (1 << 24)))) ? 1 : 0;
    }
    p.col_share = 0; p.col_rows = 0;
    constexpr int col_order = 1;
    if (col_order && p.row_perm && p.splits == 1) {
        // every column tile is cut into 8 / gcd(8, tiles_n) interleaved runs of tile-rows, so that the runs
        // divide evenly among the 8 XCDs (tiles_n = 8: one whole column per XCD; 4: two XCDs per column;
        // 5: 40 runs, five per XCD, each XCD inside at most two columns)
        int g = 8, b = p.tiles_n % 8;
        while (b) { const int t = g % b; g = b; b = t; }
        p.col_share = 8 / g;
        p.col_rows = p.col_share == 1 ? p.tiles_m : (int)cdiv(p.tiles_m, COL_CHUNK * p.col_share) * COL_CHUNK;
        grid = p.tiles_n * p.col_share * p.col_rows;
    }
    if constexpr (BM == 64 && BN == 128 && WGM == 2 && WGN == 4) {
        // three workgroups per CU (COMPACT LDS budget, 52.8 KB) for the tap-group passes of the big stencil launches: a
        // third workgroup covers the dispatch gaps and prologues of the other two -- slots occupied 0.91 -> 0.93 of 768,
        // dominant launch 816 -> 789 us alone (profiles/r02y_wg3.txt); in the three-stream pipeline the throughput is
        // unchanged (the GPU is matrix-pipe bound there).
        if (avec && p.F > 1 && p.F <= 8 && nk <= 512 && p.splits == 1) {      // (a tap group: <= 8 taps staged)
            k_gconv<BM, BN, WGM, WGN, true, 8, true><<<grid, 64 * WGM * WGN, 0, s>>>(p);
            return;
        }
    }
    // F_LDS = taps whose indices are staged in LDS: 1 for dense GEMMs, 15 for the radius-1 stencil
    if (p.F == 1) {
        if (avec) k_gconv<BM, BN, WGM, WGN, true, 1><<<grid, 64 * WGM * WGN, 0, s>>>(p);
        else k_gconv<BM, BN, WGM, WGN, false, 1><<<grid, 64 * WGM * WGN, 0, s>>>(p);
    } else {
        if (avec) k_gconv<BM, BN, WGM, WGN, true, 15><<<grid, 64 * WGM * WGN, 0, s>>>(p);
        else k_gconv<BM, BN, WGM, WGN, false, 15><<<grid, 64 * WGM * WGN, 0, s>>>(p);
    }
}

}  // namespace

namespace {
void launch_finish(const GParams &p, hipStream_t s) {
    const bool vec = p.N % 4 == 0 && p.ldy % 4 == 0 && aligned16(p.Y) && aligned16(p.partial) && (!p.bias || aligned16(p.bias)) &&
                     (!p.res || (p.ldres % 4 == 0 && aligned16(p.res))) && (!p.Y2 || (p.ldy2 % 4 == 0 && aligned16(p.Y2))) &&
                     p.M * p.N < (int64_t)0x7fffffff && p.res_mod < (int64_t)0x7fffffff;
    if (vec) {
        const int g = (int)imin(cdiv(p.M * (p.N / 4), 256), 2048);
        k_gconv_finish4<<<g, 256, 0, s>>>(p);
    } else {
        const int g = (int)imin(cdiv(p.M * p.N, 256), 2048);
        k_gconv_finish<<<g, 256, 0, s>>>(p);
    }
}
}  // namespace

extern "C" int hpl_gconv_forward(const hpl_gconv_desc *d, hplStream stream) {
    GParams p;
    int rc = fill_params(d, p, "hpl_gconv_forward");
    if (rc != HPL_OK) return rc;
    if (p.M == 0) return HPL_OK;
    hipStream_t s = to_stream(stream);
    const bool avec = (p.C % 4 == 0) && (p.lda % 4 == 0) && aligned16(p.A);
    if (avec && p.Wt3 && launch_split3(p, s)) {
        if (p.splits > 1) launch_finish(p, s);       // (mid-size stencils: partial tiles over slice ranges, summed in fixed order)
        HPL_CHECK_LAUNCH("hpl_gconv_forward");
        if (p.y_amax && !p.y_amax_done) return amax_launch(p.Y, p.ldy, p.M, p.N, p.y_amax, s, p.y_guard);
        return HPL_OK;
    }
    // Tile selection.
    const int64_t t128 = cdiv(p.M, 128), t64 = cdiv(p.M, 64);
    const int nsel = p.N;
    if (nsel > 64) {
        // Measured on the model's shapes (a sweep of ten tile configurations, rounds 1-2): 4 waves per SIMD beat
        // bigger tiles everywhere -- 64x128 with 8 waves (2 workgroups per CU) when it yields >= 512
        // tiles, else 64x64 with 4 waves (4 workgroups per CU); 64-row tiles also skip more absent
        // taps than 128-row tiles (58.6 % vs 62.5 % of the slices executed on bcn1_).
        const int64_t tn = cdiv(p.N, 128);
        if (t64 * tn >= 512) launch_cfg<64, 128, 2, 4>(p, avec, s);
        else launch_cfg<64, 64, 2, 2>(p, avec, s);
    } else if (nsel > 32) {
        launch_cfg<64, 64, 2, 2>(p, avec, s);       // 4 workgroups per CU; 128x64 measured 50 % slower at M = 70 k
    } else {
        if (t128 >= 512) launch_cfg<128, 32, 4, 1>(p, avec, s);
        else launch_cfg<64, 32, 2, 1>(p, avec, s);
    }
    if (p.splits > 1) launch_finish(p, s);
    HPL_CHECK_LAUNCH("hpl_gconv_forward");
    if (p.y_amax) return amax_launch(p.Y, p.ldy, p.M, p.N, p.y_amax, s, p.y_guard);
    return HPL_OK;
}

namespace {
// hpl_tile_index: per 64-row (BM) tile of a tap-ordered launch, the source row of every (tap, tile row) -- what the
// prologue of k_gconv would gather through row_perm and nbr -- and the tile's tap-presence masks.
__global__ void __launch_bounds__(256) k_tile_index(const int32_t *__restrict__ nbr, int64_t stride, int F, int64_t M,
                                                    const int32_t *__restrict__ perm, int BM, int32_t *__restrict__ tile_idx,
                                                    int32_t *__restrict__ tile_mask) {
    __shared__ int masks[8];
    const int t = threadIdx.x;
    if (t < 8) masks[t] = 0;
    __syncthreads();
    const int64_t tile = blockIdx.x, m0 = tile * BM;
    int32_t *out = tile_idx + tile * F * BM;
    // 256 threads, BM = 64 or 128: a thread keeps its tile row r = t % BM for every tap it handles (256 % BM == 0), so
    // it collects its taps in a register; the 32 lanes of a half-wave are the 32 rows of one block
    const int r = t % BM;
    const int64_t m = m0 + r;
    const int v = (m < M) ? (perm ? perm[m] : (int)m) : -1;
    int mybits = 0;
    for (int f = t / BM; f < F; f += 256 / BM) {
        const int row = (v >= 0) ? nbr[(int64_t)f * stride + v] : -1;
        out[f * BM + r] = row;
        mybits |= (row >= 0) ? (1 << f) : 0;
    }
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) mybits |= __shfl_xor(mybits, o, 64);       // OR over the 32 lanes of each half-wave
    if ((t & 31) == 0 && mybits) {
        atomicOr(&masks[0], mybits);
        atomicOr(&masks[2 + (r >> 5)], mybits);
    }
    __syncthreads();
    if (t < 8) tile_mask[tile * 8 + t] = masks[t];
}
}  // namespace

namespace {
// Schedule of the tiles: tile_mask[j][6] = the tile that runs j-th, most taps first (its work is proportional to the
// number of taps present in it).  Sorting the rows by tap mask puts similar rows into a tile but orders the tiles by
// the mask's numeric value, which is not their weight; with the heaviest tiles first the workgroups of a launch
// finish within one light tile of each other instead of one average tile.
__global__ void __launch_bounds__(1024) k_tile_rank(int32_t *__restrict__ tile_mask, int tiles) {
    // stable counting sort by tap count, descending (equal counts keep tile order): chunks of 1024 tiles, one per thread;
    // inside a chunk a tile's rank in its class = ballot prefix inside its wave + the class counts of the waves before it
    __shared__ int hist[16], base[16], running[16], wcnt[16][16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t < 16) { hist[t] = 0; running[t] = 0; }
    __syncthreads();
    for (int j = t; j < tiles; j += 1024) atomicAdd(&hist[__popc(tile_mask[(int64_t)j * 8] & 0x7fff)], 1);
    __syncthreads();
    if (t == 0) {
        int off = 0;
        for (int b = 15; b >= 0; --b) { base[b] = off; off += hist[b]; }
    }
    __syncthreads();
    for (int j0 = 0; j0 < tiles; j0 += 1024) {
        const int j = j0 + t;
        const int c = j < tiles ? __popc(tile_mask[(int64_t)j * 8] & 0x7fff) : -1;
        int mine = 0;
#pragma unroll
        for (int cls = 0; cls < 16; ++cls) {
            const unsigned long long bal = __ballot(c == cls);
            if (c == cls) mine = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wcnt[wave][cls] = __popcll(bal);
        }
        __syncthreads();
        if (c >= 0) {
            int before = 0;
            for (int w = 0; w < wave; ++w) before += wcnt[w][c];
            tile_mask[(int64_t)(base[c] + running[c] + before + mine) * 8 + 6] = j;
        }
        __syncthreads();
        if (t < 16) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wcnt[w][t];
            running[t] += tot;
        }
        __syncthreads();
    }
}
}  // namespace

extern "C" int hpl_tile_index(const int32_t *nbr, int64_t nbr_stride, int F, int64_t M, const int32_t *row_perm,
                              int BM, int32_t *tile_idx, int32_t *tile_mask, hplStream stream) {
    HPL_REQUIRE(nbr && tile_idx && tile_mask && F >= 1 && F <= 15 && M > 0 && (BM == 64 || BM == 128),
                "hpl_tile_index: bad arguments");
    k_tile_index<<<(int)cdiv(M, BM), 256, 0, to_stream(stream)>>>(nbr, nbr_stride, F, M, row_perm, BM, tile_idx, tile_mask);
    k_tile_rank<<<1, 1024, 0, to_stream(stream)>>>(tile_mask, (int)cdiv(M, BM));
    HPL_CHECK_LAUNCH("hpl_tile_index");
    return HPL_OK;
}

extern "C" int hpl_gconv_forward_naive(const hpl_gconv_desc *d, hplStream stream) {
    GParams p;
    int rc = fill_params(d, p, "hpl_gconv_forward_naive");
    if (rc != HPL_OK) return rc;
    if (p.M == 0) return HPL_OK;
    const int grid = (int)imin(cdiv(p.M * p.N, 256), 8192);
    k_gconv_naive<<<grid, 256, 0, to_stream(stream)>>>(p);
    HPL_CHECK_LAUNCH("hpl_gconv_forward_naive");
    return HPL_OK;
}

// ------------------------------------------------------------------------------------------
// Weight layouts
// ------------------------------------------------------------------------------------------
namespace {
__global__ void k_weight_relayout(const float *__restrict__ W, int64_t base, int R, int Q, int F,
                                  int64_t sr, int64_t sq, int64_t sf, const int32_t *__restrict__ fmap,
                                  float *__restrict__ Wt, int64_t k_rows, int64_t ldw) {
    // zero-fill pass and scatter pass are fused: every destination element computes its source
    const int64_t total = k_rows * ldw;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int64_t k = i / ldw;
        const int q = (int)(i - k * ldw);
        float v = 0.f;
        if (q < Q && k < (int64_t)F * R) {
            const int fdst = (int)(k / R);
            const int r = (int)(k - (int64_t)fdst * R);
            int f = fdst;
            if (fmap) {   // fmap is a permutation given as src -> dst; find the source tap
                f = -1;
                for (int g = 0; g < F; ++g)
                    if (fmap[g] == fdst) { f = g; break; }
            }
            if (f >= 0) v = W[base + r * sr + q * sq + f * sf];
        }
        Wt[i] = v;
    }
}

// One launch for many re-layouts (all conv weights of a model, both directions): job j owns
// destination elements [prefix[j], prefix[j+1]) of one buffer; mirror = taps stored in the order
// (F - f) % F (the mirrored-gather data gradient).
//
// Both are transposes with a 60-byte inner run (F = 15 taps), so a thread per element reads or writes 4 bytes of every 128-byte
// line it touches (77 MB of weights took 380 us in, 63 launches of 7-120 us out, per training step).  A work unit is 16
// (F <= 3: 64) consecutive r of one job: per chunk of 32 q its source -- for the two layouts the model has, runs of 16 * F (forward image:
// sr == F) or 32 * F (data-gradient image: sq == F) contiguous floats -- goes through LDS, and every global access is a full line.
constexpr int RL_TQ = 32, RL_FMAX = 15, RL_JOBS = 1024;
constexpr int RL_LDMAX = 16 * RL_FMAX + 1;
__host__ __device__ __forceinline__ int rl_tr(int F) { return F <= 3 ? 64 : 16; }      // r per unit: runs of >= 64 floats

// units of a job: r-blocks x chunks of 32 columns; "taps as column blocks" images (mirror == 2): blocks of 8 rows x 32 columns
__device__ __forceinline__ int rl_units(const hpl_relayout_job &jb) {
    const int qchunks = (int)((jb.ldw + RL_TQ - 1) / RL_TQ);
    if (jb.mirror == 2) return ((jb.R + 31) / 32 * 32 / 8) * qchunks;
    return ((jb.R + rl_tr(jb.F) - 1) / rl_tr(jb.F)) * qchunks;
}

__device__ __forceinline__ int relayout_unit_job(const int *upre, int njobs, int u) {
    int lo = 0, hi = njobs - 1;                          // last job with upre[job] <= u
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (upre[mid] <= u) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ void __launch_bounds__(256) k_weight_relayout_batch(const hpl_relayout_job *__restrict__ jobs, int njobs,
                                                               const int64_t *__restrict__ prefix, float *__restrict__ dst) {
    __shared__ float tile[RL_TQ * RL_LDMAX];
    __shared__ int upre[RL_JOBS + 1];
    const int t = threadIdx.x;
    for (int j = t; j < njobs; j += 256)             // units of a job: r-blocks x chunks of 32 columns (exact in a float: < 2^24)
        tile[j] = (float)rl_units(jobs[j]);
    __syncthreads();
    for (int j = t; j <= njobs; j += 256) {
        int acc = 0;
        for (int i = 0; i < j; ++i) acc += (int)tile[i];
        upre[j] = acc;
    }
    __syncthreads();
    const int units = upre[njobs];
    const int lane = t & 31, grp = t >> 5;
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        const int j = relayout_unit_job(upre, njobs, u);
        const hpl_relayout_job jb = jobs[j];
        float *out = dst + prefix[j];
        const int F = jb.F, R = jb.R, Q = jb.Q;
        const int64_t ldw = jb.ldw;
        const int TR = rl_tr(F), LD = TR * F + 1;              // (odd: the store reads one q per lane)
        const int qchunks = (int)((ldw + RL_TQ - 1) / RL_TQ), ul = u - upre[j];
        if (jb.mirror == 2) {       // element (r, f*Q + q) = W[base + r*sr + q*sq + f*sf]; rows past R / columns past F*Q are zero
            const int r = (ul / qchunks) * 8 + grp, col = (ul % qchunks) * RL_TQ + lane;
            if (col < ldw) {
                const int f = col / Q, q = col - f * Q;
                out[(int64_t)r * ldw + col] = (r < R && f < F) ? jb.W[jb.base + (int64_t)r * jb.sr + (int64_t)q * jb.sq + (int64_t)f * jb.sf] : 0.f;
            }
            continue;
        }
        const int r0 = (ul / qchunks) * TR, nr = min(TR, R - r0), q0 = (ul % qchunks) * RL_TQ;
        const bool mode_a = F <= RL_FMAX && jb.sf == 1 && jb.sr == F, mode_b = F <= RL_FMAX && jb.sf == 1 && jb.sq == F;
        const bool staged = mode_a || mode_b;
        {
            const int nq = max(0, min(RL_TQ, Q - q0));
            if (mode_a) {               // fixed q: (r, f) contiguous
                const int run = nr * F;
                for (int i = t; i < nq * run; i += 256) {
                    const int qi = i / run, x = i - qi * run;
                    tile[qi * LD + x] = jb.W[jb.base + (int64_t)r0 * F + x + (int64_t)(q0 + qi) * jb.sq];
                }
            } else if (mode_b) {        // fixed r: (q, f) contiguous
                const int run = nq * F;
                for (int i = t; i < nr * run; i += 256) {
                    const int ri = i / run, x = i - ri * run, qi = x / F, f = x - qi * F;
                    tile[qi * LD + ri * F + f] = jb.W[jb.base + (int64_t)(r0 + ri) * jb.sr + (int64_t)q0 * F + x];
                }
            }
            if (staged) __syncthreads();
            for (int row = grp; row < F * TR; row += 8) {
                const int f = row / TR, ri = row - f * TR;
                if (ri >= nr || q0 + lane >= ldw) continue;
                const int fd = jb.mirror ? (F - f) % F : f;
                float v = 0.f;
                if (lane < nq)
                    v = staged ? tile[lane * LD + ri * F + f]
                               : jb.W[jb.base + (int64_t)(r0 + ri) * jb.sr + (int64_t)(q0 + lane) * jb.sq + (int64_t)f * jb.sf];
                out[((int64_t)fd * R + r0 + ri) * ldw + q0 + lane] = v;
            }
            if (staged) __syncthreads();
        }
        if (r0 + TR >= R) {          // the last r-block clears its columns of the rows that pad F * R to a multiple of 32
            const int64_t k_used = (int64_t)F * R, k_rows = (k_used + 31) / 32 * 32;
            for (int64_t k = k_used + grp; k < k_rows; k += 8)
                if (q0 + lane < ldw) out[k * ldw + q0 + lane] = 0.f;
        }
    }
}

// inverse: W[base + r*sr + q*sq + f*sf] (+)= Wt[(f*R + r)*ldw + q]; small weights (a few workgroups' worth: launch latency is all
// there is) one element per thread, the others with the same staging (reads of Wt are rows of 32 q)
__global__ void k_weight_unlayout_small(const float *__restrict__ Wt, int64_t ldw, int R, int Q, int F,
                                        float *__restrict__ W, int64_t base, int64_t sr, int64_t sq, int64_t sf,
                                        int accumulate) {
    const int64_t total = (int64_t)F * R * Q;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int q = (int)(i % Q);
        const int64_t k = i / Q;
        const int f = (int)(k / R);
        const int r = (int)(k - (int64_t)f * R);
        const float v = Wt[k * ldw + q];
        float *d = W + base + r * sr + q * sq + f * sf;
        *d = accumulate ? *d + v : v;
    }
}

__global__ void __launch_bounds__(256) k_weight_unlayout(const float *__restrict__ Wt, int64_t ldw, int R, int Q, int F,
                                                         float *__restrict__ W, int64_t base, int64_t sr, int64_t sq, int64_t sf,
                                                         int accumulate) {
    __shared__ float tile[RL_TQ * RL_LDMAX];
    const int t = threadIdx.x, lane = t & 31, grp = t >> 5;
    const bool mode_a = F <= RL_FMAX && sf == 1 && sr == F, mode_b = F <= RL_FMAX && sf == 1 && sq == F;
    const bool staged = mode_a || mode_b;
    const int TR = rl_tr(F), LD = TR * F + 1;
    const int units = (R + TR - 1) / TR, qchunks = (Q + RL_TQ - 1) / RL_TQ;
    for (int64_t w = blockIdx.x; w < (int64_t)units * qchunks; w += gridDim.x) {
        const int r0 = (int)(w / qchunks) * TR, q0 = (int)(w % qchunks) * RL_TQ;
        const int nr = min(TR, R - r0), nq = min(RL_TQ, Q - q0);
        for (int row = grp; row < F * TR; row += 8) {
            const int f = row / TR, ri = row - f * TR;
            if (ri >= nr || lane >= nq) continue;
            const float v = Wt[((int64_t)f * R + r0 + ri) * ldw + q0 + lane];
            if (staged) tile[lane * LD + ri * F + f] = v;
            else {
                float *d = W + base + (int64_t)(r0 + ri) * sr + (int64_t)(q0 + lane) * sq + (int64_t)f * sf;
                *d = accumulate ? *d + v : v;
            }
        }
        if (!staged) continue;
        __syncthreads();
        if (mode_a) {
            const int run = nr * F;
            for (int i = t; i < nq * run; i += 256) {
                const int qi = i / run, x = i - qi * run;
                float *d = W + base + (int64_t)r0 * F + x + (int64_t)(q0 + qi) * sq;
                *d = accumulate ? *d + tile[qi * LD + x] : tile[qi * LD + x];
            }
        } else {
            const int run = nq * F;
            for (int i = t; i < nr * run; i += 256) {
                const int ri = i / run, x = i - ri * run, qi = x / F, f = x - qi * F;
                float *d = W + base + (int64_t)(r0 + ri) * sr + (int64_t)q0 * F + x;
                *d = accumulate ? *d + tile[qi * LD + ri * F + f] : tile[qi * LD + ri * F + f];
            }
        }
        __syncthreads();
    }
}

// The inverse of k_weight_relayout_batch for the weight gradients of a training step: job j's image [k_rows][ldw] at
// src + prefix[j] -> jb.W[base + r*sr + q*sq + f*sf] (jb.W is the gradient tensor, in the parameter's layout).  Same units,
// same staging (rows of 32 q are read as full lines, the parameter side is written in runs of 16 * F or 32 * F floats).
__global__ void __launch_bounds__(256) k_weight_unlayout_batch(const hpl_relayout_job *__restrict__ jobs, int njobs,
                                                               const int64_t *__restrict__ prefix, const float *__restrict__ src) {
    __shared__ float tile[RL_TQ * RL_LDMAX];
    __shared__ int upre[RL_JOBS + 1];
    const int t = threadIdx.x;
    for (int j = t; j < njobs; j += 256) tile[j] = (float)rl_units(jobs[j]);
    __syncthreads();
    for (int j = t; j <= njobs; j += 256) {
        int acc = 0;
        for (int i = 0; i < j; ++i) acc += (int)tile[i];
        upre[j] = acc;
    }
    __syncthreads();
    const int units = upre[njobs];
    const int lane = t & 31, grp = t >> 5;
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        const int j = relayout_unit_job(upre, njobs, u);
        const hpl_relayout_job jb = jobs[j];
        const float *in = src + (prefix[j] - prefix[0]);
        float *W = const_cast<float *>(jb.W);
        const int F = jb.F, R = jb.R, Q = jb.Q;
        const int64_t ldw = jb.ldw;
        const int TR = rl_tr(F), LD = TR * F + 1;
        const int qchunks = (int)((ldw + RL_TQ - 1) / RL_TQ), ul = u - upre[j];
        if (jb.mirror == 2) {
            const int r = (ul / qchunks) * 8 + grp, col = (ul % qchunks) * RL_TQ + lane;
            const int f = col / Q, q = col - f * Q;
            if (r < R && f < F) W[jb.base + (int64_t)r * jb.sr + (int64_t)q * jb.sq + (int64_t)f * jb.sf] = in[(int64_t)r * ldw + col];
            continue;
        }
        const int r0 = (ul / qchunks) * TR, nr = min(TR, R - r0), q0 = (ul % qchunks) * RL_TQ;
        const int nq = max(0, min(RL_TQ, Q - q0));
        const bool mode_a = F <= RL_FMAX && jb.sf == 1 && jb.sr == F, mode_b = F <= RL_FMAX && jb.sf == 1 && jb.sq == F;
        const bool staged = mode_a || mode_b;
        for (int row = grp; row < F * TR; row += 8) {
            const int f = row / TR, ri = row - f * TR;
            if (ri >= nr || lane >= nq) continue;
            const float v = in[((int64_t)f * R + r0 + ri) * ldw + q0 + lane];
            if (staged) tile[lane * LD + ri * F + f] = v;
            else W[jb.base + (int64_t)(r0 + ri) * jb.sr + (int64_t)(q0 + lane) * jb.sq + (int64_t)f * jb.sf] = v;
        }
        if (!staged) continue;
        __syncthreads();
        if (mode_a) {
            const int run = nr * F;
            for (int i = t; i < nq * run; i += 256) {
                const int qi = i / run, x = i - qi * run;
                W[jb.base + (int64_t)r0 * F + x + (int64_t)(q0 + qi) * jb.sq] = tile[qi * LD + x];
            }
        } else {
            const int run = nq * F;
            for (int i = t; i < nr * run; i += 256) {
                const int ri = i / run, x = i - ri * run, qi = x / F, f = x - qi * F;
                W[jb.base + (int64_t)(r0 + ri) * jb.sr + (int64_t)q0 * F + x] = tile[qi * LD + ri * F + f];
            }
        }
        __syncthreads();
    }
}
}  // namespace

extern "C" int hpl_weight_unlayout_batch(const hpl_relayout_job *jobs, int njobs, const int64_t *prefix, int64_t total,
                                         const float *src, hplStream stream) {
    HPL_REQUIRE(jobs && prefix && src && njobs > 0 && total > 0, "hpl_weight_unlayout_batch: bad arguments");
    HPL_REQUIRE(njobs <= RL_JOBS, "hpl_weight_unlayout_batch: at most 1024 jobs per call");
    const int grid = (int)imin(cdiv(total, 2048), 8192);
    k_weight_unlayout_batch<<<grid, 256, 0, to_stream(stream)>>>(jobs, njobs, prefix, src);
    HPL_CHECK_LAUNCH("hpl_weight_unlayout_batch");
    return HPL_OK;
}

extern "C" int hpl_weight_relayout(const float *W, int64_t base, int R, int Q, int F, int64_t sr,
                                   int64_t sq, int64_t sf, const int32_t *fmap, float *Wt,
                                   int64_t k_rows, int64_t ldw, hplStream stream) {
    HPL_REQUIRE(W && Wt && R > 0 && Q > 0 && F > 0, "hpl_weight_relayout: bad arguments");
    HPL_REQUIRE(k_rows >= (int64_t)F * R && ldw >= Q, "hpl_weight_relayout: destination too small");
    const int grid = (int)imin(cdiv(k_rows * ldw, 256), 8192);
    k_weight_relayout<<<grid, 256, 0, to_stream(stream)>>>(W, base, R, Q, F, sr, sq, sf, fmap, Wt, k_rows, ldw);
    HPL_CHECK_LAUNCH("hpl_weight_relayout");
    return HPL_OK;
}

extern "C" int hpl_weight_relayout_batch(const hpl_relayout_job *jobs, int njobs, const int64_t *prefix,
                                         int64_t total, float *dst, hplStream stream) {
    HPL_REQUIRE(jobs && prefix && dst && njobs > 0 && total > 0, "hpl_weight_relayout_batch: bad arguments");
    HPL_REQUIRE(njobs <= RL_JOBS, "hpl_weight_relayout_batch: at most 1024 jobs per call");
    const int grid = (int)imin(cdiv(total, 2048), 8192);       // (a unit is <= 64 r x 3 taps or 16 r x 15 taps, x 32 columns)
    k_weight_relayout_batch<<<grid, 256, 0, to_stream(stream)>>>(jobs, njobs, prefix, dst);
    HPL_CHECK_LAUNCH("hpl_weight_relayout_batch");
    return HPL_OK;
}

extern "C" int hpl_weight_unlayout(const float *Wt, int64_t ldw, int R, int Q, int F, float *W,
                                   int64_t base, int64_t sr, int64_t sq, int64_t sf, int accumulate,
                                   hplStream stream) {
    HPL_REQUIRE(W && Wt && R > 0 && Q > 0 && F > 0 && ldw >= Q, "hpl_weight_unlayout: bad arguments");
    if ((int64_t)F * R * Q < (int64_t)(1 << 19)) {
        const int g1 = (int)imin(cdiv((int64_t)F * R * Q, 256), 8192);
        k_weight_unlayout_small<<<g1, 256, 0, to_stream(stream)>>>(Wt, ldw, R, Q, F, W, base, sr, sq, sf, accumulate);
        HPL_CHECK_LAUNCH("hpl_weight_unlayout");
        return HPL_OK;
    }
    const int grid = (int)imin(cdiv(R, rl_tr(F)) * cdiv(Q, RL_TQ), 8192);
    k_weight_unlayout<<<grid, 256, 0, to_stream(stream)>>>(Wt, ldw, R, Q, F, W, base, sr, sq, sf, accumulate);
    HPL_CHECK_LAUNCH("hpl_weight_unlayout");
    return HPL_OK;
}

// ------------------------------------------------------------------------------------------
// Weight gradient:  dWt[k, n] += sum_m A[nbr[f][m], c] * dY[m, n],  k = f*C + c
// "TN" GEMM reducing over vertices.  A workgroup owns a 128(k) x BN(n) tile of dWt and a
// slab of vertices (split over gridDim.y); both operands land in LDS row-major exactly as
// they lie in memory (gathered row slice [m][128 k], dY row slice [m][BN]) and are read
// column-wise by the MFMA lanes, conflict-free.  Partial tiles are combined with fp32
// atomics (L2 atomics on gfx950), dWt is zero-initialised by the caller.
// ------------------------------------------------------------------------------------------
namespace {
using hpl_gc::WParams;

template <int BN, bool VEC, bool TAP, int NT = 256, bool DEEP = false>
__global__ void __launch_bounds__(NT) k_wgrad(const WParams p) {
    constexpr int BKR = 128;                 // rows of dWt per tile (flat k)
    constexpr int BMS = 32;                  // vertices per step
    // wave grid: 4 waves = 2x2 (BN >= 64) or 4x1; 8 waves (BN = 128 only) = 2x4
    constexpr int WGN = (NT == 512) ? 4 : ((BN >= 64) ? 2 : 1);
    constexpr int WGM_EFF = (NT / 64) / WGN;
    constexpr int WTM = BKR / WGM_EFF, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_F4 = BKR / 4;            // float4 per gathered row slice
    constexpr int A_ROWS_PER_PASS = NT / A_F4;
    constexpr int A_PASSES = BMS / A_ROWS_PER_PASS;
    constexpr int B_F4 = BN / 4;
    constexpr int B_ROWS_PER_PASS = NT / B_F4;
    constexpr int B_PASSES = (BMS + B_ROWS_PER_PASS - 1) / B_ROWS_PER_PASS;
    static_assert(TM >= 1 && TN >= 1 && A_PASSES >= 1, "wave grid does not fit the tile");
    __shared__ __attribute__((aligned(16))) float smem[2 * BMS * BKR + 2 * BMS * BN];
    float *As = smem;
    float *Bs = smem + 2 * BMS * BKR;

    const int tile_k = blockIdx.x / p.tiles_n, tile_n = blockIdx.x % p.tiles_n;
    const int n0 = tile_n * BN;
    // TAP: [mb, me) indexes the tap's vertex list, c0_tile is the first channel of the block
    const int tap = TAP ? tile_k / p.c_tiles : 0;
    const int c0_tile = TAP ? (tile_k - tap * p.c_tiles) * BKR : 0;
    const int k0 = TAP ? tap * p.C + c0_tile : tile_k * BKR;
    const int32_t *vm = TAP ? p.tap_m + p.tap_ptr[tap] : nullptr;
    const int32_t *vrow = TAP ? p.tap_row + p.tap_ptr[tap] : nullptr;
    const int64_t m_total = TAP ? (int64_t)(p.tap_ptr[tap + 1] - p.tap_ptr[tap]) : p.M;
    // the vertex loop is cut into gridDim.y slabs; a tap's own list is cut evenly (lists differ in length)
    const int64_t per = TAP ? ((m_total + gridDim.y - 1) / gridDim.y + BMS - 1) / BMS * BMS : p.m_per_split;
    const int64_t mb = (int64_t)blockIdx.y * per;
    const int64_t me = imin(m_total, mb + per);
    if (mb >= me) return;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 31, hi = lane >> 5;

    const int ak4 = t % A_F4, arow0 = t / A_F4;
    const int kk = k0 + ak4 * 4;                          // fixed flat k of this thread's float4
    const int f_t = TAP ? tap : kk / p.C, c_t = kk - f_t * p.C;
    const bool k_ok = TAP ? (c_t < p.C) : (kk < p.K);
    const int bn4 = t % B_F4, brow0 = t / B_F4;

    // Two register sets: the rows of step s+2 are being fetched while those of step s+1 wait in the
    // other set for their turn in LDS (DEEP; tap mode, whose rows come from L2 misses more often).
    float4 ra0[A_PASSES], rb0[B_PASSES], ra1[DEEP ? A_PASSES : 1], rb1[DEEP ? B_PASSES : 1];
    // Indices of a step are fetched one step before its data (source row of every gathered A row,
    // vertex of every dY row): the data loads of step s+1 and the index loads of step s+2 are in
    // flight while step s is multiplied.
    int rowi[A_PASSES], mi[B_PASSES];
    const bool do_bias = !TAP && p.dbias != nullptr && tile_k == 0;      // workgroup-uniform
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_idx = [&](int64_t ms) {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const int64_t j = ms + arow0 + i * A_ROWS_PER_PASS;
            int row = -1;
            if (j < me && k_ok) {
                if (TAP) row = vrow[j];
                else if (VEC) row = p.nbr ? p.nbr[(int64_t)f_t * p.nbr_stride + j] : (int)((int64_t)f_t * p.reg_stride + j);
                else row = 0;       // scalar path resolves rows per element below
            }
            rowi[i] = row;
        }
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int r = brow0 + i * B_ROWS_PER_PASS;
            const int64_t j = ms + r;
            mi[i] = (r < BMS && j < me) ? (TAP ? vm[j] : (int)j) : -1;
        }
    };
    auto load_data = [&](int64_t ms, float4 *ra, float4 *rb) {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rowi[i] >= 0) {
                if (VEC) {
                    v = *reinterpret_cast<const float4 *>(p.A + (int64_t)rowi[i] * p.lda + c_t);
                } else {
                    const int64_t m = ms + arow0 + i * A_ROWS_PER_PASS;
                    float e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = kk + j;
                        e[j] = 0.f;
                        if (k < p.K) {
                            const int f = k / p.C, c = k - f * p.C;
                            int64_t row = p.nbr ? (int64_t)p.nbr[(int64_t)f * p.nbr_stride + m] : (int64_t)f * p.reg_stride + m;
                            if (row >= 0) e[j] = p.A[row * p.lda + c];
                        }
                    }
                    v = make_float4(e[0], e[1], e[2], e[3]);
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int64_t m = mi[i];
            const int col = n0 + bn4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m >= 0) {
                if (VEC && col + 3 < p.N) v = *reinterpret_cast<const float4 *>(p.dY + m * p.lddy + col);
                else {
                    float e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) e[j] = (col + j < p.N) ? p.dY[m * p.lddy + col + j] : 0.f;
                    v = make_float4(e[0], e[1], e[2], e[3]);
                }
            }
            rb[i] = v;
            if (!TAP && do_bias) { bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w; }
        }
    };
    auto store_lds = [&](int buf, const float4 *ra, const float4 *rb) {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i)
            *reinterpret_cast<float4 *>(As + buf * BMS * BKR + (arow0 + i * A_ROWS_PER_PASS) * BKR + ak4 * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int r = brow0 + i * B_ROWS_PER_PASS;
            if (r < BMS) *reinterpret_cast<float4 *>(Bs + buf * BMS * BN + r * BN + bn4 * 4) = rb[i];
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int64_t nsteps = (me - mb + BMS - 1) / BMS;
    auto multiply = [&](int cur) {
        const float *a = As + cur * BMS * BKR + wm * WTM + li;
        const float *b = Bs + cur * BMS * BN + wn * WTN + li;
#pragma unroll
        for (int q = 0; q < BMS / 2; ++q) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = a[(q * 2 + hi) * BKR + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = b[(q * 2 + hi) * BN + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };
    if (DEEP) {
        // step st: LDS buffer `cur` holds st, register set (st+1)&1 holds st+1, set st&1 is free for st+2,
        // the index registers hold st+2 and are refilled for st+3 once its loads are issued
        load_idx(mb);
        load_data(mb, ra0, rb0);
        if (nsteps > 1) { load_idx(mb + BMS); load_data(mb + BMS, ra1, rb1); }
        if (nsteps > 2) load_idx(mb + 2 * BMS);
        store_lds(0, ra0, rb0);
        __syncthreads();
        auto body = [&](int64_t st, int cur, float4 *fa, float4 *fb, float4 *na, float4 *nb) {
            if (st + 2 < nsteps) load_data(mb + (st + 2) * BMS, fa, fb);
            if (st + 3 < nsteps) load_idx(mb + (st + 3) * BMS);
            multiply(cur);
            if (st + 1 < nsteps) store_lds(cur ^ 1, na, nb);
            __syncthreads();
        };
        for (int64_t st = 0; st < nsteps; st += 2) {
            body(st, 0, ra0, rb0, ra1, rb1);
            if (st + 1 < nsteps) body(st + 1, 1, ra1, rb1, ra0, rb0);
        }
    } else {
        load_idx(mb);
        load_data(mb, ra0, rb0);
        if (nsteps > 1) load_idx(mb + BMS);
        store_lds(0, ra0, rb0);
        __syncthreads();
        int cur = 0;
        for (int64_t st = 0; st < nsteps; ++st) {
            const bool more = st + 1 < nsteps;
            if (more) load_data(mb + (st + 1) * BMS, ra0, rb0);
            if (st + 2 < nsteps) load_idx(mb + (st + 2) * BMS);
            multiply(cur);
            if (more) store_lds(cur ^ 1, ra0, rb0);
            __syncthreads();
            cur ^= 1;
        }
    }
    if (!TAP && do_bias) {
        // column sums of this workgroup's dY slab: threads with the same float4 column combine in LDS
        float4 *red = reinterpret_cast<float4 *>(As);     // the last step's barrier has passed
        red[t] = bsum;
        __syncthreads();
        if (t < B_F4) {
            float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int r = 0; r < NT / B_F4; ++r) {
                const float4 v = red[t + r * B_F4];
                sacc.x += v.x; sacc.y += v.y; sacc.z += v.z; sacc.w += v.w;
            }
            const int col = n0 + t * 4;
            if (col + 0 < p.N) atomicAdd(p.dbias + col + 0, sacc.x);
            if (col + 1 < p.N) atomicAdd(p.dbias + col + 1, sacc.y);
            if (col + 2 < p.N) atomicAdd(p.dbias + col + 2, sacc.z);
            if (col + 3 < p.N) atomicAdd(p.dbias + col + 3, sacc.w);
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WTN + j * 32 + li;
            if (n >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kr = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;   // row inside the tile
                const int k = k0 + kr;
                if (TAP ? (c0_tile + kr < p.C) : (k < p.K)) atomicAdd(p.dWt + (int64_t)k * p.ldw + n, acc[i][j][r]);
            }
        }
}
}  // namespace

extern "C" int hpl_gconv_wgrad(const float *A, int64_t lda, int64_t rows_a, const int32_t *nbr,
                               int64_t nbr_stride, int64_t reg_stride, int64_t M, int C, int F,
                               const float *dY, int64_t lddy, int N, float *dWt, int64_t ldw,
                               const int32_t *tap_m, const int32_t *tap_row, const int32_t *tap_ptr,
                               int64_t tap_max, float *dbias, hplStream stream) {
    return hpl_gconv_wgrad_scaled(A, lda, rows_a, nbr, nbr_stride, reg_stride, M, C, F, dY, lddy, N, dWt, ldw, tap_m, tap_row,
                                  tap_ptr, tap_max, dbias, nullptr, nullptr, stream);
}

extern "C" int hpl_gconv_wgrad_scaled(const float *A, int64_t lda, int64_t rows_a, const int32_t *nbr,
                                      int64_t nbr_stride, int64_t reg_stride, int64_t M, int C, int F,
                                      const float *dY, int64_t lddy, int N, float *dWt, int64_t ldw,
                                      const int32_t *tap_m, const int32_t *tap_row, const int32_t *tap_ptr,
                                      int64_t tap_max, float *dbias, const float *a_amax, const float *dy_amax,
                                      hplStream stream) {
    HPL_REQUIRE(A && dY && dWt, "hpl_gconv_wgrad: null pointer");
    HPL_REQUIRE(M >= 0 && C > 0 && F > 0 && N > 0 && lda >= C && lddy >= N && ldw >= N,
                "hpl_gconv_wgrad: bad sizes");
    HPL_REQUIRE(nbr || F == 1 || reg_stride > 0, "hpl_gconv_wgrad: F > 1 needs a table or reg_stride");
    HPL_REQUIRE(nbr || (int64_t)F * reg_stride + M < (int64_t)INT32_MAX, "hpl_gconv_wgrad: regular table too large");
    if (M == 0) return HPL_OK;
    WParams p;
    p.A = A; p.lda = lda; p.nbr = nbr; p.nbr_stride = nbr_stride; p.reg_stride = reg_stride;
    p.M = M; p.C = C; p.F = F; p.K = F * C; p.dY = dY; p.lddy = lddy; p.N = N; p.dWt = dWt; p.ldw = ldw;
    p.rows_a = rows_a;
    p.a_amax = a_amax; p.dy_amax = dy_amax;
    const bool vec = (C % 4 == 0) && (lda % 4 == 0) && (lddy % 4 == 0) && aligned16(A) && aligned16(dY);
    const int bn = N > 64 ? 128 : (N > 32 ? 64 : 32);
    // Tap mode (per-tap lists of present vertices given, wide layers): k tiles are aligned to taps so
    // that a tile walks only the vertices whose tap is present -- exact skipping of absent neighbours
    // (42 % / 72 % of the vertex-tap pairs exist at levels 0 / 1).  It pads C to a multiple of 128,
    // so it is used when that costs < 25 %.
    const int c_tiles = (int)cdiv(C, 128);
    const bool tap = tap_m && tap_row && tap_ptr && nbr && vec && F > 1 && tap_max > 0 && c_tiles * 128 * 4 <= C * 5;
    p.tap_m = tap ? tap_m : nullptr;
    p.tap_row = tap ? tap_row : nullptr;
    p.tap_ptr = tap ? tap_ptr : nullptr;
    p.c_tiles = c_tiles;
    p.dbias = tap ? nullptr : dbias;
    if (tap && dbias) colsum_accumulate(dY, lddy, M, N, dbias, to_stream(stream));
    const int tiles_k = tap ? F * c_tiles : (int)cdiv(p.K, 128);
    const int64_t m_len = tap ? tap_max : M;                 // longest vertex loop of a tile
    if (vec) {
        // wide layers: split operands on the bf16 MFMA (wgrad3.hip)
        WParams q = p;
        if (launch_wgrad3(q, tap, m_len, to_stream(stream))) {
            if (!tap && dbias) colsum_accumulate(dY, lddy, M, N, dbias, to_stream(stream));
            HPL_CHECK_LAUNCH("hpl_gconv_wgrad");
            return HPL_OK;
        }
    }
    p.tiles_n = (int)cdiv(N, bn);
    const int tiles = tiles_k * p.tiles_n;
    // split the vertex axis so that ~4 workgroups per CU exist, each with >= 256 vertices
    // Slabs of the vertex loop per tile: workgroups run 2 per CU (512 at a time); pick the count that
    // minimises rounds x (slab length + the cost of the atomic epilogue, ~256 vertices' worth).  In tap
    // mode the lists are ~half of M and unequal; finer slabs even them out (~16 workgroups per slot).
    constexpr int force_splits = 0;
    int64_t splits = 1;
    {
        const int64_t len = tap ? imax(1, m_len / 2) : m_len;
        const int64_t smax = imax(1, imin(256, cdiv(len, 64)));      // (round 5: was 64 slabs of >= 256 vertices -- the 1x1 layers of levels 0-1, ONE tile, ran as 63 workgroups of 13 dependent steps)
        int64_t best = INT64_MAX;
        for (int64_t sp = 1; sp <= smax; ++sp) {
            const int64_t cost = cdiv((int64_t)tiles * sp, 512) * (cdiv(cdiv(len, sp), 32) * 32 + 256);
            if (cost < best) { best = cost; splits = sp; }
        }
        if (tap) splits = imax(1, imin(cdiv(8192, tiles), cdiv(len, 512)));      // measured: finer is better
        if (force_splits > 0) splits = force_splits;
    }
    p.m_per_split = cdiv(cdiv(m_len, splits), 32) * 32;
    if (!tap) splits = cdiv(m_len, p.m_per_split);
    dim3 grid(tiles, (unsigned)splits);
    hipStream_t s = to_stream(stream);
#define LAUNCH(BN_)                                                          \
    do {                                                                     \
        if (tap) k_wgrad<BN_, true, true><<<grid, 256, 0, s>>>(p);           \
        else if (vec) k_wgrad<BN_, true, false><<<grid, 256, 0, s>>>(p);     \
        else k_wgrad<BN_, false, false><<<grid, 256, 0, s>>>(p);             \
    } while (0)
    if (bn == 128 && vec) {     // 8 waves (2x4): 4 waves per SIMD, 2-5 % over 4 waves
        constexpr bool deep = true;
        if (tap && deep) k_wgrad<128, true, true, 512, true><<<grid, 512, 0, s>>>(p);
        else if (tap) k_wgrad<128, true, true, 512><<<grid, 512, 0, s>>>(p);
        else if (deep) k_wgrad<128, true, false, 512, true><<<grid, 512, 0, s>>>(p);
        else k_wgrad<128, true, false, 512><<<grid, 512, 0, s>>>(p);
    } else
    if (bn == 128) LAUNCH(128); else if (bn == 64) LAUNCH(64); else LAUNCH(32);
#undef LAUNCH
    HPL_CHECK_LAUNCH("hpl_gconv_wgrad");
    return HPL_OK;
}
