// gconv3b.hip -- split-operand gather-GEMM, second form: BOTH operands arrive pre-split.
//
//   Y[m, n] = act(bias[n] + res + sum_{f<F} sum_{c<C} A[nbr[f][m], c] * Wt[f*C + c, n])        (as gconv.hip)
//
// gconv3.hip splits the gathered fp32 rows while it stages them: every element is split once per (column tile, tap
// that gathers it) -- ~30 times on the dominant layer -- and the conversion is 60 of the ~110 vector instructions a wave
// issues per 24 MFMAs, which is what holds that kernel at 42 % of the matrix pipe (profiles/r03d_split3_pmc_*.txt).
// Here the activation matrix is split ONCE by a streaming pre-pass (hpl_rows_split3: three bf16 planes [rows][Cp],
// Cp = C rounded up to 8) and the GEMM moves nothing through registers on its way to LDS:
//   - gathered rows: LDS-direct loads, one per (16 rows x 64 bytes) of a plane -- a lane fetches the 16 bytes (8
//     channels) of one row, the row's four lanes of a 32-wide slice read 64 contiguous bytes; the LDS image is
//     [plane][row][4 x 16 B] with the k-block position XOR-swizzled by (row / 4) % 4 on the SOURCE side (an LDS-direct
//     load writes lane-linear), which makes the ds_read_b128 fragment reads conflict-free;
//   - weights: LDS-direct loads of the split image, already in B-fragment order (as gconv3.hip), rows f*Cp + c (each
//     tap padded to Cp, so that an 8-channel block never straddles two taps = two source rows).
// Geometry: 128 x (128*TN) tile, 8 waves (2 x 4), wave tile 64 x (32*TN); one workgroup per CU (2 waves per SIMD).
// Pipeline: NSTAGE LDS stages of one 32-wide slice each; a slice is two MFMA k-steps; the fragments of the next k-step
// are read while the current one is multiplied (two register sets), so there is ONE barrier per slice, in its middle:
//     k-step 0 of slice s   | read fragments (s, k-step 1)
//     wait: slice s+1 landed (later slices stay in flight); barrier; issue the loads of slice s+NSTAGE into s's stage
//     k-step 1 of slice s   | read fragments (s+1, k-step 0)
// A wave issues ~2 other instructions per MFMA (TN = 2) instead of ~9.
#include "common.h"
#include "gconv_common.h"

#include <stdlib.h>
#include <string>
#include <type_traits>

using namespace hpl;
using namespace hpl_gc;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

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

constexpr int BMB = 128;

template <int TN, int NSTAGE, int F_LDS>
__global__ void __launch_bounds__(512, 2) k_gconv3b(const GParams p) {
    constexpr int BM = BMB, BN = 128 * TN, NT = 512;
    constexpr int A_PLANE = BM * 64;                 // bytes: [row][4 k-blocks x 16 B]
    constexpr int A_STAGE = 3 * A_PLANE;
    constexpr int B_STAGE = 3 * 4 * BN * 16;         // [plane][k-block][n][16 B]
    constexpr int STAGE = A_STAGE + B_STAGE;
    constexpr int LOADS = 3 + 3 * TN;                // LDS-direct loads a wave issues per slice
    constexpr int KLIST = 1024;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE + (F_LDS * BM + BM + 8) * 4 + KLIST * 2];
    int *Is = reinterpret_cast<int *>(smem + NSTAGE * STAGE);
    int *Vs = Is + F_LDS * BM;
    int *tapmask_s = Vs + BM;
    unsigned short *Ks = reinterpret_cast<unsigned short *>(tapmask_s + 8);

    int tile_m, tile_n;
    tile_coords(p, tile_m, tile_n);
    if (tile_m < 0) return;
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, hi = lane >> 5;
    const int Cp = p.Cp;

    const bool probe = p.clock_probe && (blockIdx.x & 63) == 0 && t == 0;
    long long probe_c = 0, probe_w = 0;
    if (probe) { probe_c = (long long)__builtin_readcyclecounter(); probe_w = (long long)__builtin_amdgcn_s_memrealtime(); }

    // ---- tile prologue (as gconv3.hip)
    if (p.tile_idx && p.tile_bm == BM) {
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
        int mybits = 0;
        for (int i = t; i < F_LDS * BM; i += NT) {
            const int f = i / BM, r = i - f * BM;
            const int v = Vs[r];
            int row = -1;
            if (v >= 0 && f < p.F) row = p.nbr ? p.nbr[(int64_t)f * p.nbr_stride + v] : (int)((int64_t)f * p.reg_stride + v);
            Is[i] = row;
            mybits |= (row >= 0) ? (1 << f) : 0;
        }
        if (mybits) {
            atomicOr(tapmask_s, mybits);
            atomicOr(tapmask_s + 2 + ((t % BM) >> 5), mybits);
        }
    }
    __syncthreads();
    const int tapmask = __builtin_amdgcn_readfirstlane(tapmask_s[0]);
    int bmask[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) bmask[i] = __builtin_amdgcn_readfirstlane(tapmask_s[2 + wm * 2 + i]);
    // slices of the padded contraction index k' = f*Cp + c that this tile needs (Cp >= 32: a slice touches <= 2 taps)
    const int nk = (p.F * Cp + BK - 1) / BK;
    if (wave == 0) {
        int count = 0;
        for (int base = 0; base < nk; base += 64) {
            const int kt = base + lane;
            bool need = false;
            int f_lo = 0;
            if (kt < nk) {
                f_lo = (kt * BK) / Cp;
                const int f_hi = min((kt * BK + BK - 1) / Cp, p.F - 1);
                int bits = 0;
                for (int f = f_lo; f <= f_hi; ++f) bits |= 1 << f;
                need = (tapmask & bits) != 0;
            }
            const unsigned long long bal = __ballot(need);
            if (need) {
                const int e = kt | (f_lo << 10) | (((kt * BK + BK - 1) / Cp > f_lo && f_lo + 1 < p.F) ? (1 << 14) : 0);
                Ks[count + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)e;
            }
            count += __popcll(bal);
        }
        if (lane == 0) tapmask_s[1] = count;
    }
    __syncthreads();
    const int nsl = __builtin_amdgcn_readfirstlane(tapmask_s[1]);

    // ---- operand descriptors
    constexpr unsigned OOB = 0x80000000u;
    __amdgpu_buffer_rsrc_t rsrc_a[3], rsrc_b[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        rsrc_a[pl] = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char *>(reinterpret_cast<const unsigned char *>(p.A3) + (int64_t)pl * p.a3_plane_stride),
            (short)0, (int)p.a3_bytes, 0x00020000);
        rsrc_b[pl] = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char *>(reinterpret_cast<const unsigned char *>(p.Wt3) + (int64_t)pl * p.w3_plane_stride),
            (short)0, (int)p.w3_bytes, 0x00020000);
    }
    const unsigned a_row_bytes = (unsigned)p.a3_ld * 2u;
    const unsigned ldw16 = (unsigned)p.ldw * 16u;
    // gathered rows: wave w stages tile rows [16w, 16w+16) of every plane; lane = (row r, position s), the position
    // holds k-block s ^ (row / 4) % 4
    const int a_row = wave * 16 + (lane >> 2);
    const int a_kb = (lane & 3) ^ ((a_row >> 2) & 3);
    int f0_u = 0, c0_u = 0, k_u = 0;
    auto issue = [&](int kt, int st) {
        const int k0 = kt * BK;
        c0_u += k0 - k_u;
        k_u = k0;
        while (c0_u >= Cp) { c0_u -= Cp; ++f0_u; }
        int c = c0_u + a_kb * 8, f = f0_u;
        if (c >= Cp) { c -= Cp; ++f; }
        const int row = Is[min(f, F_LDS - 1) * BM + a_row];
        const unsigned aoff = (f < p.F && row >= 0) ? (unsigned)row * a_row_bytes + (unsigned)c * 2u : OOB;
        unsigned char *sb = smem + st * STAGE;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a[pl], (__attribute__((address_space(3))) void *)(sb + pl * A_PLANE + wave * 1024),
                                                     16, (int)aoff, 0, 0, 0);
        // weights: (k-block, 64-column block) pairs of the slice, TN per wave, three planes each
#pragma unroll
        for (int q = 0; q < TN; ++q) {
            const int combo = wave + 8 * q, kb = combo / (2 * TN), nb = combo % (2 * TN);
            const unsigned col = (unsigned)(n0 + nb * 64 + lane);
            const unsigned boff = (col < (unsigned)p.ldw) ? (unsigned)(kt * (BK / 8) + kb) * ldw16 + col * 16u : OOB;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rsrc_b[pl], (__attribute__((address_space(3))) void *)(sb + A_STAGE + ((pl * 4 + kb) * BN + nb * 64) * 16), 16,
                    (int)boff, 0, 0, 0);
        }
    };

    floatx16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses inside a stage for k-step j: A row block i -> a_rofs[j][i], B column block -> b_rofs + ...
    unsigned a_rofs[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wm * 64 + i * 32 + li;
            a_rofs[j][i] = (unsigned)(row * 64 + (((2 * j + hi) ^ ((row >> 2) & 3)) << 4));
        }
    const unsigned b_rofs = (unsigned)(A_STAGE + (hi * BN + wn * 32 * TN + li) * 16);

    u32x4 af[2][3][2], bf[2][3][TN];
    auto read_frags = [&](auto set_tag, int st, int j) {
        constexpr int SET = decltype(set_tag)::value;
        const unsigned char *sb = smem + st * STAGE;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[SET][pl][i] = *reinterpret_cast<const u32x4 *>(sb + (j ? a_rofs[1][i] : a_rofs[0][i]) + pl * A_PLANE);
#pragma unroll
            for (int q = 0; q < TN; ++q)
                bf[SET][pl][q] = *reinterpret_cast<const u32x4 *>(sb + b_rofs + ((pl * 4 + 2 * j) * BN + q * 32) * 16);
        }
    };
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
    auto multiply = [&](auto set_tag, bool need0, bool need1) {
        constexpr int SET = decltype(set_tag)::value;
        if (need0 && need1) {
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[SET][PA[q]][i]),
                                                                            __builtin_bit_cast(bf16x8, bf[SET][PB[q]][j]), acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (!(i ? need1 : need0)) continue;
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[SET][PA[q]][i]),
                                                                            __builtin_bit_cast(bf16x8, bf[SET][PB[q]][j]), acc[i][j], 0, 0, 0);
            }
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    auto wait_vm = [](auto n_tag) {               // vmcnt <= N, lgkmcnt = 0
        constexpr int N = decltype(n_tag)::value;
        __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4) | (0x0 << 8));
    };

    if (nsl > 0) {
#pragma unroll
        for (int s = 0; s < NSTAGE; ++s)
            if (s < nsl) issue((int)(Ks[s] & 1023), s);
        // slice 0 has landed when at most the loads of the slices behind it are in flight
        if (nsl >= NSTAGE) wait_vm(std::integral_constant<int, (NSTAGE - 1) * LOADS>{});
        else wait_vm(S0{});
        asm volatile("s_barrier" ::: "memory");
        read_frags(S0{}, 0, 0);
        int st = 0;
        for (int s = 0; s < nsl; ++s) {
            const int e = (int)Ks[s];
            const int f_lo = (e >> 10) & 15, two = (e >> 14) & 1;
            const bool need0 = ((bmask[0] >> f_lo) | (two ? (bmask[0] >> (f_lo + 1)) : 0)) & 1;
            const bool need1 = ((bmask[1] >> f_lo) | (two ? (bmask[1] >> (f_lo + 1)) : 0)) & 1;
            const int stn = st + 1 == NSTAGE ? 0 : st + 1;
            read_frags(S1{}, st, 1);
            multiply(S0{}, need0, need1);
            // slice s+1 landed (later slices may be in flight), every wave has read the last fragments of slice s
            if (s + NSTAGE - 1 < nsl) wait_vm(std::integral_constant<int, (NSTAGE - 2) * LOADS>{});
            else wait_vm(S0{});
            asm volatile("s_barrier" ::: "memory");
            if (s + NSTAGE < nsl) issue((int)(Ks[s + NSTAGE] & 1023), st);
            if (s + 1 < nsl) read_frags(S0{}, stn, 0);
            multiply(S1{}, need0, need1);
            st = stn;
        }
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * 32 * TN + j * 32 + li;
            if (n >= p.N) continue;
            const float bsv = p.bias ? p.bias[n] : 0.f;
            const int res_mod = (int)p.res_mod;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = Vs[wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi];
                if (m < 0) continue;
                float v = acc[i][j][r] + bsv;
                if (p.res) v += p.res[(int64_t)((int)m < res_mod ? (int)m : (int)m % res_mod) * p.ldres + n];
                if (p.act == HPL_ACT_LEAKY) v = v > 0.f ? v : p.slope * v;
                p.Y[m * p.ldy + n] = v;
                if (p.Y2 && m < p.rows2) p.Y2[m * p.ldy2 + n] = v;
            }
        }
    if (probe) {
        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe),
                  (unsigned long long)((long long)__builtin_readcyclecounter() - probe_c));
        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe) + 1,
                  (unsigned long long)((long long)__builtin_amdgcn_s_memrealtime() - probe_w));
    }
}

// activation rows fp32 [rows][lda] (C channels) -> three bf16 planes [rows][Cp], channels C..Cp-1 zero
__global__ void __launch_bounds__(256) k_rows_split3(const float *__restrict__ A, int64_t lda, int64_t rows, int C, int Cp,
                                                      unsigned char *__restrict__ dst, int64_t plane_stride, bool vec) {
    const int blocks = Cp / 8;
    const int64_t total = rows * blocks;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int64_t r = i / blocks;
        const int c = (int)(i - r * blocks) * 8;
        float x[8];
        const float *src = A + r * lda + c;
        if (vec && c + 8 <= C) {
            const float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = (c + j < C) ? src[j] : 0.f;
        }
        u32x4 h, m, l;
        unsigned a, b, cc;
        split2(x[0], x[1], a, b, cc); h.x = a; m.x = b; l.x = cc;
        split2(x[2], x[3], a, b, cc); h.y = a; m.y = b; l.y = cc;
        split2(x[4], x[5], a, b, cc); h.z = a; m.z = b; l.z = cc;
        split2(x[6], x[7], a, b, cc); h.w = a; m.w = b; l.w = cc;
        unsigned char *o = dst + (r * Cp + c) * 2;
        *reinterpret_cast<u32x4 *>(o) = h;
        *reinterpret_cast<u32x4 *>(o + plane_stride) = m;
        *reinterpret_cast<u32x4 *>(o + 2 * plane_stride) = l;
    }
}

// Wt rows f*C + c -> split image rows f*Cp + c (zero rows for c >= C), planes [rows/8][ldw][8]
__global__ void __launch_bounds__(256) k_weight_split3p(const float *__restrict__ Wt, int F, int C, int Cp, int64_t ldw,
                                                         int64_t rows_out, unsigned char *__restrict__ dst, int64_t plane_stride) {
    const int64_t total = rows_out / 8 * ldw;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int64_t kb = i / ldw, n = i - kb * ldw;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t kp = kb * 8 + j;
            const int f = (int)(kp / Cp), c = (int)(kp - (int64_t)f * Cp);
            x[j] = (f < F && c < C) ? Wt[((int64_t)f * C + c) * ldw + n] : 0.f;
        }
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

}  // namespace

extern "C" int hpl_rows_split3(const float *A, int64_t lda, int64_t rows, int C, void *dst, int64_t plane_stride,
                               hplStream stream) {
    const int Cp = (C + 7) / 8 * 8;
    HPL_REQUIRE(A && dst && rows > 0 && C > 0 && lda >= C && plane_stride >= rows * Cp * 2 && plane_stride % 16 == 0 &&
                    aligned16(dst),
                "hpl_rows_split3: bad arguments (rows=%lld C=%d)", (long long)rows, C);
    const bool vec = lda % 4 == 0 && aligned16(A);
    const int grid = (int)imin(cdiv(rows * (Cp / 8), 256), 65536);
    k_rows_split3<<<grid, 256, 0, to_stream(stream)>>>(A, lda, rows, C, Cp, reinterpret_cast<unsigned char *>(dst), plane_stride, vec);
    HPL_CHECK_LAUNCH("hpl_rows_split3");
    return HPL_OK;
}

extern "C" int hpl_weight_split3p(const float *Wt, int F, int C, int64_t ldw, void *dst, int64_t plane_stride,
                                  hplStream stream) {
    HPL_REQUIRE(Wt && dst && F > 0 && C > 0 && ldw > 0, "hpl_weight_split3p: bad arguments");
    const int Cp = (C + 7) / 8 * 8;
    const int64_t rows_out = cdiv((int64_t)F * Cp, 32) * 32;
    HPL_REQUIRE(plane_stride >= rows_out * ldw * 2 && plane_stride % 16 == 0 && aligned16(dst),
                "hpl_weight_split3p: destination too small (needs %lld bytes per plane)", (long long)(rows_out * ldw * 2));
    const int grid = (int)imin(cdiv(rows_out / 8 * ldw, 256), 16384);
    k_weight_split3p<<<grid, 256, 0, to_stream(stream)>>>(Wt, F, C, Cp, ldw, rows_out, reinterpret_cast<unsigned char *>(dst), plane_stride);
    HPL_CHECK_LAUNCH("hpl_weight_split3p");
    return HPL_OK;
}

bool hpl_gc::launch_split3b(GParams &p, hipStream_t s) {
    // both operands pre-split: A3 planes [rows][Cp] and the padded weight image (rows f*Cp + c)
    if (p.scat || p.C < 32 || p.F > 15 || p.N % 128 != 0 || (int64_t)p.F * p.Cp > 32768) return false;
    static const int min_rows = getenv("HPL_SPLIT3_MIN_ROWS") ? atoi(getenv("HPL_SPLIT3_MIN_ROWS")) : 8192;
    if (p.M < min_rows || p.N < 256) return false;
    static const int wide = getenv("HPL_SPLIT3_BN") ? atoi(getenv("HPL_SPLIT3_BN")) : 256;
    const bool bn256 = wide == 256 && p.N % 256 == 0;
    const int BN = bn256 ? 256 : 128;
    p.tiles_m = (int)cdiv(p.M, BMB);
    p.tiles_n = p.N / BN;
    if (p.tile_bm != BMB) p.tile_idx = nullptr;
    p.splits = 1; p.partial = nullptr;
    int grid = p.tiles_m * p.tiles_n;
    p.col_share = 0; p.col_rows = 0;
    if (p.row_perm) {
        int g = 8, b = p.tiles_n % 8;
        while (b) { const int tt = g % b; g = b; b = tt; }
        p.col_share = 8 / g;
        p.col_rows = p.col_share == 1 ? p.tiles_m : (int)cdiv(p.tiles_m, COL_CHUNK * p.col_share) * COL_CHUNK;
        grid = p.tiles_n * p.col_share * p.col_rows;
    }
    const bool f8 = p.F <= 8;
    if (bn256) {
        if (f8) k_gconv3b<2, 2, 8><<<grid, 512, 0, s>>>(p); else k_gconv3b<2, 2, 15><<<grid, 512, 0, s>>>(p);
    } else {
        if (f8) k_gconv3b<1, 3, 8><<<grid, 512, 0, s>>>(p); else k_gconv3b<1, 3, 15><<<grid, 512, 0, s>>>(p);
    }
    return true;
}
