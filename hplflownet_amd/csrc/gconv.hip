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

using namespace hpl;

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;   // contraction slice per step (floats) = one 128-byte line per gathered row

struct GParams {
    const float *A; int64_t lda; int64_t rows_a;
    const int32_t *nbr; int64_t nbr_stride; int64_t reg_stride;
    int64_t M; int C; int F; int K;
    const float *Wt; int64_t ldw; int N; int act; float slope;
    const float *bias; const float *res; int64_t ldres; int64_t res_mod;
    float *Y; int64_t ldy;
    const int32_t *scat; int64_t scat_stride; int scat_c;
    int tiles_m; int tiles_n;
};

__device__ __forceinline__ int64_t src_row(const GParams &p, int f, int64_t m) {
    if (f >= p.F || m >= p.M) return -1;
    if (p.nbr) return (int64_t)p.nbr[(int64_t)f * p.nbr_stride + m];
    return (int64_t)f * p.reg_stride + m;
}

// XCD-aware tile order: the hardware deals consecutive workgroup ids round-robin over the 8
// XCDs (private 4 MiB L2 each).  Re-number so that each XCD owns a contiguous run of tiles,
// and walk tiles_n fastest inside a band of 8 tile-rows so that concurrently resident
// workgroups of one XCD share both gathered A rows and weight panels in that L2.
__device__ __forceinline__ void tile_coords(const GParams &p, int &tm, int &tn) {
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = bid % 8, pos = bid / 8;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;   // bijective
    constexpr int BAND = 8;
    const int band_sz = BAND * p.tiles_n;
    const int band = id / band_sz;
    const int in_band = id - band * band_sz;
    const int rows_in_band = min(BAND, p.tiles_m - band * BAND);
    tm = band * BAND + in_band % rows_in_band;
    tn = in_band / rows_in_band;
}

template <int BM, int BN, int WGM, int WGN, bool AVEC>
__global__ void __launch_bounds__(64 * WGM * WGN) k_gconv(const GParams p) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int LDA_S = BM + 2;   // +2: the transposing ds_write_b32 of 8 lanes x 4 k stay <= 2-way
    constexpr int LDB_S = BN;
    constexpr int A_ROWS_PER_PASS = NT / 8;              // 8 float4 per gathered row slice
    constexpr int A_PASSES = BM / A_ROWS_PER_PASS;
    constexpr int B_F4_PER_ROW = BN / 4;
    constexpr int B_ROWS_PER_PASS = NT / B_F4_PER_ROW;
    constexpr int B_PASSES = BK / B_ROWS_PER_PASS;
    static_assert(A_PASSES >= 1 && B_PASSES >= 1, "tile too small for the thread count");

    __shared__ __attribute__((aligned(16))) float smem[2 * BK * LDA_S + 2 * BK * LDB_S];
    float *As = smem;
    float *Bs = smem + 2 * BK * LDA_S;

    int tile_m, tile_n;
    tile_coords(p, tile_m, tile_n);
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 31, hi = lane >> 5;

    // ---- A staging state: this thread always loads float4 column kq of the slice
    const int kq = t & 7;
    const int arow0 = t >> 3;
    int f_t = (kq * 4) / p.C;          // (f, c) of flat index k0 + kq*4 ; advanced by BK per step
    int c_t = (kq * 4) - f_t * p.C;
    // ---- B staging state
    const int bn4 = t % B_F4_PER_ROW;
    const int brow0 = t / B_F4_PER_ROW;

    float4 ra[A_PASSES];
    float4 rb[B_PASSES];

    auto load_regs = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const int64_t m = m0 + arow0 + i * A_ROWS_PER_PASS;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (AVEC) {
                const int64_t row = src_row(p, f_t, m);
                if (row >= 0) v = *reinterpret_cast<const float4 *>(p.A + row * p.lda + c_t);
            } else {   // generic path: any C / alignment, element by element
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + kq * 4 + j;
                    const int f = k / p.C, c = k - f * p.C;
                    const int64_t row = src_row(p, f, m);
                    e[j] = (row >= 0) ? p.A[row * p.lda + c] : 0.f;
                }
                v = make_float4(e[0], e[1], e[2], e[3]);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int kr = brow0 + i * B_ROWS_PER_PASS;
            const int col = n0 + bn4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col < p.ldw) v = *reinterpret_cast<const float4 *>(p.Wt + (int64_t)(k0 + kr) * p.ldw + col);
            rb[i] = v;
        }
        // advance (f, c) to the next slice
        c_t += BK;
        while (c_t >= p.C) { c_t -= p.C; ++f_t; }
    };

    auto store_lds = [&](int buf) {
        float *a = As + buf * BK * LDA_S;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const int r = arow0 + i * A_ROWS_PER_PASS;
            a[(kq * 4 + 0) * LDA_S + r] = ra[i].x;
            a[(kq * 4 + 1) * LDA_S + r] = ra[i].y;
            a[(kq * 4 + 2) * LDA_S + r] = ra[i].z;
            a[(kq * 4 + 3) * LDA_S + r] = ra[i].w;
        }
        float *b = Bs + buf * BK * LDB_S;
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int kr = brow0 + i * B_ROWS_PER_PASS;
            *reinterpret_cast<float4 *>(b + kr * LDB_S + bn4 * 4) = rb[i];
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    load_regs(0);
    store_lds(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1) < nk;
        if (more) load_regs((kt + 1) * BK);
        const float *a = As + cur * BK * LDA_S + wm * WTM + li;
        const float *b = Bs + cur * BK * LDB_S + wn * WTN + li;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = a[(kk * 2 + hi) * LDA_S + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = b[(kk * 2 + hi) * LDB_S + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_lds(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WTN + j * 32 + li;
            if (n >= p.N) continue;
            const float bsv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bsv;
                if (p.res) v += p.res[(m % p.res_mod) * p.ldres + n];
                if (p.act == HPL_ACT_LEAKY) v = v > 0.f ? v : p.slope * v;
                if (p.scat) {
                    const int k = n / p.scat_c, c = n - k * p.scat_c;
                    const int32_t tgt = p.scat[(int64_t)k * p.scat_stride + m];
                    if (tgt >= 0) atomicAdd(p.Y + (int64_t)tgt * p.ldy + c, v);
                } else {
                    p.Y[m * p.ldy + n] = v;
                }
            }
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
        }
    }
}

int fill_params(const hpl_gconv_desc *d, GParams &p, const char *who) {
    HPL_REQUIRE(d, "%s: null descriptor", who);
    HPL_REQUIRE(d->A && d->Wt && d->Y, "%s: null A / Wt / Y", who);
    HPL_REQUIRE(d->M >= 0 && d->C > 0 && d->F > 0 && d->N > 0, "%s: bad sizes M=%lld C=%d F=%d N=%d", who,
                (long long)d->M, d->C, d->F, d->N);
    HPL_REQUIRE((int64_t)d->F * d->C < (int64_t)INT32_MAX, "%s: contraction too long", who);
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
    p.scat = d->scat; p.scat_stride = d->scat_stride; p.scat_c = d->scat_c;
    p.tiles_m = p.tiles_n = 0;
    return HPL_OK;
}

template <int BM, int BN, int WGM, int WGN>
void launch_cfg(GParams &p, bool avec, hipStream_t s) {
    p.tiles_m = (int)cdiv(p.M, BM);
    p.tiles_n = (int)cdiv(p.N, BN);
    const int grid = p.tiles_m * p.tiles_n;
    if (avec) k_gconv<BM, BN, WGM, WGN, true><<<grid, 64 * WGM * WGN, 0, s>>>(p);
    else k_gconv<BM, BN, WGM, WGN, false><<<grid, 64 * WGM * WGN, 0, s>>>(p);
}

}  // namespace

extern "C" int hpl_gconv_forward(const hpl_gconv_desc *d, hplStream stream) {
    GParams p;
    int rc = fill_params(d, p, "hpl_gconv_forward");
    if (rc != HPL_OK) return rc;
    if (p.M == 0) return HPL_OK;
    hipStream_t s = to_stream(stream);
    const bool avec = (p.C % 4 == 0) && (p.lda % 4 == 0) && aligned16(p.A);
    // tile selection: widest N tile that is not mostly padding; BM = 64 when 128-row tiles
    // could not give every CU at least ~2 workgroups
    const int64_t CU2 = 512;
    if (p.N > 64) {
        launch_cfg<128, 128, 2, 2>(p, avec, s);
    } else if (p.N > 32) {
        if (cdiv(p.M, 128) >= CU2) launch_cfg<128, 64, 2, 2>(p, avec, s);
        else launch_cfg<64, 64, 2, 2>(p, avec, s);
    } else {
        if (cdiv(p.M, 128) >= CU2) launch_cfg<128, 32, 4, 1>(p, avec, s);
        else launch_cfg<64, 32, 2, 1>(p, avec, s);
    }
    HPL_CHECK_LAUNCH("hpl_gconv_forward");
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

__global__ void k_weight_unlayout(const float *__restrict__ Wt, int64_t ldw, int R, int Q, int F,
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
        float *dst = W + base + r * sr + q * sq + f * sf;
        *dst = accumulate ? *dst + v : v;
    }
}
}  // namespace

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

extern "C" int hpl_weight_unlayout(const float *Wt, int64_t ldw, int R, int Q, int F, float *W,
                                   int64_t base, int64_t sr, int64_t sq, int64_t sf, int accumulate,
                                   hplStream stream) {
    HPL_REQUIRE(W && Wt && R > 0 && Q > 0 && F > 0 && ldw >= Q, "hpl_weight_unlayout: bad arguments");
    const int grid = (int)imin(cdiv((int64_t)F * R * Q, 256), 8192);
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
struct WParams {
    const float *A; int64_t lda; const int32_t *nbr; int64_t nbr_stride; int64_t reg_stride;
    int64_t M; int C; int F; int K;
    const float *dY; int64_t lddy; int N;
    float *dWt; int64_t ldw;
    int tiles_n; int64_t m_per_split;
};

template <int BN, bool VEC>
__global__ void __launch_bounds__(256) k_wgrad(const WParams p) {
    constexpr int BKR = 128;                 // rows of dWt per tile (flat k)
    constexpr int BMS = 32;                  // vertices per step
    constexpr int WGM = 2, WGN = (BN >= 64) ? 2 : 1;
    constexpr int WGM_EFF = (BN >= 64) ? 2 : 4;
    constexpr int WTM = BKR / WGM_EFF, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_F4 = BKR / 4;            // float4 per gathered row slice
    constexpr int A_ROWS_PER_PASS = 256 / A_F4;
    constexpr int A_PASSES = BMS / A_ROWS_PER_PASS;
    constexpr int B_F4 = BN / 4;
    constexpr int B_ROWS_PER_PASS = 256 / B_F4;
    constexpr int B_PASSES = (BMS + B_ROWS_PER_PASS - 1) / B_ROWS_PER_PASS;
    (void)WGM;
    __shared__ __attribute__((aligned(16))) float smem[2 * BMS * BKR + 2 * BMS * BN];
    float *As = smem;
    float *Bs = smem + 2 * BMS * BKR;

    const int tile_k = blockIdx.x / p.tiles_n, tile_n = blockIdx.x % p.tiles_n;
    const int k0 = tile_k * BKR, n0 = tile_n * BN;
    const int64_t mb = (int64_t)blockIdx.y * p.m_per_split;
    const int64_t me = imin(p.M, mb + p.m_per_split);
    if (mb >= me) return;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 31, hi = lane >> 5;

    const int ak4 = t % A_F4, arow0 = t / A_F4;
    const int kk = k0 + ak4 * 4;                          // fixed flat k of this thread's float4
    const int f_t = kk / p.C, c_t = kk - f_t * p.C;
    const bool k_ok = kk < p.K;
    const int bn4 = t % B_F4, brow0 = t / B_F4;

    float4 ra[A_PASSES], rb[B_PASSES];
    auto load_regs = [&](int64_t ms) {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const int64_t m = ms + arow0 + i * A_ROWS_PER_PASS;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < me && k_ok) {
                if (VEC) {
                    int64_t row = p.nbr ? (int64_t)p.nbr[(int64_t)f_t * p.nbr_stride + m] : (int64_t)f_t * p.reg_stride + m;
                    if (row >= 0) v = *reinterpret_cast<const float4 *>(p.A + row * p.lda + c_t);
                } else {
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
            const int r = brow0 + i * B_ROWS_PER_PASS;
            const int64_t m = ms + r;
            const int col = n0 + bn4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < BMS && m < me) {
                if (VEC && col + 3 < p.N) v = *reinterpret_cast<const float4 *>(p.dY + m * p.lddy + col);
                else {
                    float e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) e[j] = (col + j < p.N) ? p.dY[m * p.lddy + col + j] : 0.f;
                    v = make_float4(e[0], e[1], e[2], e[3]);
                }
            }
            rb[i] = v;
        }
    };
    auto store_lds = [&](int buf) {
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
    load_regs(mb);
    store_lds(0);
    __syncthreads();
    int cur = 0;
    for (int64_t st = 0; st < nsteps; ++st) {
        const bool more = st + 1 < nsteps;
        if (more) load_regs(mb + (st + 1) * BMS);
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
        if (more) store_lds(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WTN + j * 32 + li;
            if (n >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (k < p.K) atomicAdd(p.dWt + (int64_t)k * p.ldw + n, acc[i][j][r]);
            }
        }
}
}  // namespace

extern "C" int hpl_gconv_wgrad(const float *A, int64_t lda, int64_t rows_a, const int32_t *nbr,
                               int64_t nbr_stride, int64_t reg_stride, int64_t M, int C, int F,
                               const float *dY, int64_t lddy, int N, float *dWt, int64_t ldw,
                               hplStream stream) {
    (void)rows_a;
    HPL_REQUIRE(A && dY && dWt, "hpl_gconv_wgrad: null pointer");
    HPL_REQUIRE(M >= 0 && C > 0 && F > 0 && N > 0 && lda >= C && lddy >= N && ldw >= N,
                "hpl_gconv_wgrad: bad sizes");
    HPL_REQUIRE(nbr || F == 1 || reg_stride > 0, "hpl_gconv_wgrad: F > 1 needs a table or reg_stride");
    if (M == 0) return HPL_OK;
    WParams p;
    p.A = A; p.lda = lda; p.nbr = nbr; p.nbr_stride = nbr_stride; p.reg_stride = reg_stride;
    p.M = M; p.C = C; p.F = F; p.K = F * C; p.dY = dY; p.lddy = lddy; p.N = N; p.dWt = dWt; p.ldw = ldw;
    const bool vec = (C % 4 == 0) && (lda % 4 == 0) && (lddy % 4 == 0) && aligned16(A) && aligned16(dY);
    const int bn = N > 64 ? 128 : (N > 32 ? 64 : 32);
    const int tiles_k = (int)cdiv(p.K, 128);
    p.tiles_n = (int)cdiv(N, bn);
    const int tiles = tiles_k * p.tiles_n;
    // split the vertex axis so that ~4 workgroups per CU exist, each with >= 256 vertices
    int64_t splits = imax(1, imin(cdiv(1024, tiles), cdiv(M, 256)));
    p.m_per_split = cdiv(cdiv(M, splits), 32) * 32;
    splits = cdiv(M, p.m_per_split);
    dim3 grid(tiles, (unsigned)splits);
    hipStream_t s = to_stream(stream);
#define LAUNCH(BN_) (vec ? k_wgrad<BN_, true><<<grid, 256, 0, s>>>(p) : k_wgrad<BN_, false><<<grid, 256, 0, s>>>(p))
    if (bn == 128) LAUNCH(128); else if (bn == 64) LAUNCH(64); else LAUNCH(32);
#undef LAUNCH
    HPL_CHECK_LAUNCH("hpl_gconv_wgrad");
    return HPL_OK;
}
