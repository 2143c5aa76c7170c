// gconv_common.h -- what the gather-GEMM kernels of gconv.hip (fp32 MFMA) and gconv3.hip (3 x bf16 split operands on the
// bf16 MFMA) share: launch parameters, buffer descriptors, the XCD-aware tile order.
#pragma once
#include "common.h"

namespace hpl_gc {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef int int32x4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));

// buffer_load_dwordx4 through the LLVM intrinsic (hipcc 7.2's __builtin_amdgcn_raw_buffer_load_b128
// lowers to a single-dword load, so the intrinsic is bound by name instead)
__device__ float4_t buffer_load_f32x4(int32x4_t srsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.load.v4f32");


// raw buffer descriptor: base pointer, stride 0, extent in bytes, gfx9 dword-3 flags (32-bit data format)
__device__ __forceinline__ int32x4_t make_rsrc(const void *base, int bytes) {
    union {
        int32x4_t v;
        struct { const void *p; int range; int cfg; } s;
    } u;
    u.s.p = base;
    u.s.range = bytes;
    u.s.cfg = 0x00020000;
    return u.v;
}

// fp16-pair operands (round 5): a matrix is multiplied by the power of two s that puts its largest magnitude `amax` into
// [2^14, 2^15) before it is split into fp16 hi + lo (hi cannot overflow -- fp16 reaches 65 504 --, lo of every element within
// 2^-18 of the largest is a normal fp16 number, smaller ones keep an absolute error of 2^-40 amax); the product is multiplied by
// 1 / s afterwards.  amax == 0 or denormal: s = 1.  NaN / inf in amax: the products are NaN / inf as in fp32.
__host__ __device__ __forceinline__ float split_scale(float amax) {
    unsigned bits;
    __builtin_memcpy(&bits, &amax, 4);
    const int e = (int)((bits >> 23) & 0xffu);
    int sb = 268 - e;                        // biased exponent of 2^(14 - (e - 127))
    sb = sb < 2 ? 2 : (sb > 252 ? 252 : sb);
    const unsigned sbits = e == 0 ? 0x3f800000u : ((unsigned)sb << 23);
    float s;
    __builtin_memcpy(&s, &sbits, 4);
    return s;
}
__host__ __device__ __forceinline__ float split_unscale(float s) {      // 1 / s for a power of two
    unsigned bits;
    __builtin_memcpy(&bits, &s, 4);
    const unsigned ibits = (254u - ((bits >> 23) & 0xffu)) << 23;
    float r;
    __builtin_memcpy(&r, &ibits, 4);
    return r;
}

constexpr int COL_CHUNK = 4;
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
    int64_t a_bytes; int64_t w_bytes;   // extents of A and Wt for the buffer descriptors
    const int32_t *row_perm;            // optional permutation of the output rows (tile row -> vertex)
    const int32_t *tile_idx;            // optional precomputed [tiles_m][F][BM] source rows + [tiles_m][8] masks
    const int32_t *tile_mask;           //   (hpl_tile_index; only when its BM is this launch's BM)
    int tile_bm;
    long long *clock_probe;             // optional: sampled workgroups add their residence in shader cycles / 100 MHz wall ticks
    int perm_chunk;                     // tile-rows per XCD chunk in permuted launches
    int col_share;                      // > 0: column-major XCD order, XCDs per column tile (see tile_coords)
    int col_rows;                       // tile-rows per virtual column in that order
    int splits; float *partial;         // split-K over the slice list: partial[split][M][N]
    float *Y2; int64_t ldy2; int64_t rows2;   // optional second destination: rows < rows2 are also written to Y2
    float *ws; int64_t ws_bytes;
    const void *Wt3; int64_t w3_plane_stride;      // split weight image (hpl_weight_split3 / hpl_weight_split2h) or nullptr
    int planes;                         // 3: bf16 triples; 2: fp16 pairs, with the largest magnitudes of A / of the weight image:
    const float *a_amax; const float *w_amax;      //   DEVICE scalars (hpl_amax; hpl_weight_split2h)
    float *y_amax; int y_amax_done;     // optional: largest |Y| stored -> *y_amax (done = 1: the kernel's epilogue did it)
    const unsigned *a_guard; unsigned *y_guard; int *guard_trips;      // range guard of the pair form (hpl_gconv_desc.a_guard ...)
    int guard_partials;                 // split-K launch with a guarded operand: a tripped guard leaves a SECOND set of `splits` partial tiles
    int epi_fast;                       // 32-bit buffer addressing in the epilogue (set by the launch functions)
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
// (block: the workgroup id the tile is derived from -- blockIdx.x, or the virtual id of a persistent launch: gconv3.hip's guard launches)
__device__ __forceinline__ void tile_coords(const GParams &p, int &tm, int &tn, const unsigned block) {
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = (int)block % nwg;       // (split-K: grid = splits x tiles)
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = bid % 8, pos = bid / 8;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;   // bijective
    if (p.row_perm && p.col_share < 0) {
        // Row-major order (round 6; the split-operand launches, gconv3.hip launch_split3): XCD x owns the scheduled tile-rows x, x + 8, ... (heaviest first) and
        // walks the column tiles of a tile-row back to back, so the tiles_n column tiles of one tile-row are resident on ONE XCD
        // at the same time: its gathered A rows are fetched into that L2 once, not once per column tile; the weight panels of all
        // column tiles then stream through every L2 (from the Infinity Cache).  Grid = 8 * col_rows * tiles_n.
        const int x2 = (int)block % 8, pos2 = (int)block / 8;
        const int j = pos2 / p.tiles_n;
        tn = pos2 - j * p.tiles_n;
        const int row = j * 8 + x2;
        tm = row < p.tiles_m ? (p.tile_idx ? p.tile_mask[(int64_t)row * 8 + 6] : p.tiles_m - 1 - row) : -1;
        return;
    }
    if (p.row_perm && p.col_share > 0) {
        // Column-major order: an XCD works through ONE column tile at a time (its weight panel stays in that
        // L2), heaviest tile-rows first.  Column tiles shared by col_share XCDs deal their tile-rows round
        // robin.  The grid is rounded up to 8 equal runs; surplus workgroups get tm = -1.
        const int gsz = p.tiles_n * p.col_share * p.col_rows;
        const int q2 = gsz / 8, r2 = gsz % 8;
        const int b2 = p.splits > 1 ? (int)block % gsz : (int)block;      // (split-K: grid = splits x gsz)
        const int x2 = b2 % 8, pos2 = b2 / 8;
        const int id2 = (x2 < r2 ? x2 * (q2 + 1) : r2 * (q2 + 1) + (x2 - r2) * q2) + pos2;
        const int v = id2 / p.col_rows, pos = id2 - v * p.col_rows;
        tn = v / p.col_share;
        // XCDs sharing a column tile take its scheduled tile-rows in chunks of COL_CHUNK: tiles that are neighbours in
        // the schedule (= in the row order) run on one XCD and share gathered rows in its L2
        const int sh = v - tn * p.col_share;
        const int row = p.col_share == 1 ? pos : (pos / COL_CHUNK) * COL_CHUNK * p.col_share + sh * COL_CHUNK + pos % COL_CHUNK;
        tm = row < p.tiles_m ? (p.tile_idx ? p.tile_mask[(int64_t)row * 8 + 6] : p.tiles_m - 1 - row) : -1;
        return;
    }
    if (p.row_perm) {
        // Rows are sorted by tap mask: tile-rows differ in work (few taps ... all taps) and have no
        // spatial coherence.  Tile-rows are dealt to the 8 XCDs in chunks of PERM_CHUNK consecutive
        // (= similar mask, similar slice list) tile-rows, heaviest chunk first, in snake order
        // (0..7, 7..0, ...) so that XCDs finish together; the co-resident workgroups of an XCD then
        // walk nearly the same slices and share weight panels in its L2, and the column tiles of
        // one tile-row stay adjacent and share its gathered rows.
        const int G = p.perm_chunk;
        const int T = p.tiles_m;
        const int nchunks = (T + G - 1) / G;
        auto chunk_rows = [&](int c) { return min(G, T - c * G); };
        auto chunk_of = [&](int x, int r) { return r * 8 + ((r & 1) ? 7 - x : x); };   // round r of XCD x
        int sq = id / p.tiles_n;            // position in the XCD-major sequence of tile-rows
        tn = id - sq * p.tiles_n;
        int x = 0;
        for (; x < 8; ++x) {
            int rows_x = 0;
            for (int r = 0; r * 8 < nchunks; ++r) {      // rounds that contain at least one chunk
                const int c = chunk_of(x, r);
                if (c < nchunks) rows_x += chunk_rows(c);
            }
            if (sq < rows_x) break;
            sq -= rows_x;
        }
        int row = 0;
        for (int r = 0;; ++r) {
            const int c = chunk_of(x, r);
            if (c >= nchunks) continue;
            const int n = chunk_rows(c);
            if (sq < n) { row = c * G + sq; break; }
            sq -= n;
        }
        tm = T - 1 - row;
        return;
    }
    constexpr int BAND = 8;
    const int band_sz = BAND * p.tiles_n;
    const int band = id / band_sz;
    const int in_band = id - band * band_sz;
    const int rows_in_band = min(BAND, p.tiles_m - band * BAND);
    tm = band * BAND + in_band % rows_in_band;
    tn = in_band / rows_in_band;
}
__device__ __forceinline__ void tile_coords(const GParams &p, int &tm, int &tn) { tile_coords(p, tm, tn, blockIdx.x); }


// host side (gconv.hip / gconv3.hip)
int fill_params(const hpl_gconv_desc *d, GParams &p, const char *who);
// the split-operand kernel (gconv3.hip): true if it took the launch
bool launch_split3(GParams &p, hipStream_t s);
// HPL_MATH=f32 keeps every launch on the fp32 MFMA (A/B runs, parity baselines)
bool split3_enabled();
// operand planes of the split kernels: 2 = fp16 pairs (default), 3 = bf16 triples (HPL_MATH=bf16x3), 0 = HPL_MATH=f32
int split_planes();
// *slot = max(*slot, largest magnitude of X[0 .. rows)[0 .. cols)) (gconv3.hip; the caller cleared the slot)
// guard (optional, cleared by the caller): the range-guard word of X (hpl_amax_rows)
int amax_launch(const float *X, int64_t ld, int64_t rows, int cols, float *slot, hipStream_t s, unsigned *guard = nullptr);
// the pair form's range guard: a second launch over the residuals when the largest magnitude and the smallest non-zero row maximum
// are >= 2^18 apart (GParams.a_guard: ~bits of that row maximum, 0 = unknown)
constexpr int GUARD_GAP = 18;
__device__ __forceinline__ bool guard_tripped(const float *a_amax, const unsigned *a_guard) {
    if (!a_guard || !a_amax) return false;
    unsigned am;
    const float amf = a_amax[0];
    __builtin_memcpy(&am, &amf, 4);
    const unsigned g = a_guard[0];
    const unsigned rmin = g ? ~g : am;
    return (am >> 23) >= (rmin >> 23) + (unsigned)GUARD_GAP && (am >> 23) < 255u;
}
constexpr int RESID_SHIFT = 24;          // residual scale 2^24: (a s - hi - lo) 2^24 < 2^15.01 for every a s < 2^15
// the cheap part of launch_split3's test (the callers decide with it whether to compute the largest magnitude of A)
inline bool split3_maybe(int64_t M, int C, int F, int N) {
    if (!(C >= 32 && N >= 256 && M >= 1024 && F <= 15)) return false;
    // ... and the shape tests of launch_split3 (round 6: the 1x1 convs of level 3 -- 1 787 rows -- had their operand reduced, 7 us
    // each, and then ran on the fp32 kernel): dense launches below 8 192 rows that do not fill half the CUs with 128 x 256 tiles, and
    // mid-size stencils that cannot be split over K into one round of workgroups, stay on the fp32 MFMA
    const int64_t tiles256 = ((M + 127) / 128) * ((N + 255) / 256);
    if (tiles256 >= 128) return true;
    if (F == 1) return M >= 8192;
    if (M >= 16384) return true;
    const int64_t nk = ((int64_t)F * C + BK - 1) / BK;
    const int64_t by_tiles = 256 / tiles256, by_k = nk / 16;
    return N % 256 == 0 && (by_tiles < by_k ? by_tiles : by_k) >= 2;
}

// weight gradient (gconv.hip: fp32 MFMA; wgrad3.hip: split operands on the bf16 MFMA)
struct WParams {
    const float *A; int64_t lda; const int32_t *nbr; int64_t nbr_stride; int64_t reg_stride;
    int64_t M; int C; int F; int K;
    const float *dY; int64_t lddy; int N;
    float *dWt; int64_t ldw;
    int tiles_n; int64_t m_per_split;
    // tap mode: k tiles are (tap f, 128-channel block) and the vertex loop of tap f runs over its
    // compacted list of present (vertex, source row) pairs tap_m / tap_row[tap_ptr[f] .. tap_ptr[f+1])
    // (hpl_tap_lists)
    const int32_t *tap_m; const int32_t *tap_row; const int32_t *tap_ptr; int c_tiles;
    float *dbias;        // optional: dbias[n] += sum_m dY[m, n] (by the k-tile-0 workgroups; not in tap mode)
    int64_t rows_a;      // rows of A (0 = unknown)
    const float *a_amax; const float *dy_amax;     // optional DEVICE scalars (hpl_amax of A / dY): both given = fp16-pair operands
};
// the split-operand weight gradient (wgrad3.hip): true if it took the launch.  tap: p.tap_* are set (k tiles = (tap, channel
// block)); m_len = longest vertex loop of a tile
bool launch_wgrad3(WParams &p, bool tap, int64_t m_len, hipStream_t s);

}  // namespace hpl_gc
