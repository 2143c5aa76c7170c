/*
 * hpl_bcl.h -- C ABI of libhplbcl.so: the MI355X (gfx950) implementation of
 * HPLFlowNet's bilateral-convolution hot path.
 *
 * This is the drop-in boundary (SURVEY.md §8 b).  The reference has no native ops
 * for the layer half of the path (it composes ATen calls) and a 4-function khash
 * FFI for the lattice half; each entry point below names the reference lines it
 * replaces (paths relative to the reference repository root).  All pointers are
 * DEVICE pointers unless marked HOST; all kernels are enqueued on `stream`
 * (a hipStream_t passed as void*) and return without synchronising.  No torch
 * types appear here: hplflownet_amd/_lib.py binds this header with ctypes and
 * passes tensor.data_ptr() values.
 *
 * Conventions
 *   - activations are CHANNEL-LAST float32: X[row * ld + channel]; `ld` (row stride
 *     in floats) lets a kernel read or write a column slice of a wider buffer, which
 *     is how the torch.cat calls of models/HPLFlowNet.py:242-423 disappear;
 *   - index tables are int32 (converted once per sample from the reference's int64
 *     wire format); -1 means "no such lattice vertex" and reads as
 *     an all-zero row (the reference's "+1 / zero column" trick,
 *     models/bilateralNN.py:158-162,215-217);
 *   - float4 fast paths need ld % 4 == 0, channel counts % 4 == 0 and 16-byte
 *     aligned bases; other shapes take a scalar path (same results).
 *   - return value: 0 on success, negative HPL_E* on failure; hpl_last_error()
 *     gives the message of the calling thread's last failure.
 */
#ifndef HPL_BCL_H
#define HPL_BCL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPL_OK 0
#define HPL_EINVAL (-1)   /* bad argument (shape, alignment, null pointer) */
#define HPL_EHIP (-2)     /* a HIP runtime call or kernel launch failed */
#define HPL_ENODEV (-3)   /* no gfx950 device visible */

#define HPL_ACT_NONE 0
#define HPL_ACT_LEAKY 1   /* y > 0 ? y : slope * y   (models/module_utils.py:6,14) */

typedef void *hplStream;

int hpl_version(void);
const char *hpl_last_error(void);
/* HOST outputs. arch receives e.g. "gfx950". */
int hpl_device_info(int device, int *cu_count, int *wave_size, char *arch, int arch_len);

/* ------------------------------------------------------------------------ *
 * Index tables
 * ------------------------------------------------------------------------ */
/* int64 -> int32 narrowing of any reference table (lattice_offset, blur_neighbors,
 * pc1_corr_indices; transforms/transforms.py:471-483). */
int hpl_index_narrow(const int64_t *src, int32_t *dst, int64_t n, hplStream stream);

/* pc2_corr_indices [F][K][H] int64 (models/bnn_flow.py:195-197, index order
 * transforms/transforms.py:232-241) -> int32 [K][F*H]: row k of the result is the
 * neighbour table of the F*H "virtual vertices" m = f*H + h used by the
 * patch-correlation gather-GEMM. */
int hpl_corr2_permute(const int64_t *src, int32_t *dst, int F, int K, int64_t H, hplStream stream);
/* same from an int32 [F][K][H] source (device-built lattices) */
int hpl_corr2_permute32(const int32_t *src, int32_t *dst, int F, int K, int64_t H, hplStream stream);

/* CSR of the splat: for `off` (int32, n_entries values in [0,H); for a lattice_offset table
 * entry e = r*N + n and n_entries = 4*N) and per-entry weights `bary` ([n_entries]) produce
 *   csr_ptr [H+1]        segment bounds per vertex,
 *   csr_pt  [n_entries]  source row (e % pt_mod; pt_mod = N gives the point index) of each
 *                        contribution, ascending e within a vertex,
 *   csr_w   [n_entries]  its weight,
 *   norm    [H]          1 / (sum of weights + 1e-5)   (models/bilateralNN.py:168-183).
 * `scratch` needs (H + 1) + n_entries + 1026 int32.  Replaces the COO coalesce inside
 * torch.sparse.FloatTensor(...).to_dense() (models/bilateralNN.py:24-29): the sort is
 * done once per lattice instead of once per layer call, and is deterministic. */
int hpl_csr_build(const int32_t *off, const float *bary, int64_t n_entries, int64_t pt_mod, int64_t H,
                  int32_t *csr_ptr, int32_t *csr_pt, float *csr_w, float *norm,
                  int32_t *scratch, hplStream stream);
/* The same for a pair of clouds treated as one: points [0,N0) + [N0,N0+N1), vertices [0,H0) +
 * [H0,H0+H1) (off1 values are shifted by H0 here).  off0, off1, bary0, bary1 are the per-cloud [4][N] tables.
 * Outputs sized for H = H0+H1 and 4*(N0+N1) entries; scratch (H+1) + 4*(N0+N1) + 1026 int32.
 * A vertex belongs to one cloud, so each segment is identical to the per-cloud CSR. */
int hpl_csr_build_pair(const int32_t *off0, const float *bary0, int64_t N0, int64_t H0,
                       const int32_t *off1, const float *bary1, int64_t N1, int64_t H1,
                       int32_t *csr_ptr, int32_t *csr_pt, float *csr_w, float *norm,
                       int32_t *scratch, hplStream stream);

/* ------------------------------------------------------------------------ *
 * Splat / slice  (HBM-bound gathers)
 * ------------------------------------------------------------------------ */
/* out[v, c] = (norm ? norm[v] : 1) * sum_{j in csr(v)} csr_w[j] * feat[csr_pt[j], c]
 * Forward splat + density normalisation: models/bilateralNN.py:151-186 and
 * models/bnn_flow.py:119-151 (SparseSum.forward, bilateralNN.py:9-30).
 * With norm == NULL it is also the backward of hpl_slice w.r.t. Y. */
int hpl_splat(const float *feat, int64_t ldf, int C, const int32_t *csr_ptr, const int32_t *csr_pt,
              const float *csr_w, const float *norm, int64_t H, float *out, int64_t ldo,
              hplStream stream);

/* out[n, c] = (bias ? bias[c] : 0) + sum_{r<4} bary[r*N+n] * (vscale ? vscale[v] : 1) * Y[v, c],
 * v = off[r*N+n].   Slice + bias: models/bilateralNN.py:223-238.  With vscale = norm it is
 * also the backward of hpl_splat w.r.t. feat (SparseSum.backward, bilateralNN.py:33-40). */
int hpl_slice(const float *Y, int64_t ldy, int C, const float *bary, const int32_t *off, int64_t N,
              const float *vscale, const float *bias, float *out, int64_t ldo, hplStream stream);
/* The same two kernels ADDING to what `out` holds: a feature matrix with several consumers collects their gradients
 * (autograd's accumulation of models/HPLFlowNet.py's torch.cat / reuse of a tensor). */
int hpl_splat_add(const float *feat, int64_t ldf, int C, const int32_t *csr_ptr, const int32_t *csr_pt,
                  const float *csr_w, const float *norm, int64_t H, float *out, int64_t ldo,
                  hplStream stream);
int hpl_slice_add(const float *Y, int64_t ldy, int C, const float *bary, const int32_t *off, int64_t N,
                  const float *vscale, const float *bias, float *out, int64_t ldo, hplStream stream);

/* Patch correlation from per-tap projections.  The pc2 half of the Conv3d((1,15,1)) of models/bnn_flow.py:195-202 gathers the
 * same pc2 vertex row for up to 225 (filter tap, correlation tap) pairs and multiplies each copy by W_k; projecting every pc2
 * vertex ONCE per correlation tap, Z[v, k*N + n] = sum_c f2[v, c] * W[n, c, k] (a small dense GEMM), leaves
 *   Y[m, n] = act(bias[n] + res[(m % res_mod) * ldres + n] + sum_{k<K} Z[nbr[k][m] * ldz + k*col_step + n])    (nbr < 0: nothing)
 * with col_step = N -- a gather of N-float rows instead of C-float rows and no 8.7-GFLOP contraction (N = 32, C = 64,
 * M = 15 * H1).  N % 4 == 0, K <= 15, 16-byte aligned rows; taps are added in ascending k (deterministic).
 * Its gradient w.r.t. Z is the same kernel through the INVERSE table (hpl_table_invert) with col_step = 0:
 * dZ[(v*K + k)*N + n] = sum_f G[inv[f][v*K + k] * ldg + n] -- no atomics. */
int hpl_gather_sum(const float *Z, int64_t ldz, const int32_t *nbr, int64_t nbr_stride, int64_t M, int K, int N, int col_step,
                   const float *bias, const float *res, int64_t ldres, int64_t res_mod, int act, float slope, float *Y,
                   int64_t ldy, hplStream stream);
/* T [K][stride >= F*H0] with values in [0, H1) or -1 whose F blocks of H0 columns are injective (the permuted pc2_corr_indices:
 * for a fixed (filter tap f, correlation tap k) the map h -> T2(key1_h + offset) is one to one -- also under the reference's
 * unchecked key packing, which is linear in the key) -> inv int32 [F][K*H1]: inv[f][v*K + k] = f*H0 + h where
 * T[k][f*H0 + h] = v, else -1. */
int hpl_table_invert(const int32_t *T, int64_t stride, int K, int64_t H0, int F, int64_t H1, int32_t *inv, hplStream stream);

/* ------------------------------------------------------------------------ *
 * Per-vertex dense contraction: gather-GEMM on fp32 MFMA
 * ------------------------------------------------------------------------ */
/* Weight re-layout.  Source element (r, q, f) lives at W[base + r*sr + q*sq + f*sf];
 * destination Wt[(fdst*R + r) * ldw + q] with fdst = fmap ? fmap[f] : f.  Rows beyond
 * F*R (up to k_rows) and columns beyond Q (up to ldw) are zero-filled.
 *   forward  of Conv2d (O,C,F,1) / Conv3d (O,Ctot,1,K,1) / Conv1d (O,C,1):
 *            r = input channel, q = output channel  -> Wt[(f*C + c)][o]
 *   backward-data: r = output channel, q = input channel, fmap[f] = (15-f)%15 style
 *            mirror. */
int hpl_weight_relayout(const float *W, int64_t base, int R, int Q, int F, int64_t sr, int64_t sq,
                        int64_t sf, const int32_t *fmap, float *Wt, int64_t k_rows, int64_t ldw,
                        hplStream stream);
/* Many re-layouts in one launch (training re-lays every conv weight after each optimiser step, in
 * the forward and in the data-gradient direction).  jobs / prefix live in DEVICE memory; job j writes
 * destination elements [prefix[j], prefix[j+1]) of dst as an image [k_rows][ldw] with the meaning of
 * hpl_weight_relayout; mirror != 0 stores tap f at position (F - f) % F.  total = prefix[njobs]. */
typedef struct hpl_relayout_job {
    const float *W;         /* source parameter */
    int64_t base, sr, sq, sf;
    int32_t R, Q, F, mirror;
    int64_t ldw;            /* row length of the destination image (multiple of 4) */
} hpl_relayout_job;
int hpl_weight_relayout_batch(const hpl_relayout_job *jobs /* DEVICE */, int njobs,
                              const int64_t *prefix /* DEVICE, njobs + 1 */, int64_t total, float *dst,
                              hplStream stream);
/* mirror == 2 ("taps as column blocks"): the image is [roundup(R, 32)][ldw >= F*Q] with element (r, f*Q + q) =
 * W[base + r*sr + q*sq + f*sf] -- the operand of the scatter / regular data gradients of the correlation layer
 * (models/bnn_flow.py:202-205 backward: G[m, (f, c)] = g[m] . W[:, c, f]). */
/* The inverse of the batch call, for the weight gradients of a whole training step: job j reads its image
 * [k_rows][ldw] at src + (prefix[j] - prefix[0]) and writes W[base + r*sr + q*sq + f*sf] = src_j[(f*R + r)*ldw + q]
 * (mirror 0 or 2); W is then the job's gradient tensor in the parameter's own layout (main.py:214 loss.backward()).
 * total = elements of all images (sizes the grid). */
int hpl_weight_unlayout_batch(const hpl_relayout_job *jobs /* DEVICE */, int njobs,
                              const int64_t *prefix /* DEVICE, njobs + 1 */, int64_t total, const float *src,
                              hplStream stream);

/* Split weight image for the bf16-MFMA path: Wt [k_rows][ldw] fp32 (k_rows % 8 == 0) -> three planes p = 0 (hi),
 * 1 (mid), 2 (lo) of bf16, each [k_rows/8][ldw][8] (element (k, n) at ((k/8)*ldw + n)*8 + k%8: the 8 consecutive k of
 * one output column are one 16-byte MFMA B fragment), plane p at dst + p*plane_stride bytes; hi = bf16_rne(w),
 * mid = bf16_rne(w - hi), lo = bf16_rne(w - hi - mid): w == hi + mid + lo exactly.  plane_stride >= k_rows*ldw*2. */
int hpl_weight_split3(const float *Wt, int64_t k_rows, int64_t ldw, void *dst, int64_t plane_stride, hplStream stream);
/* The fp16-pair form (hpl_gconv_desc.wt3_planes == 2): *amax (DEVICE) = the image's largest magnitude, s = the power of two
 * that puts it into [2^14, 2^15); two planes of fp16 in the same fragment order, hi = f16_rne(w s), lo = f16_rne(w s - hi). */
int hpl_weight_split2h(const float *Wt, int64_t k_rows, int64_t ldw, void *dst, int64_t plane_stride, float *amax,
                       hplStream stream);
/* The same for many images in one launch (a training step re-splits every wide image after the optimiser step): jobs in DEVICE
 * memory, max_elems = the largest k_rows * ldw among them (sizes the grid). */
typedef struct hpl_split3_job {
    const float *Wt;
    void *dst;
    int64_t k_rows, ldw, plane_stride;
    float *amax;              /* planes == 2: DEVICE scalar the job's largest magnitude is written to */
    int32_t planes;           /* 2: fp16 pairs (hpl_weight_split2h); 0 / 3: bf16 triples */
    int32_t pad_;
} hpl_split3_job;
/* any_pairs: nonzero when any job has planes == 2 (their largest magnitudes are reduced first: two more launches). */
int hpl_weight_split3_batch(const hpl_split3_job *jobs /* DEVICE */, int njobs, int64_t max_elems, int any_pairs,
                            hplStream stream);

/* inverse scatter for weight gradients: W[base + r*sr + q*sq + f*sf] (+)= Wt[(f*R + r)*ldw + q] */
int hpl_weight_unlayout(const float *Wt, int64_t ldw, int R, int Q, int F, float *W, int64_t base,
                        int64_t sr, int64_t sq, int64_t sf, int accumulate, hplStream stream);

typedef struct hpl_gconv_desc {
    /* A operand: rows of a channel-last matrix, gathered through a neighbour table */
    const float *A;         /* [rows_a][lda] */
    int64_t lda;
    int64_t rows_a;         /* rows addressable in A (indices are checked against it in debug) */
    const int32_t *nbr;     /* nbr[f * nbr_stride + m] in [-1, rows_a); NULL: see reg_stride */
    int64_t nbr_stride;
    int64_t reg_stride;     /* used when nbr == NULL: source row = f * reg_stride + m
                               (F == 1, reg_stride == 0 is a plain GEMM) */
    int64_t M;              /* output rows (lattice vertices, or F*H virtual vertices) */
    int32_t C;              /* channels taken from each gathered row */
    int32_t F;              /* filter taps, <= 15 per call (radius 1); contraction length K = F * C <= 32768.
                             * Wider stencils (radius 2: 65 taps) = consecutive calls over tap ranges, each a row
                             * range of nbr and of Wt (w_rows), accumulating through res = Y */
    /* B operand: re-laid-out weights (hpl_weight_relayout) */
    const float *Wt;        /* [>= roundup(F*C, 32)][ldw], zero padded */
    int64_t ldw;            /* multiple of 4, >= N */
    int32_t N;              /* output channels */
    int32_t act;            /* HPL_ACT_* applied after bias and residual */
    float slope;
    const float *bias;      /* [N] or NULL */
    const float *res;       /* optional addend res[(m % res_mod) * ldres + n] (before act) */
    int64_t ldres;
    int64_t res_mod;
    float *Y;               /* [M][ldy] */
    int64_t ldy;
    /* optional scatter epilogue (backward-data through a non-symmetric table): when
     * scat != NULL the result is NOT stored to Y[m]; column n = k*scat_c + c is
     * atomically added to Y[scat[k*scat_stride + m] * ldy + c] (skipped when < 0). */
    const int32_t *scat;
    int64_t scat_stride;
    int32_t scat_c;
    /* rows of Wt that exist (0 = roundup(F*C, 32), i.e. a zero-padded image).  A smaller value
     * (>= F*C) lets Wt be a row range of a bigger image -- one tap group of it: rows past w_rows
     * read as zero. */
    int32_t w_rows;
    /* optional permutation of the M output rows (hpl_tap_order): tile row j computes and writes
     * output row row_perm[j]; the result is unchanged, rows of one tile share absent taps so that
     * whole contraction slices can be skipped.  NULL = identity. */
    const int32_t *row_perm;
    /* optional workspace for split-K (M*N*4 bytes per split, up to 16 splits; small and mid-size launches);
     * NULL = never split.  Partial tiles are summed in split order (cut points are slice indices): results are
     * deterministic and independent of the row order. */
    float *ws;
    int64_t ws_bytes;
    /* optional, with row_perm: per-tile gather indices and tap masks precomputed by hpl_tile_index for tiles of
     * tile_bm rows.  Used when the launch picks that tile height (ignored otherwise): the tile prologue is then
     * one round of coalesced loads instead of the dependent chain row_perm -> nbr -> masks. */
    const int32_t *tile_idx;
    const int32_t *tile_mask;
    int32_t tile_bm;
    /* optional diagnostic (DEVICE, 2 x int64, zeroed by the caller): every 64th workgroup adds its residence time in
     * shader cycles to [0] and in 100 MHz wall ticks to [1] -- [0] / [1] * 100 MHz is the clock the chip sustains
     * under this kernel (it clocks to its power budget). */
    int64_t *clock_probe;
    /* optional second destination (forward only, not with scat): rows m < rows2 of the result are ALSO stored to
     * Y2[m * ldy2 + n] -- a layer's output that feeds two concatenation buffers is written once by its producer
     * instead of copied (the reference builds both with torch.cat, models/HPLFlowNet.py:299-393). */
    float *Y2;
    int64_t ldy2;
    int64_t rows2;
    /* optional: the same weights as three bf16 planes (hpl_weight_split3 of the image Wt points into, at the same
     * first row, which must be a multiple of 8).  Launches that qualify (wide row-ordered stencil passes) then run
     * on the bf16 matrix pipe with every fp32 operand carried EXACTLY as hi + mid + lo bf16 terms and the six
     * leading partial products accumulated in fp32 (csrc/gconv3.hip: per-product error <= 2^-24 relative, the fp32
     * rounding class; 16/6 of the fp32-MFMA rate).  NULL: fp32 MFMA. */
    const void *Wt3;
    int64_t wt3_plane_stride;   /* bytes between the planes */
    /* wt3_planes == 2 (round 5): Wt3 is hpl_weight_split2h of the image -- two fp16 planes hi / lo of w * s_w -- and the
     * launch splits A the same way: a * s_a = hi + lo (one rounding each), the three partial products hi*hi + hi*lo + lo*hi
     * on the fp16 MFMA, fp32 accumulate, the result times 1 / (s_a s_w).  s = the power of two that puts the matrix's
     * largest magnitude into [2^14, 2^15): a_amax / w_amax are DEVICE scalars holding those magnitudes (hpl_amax of the
     * rows and channels of A the launch can read -- or of any superset --; the one hpl_weight_split2h wrote).  Per product
     * the error is <= 2^-21 |a b| for every a within 2^-18 of the largest (smaller ones: 2^-40 of the largest, absolute), per
     * output <= (2^-21 + (3K/16) 2^-24) sum|a||w| -- below the (K - 1) 2^-24 sum|a||w| of an fp32 dot product of length K >= 12
     * in any order; measured against float64 the sums are closer than the fp32 MFMA's and the triples'
     * (tests/test_gpu_split3.py) at half the bf16-triple form's MFMA work.  Both scalars must be given
     * (else the launch runs on the fp32 MFMA).  wt3_planes == 0 / 3: bf16 triples as above. */
    int32_t wt3_planes;
    const float *a_amax;
    const float *w_amax;
    /* optional (DEVICE, cleared by the caller; not with scat): *y_amax = max(*y_amax, largest |Y[m][n]| this call stores) -- the
     * a_amax of a wide launch that reads Y next, from the epilogue's registers where the launch allows, else by one more pass. */
    float *y_amax;
    /* Range guard of the fp16-pair form (round 6; all optional, wt3_planes == 2 only).  ONE scale per matrix leaves a row whose
     * entries are all below 2^-18 of the matrix's largest magnitude with an ABSOLUTE error (2^-40 of that magnitude) -- fine for
     * the sums of loud rows, not for a quiet row's own outputs.  a_guard: DEVICE word written by hpl_amax_rows (or left by a
     * producer through y_guard): the smallest non-zero ROW maximum of A, stored as ~bits (0 = unknown: no guard).  When the
     * largest magnitude exceeds it by 2^18 or more (exponent gap), the launch runs a SECOND pass over its slice list with the
     * residuals of the first split, (a s - hi - lo) 2^24, again as fp16 pairs, into the same accumulators: every element then
     * carries >= 21 significand bits down to 2^-41 of the largest magnitude (2^-63 of it, absolute, below that) -- fp32-class
     * results per OUTPUT ROW over 12 decades of row loudness, at twice the matrix-pipe work of that launch only.  y_guard: the companion of y_amax (cleared by the
     * caller): the guard word of Y from this launch's epilogue (row maxima over the 64 columns a wave holds: never larger
     * than the true row maximum, i.e. conservative), else by hpl_amax_rows.  guard_trips: DEVICE counter, +1 per launch that
     * took the second pass. */
    const uint32_t *a_guard;
    uint32_t *y_guard;
    int32_t *guard_trips;
} hpl_gconv_desc;

/* Largest magnitude of X[0 .. rows)[0 .. cols) (row stride ld) -> *slot (DEVICE; NaN if X holds one). */
int hpl_amax(const float *X, int64_t ld, int64_t rows, int32_t cols, float *slot, hplStream stream);
/* The same, and the range-guard word of X (hpl_gconv_desc.a_guard): *guard = max over the rows with a non-zero entry of
 * ~bits(largest magnitude of the row), 0 if every row is zero.  Both slots are cleared by the call. */
int hpl_amax_rows(const float *X, int64_t ld, int64_t rows, int32_t cols, float *slot, uint32_t *guard, hplStream stream);

/* Row order for tap skipping: perm = the M vertices grouped by their F-bit tap-presence mask (bit f set iff
 * nbr[f*nbr_stride + m] >= 0; F <= 15), the groups in Gray-code order of their masks (neighbouring groups differ in
 * one tap), ties by ascending row id (a stable radix sort: the order is deterministic and does not affect results).  scratch: hpl_tap_order_scratch_ints(M) int32, 8-byte aligned. */
int64_t hpl_tap_order_scratch_ints(int64_t M);
int hpl_tap_order(const int32_t *nbr, int64_t nbr_stride, int F, int64_t M, int32_t *perm,
                  int32_t *scratch, hplStream stream);
/* Per-tile gather indices of a row-ordered launch: tile j covers output rows row_perm[j*BM .. j*BM+BM) (identity
 * when row_perm is NULL); tile_idx[j][f][r] = nbr[f][row_perm[j*BM + r]] (-1 past M), tile_mask[j][0] = taps present
 * in the tile, [2 + b] = taps present in its b-th block of 32 rows, [j][6] = the tile that is scheduled j-th (most taps
 * first: a launch's workgroups then finish within one light tile of each other; equal tap counts in tile order, so
 * that tiles scheduled back to back are neighbours in the row order), the rest 0.  Sizes: ceil(M/BM)*F*BM and
 * ceil(M/BM)*8 int32.  Built once per lattice and row order (models/bilateralNN.py:215-217 gathers through the same
 * table in every layer call). */
int hpl_tile_index(const int32_t *nbr, int64_t nbr_stride, int F, int64_t M, const int32_t *row_perm, int BM,
                   int32_t *tile_idx, int32_t *tile_mask, hplStream stream);

/* Y[m, n] = act(bias[n] + res[...] + sum_{f<F, c<C} A[nbr[f][m], c] * Wt[f*C + c, n]).
 * One call covers: the blur Conv2d((15,1)) over gathered neighbours
 * (models/bilateralNN.py:199-221), every 1x1 Conv2d/Conv3d/Conv1d that follows
 * (bilateralNN.py:219, bnn_flow.py:202-205, HPLFlowNet.py:239-240,426-428), the patch
 * correlation Conv3d((1,15,1)) split into its f-independent pc1 half and its pc2 half
 * (bnn_flow.py:189-202) and the displacement filter Conv2d((15,1))
 * (bnn_flow.py:205).  The gathered tensor is never materialised. */
int hpl_gconv_forward(const hpl_gconv_desc *desc /* HOST */, hplStream stream);
/* same contract, one thread per output element, no MFMA: test/debug reference only */
int hpl_gconv_forward_naive(const hpl_gconv_desc *desc /* HOST */, hplStream stream);

/* dWt[f*C + c, n] (+)= sum_m A[nbr[f][m], c] * dY[m, n]    (weight gradient; split over m
 * with fp32 atomics, so dWt must be zero-initialised by the caller unless accumulating).
 * tap_m / tap_row / tap_ptr (optional, from hpl_tap_lists; tap_max = longest list, M if the
 * centre tap is always present): the sum of tap f then runs over its present vertices only.
 * dbias (optional, [N], zero-initialised by the caller): dbias[n] += sum_m dY[m, n], the bias
 * gradient, from the dY tiles the kernel loads anyway.
 * Wide layers (N >= 256, N % 4 == 0, C >= 128, C % 4 == 0, M >= 8192, 16-byte aligned operands, tap lists given or a dense
 * 1x1 layer without a table, rows_a * lda * 4 and M * lddy * 4 below 2^31) run with both fp32 operands split exactly into
 * three bf16 terms on the bf16 matrix pipe (csrc/wgrad3.hip: fp32-class accuracy, tests/test_gpu_wgrad3.py); HPL_WGRAD3=0 or
 * HPL_MATH=f32 in the environment keep the fp32-MFMA kernel. */
int hpl_gconv_wgrad(const float *A, int64_t lda, int64_t rows_a, const int32_t *nbr,
                    int64_t nbr_stride, int64_t reg_stride, int64_t M, int C, int F,
                    const float *dY, int64_t lddy, int N, float *dWt, int64_t ldw,
                    const int32_t *tap_m, const int32_t *tap_row, const int32_t *tap_ptr,
                    int64_t tap_max, float *dbias, hplStream stream);
/* The same with the largest magnitudes of A (the rows and channels the launch can read) and of dY (hpl_amax; DEVICE scalars):
 * the wide layers then run with both operands as scaled fp16 pairs on the fp16 MFMA (hpl_gconv_desc.wt3_planes == 2: the same
 * arithmetic, half the MFMA work of the bf16 triples).  Either NULL (or HPL_MATH=bf16x3): as hpl_gconv_wgrad. */
int hpl_gconv_wgrad_scaled(const float *A, int64_t lda, int64_t rows_a, const int32_t *nbr,
                           int64_t nbr_stride, int64_t reg_stride, int64_t M, int C, int F,
                           const float *dY, int64_t lddy, int N, float *dWt, int64_t ldw,
                           const int32_t *tap_m, const int32_t *tap_row, const int32_t *tap_ptr,
                           int64_t tap_max, float *dbias, const float *a_amax, const float *dy_amax, hplStream stream);

/* out[n] = sum_m X[m*ld + n]   (bias gradients) */
int hpl_colsum(const float *X, int64_t ld, int64_t M, int N, float *out, hplStream stream);
/* dX = dY * (Y > 0 ? 1 : slope)   element-wise on [M][N] views (LeakyReLU backward) */
int hpl_leaky_bwd(const float *dY, int64_t lddy, const float *Y, int64_t ldy, float slope,
                  float *dX, int64_t lddx, int64_t M, int N, hplStream stream);
/* The same; amax (optional, DEVICE): *amax = max(*amax, largest |dX|) -- the operand scale of the wide gradient launches that
 * read dX (hpl_gconv_desc.a_amax), taken while the values are in registers.  The caller clears *amax. */
int hpl_leaky_bwd_amax(const float *dY, int64_t lddy, const float *Y, int64_t ldy, float slope,
                       float *dX, int64_t lddx, int64_t M, int N, float *amax, hplStream stream);
/* out[h, n] (+)= sum_j X[(j*mod + h)*ldx + n], j < rows / mod: the gradient of a residual that was broadcast over the
 * rows / mod blocks of a result (the f-independent pc1 half of the patch correlation, models/bnn_flow.py:192). */
int hpl_psum(const float *X, int64_t ldx, int64_t rows, int64_t mod, int N, float *out, int64_t ldo, int accumulate,
             hplStream stream);
/* out[(f*M + m)*ldo + c] (+)= G[m*ldg + f*C + c]: the data gradient of the displacement filter (whose tap f of vertex m
 * reads row f*M + m, models/bnn_flow.py:205) from the plain GEMM G = g . W. */
int hpl_regroup(const float *G, int64_t ldg, int64_t M, int F, int C, float *out, int64_t ldo, int accumulate,
                hplStream stream);
/* EPE3DLoss (models/epe3d_loss.py:9-10, main.py:213) and its gradient in one launch: pred [N][3] point-major (the
 * flow hpl_plan_run writes), sf (3, N); *loss = mean_n ||pred_n - sf_n||_2 (one workgroup, fixed-order sum:
 * deterministic), grad[n][c] = (pred - sf) / (N * ||.||) (0 where the norm is 0). */
int hpl_epe3d(const float *pred, const float *sf, int64_t N, float *grad, float *loss, hplStream stream);
/* optimizer.step() of main.py:216 for the Adam of main.py:138-140 (lr 1e-4, weight_decay 0, no amsgrad) over FLAT fp32 arrays,
 * step >= 1 counting this one: m = lerp(m, g, 1 - beta1); v = beta2 v + (1 - beta2) g^2;
 * p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps) -- torch's fused Adam operation by operation, the
 * scalars computed in double and rounded once as there.  The training plan keeps parameters, gradients and both moments of a
 * model in four arrays of one layout, so a step is ONE launch over 19.3 M elements instead of four multi-tensor launches.
 * All pointers 16-byte aligned. */
int hpl_adam_flat(float *p, const float *g, float *m, float *v, int64_t n, double lr, double beta1, double beta2, double eps,
                  int64_t step, hplStream stream);

/* Per-tap lists of present vertices: list_m[tap_ptr[f] .. tap_ptr[f+1]) = { m : nbr[f][m] >= 0 }
 * ascending, list_row = their source rows nbr[f][m]; both hold up to F*M entries, tap_ptr F+1
 * (DEVICE).  scratch: 2*F*ceil(M/1024) + 1100 int32.  Consumer: hpl_gconv_wgrad (exact skipping
 * of absent neighbours, indices prefetched as plain streams). */
int hpl_tap_lists(const int32_t *nbr, int64_t nbr_stride, int F, int64_t M, int32_t *list_m,
                  int32_t *list_row, int32_t *tap_ptr, int32_t *scratch, hplStream stream);

/* *flag (int32, DEVICE, set to 1 by the caller) is cleared unless the table is symmetric:
 * nbr[0][m] == m and nbr[f][m] = g >= 0  =>  nbr[F-f][g] == m (f >= 1).  A symmetric blur / corr
 * table lets the backward w.r.t. the features run as a gather with mirrored taps instead of
 * fp32 atomics (true by construction unless the reference's unchecked key
 * packing aliased a neighbour key, A.2).  Several tables can share one stream and be read back once. */
int hpl_table_symmetric(const int32_t *nbr, int64_t nbr_stride, int F, int64_t M, int32_t *flag,
                        hplStream stream);

/* dst[m*ldd + c] = src[c*lds + m]  (channel-first <-> channel-last at the API edge) */
int hpl_transpose(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t rows_src,
                  int64_t cols_src, hplStream stream);

/* ------------------------------------------------------------------------ *
 * Lattice construction on the device (replaces the Numba + CFFI khash path:
 * transforms/transforms.py:133-261,300-353,358-485 and models/khash_int2int.h:8-33)
 * ------------------------------------------------------------------------ */
/* Float part, transforms/transforms.py:300-353.  pc (3, N) float32 (unscaled), the level
 * scale is applied inside (:377-378).  keys: int32 [4 coord][N][4 remainder]; bary (4, N);
 * emg (el_minus_gr): (4, N) like the reference when emg_ld == 0, else point-major
 * emg[n*emg_ld + j] (emg_ld >= 4: the channel-last layout the layers consume, so that both
 * clouds of a pair can be written into one matrix).  Bit-identical to oracle/lattice_oracle.c
 * (explicit fmaf chain, half-even rounding, stable descending rank). */
int hpl_lattice_keys(const float *pc, int64_t N, float scale, int32_t *keys, float *bary,
                     float *emg, int64_t emg_ld, hplStream stream);

/* The same for both clouds of a level in one launch.  Points: (3, n) arrays pc1 / pc2, or -- when
 * pc1 == pc2 == NULL -- the vertices of the previous level given by their integer keys vk1 / vk2
 * ([4][vstride], first n columns) and `divisor`, computed on the fly exactly as
 * hpl_lattice_next_points does (the (3, H) arrays of transforms.py:461-467 are never written). */
int hpl_lattice_keys_pair(const float *pc1, const float *pc2, const int32_t *vk1, const int32_t *vk2,
                          int64_t vstride1, int64_t vstride2, float divisor, int64_t n1, int64_t n2,
                          float scale, int32_t *keys1, int32_t *keys2, float *bary1, float *bary2,
                          float *emg1, float *emg2, int64_t emg_ld, hplStream stream);

/* Integer part, stage 1 (transforms/transforms.py:171-207 and :384-391): per-coordinate
 * key range over both clouds, mixed-radix packing (key2int, :70-86), one open-addressing
 * 64-bit table per cloud -- each workgroup first deduplicates its 1024 keys in an LDS
 * table, then only distinct keys CAS into the global table -- and vertex ids in
 * first-appearance order (points outer, remainder inner).  Outputs:
 *   off1/off2 int32 [4][n]   lattice_offset
 *   vkeys1/2  int32 [4][4n]  key of every vertex (first H columns valid)
 *   counts    int32 [2]      H1, H2  (DEVICE; the host reads them to size stage 2)
 * `workspace` (hpl_lattice_workspace_bytes(n1, n2) bytes) keeps the tables for stage 2. */
int64_t hpl_lattice_workspace_bytes(int64_t n1, int64_t n2);
int hpl_lattice_hash(const int32_t *keys1, int64_t n1, const int32_t *keys2, int64_t n2,
                     int32_t *off1, int32_t *off2, int32_t *vkeys1, int32_t *vkeys2,
                     int32_t *counts, void *workspace, int64_t workspace_bytes, hplStream stream);

/* Integer part, stage 2 (transforms/transforms.py:209-255): neighbour tables by hash lookup.
 * Radius -1 skips a table (pointer may be NULL).  H1, H2 are the counts read back from
 * stage 1.  blur1 [F][H1], blur2 [F][H2], corr1 [K][H1] (may be NULL with corr2 given: the caller
 * reuses blur1 when the radii are equal); corr2 is written directly in the
 * kernel-ready permuted layout [K][F*H1] (see hpl_corr2_permute).  Misses are -1; like the
 * reference, neighbour keys are packed without a range check.
 * blur_stride == 0: both blur tables are dense ([F][H1], [F][H2]).  Otherwise both have row
 * stride blur_stride and the ids in blur2 are shifted by blur2_shift: with blur2 = blur1 + H1,
 * blur_stride = H1 + H2, blur2_shift = H1 the two tables form ONE table [F][H1+H2] of the pair
 * (cloud 2's vertices numbered behind cloud 1's) -- the Down layers then run once per pair. */
int hpl_lattice_neighbors(const void *workspace, int64_t n1, int64_t n2,
                          const int32_t *vkeys1, const int32_t *vkeys2, int64_t H1, int64_t H2,
                          int bcn_radius, int corr_filter_radius, int corr_corr_radius,
                          int32_t *blur1, int32_t *blur2, int64_t blur_stride, int64_t blur2_shift,
                          int32_t *corr1, int32_t *corr2, hplStream stream);

/* Next level's points (transforms.py:461-467): out (3, H) = E^T (vkeys / divisor), divisor =
 * (float)(expected_std * scale) computed by the caller in double like the reference. */
int hpl_lattice_next_points(const int32_t *vkeys, int64_t vstride, int64_t H, float divisor,
                            float *out, hplStream stream);


/* ------------------------------------------------------------------------ *
 * Native forward executor: a whole model forward in ONE call
 *
 * The reference's forward (models/HPLFlowNet.py:238-430, models/HPLFlowNet_shallow.py:171-311) is ~130 kernel
 * launches per pair whose shapes follow the pair's vertex counts.  Issuing them from Python costs more host time
 * than the GPU needs on dense clouds; here the host side prepares a PLAN once per model -- the list of
 * operations with symbolic row counts, weight images and biases already resolved to device pointers -- and every
 * forward is one hpl_plan_run() that walks it with the lattice tables of the pair (hpl_level_tables).
 * The plan knows nothing about HPLFlowNet: hplflownet_amd/plan.py writes the program from the model's wiring.
 * ------------------------------------------------------------------------ */
/* Row-count symbols.  Level L (0-based) owns symbols HPL_SYM_LEVEL0 + 8*L + k. */
#define HPL_SYM_ZERO (-1)
#define HPL_SYM_N0 0          /* points of cloud 1 */
#define HPL_SYM_N1 1          /* points of cloud 2 */
#define HPL_SYM_NP 2          /* N0 + N1 */
#define HPL_SYM_LEVEL0 8
#define HPL_SYM_H0 0          /* vertices of cloud 1 at level L */
#define HPL_SYM_H1 1          /* vertices of cloud 2 */
#define HPL_SYM_HP 2          /* H0 + H1 */
#define HPL_SYM_FH0 3         /* 15 * H0: virtual vertices of the patch correlation */
#define HPL_SYM_IN0 4         /* input points of cloud 1 at level L (N0, or H0 of level L-1) */
#define HPL_SYM_INP 5         /* input points of both clouds */
#define HPL_SYM_FH1 6         /* 15 * H1: rows of the inverse pc2 correlation table per tap */
#define HPL_MAX_LEVELS 8

#define HPL_OP_GCONV 1        /* hpl_gconv_forward (incl. tap-group passes) */
#define HPL_OP_SPLAT 2        /* hpl_splat */
#define HPL_OP_SLICE 3        /* hpl_slice */
#define HPL_OP_COPY 4         /* out[:, cols] = a[:, cols]   (the torch.cat calls of the reference forward) */
#define HPL_OP_LOAD 5         /* out rows = transpose of an external (3, N) cloud (ext: 0 = pc1, 1 = pc2) */

/* Training (hpl_plan_run_range): the backward of the same program as more operations of the same plan.  Each is the
 * hand-written gradient of one forward op (what autograd derives for models/bilateralNN.py:151-238 and
 * models/bnn_flow.py:119-208), written by hplflownet_amd/plan.py in reverse order. */
#define HPL_OP_WGRAD 6        /* hpl_gconv_wgrad: a = the forward op's input, b = the gradient of its output; weight = index of
                                 the gradient image, bias = index of the bias-gradient vector or -1 */
#define HPL_OP_LEAKY_BWD 7    /* hpl_leaky_bwd: a = dY, b = Y, out = dX (may be a) */
#define HPL_OP_COLSUM 8       /* hpl_colsum of a [m_sym][C] into bias vector `bias` */
#define HPL_OP_SPLAT_BWD 9    /* gradient of HPL_OP_SPLAT: hpl_slice with the normaliser as vertex scale (pair CSR: one per cloud) */
#define HPL_OP_PSUM 10        /* hpl_psum: a [F*m_sym][N] -> out [m_sym][N] */
#define HPL_OP_REGROUP 11     /* hpl_regroup: a [m_sym][F*C] -> out [F*m_sym][C] */
#define HPL_OP_ZERO 12        /* out = 0 */
#define HPL_OP_EPE3D 13       /* hpl_epe3d on the run's flow / sf / loss; out = gradient [N0][3] */
#define HPL_OP_VCOPY 14       /* bias vector `bias` = bias vector `weight` (N entries): a gradient shared by two parameters */
#define HPL_OP_UNLAYOUT 15    /* hpl_weight_unlayout_batch of bucket `aux` (hpl_plan_set_unlayout) */
#define HPL_OP_GSUM 16        /* hpl_gather_sum: a = Z, out [m_sym][N], F = taps, table HPL_TBL_CORR2 (forward, col_step = N) or, with
                                 HPL_FLAG_INVERSE, through the inverse table held (as int32) by buffer b (col_step = 0, out = dense rows of N) */
#define HPL_OP_INVERT 17      /* hpl_table_invert of the level's corr2 table into buffer `out` (as int32 [15][15*H1]) */

#define HPL_FLAG_ACCUM 1      /* the op adds to what `out` holds (gconv: first pass with res = out) */
#define HPL_FLAG_SCATTER 2    /* gconv: scatter epilogue through the level's corr2 table, aux = channels per tap (scat_c) */
#define HPL_FLAG_TAPS 4       /* wgrad: sum over the per-tap vertex lists of the level's cloud-1 blur table (when the run has them) */
#define HPL_FLAG_INVERSE 16   /* HPL_OP_GSUM: see there */
#define HPL_FLAG_SIDE 8       /* the op is a leaf of the backward graph: it may run on the side stream of hpl_plan_run_range */
#define HPL_FLAG_NOGUARD 32   /* gconv: the fp16-pair form of this op runs WITHOUT its range guard (hpl_gconv_desc.a_guard): the data gradients
                                 of the training program -- a quiet row of a gradient matrix is a vertex whose share of every weight gradient
                                 lies below the rounding of the sums; the guard would run their launches twice (round 6: 1.4 ms of a step) */

#define HPL_TBL_NONE 0
#define HPL_TBL_BLUR_PAIR 1   /* blur table of the stacked pair [15][H0+H1]          (Down convs) */
#define HPL_TBL_BLUR0 2       /* its cloud-1 columns                                  (Up convs) */
#define HPL_TBL_CORR1 3       /* pc1_corr_indices [15][H0] */
#define HPL_TBL_CORR2 4       /* pc2_corr_indices, permuted [15][15*H0] */
#define HPL_TBL_REGULAR 5     /* no table: tap f of row m reads row f*sym[reg_stride_sym] + m */
#define HPL_TBL_CSR_PAIR 6    /* splat CSR of the stacked pair */
#define HPL_TBL_CSR_C0 7      /* splat CSR of cloud 1 alone (prefix of the pair CSR) */
#define HPL_TBL_CLOUD0 8      /* barycentric / offsets of cloud 1's input points (slice) */

#define HPL_ORD_NONE 0
#define HPL_ORD_PERM 1        /* rows in the table's single-pass tap order when the lattice has one */
#define HPL_ORD_GROUPS 2      /* tap-group passes when the lattice has group orders, else as HPL_ORD_PERM */

#define HPL_COND_ALWAYS 0
#define HPL_COND_SHRINK 1     /* the level's slice shrinks the row count: input points of cloud 1 < its vertices
                                 (the slice-before-1x1 order of an Up layer pays, bcl.BilateralConvFlex.forward_cl) */
#define HPL_COND_NOT_SHRINK 2

#define HPL_BUF_OUT (-2)      /* hpl_ref.buf: the caller's output matrix [N0][3] */

typedef struct hpl_ref {      /* rows [sym[row_off_sym], + sym[rows_sym]) x columns [col_off, col_off+cols) of buffer `buf` */
    int32_t buf;              /* index into the plan's buffers, HPL_BUF_OUT, or -1 = absent */
    int32_t row_off_sym;      /* HPL_SYM_ZERO: from the first row */
    int32_t rows_sym;         /* HPL_SYM_ZERO: to the last row */
    int32_t col_off;
    int32_t cols;
} hpl_ref;

typedef struct hpl_buf {      /* an activation matrix the executor allocates from the workspace per run */
    int32_t rows_sym;
    int32_t cols;
} hpl_buf;

typedef struct hpl_weight {   /* a re-laid weight image (hpl_weight_relayout) */
    const float *Wt;
    int64_t ldw;
    int64_t rows;
    const void *Wt3;          /* optional: hpl_weight_split3 / hpl_weight_split2h of the whole image (NULL: the layer stays on the fp32 MFMA) */
    int64_t wt3_plane_stride;
    const float *w_amax;      /* wt3_planes == 2: the scalar hpl_weight_split2h wrote */
    int32_t wt3_planes;
    int32_t pad_;
} hpl_weight;

typedef struct hpl_op {
    int32_t kind;             /* HPL_OP_* */
    int32_t tag;              /* profiling class (hpl_plan_profile) */
    hpl_ref a, out, res;
    int32_t m_sym;            /* rows produced: gconv M, splat H, slice N, copy / load rows */
    int32_t res_mod_sym;      /* gconv residual period (HPL_SYM_ZERO: the residual has M rows) */
    int32_t level;            /* lattice level whose tables are used */
    int32_t table;            /* HPL_TBL_* */
    int32_t order;            /* HPL_ORD_* */
    int32_t F, C, N;          /* taps, channels taken from `a`, output channels */
    int32_t weight;           /* index into the plan's weight images, -1 = none */
    int32_t bias;             /* index into the plan's bias pointers, -1 = none */
    int32_t act;              /* HPL_ACT_* */
    float slope;
    int32_t use_norm;         /* splat: density normalisation */
    int32_t reg_stride_sym;   /* HPL_TBL_REGULAR */
    int32_t ext;              /* HPL_OP_LOAD: 0 = pc1, 1 = pc2 */
    int32_t cond;             /* HPL_COND_*: run the op only if the condition holds at level cond_level */
    int32_t cond_level;
    hpl_ref out2;             /* gconv: optional second destination (buf == -1: none), see hpl_gconv_desc.Y2 */
    int32_t rows2_sym;        /* rows of the result that go to out2 as well */
    hpl_ref b;                /* second input of the backward ops (buf == -1: none) */
    int32_t flags;            /* HPL_FLAG_* */
    int32_t aux;
} hpl_op;

/* Kernel-ready tables of one lattice level of a pair (what hplflownet_amd.lattice builds on the device;
 * the schema of `generated_data`, transforms/transforms.py:471-483, in int32 / CSR form). */
typedef struct hpl_level_tables {
    int64_t n0, n1;                   /* input points per cloud */
    int64_t H0, H1;                   /* vertices per cloud */
    const float *emg_pair;            /* el_minus_gr, both clouds, point-major [n0+n1][4] */
    const int32_t *csr_ptr;           /* pair CSR (hpl_csr_build_pair): [H0+H1+1] */
    const int32_t *csr_pt;            /* [4*(n0+n1)] */
    const float *csr_w;
    const float *csr_norm;            /* [H0+H1] */
    const float *bary0;               /* cloud 1: [4][n0] */
    const int32_t *off0;
    const int32_t *blur;              /* pair blur table [15][blur_stride], NULL if the level has none */
    int64_t blur_stride;
    const int32_t *blur_perm;         /* hpl_tap_order of the pair table or NULL */
    const int32_t *up_perm;           /* hpl_tap_order of its cloud-1 columns or NULL */
    int32_t n_up_groups;              /* >= 2: tap groups [cut[i], cut[i+1]) with their own row orders */
    int32_t up_group_cut[5];
    const int32_t *up_group_perm[4];
    const int32_t *corr1;             /* [15][corr1_stride] or NULL */
    int64_t corr1_stride;
    const int32_t *corr1_perm;
    const int32_t *corr2;             /* [15][15*H0] or NULL */
    /* optional per-tile index tables (hpl_tile_index, tiles of tile_bm rows) of the row orders above; NULL = none */
    int32_t tile_bm;
    int32_t group_tile_bm;            /* tile height of the up_group tables (128: the split-operand kernel's tiles) */
    const int32_t *blur_perm_tidx, *blur_perm_tmask;
    const int32_t *up_perm_tidx, *up_perm_tmask;
    const int32_t *up_group_tidx[4], *up_group_tmask[4];
    const int32_t *corr1_perm_tidx, *corr1_perm_tmask;
    /* training only (zero otherwise): cloud 2's barycentric weights / offsets [4][n1] (the splat's gradient), and the per-tap
     * vertex lists of the cloud-1 blur table (hpl_tap_lists; up_tap_max = its longest list or H0) */
    const float *bary1;
    const int32_t *off1;
    const int32_t *up_tap_m, *up_tap_row, *up_tap_ptr;
    int64_t up_tap_max;
} hpl_level_tables;

typedef struct hpl_plan hpl_plan;   /* not thread-safe: one thread runs a given plan at a time (any number of streams) */

/* HOST arrays, copied.  The weight images and biases must stay alive (and may be refreshed in place). */
hpl_plan *hpl_plan_create(const hpl_op *ops, int n_ops, const hpl_buf *bufs, int n_bufs,
                          const hpl_weight *weights, int n_weights, const float *const *biases, int n_biases);
void hpl_plan_destroy(hpl_plan *plan);
/* bytes of workspace a run with these tables needs (activations + 64 MB of split-K partials) */
int64_t hpl_plan_workspace_bytes(const hpl_plan *plan, const hpl_level_tables *levels /* HOST */, int n_levels);
/* One forward: pc1 (3, n0), pc2 (3, n1) float32 -> out [n0][3] (the flow, point-major); everything is enqueued
 * on `stream`; `workspace` must not be reused before the run has finished on the device. */
int hpl_plan_run(hpl_plan *plan, const hpl_level_tables *levels /* HOST */, int n_levels, const float *pc1,
                 const float *pc2, float *out, void *workspace, int64_t workspace_bytes, hplStream stream);
/* Ops [op_begin, op_end) of the plan only (same carving of the workspace in every call: a step may be issued in
 * pieces, e.g. to start a gradient all-reduce between them).  sf (3, n0) and loss (DEVICE, one float) serve HPL_OP_EPE3D
 * (NULL: such an op is an error); side_stream (or NULL): ops flagged HPL_FLAG_SIDE are enqueued there behind an event on
 * `stream` (an un-layout flagged that way follows the weight gradients it reads on that stream); `stream` waits for them before
 * an HPL_OP_UNLAYOUT that runs on it and, with join != 0, at the end of the range (always join in the LAST range of a step). */
int hpl_plan_run_range(hpl_plan *plan, const hpl_level_tables *levels /* HOST */, int n_levels, const float *pc1,
                       const float *pc2, const float *sf, float *out, float *loss, void *workspace,
                       int64_t workspace_bytes, hplStream stream, hplStream side_stream, int op_begin, int op_end, int join);
/* The un-layout jobs of HPL_OP_UNLAYOUT: bucket i = jobs [bucket_first[i], bucket_first[i+1]) of the DEVICE tables
 * (hpl_weight_unlayout_batch); src = the gradient images (image of job j at src + prefix[j]). */
int hpl_plan_set_unlayout(hpl_plan *plan, const hpl_relayout_job *jobs /* DEVICE */, const int64_t *prefix /* DEVICE */,
                          const float *src, const int32_t *bucket_first /* HOST, n_buckets + 1 */,
                          const int64_t *bucket_offset /* HOST, n_buckets + 1: prefix[bucket_first[i]] */, int n_buckets);
/* Profiling: with tag >= 0 every following run brackets the ops of that tag with HIP events on the run's stream
 * (tag < 0: off).  hpl_plan_profile_read waits for the recorded events and returns the number of bracketed
 * launches, their total duration in ms, and resets the record. */
int hpl_plan_profile(hpl_plan *plan, int tag);
/* Launches of this plan's runs so far that took the second pass of the fp16-pair form's range guard (hpl_gconv_desc.a_guard):
 * an operand with a row 2^18 or more below the matrix's largest magnitude.  Synchronises with the device. */
int hpl_plan_guard_trips(hpl_plan *plan, int64_t *count);
/* clock_probe (DEVICE, 2 x int64, or NULL) is handed to the gather-GEMM launches of the profiled tag
 * (hpl_gconv_desc.clock_probe): they accumulate their samples there. */
int hpl_plan_clock_probe(hpl_plan *plan, int64_t *clock_probe);
int hpl_plan_profile_read(hpl_plan *plan, int *launches, float *total_ms);


/* ------------------------------------------------------------------------ *
 * Native lattice builder: GenerateDataUnsymmetric.__call__ (transforms/transforms.py:358-485) as a state machine
 * over the stage functions above.  One builder = one pair under construction: hpl_lattice_begin launches level 0
 * up to the read-back of its vertex counts (they size the next arrays); hpl_lattice_advance launches the rest of
 * that level (neighbour tables, splat CSR, row orders, per-tile index tables) and the next level up to ITS
 * read-back, and so on.  Every table is carved out of the caller's `arena` (DEVICE, 256-byte aligned); the
 * finished hpl_level_tables point into it.  The host never blocks unless it calls advance before ready.
 * ------------------------------------------------------------------------ */
#define HPL_ENOMEM (-4)   /* the arena is too small: retry with a bigger one */

typedef struct hpl_lattice_spec {
    int32_t n_levels;
    float scale[HPL_MAX_LEVELS];                 /* scales_filter_map[k][0] */
    int32_t bcn_radius[HPL_MAX_LEVELS];          /* [k][1] */
    int32_t corr_filter_radius[HPL_MAX_LEVELS];  /* [k][2], -1 = no correlation at this level */
    int32_t corr_corr_radius[HPL_MAX_LEVELS];    /* [k][3] */
    float next_divisor[HPL_MAX_LEVELS];          /* (float)(expected_std * scale), transforms.py:462-463 */
    int32_t wide_up[HPL_MAX_LEVELS];             /* 1: the level's Up conv runs as tap-group passes, 0: single pass, -1: both */
    int32_t n_groups;                            /* tap groups (>= 2) and their cuts, e.g. {0, 8, 15} */
    int32_t group_cut[5];
    float groups_min_sparsity;                   /* groups only where H0 / n0 >= this */
    int64_t perm_min_rows;                       /* single-pass row orders only for tables with at least this many rows */
    int64_t groups_min_rows;                     /* tap-group row orders (wide, sparse levels) from this many cloud-1 vertices on (0 = perm_min_rows) */
    int32_t group_tile_bm;                       /* tile height of the tap-group tile tables: 64 or 128 (0 = 64) */
    int32_t fused;                               /* != 0: hpl_lattice_begin enqueues the whole build, the vertex counts stay on the
                                                    device and are read back once (csrc/lattice_fused.hip); radius-1 specs only */
} hpl_lattice_spec;

typedef struct hpl_lattice hpl_lattice;   /* one pair under construction per builder; use several builders to overlap pairs */

hpl_lattice *hpl_lattice_create(const hpl_lattice_spec *spec /* HOST */);
void hpl_lattice_destroy(hpl_lattice *b);
/* Fused builds size every array by a bound: a level's vertices per cloud <= min(4 x its input points, bounds[L]), bounds[L]
 * = 0 meaning 16 x max(n0, n1).  hpl_lattice_arena_bytes: the arena a build of (n0, n1) points needs under the current bounds
 * (0 for a staged builder: its arena is a guess that HPL_ENOMEM corrects).  A pair that outgrows a bound is rebuilt by the
 * staged driver inside the same hpl_lattice_advance protocol (hpl_lattice_stats counts it), so bounds only cost memory and
 * idle workgroups -- e.g. twice the largest counts seen so far.  hpl_lattice_stats: out[0] = kernel launches of the last
 * fused enqueue, out[1] = 1 if the last finished build came from the fused driver, out[2] = builds of this handle that fell back to the staged one. */
int64_t hpl_lattice_arena_bytes(const hpl_lattice *b, int64_t n0, int64_t n1);
int hpl_lattice_set_bounds(hpl_lattice *b, const int64_t *bounds /* HOST, HPL_MAX_LEVELS entries, or NULL */);
int hpl_lattice_stats(const hpl_lattice *b, int32_t *out /* HOST, 3 */);
/* pc1 (3, n0), pc2 (3, n1) float32 DEVICE, must stay valid until the build is complete */
int hpl_lattice_begin(hpl_lattice *b, const float *pc1, const float *pc2, int64_t n0, int64_t n1, void *arena,
                      int64_t arena_bytes, hplStream stream);
/* 1 if hpl_lattice_advance would not block (the pending read-back has landed, or the build is done), else 0 */
int hpl_lattice_ready(hpl_lattice *b);
/* *done = 1 once every launch of the build has been enqueued (the tables are valid for work ordered behind them
 * on the same stream).  HPL_ENOMEM: the arena overflowed, the build is abandoned. */
int hpl_lattice_advance(hpl_lattice *b, int *done);
/* tables of the finished build (n_levels entries, owned by the builder, valid until its next begin) */
const hpl_level_tables *hpl_lattice_tables(const hpl_lattice *b);
/* extra per-level pointers the forward does not need (cloud 2's barycentric / offsets): out[2*L] = bary1,
 * out[2*L+1] = off1; and the bytes of the arena in use */
int hpl_lattice_extras(const hpl_lattice *b, const void **out /* HOST, 2 * n_levels */, int64_t *arena_used);

#ifdef __cplusplus
}
#endif
#endif /* HPL_BCL_H */
