// lattice_fused.h -- interface between csrc/lattice_builder.hip (the hpl_lattice_* state machine) and
// csrc/lattice_fused.hip (the whole 7-level build of a pair enqueued in one go, vertex counts kept on the device).
#pragma once
#include "common.h"

namespace hpl {
namespace fused {

constexpr int MAX_JOBS = 6;        // row orders of a level: pair table, single-pass cloud-1 order, <= 4 tap groups
constexpr int DIM_INTS = 32;       // ints per record of the dims block; record 0 is the header, record 1 + L level L
// level record
constexpr int D_H0 = 0, D_H1 = 1;  // vertices per cloud (written by the id stage)
constexpr int D_MM = 2;            // [8]: per-coordinate key minima [4] and maxima [4] over both clouds
// header record
constexpr int HDR_OVERFLOW = 0;    // a level produced more vertices than its bound: the build is void

struct SortJob {
    int32_t kind;        // 1: rows of the pair table (M = H0 + H1), 2: its cloud-1 columns (M = H0)
    int32_t role;        // 0 = pair order, 1 = single-pass cloud-1 order, 2 + g = tap group g
    int32_t f0, F;       // taps [f0, f0 + F) of the blur table form the presence mask
    int32_t bm;          // tile height of its tile tables
    int32_t two_pass;    // F > 8: the key has two 8-bit digits
    int32_t chunks_b;    // bound on its 2048-row chunks
    int32_t pad_;
    uint32_t *key;       // [Mb] Gray rank of the row's tap mask
    int32_t *hist1, *hist2;      // [chunks_b][256] digit counts per chunk (pass 1: of the rows, pass 2: of pass 1's output)
    uint32_t *tkey;      // pass-1 output (two-pass jobs)
    int32_t *tval;
    int32_t *perm;       // [Mb]
    int32_t *tidx;       // [tiles][F][bm]
    int32_t *tmask;      // [tiles][8]
};

// Everything a kernel needs to know about one level: bounds, spec, device pointers.  The array of levels lives at the
// head of the arena (copied there once per build); kernels take a pointer to it.
struct Level {
    int32_t index, n_levels;
    int32_t n_host[2];            // level 0: points per cloud (deeper levels: the previous level's vertex counts)
    int32_t nb[2], Hb[2];         // bounds the arrays are sized for: input points / vertices per cloud
    float scale, prev_div;
    int32_t prev_vstride[2], vstride[2];
    int32_t has_blur, has_corr, wide, n_groups;
    int32_t perm_min_rows, groups_min_rows;
    float min_sparsity;
    int32_t pad0_;
    const float *pc[2];           // level 0 only
    const int32_t *prev_vk[2];
    int32_t *hdr, *dims;
    const int32_t *prev_dims;     // nullptr at level 0
    float *emg;
    int32_t *keys[2];
    float *bary[2];
    int32_t *off[2], *vk[2];
    // open-addressing table of a cloud: 16-byte slots {packed key (two words), first entry that named it, vertex id} -- a probe is
    // ONE 16-byte load (round 6; rounds 4-5: three arrays, two dependent-free loads per probe and slot)
    int4 *tslot[2];
    int32_t *slot[2], *bsum[2];
    int32_t *blur, *corr2;
    int32_t *cnt, *cursor, *csum, *ent, *csr_ptr, *csr_pt;
    float *csr_w, *norm;
    int32_t n_jobs, pad_;
    SortJob job[MAX_JOBS];
};

struct Plan {
    Level lv[HPL_MAX_LEVELS];     // host copy
    int n_levels = 0;
    Level *d_levels = nullptr;    // in the arena
    int32_t *d_dims = nullptr;    // in the arena: (1 + n_levels) records
    int64_t bytes = 0;            // arena bytes in use
    int launches = 0;             // kernel launches of the last enqueue
};

// default per-cloud bound on a level's vertex count: min(4 x the bound of its input points, row_cap)
int64_t default_row_cap(int64_t n0, int64_t n1);

// Lay the build out in `arena` (nullptr: size query only).  bounds[L] > 0 overrides the vertex bound of level L (per
// cloud).  Returns the bytes needed, or -1 (plan.bytes is set either way).
int64_t layout(const hpl_lattice_spec &spec, int64_t n0, int64_t n1, const int64_t *bounds, const float *pc1,
               const float *pc2, char *arena, Plan &plan);

// true if this spec can be built by the fused path (radius-1 stencils, corr1 sharing the blur table, <= 4 groups)
bool supported(const hpl_lattice_spec &spec);

// Enqueue the whole build.  lv_stage: pinned host memory of sizeof(Level) * HPL_MAX_LEVELS the level array is copied
// from (must stay untouched until the copy has run).  dims_host (pinned, (1 + HPL_MAX_LEVELS) * DIM_INTS ints) receives the
// dims block as soon as the last level's vertex counts exist; counts_ev is recorded behind that copy.
int enqueue(Plan &plan, Level *lv_stage, int32_t *dims_host, hipEvent_t counts_ev, hipStream_t s);

}  // namespace fused
}  // namespace hpl
