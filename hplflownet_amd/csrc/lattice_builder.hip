// lattice_builder.hip -- the multi-scale lattice driver (transforms/transforms.py:358-485) as a native state machine
// (include/hpl_bcl.h "Native lattice builder").  It issues exactly the stage calls hplflownet_amd/lattice.py issues
// (hpl_lattice_keys_pair, hpl_lattice_hash, hpl_lattice_neighbors, hpl_csr_build_pair, hpl_tap_order,
// hpl_tile_index) with the same arguments -- the tables are bit-identical -- but without ~35 ctypes round trips and
// ~60 tensor allocations per pair: every array is carved out of one caller-owned arena.
//
// Two drivers behind the same hpl_lattice_* calls:
//   staged (spec.fused == 0, or the fallback of a fused build that overflowed its bounds): level by level, the vertex
//          counts of each level read back to size the next arrays -- 7 host round trips and ~260 launches per pair;
//   fused  (spec.fused != 0; csrc/lattice_fused.hip): hpl_lattice_begin enqueues the WHOLE build (33 launches for 7
//          levels), the counts stay on the device and come back once; hpl_lattice_advance only waits for that one
//          read-back and fills in the tables.
#include "common.h"
#include "lattice_fused.h"

#include <new>
#include <stdlib.h>

using namespace hpl;

namespace {
constexpr int TILE_BM = 64;
inline int filter_size(int r) { return (r + 1) * (r + 1) * (r + 1) * (r + 1) - r * r * r * r; }
}  // namespace

struct hpl_lattice {
    hpl_lattice_spec spec;
    hpl_level_tables tab[HPL_MAX_LEVELS];
    const void *bary1[HPL_MAX_LEVELS];
    const void *off1[HPL_MAX_LEVELS];
    // build state
    char *arena = nullptr, *cur = nullptr, *end = nullptr;
    hipStream_t s = nullptr;
    hplStream hs = nullptr;
    int level = 0;                 // level whose counts are pending
    bool active = false, done = false, overflow = false;
    int64_t n[2] = {0, 0};
    int64_t n_start[2] = {0, 0};
    int64_t n_vert[2] = {0, 0};    // vertices of the level being finished
    const float *pc[2] = {nullptr, nullptr};
    // per level scratch kept until its second half
    void *ws = nullptr;
    int32_t *vk[2] = {nullptr, nullptr}, *counts = nullptr, *off[2] = {nullptr, nullptr};
    float *bary[2] = {nullptr, nullptr};
    const int32_t *prev_vk[2] = {nullptr, nullptr};
    int64_t prev_stride[2] = {0, 0};
    float prev_div = 1.f;
    int32_t *scratch = nullptr;    // reused by the CSR build and every tap order of a level
    int32_t *counts_host = nullptr;   // pinned, 2 per level
    hipEvent_t ev[HPL_MAX_LEVELS];
    bool ev_ok = false;
    // fused driver
    bool fused_run = false;           // the build in progress was enqueued by the fused driver
    fused::Plan plan;
    fused::Level *lv_stage = nullptr; // pinned: source of the level-descriptor copy
    int32_t *dims_host = nullptr;     // pinned: the one read-back
    hipEvent_t counts_ev = nullptr;
    int64_t bounds[HPL_MAX_LEVELS] = {0};
    int32_t stat_launches = 0, stat_fallbacks = 0, stat_fused = 0;
    bool last_fused = false;          // the last finished build came from the fused driver

    template <class T> T *take(int64_t count) {
        const int64_t bytes = (count * (int64_t)sizeof(T) + 255) / 256 * 256;
        if (cur + bytes > end) { overflow = true; return nullptr; }
        T *p = reinterpret_cast<T *>(cur);
        cur += bytes;
        return p;
    }
};

namespace {

// first half of a level: keys + barycentric + hash, then the asynchronous read-back of the two vertex counts
int level_head(hpl_lattice *b) {
    const int L = b->level;
    const int64_t n0 = b->n[0], n1 = b->n[1];
    float *emg = b->take<float>((n0 + n1) * 4);
    int32_t *keys0 = b->take<int32_t>(16 * n0), *keys1 = b->take<int32_t>(16 * n1);
    b->bary[0] = b->take<float>(4 * n0);
    b->bary[1] = b->take<float>(4 * n1);
    const int64_t wsb = hpl_lattice_workspace_bytes(n0, n1);
    b->ws = b->take<char>(wsb);
    b->off[0] = b->take<int32_t>(4 * n0);
    b->off[1] = b->take<int32_t>(4 * n1);
    b->vk[0] = b->take<int32_t>(16 * n0);
    b->vk[1] = b->take<int32_t>(16 * n1);
    b->counts = b->take<int32_t>(2);
    if (b->overflow) return HPL_ENOMEM;
    int rc;
    if (L == 0)
        rc = hpl_lattice_keys_pair(b->pc[0], b->pc[1], nullptr, nullptr, 0, 0, 1.0f, n0, n1, b->spec.scale[0], keys0, keys1,
                                   b->bary[0], b->bary[1], emg, emg + 4 * n0, 4, b->hs);
    else
        rc = hpl_lattice_keys_pair(nullptr, nullptr, b->prev_vk[0], b->prev_vk[1], b->prev_stride[0], b->prev_stride[1],
                                   b->prev_div, n0, n1, b->spec.scale[L], keys0, keys1, b->bary[0], b->bary[1], emg,
                                   emg + 4 * n0, 4, b->hs);
    if (rc) return rc;
    rc = hpl_lattice_hash(keys0, n0, keys1, n1, b->off[0], b->off[1], b->vk[0], b->vk[1], b->counts, b->ws, wsb, b->hs);
    if (rc) return rc;
    if (hipMemcpyAsync(b->counts_host + 2 * L, b->counts, 8, hipMemcpyDeviceToHost, b->s) != hipSuccess ||
        hipEventRecord(b->ev[L], b->s) != hipSuccess) {
        set_error("hpl_lattice: read-back of the vertex counts failed");
        return HPL_EHIP;
    }
    hpl_level_tables &t = b->tab[L];
    t = hpl_level_tables{};
    t.n0 = n0; t.n1 = n1;
    t.emg_pair = emg;
    t.bary0 = b->bary[0]; t.off0 = b->off[0];
    b->bary1[L] = b->bary[1]; b->off1[L] = b->off[1];
    return HPL_OK;
}

// M <= H0: rows are cloud-1 vertices; M = H0 + H1: the stacked pair
int order_of(hpl_lattice *b, const int32_t *nbr, int64_t stride, int F, int64_t M, const int32_t **perm,
             const int32_t **tidx, const int32_t **tmask, int bm = TILE_BM) {
    int32_t *p = b->take<int32_t>(M);
    const int64_t tiles = cdiv(M, bm);
    int32_t *ti = b->take<int32_t>(tiles * F * bm), *tm = b->take<int32_t>(tiles * 8);
    if (b->overflow) return HPL_ENOMEM;
    int rc = hpl_tap_order(nbr, stride, F, M, p, b->scratch, b->hs);
    if (rc) return rc;
    rc = hpl_tile_index(nbr, stride, F, M, p, bm, ti, tm, b->hs);
    if (rc) return rc;
    *perm = p; *tidx = ti; *tmask = tm;
    return HPL_OK;
}

// second half: the counts of level L have landed
int level_tail(hpl_lattice *b) {
    const int L = b->level;
    const hpl_lattice_spec &sp = b->spec;
    hpl_level_tables &t = b->tab[L];
    const int64_t n0 = b->n[0], n1 = b->n[1];
    const int64_t H0 = b->counts_host[2 * L], H1 = b->counts_host[2 * L + 1];
    HPL_REQUIRE(H0 > 0 && H1 > 0 && H0 <= 4 * n0 && H1 <= 4 * n1, "hpl_lattice: implausible vertex counts %lld / %lld at level %d",
                (long long)H0, (long long)H1, L);
    t.H0 = H0; t.H1 = H1;
    b->n_vert[0] = H0; b->n_vert[1] = H1;
    const int bcn = sp.bcn_radius[L], cf = sp.corr_filter_radius[L], cc = sp.corr_corr_radius[L];
    const int64_t Hp = H0 + H1;
    int32_t *blur = nullptr, *corr1 = nullptr, *corr2 = nullptr;
    int F = 0;
    if (bcn != -1) {
        F = filter_size(bcn);
        blur = b->take<int32_t>((int64_t)F * Hp);
    }
    if (cf != -1) {
        if (cc != bcn) corr1 = b->take<int32_t>((int64_t)filter_size(cc) * H0);
        corr2 = b->take<int32_t>((int64_t)filter_size(cc) * filter_size(cf) * H0);
    }
    // splat CSR of the pair + the scratch shared by the CSR build and the tap orders
    int32_t *csr_ptr = b->take<int32_t>(Hp + 1), *csr_pt = b->take<int32_t>(4 * (n0 + n1));
    float *csr_w = b->take<float>(4 * (n0 + n1)), *norm = b->take<float>(Hp);
    const int64_t scratch_ints = imax(Hp + 1 + 4 * (n0 + n1) + 1026, hpl_tap_order_scratch_ints(Hp));
    b->scratch = b->take<int32_t>(scratch_ints);
    if (b->overflow) return HPL_ENOMEM;
    int rc = hpl_lattice_neighbors(b->ws, n0, n1, b->vk[0], b->vk[1], H0, H1, bcn, cf, cc, blur, blur ? blur + H0 : nullptr,
                                   Hp, H0, corr1, corr2, b->hs);
    if (rc) return rc;
    rc = hpl_csr_build_pair(b->off[0], b->bary[0], n0, H0, b->off[1], b->bary[1], n1, H1, csr_ptr, csr_pt, csr_w, norm,
                            b->scratch, b->hs);
    if (rc) return rc;
    t.csr_ptr = csr_ptr; t.csr_pt = csr_pt; t.csr_w = csr_w; t.csr_norm = norm;
    t.blur = blur; t.blur_stride = Hp;
    t.tile_bm = TILE_BM;
    t.group_tile_bm = sp.group_tile_bm == 128 ? 128 : TILE_BM;
    if (cf != -1) {
        t.corr1 = corr1 ? corr1 : blur;                 // equal radii: corr1 IS the cloud-1 blur table
        t.corr1_stride = corr1 ? H0 : Hp;
        t.corr2 = corr2;
    }
    // row orders (lattice.py / flownet.DeviceLattice.prepare_tables): pair table (Down convs), cloud-1 columns (Up conv:
    // single order and / or tap groups by the model's hint), corr1 when it is a table of its own
    if (blur && F > 1 && F <= 15) {
        if (Hp >= sp.perm_min_rows) {
            rc = order_of(b, blur, Hp, F, Hp, &t.blur_perm, &t.blur_perm_tidx, &t.blur_perm_tmask);
            if (rc) return rc;
        }
        {
            const int wide = sp.wide_up[L];
            const int64_t gmin = sp.groups_min_rows > 0 ? sp.groups_min_rows : sp.perm_min_rows;
            const bool sparse = (double)H0 / (double)n0 >= (double)sp.groups_min_sparsity;
            const bool grouped = wide != 0 && sp.n_groups >= 2 && sparse && H0 >= gmin;
            if (grouped) {
                t.n_up_groups = sp.n_groups;
                for (int g = 0; g < sp.n_groups; ++g) {
                    const int f0 = sp.group_cut[g], f1 = sp.group_cut[g + 1];
                    t.up_group_cut[g] = f0; t.up_group_cut[g + 1] = f1;
                    rc = order_of(b, blur + (int64_t)f0 * Hp, Hp, f1 - f0, H0, &t.up_group_perm[g], &t.up_group_tidx[g],
                                  &t.up_group_tmask[g], t.group_tile_bm);
                    if (rc) return rc;
                }
            }
            // the single-pass order: needed unless the Up conv surely runs as groups; corr1 (same table) uses it too
            if (H0 >= sp.perm_min_rows && (!(grouped && wide == 1) || (cf != -1 && !corr1))) {
                rc = order_of(b, blur, Hp, F, H0, &t.up_perm, &t.up_perm_tidx, &t.up_perm_tmask);
                if (rc) return rc;
            }
        }
    }
    if (cf != -1) {
        if (!corr1) { t.corr1_perm = t.up_perm; t.corr1_perm_tidx = t.up_perm_tidx; t.corr1_perm_tmask = t.up_perm_tmask; }
        else if (H0 >= sp.perm_min_rows && filter_size(cc) <= 15) {
            rc = order_of(b, corr1, H0, filter_size(cc), H0, &t.corr1_perm, &t.corr1_perm_tidx, &t.corr1_perm_tmask);
            if (rc) return rc;
        }
    }
    // next level: its points are this level's vertices
    b->prev_vk[0] = b->vk[0]; b->prev_vk[1] = b->vk[1];
    b->prev_stride[0] = 4 * n0; b->prev_stride[1] = 4 * n1;
    b->prev_div = sp.next_divisor[L];
    b->n[0] = H0; b->n[1] = H1;
    return HPL_OK;
}


// fused driver: the counts have landed -> the hpl_level_tables of every level (the same decisions level_tail makes)
int fused_finish(hpl_lattice *b) {
    const hpl_lattice_spec &sp = b->spec;
    const fused::Plan &P = b->plan;
    int64_t n0 = b->n[0], n1 = b->n[1];
    for (int L = 0; L < sp.n_levels; ++L) {
        const fused::Level &V = P.lv[L];
        const int32_t *d = b->dims_host + fused::DIM_INTS * (1 + L);
        const int64_t H0 = d[fused::D_H0], H1 = d[fused::D_H1];
        HPL_REQUIRE(H0 > 0 && H1 > 0 && H0 <= 4 * n0 && H1 <= 4 * n1, "hpl_lattice: implausible vertex counts %lld / %lld at level %d",
                    (long long)H0, (long long)H1, L);
        const int64_t Hp = H0 + H1;
        hpl_level_tables &t = b->tab[L];
        t = hpl_level_tables{};
        t.n0 = n0; t.n1 = n1; t.H0 = H0; t.H1 = H1;
        t.emg_pair = V.emg;
        t.bary0 = V.bary[0]; t.off0 = V.off[0];
        b->bary1[L] = V.bary[1]; b->off1[L] = V.off[1];
        t.csr_ptr = V.csr_ptr; t.csr_pt = V.csr_pt; t.csr_w = V.csr_w; t.csr_norm = V.norm;
        t.blur = V.blur; t.blur_stride = Hp;
        t.tile_bm = TILE_BM;
        t.group_tile_bm = sp.group_tile_bm == 128 ? 128 : TILE_BM;
        const bool has_corr = sp.corr_filter_radius[L] != -1;
        if (has_corr) { t.corr1 = V.blur; t.corr1_stride = Hp; t.corr2 = V.corr2; }
        const int wide = sp.wide_up[L];
        const bool sparse = (double)H0 / (double)n0 >= (double)sp.groups_min_sparsity;
        const int64_t gmin = sp.groups_min_rows > 0 ? sp.groups_min_rows : sp.perm_min_rows;
        const bool grouped = H0 >= gmin && wide != 0 && sp.n_groups >= 2 && sparse;
        const bool single = H0 >= sp.perm_min_rows && (!(grouped && wide == 1) || has_corr);
        for (int q = 0; q < V.n_jobs; ++q) {
            const fused::SortJob &J = V.job[q];
            if (J.role == 0 && Hp >= sp.perm_min_rows) {
                t.blur_perm = J.perm; t.blur_perm_tidx = J.tidx; t.blur_perm_tmask = J.tmask;
            } else if (J.role == 1 && single) {
                t.up_perm = J.perm; t.up_perm_tidx = J.tidx; t.up_perm_tmask = J.tmask;
            } else if (J.role >= 2 && grouped) {
                const int g = J.role - 2;
                t.up_group_perm[g] = J.perm; t.up_group_tidx[g] = J.tidx; t.up_group_tmask[g] = J.tmask;
                t.up_group_cut[g] = J.f0; t.up_group_cut[g + 1] = J.f0 + J.F;
            }
        }
        if (grouped) t.n_up_groups = sp.n_groups;
        if (has_corr) { t.corr1_perm = t.up_perm; t.corr1_perm_tidx = t.up_perm_tidx; t.corr1_perm_tmask = t.up_perm_tmask; }
        n0 = H0; n1 = H1;
    }
    return HPL_OK;
}

int fused_begin(hpl_lattice *b) {
    const int64_t need = fused::layout(b->spec, b->n[0], b->n[1], b->bounds, b->pc[0], b->pc[1], b->arena, b->plan);
    if (need < 0) { set_error("hpl_lattice (fused): clouds too large"); return HPL_EINVAL; }
    if (need > b->end - b->arena) return HPL_ENOMEM;
    b->cur = b->arena + need;
    const int rc = fused::enqueue(b->plan, b->lv_stage, b->dims_host, b->counts_ev, b->s);
    b->stat_launches = b->plan.launches;
    return rc;
}

}  // namespace

extern "C" hpl_lattice *hpl_lattice_create(const hpl_lattice_spec *spec) {
    if (!spec || spec->n_levels < 1 || spec->n_levels > HPL_MAX_LEVELS || spec->n_groups > 4 ||
        (spec->group_tile_bm != 0 && spec->group_tile_bm != 64 && spec->group_tile_bm != 128)) {
        set_error("hpl_lattice_create: bad spec");
        return nullptr;
    }
    hpl_lattice *b = new (std::nothrow) hpl_lattice();
    if (!b) return nullptr;
    b->spec = *spec;
    if (hipHostMalloc(reinterpret_cast<void **>(&b->counts_host), 2 * HPL_MAX_LEVELS * sizeof(int32_t), hipHostMallocDefault) !=
        hipSuccess) {
        set_error("hpl_lattice_create: no pinned memory for the read-backs");
        delete b;
        return nullptr;
    }
    bool ok = true;
    for (int i = 0; i < HPL_MAX_LEVELS; ++i) ok = ok && hipEventCreateWithFlags(&b->ev[i], hipEventDisableTiming) == hipSuccess;
    b->ev_ok = ok;
    if (!ok) { set_error("hpl_lattice_create: event creation failed"); hpl_lattice_destroy(b); return nullptr; }
    if (spec->fused) {
        if (!fused::supported(*spec)) { set_error("hpl_lattice_create: this spec needs the staged builder (fused = 0)"); hpl_lattice_destroy(b); return nullptr; }
        if (hipHostMalloc(reinterpret_cast<void **>(&b->lv_stage), sizeof(fused::Level) * HPL_MAX_LEVELS, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void **>(&b->dims_host), sizeof(int32_t) * fused::DIM_INTS * (1 + HPL_MAX_LEVELS), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&b->counts_ev, hipEventDisableTiming) != hipSuccess) {
            set_error("hpl_lattice_create: no pinned memory / event for the fused driver");
            hpl_lattice_destroy(b);
            return nullptr;
        }
    }
    return b;
}

extern "C" void hpl_lattice_destroy(hpl_lattice *b) {
    if (!b) return;
    if (b->counts_host) (void)hipHostFree(b->counts_host);
    if (b->lv_stage) (void)hipHostFree(b->lv_stage);
    if (b->dims_host) (void)hipHostFree(b->dims_host);
    if (b->counts_ev) (void)hipEventDestroy(b->counts_ev);
    if (b->ev_ok) for (int i = 0; i < HPL_MAX_LEVELS; ++i) (void)hipEventDestroy(b->ev[i]);
    delete b;
}

extern "C" int hpl_lattice_begin(hpl_lattice *b, const float *pc1, const float *pc2, int64_t n0, int64_t n1, void *arena,
                                 int64_t arena_bytes, hplStream stream) {
    HPL_REQUIRE(b && pc1 && pc2 && n0 > 0 && n1 > 0 && arena && arena_bytes > 0, "hpl_lattice_begin: bad arguments");
    HPL_REQUIRE((reinterpret_cast<uintptr_t>(arena) & 255u) == 0, "hpl_lattice_begin: the arena must be 256-byte aligned");
    b->arena = b->cur = reinterpret_cast<char *>(arena);
    b->end = b->arena + arena_bytes;
    b->hs = stream; b->s = to_stream(stream);
    b->level = 0; b->active = true; b->done = false; b->overflow = false;
    b->n[0] = n0; b->n[1] = n1; b->pc[0] = pc1; b->pc[1] = pc2;
    b->n_start[0] = n0; b->n_start[1] = n1;
    b->fused_run = b->spec.fused != 0;
    b->last_fused = false;
    const int rc = b->fused_run ? fused_begin(b) : level_head(b);
    if (rc) b->active = false;
    return rc;
}

extern "C" int64_t hpl_lattice_arena_bytes(const hpl_lattice *b, int64_t n0, int64_t n1) {
    if (!b || n0 <= 0 || n1 <= 0) return -1;
    if (!b->spec.fused) return 0;
    fused::Plan tmp;
    return fused::layout(b->spec, n0, n1, b->bounds, nullptr, nullptr, nullptr, tmp);
}

extern "C" int hpl_lattice_set_bounds(hpl_lattice *b, const int64_t *bounds) {
    HPL_REQUIRE(b && (!b->active || b->done), "hpl_lattice_set_bounds: no builder, or a build is in progress");
    for (int L = 0; L < HPL_MAX_LEVELS; ++L) b->bounds[L] = bounds ? bounds[L] : 0;
    return HPL_OK;
}

extern "C" int hpl_lattice_stats(const hpl_lattice *b, int32_t *out) {
    HPL_REQUIRE(b && out, "hpl_lattice_stats: null argument");
    out[0] = b->stat_launches; out[1] = b->last_fused ? 1 : 0; out[2] = b->stat_fallbacks;
    return HPL_OK;
}

extern "C" int hpl_lattice_ready(hpl_lattice *b) {
    if (!b || !b->active || b->done) return 1;
    if (b->fused_run) return hipEventQuery(b->counts_ev) == hipSuccess ? 1 : 0;
    return hipEventQuery(b->ev[b->level]) == hipSuccess ? 1 : 0;
}

extern "C" int hpl_lattice_advance(hpl_lattice *b, int *done) {
    HPL_REQUIRE(b && done, "hpl_lattice_advance: null argument");
    *done = b->done ? 1 : 0;
    if (b->done) return HPL_OK;
    HPL_REQUIRE(b->active, "hpl_lattice_advance: no build in progress");
    if (b->fused_run) {
        if (hipEventSynchronize(b->counts_ev) != hipSuccess) { set_error("hpl_lattice_advance: event wait failed"); return HPL_EHIP; }
        if (!b->dims_host[fused::HDR_OVERFLOW]) {
            const int rc = fused_finish(b);
            if (rc) { b->active = false; return rc; }
            ++b->stat_fused;
            b->last_fused = true;
            b->done = true;
            *done = 1;
            return HPL_OK;
        }
        // a level outgrew its bound: the pair is rebuilt level by level with exact sizes in the same arena (the fused
        // launches still in flight on this stream only touch memory the staged build rewrites behind them)
        ++b->stat_fallbacks;
        b->last_fused = false;
        b->fused_run = false;
        b->cur = b->arena;
        b->level = 0;
        b->n[0] = b->n_start[0]; b->n[1] = b->n_start[1];
        const int rc = level_head(b);
        if (rc) b->active = false;
        return rc;
    }
    if (hipEventSynchronize(b->ev[b->level]) != hipSuccess) { set_error("hpl_lattice_advance: event wait failed"); return HPL_EHIP; }
    int rc = level_tail(b);
    if (rc) { b->active = false; return rc; }
    if (++b->level == b->spec.n_levels) {
        b->done = true;
        *done = 1;
        return HPL_OK;
    }
    rc = level_head(b);
    if (rc) b->active = false;
    return rc;
}

extern "C" const hpl_level_tables *hpl_lattice_tables(const hpl_lattice *b) { return (b && b->done) ? b->tab : nullptr; }

extern "C" int hpl_lattice_extras(const hpl_lattice *b, const void **out, int64_t *arena_used) {
    HPL_REQUIRE(b && b->done && out && arena_used, "hpl_lattice_extras: no finished build");
    for (int L = 0; L < b->spec.n_levels; ++L) { out[2 * L] = b->bary1[L]; out[2 * L + 1] = b->off1[L]; }
    *arena_used = b->cur - b->arena;
    return HPL_OK;
}
