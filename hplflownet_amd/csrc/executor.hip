// executor.hip -- native forward executor: one C call per forward (include/hpl_bcl.h, "Native forward
// executor").  The plan is a flat program over symbolic row counts written once per model by
// hplflownet_amd/plan.py from the wiring of /root/reference/models/HPLFlowNet.py:238-430 (and
// HPLFlowNet_shallow.py:171-311); a run resolves the symbols from the pair's vertex counts, carves the
// activation matrices out of the caller's workspace and enqueues every kernel through the same entry points
// the Python path uses (hpl_gconv_forward, hpl_splat, hpl_slice, hpl_transpose) -- same kernels, same
// arguments, same results, without ~130 Python round trips.  Host cost per forward: the launches themselves.
#include "common.h"
#include "gconv_common.h"
#include <stdlib.h>

#include <algorithm>
#include <new>
#include <vector>

using namespace hpl;

namespace {

constexpr int64_t SPLITK_ELEMS = 8 << 20;          // ops.py: split-K only for outputs of <= 8 M elements
constexpr int64_t SPLITK_WS_BYTES = 64ll << 20;    // 16 M floats of partial tiles (a launch fits its split count to it)
// behind it: the largest magnitudes of the matrices the wide (fp16-pair) launches of a run read, one scalar per reduction
// (round 6: a slot is TWO words -- the largest magnitude and the range-guard word of the same view, hpl_gconv_desc.a_guard)
constexpr int AMAX_SLOTS = 4096;
constexpr int64_t HEAD_BYTES = SPLITK_WS_BYTES + AMAX_SLOTS * 8;
constexpr int MAX_SYMS = HPL_SYM_LEVEL0 + 8 * HPL_MAX_LEVELS;

__global__ void k_copy_cols(const float *__restrict__ src, int64_t lds, float *__restrict__ dst, int64_t ldd,
                            int64_t rows, int cols) {
    const int64_t total = rows * cols;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        dst[r * ldd + c] = src[r * lds + c];
    }
}

__global__ void k_copy_cols4(const float4 *__restrict__ src, int64_t lds4, float4 *__restrict__ dst, int64_t ldd4,
                             int64_t rows, int cols4) {
    const int64_t total = rows * cols4;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int64_t r = i / cols4;
        const int c = (int)(i - r * cols4);
        dst[r * ldd4 + c] = src[r * lds4 + c];
    }
}

// el_minus_gr tables -> the first four columns of several concatenation buffers, one launch for the whole forward
// (the tables are lattice outputs: every one of these copies can run before the first layer)
constexpr int EMG_JOBS = 24;
struct EmgJobs {
    const float4 *src[EMG_JOBS];
    float *dst[EMG_JOBS];
    int64_t ldd[EMG_JOBS];
    int64_t rows[EMG_JOBS];
    int n;
};
__global__ void k_copy_emg_batch(const EmgJobs j) {
    const int job = blockIdx.y;
    const float4 *src = j.src[job];
    float *dst = j.dst[job];
    const int64_t ldd = j.ldd[job], rows = j.rows[job];
    const bool vec = (ldd % 4 == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = src[r];
        float *d = dst + r * ldd;
        if (vec) *reinterpret_cast<float4 *>(d) = v;
        else { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
    }
}

// HPL_RANGE_GUARD=0: the fp16-pair launches of a plan run without their range guard (hpl_gconv_desc.a_guard: no guard words, no
// second launches) -- the round-5 behaviour, for A/B runs
// (diagnostic values: 2 = guard words only -- reductions and epilogues leave them, no launch reads them --, 3 = second launches only,
// reading words nobody wrote = 0 = unknown = no trip, 4 = reductions' words + second launches, not the epilogues')
inline int range_guard_mode() {
    static const int m = [] { const char *e = getenv("HPL_RANGE_GUARD"); return e ? atoi(e) : 1; }();
    return m;
}
inline bool range_guard() { return range_guard_mode() != 0; }

struct View {                 // a resolved hpl_ref
    float *p;
    int64_t ld;
    int64_t rows;
    int cols;
};

}  // namespace

struct hpl_plan {
    std::vector<hpl_op> ops;
    std::vector<hpl_buf> bufs;
    std::vector<hpl_weight> weights;
    std::vector<const float *> biases;
    int profile_tag = -1;
    bool hoist_emg = true;                  // no other op writes the columns the el_minus_gr copies fill
    int64_t *clock_probe = nullptr;
    std::vector<hipEvent_t> pool;           // events owned by the plan (reused)
    size_t pool_used = 0;
    std::vector<float *> base;              // per run: buffer base pointers
    std::vector<int64_t> rows;              // per run: buffer row counts
    // training: un-layout buckets of the weight-gradient images (hpl_plan_set_unlayout), fence events of the side stream
    const hpl_relayout_job *ul_jobs = nullptr;
    const int64_t *ul_prefix = nullptr;
    const float *ul_src = nullptr;
    std::vector<int32_t> ul_first;
    std::vector<int64_t> ul_offset;
    std::vector<hipEvent_t> fence;
    size_t fence_used = 0;
    // largest magnitudes reduced so far in this run (hpl_amax of a view, valid until an op writes the matrix): the wide launches
    // that read the same view share one reduction; a training step keeps them from its forward range to its backward ranges
    struct AmaxEnt { int buf; int64_t row_off, rows; int col_off, cols; const float *slot; bool guarded; };      // guarded: the slot's second word is the view's guard word
    std::vector<AmaxEnt> amax;
    int amax_used = 0;
    int32_t *guard_trips = nullptr;         // device counter: launches that took the range guard's second pass (allocated at the first run)
    // workspace layout of the last run (a function of the plan and the row-count symbols): byte offset of every buffer
    std::vector<int64_t> lay_sym, lay_off;
    int64_t lay_total = 0;
};

namespace {

int resolve_syms(const hpl_level_tables *lv, int n_levels, int64_t *sym) {
    HPL_REQUIRE(lv && n_levels >= 1 && n_levels <= HPL_MAX_LEVELS, "hpl_plan: 1 .. %d lattice levels expected, got %d",
                HPL_MAX_LEVELS, n_levels);
    for (int i = 0; i < MAX_SYMS; ++i) sym[i] = 0;
    sym[HPL_SYM_N0] = lv[0].n0;
    sym[HPL_SYM_N1] = lv[0].n1;
    sym[HPL_SYM_NP] = lv[0].n0 + lv[0].n1;
    for (int L = 0; L < n_levels; ++L) {
        int64_t *s = sym + HPL_SYM_LEVEL0 + 8 * L;
        HPL_REQUIRE(lv[L].n0 > 0 && lv[L].n1 > 0 && lv[L].H0 > 0 && lv[L].H1 > 0, "hpl_plan: empty lattice level %d", L);
        HPL_REQUIRE(L == 0 || (lv[L].n0 == lv[L - 1].H0 && lv[L].n1 == lv[L - 1].H1),
                    "hpl_plan: level %d has %lld / %lld input points, level %d has %lld / %lld vertices", L,
                    (long long)lv[L].n0, (long long)lv[L].n1, L - 1, (long long)lv[L - 1].H0, (long long)lv[L - 1].H1);
        s[HPL_SYM_H0] = lv[L].H0;
        s[HPL_SYM_H1] = lv[L].H1;
        s[HPL_SYM_HP] = lv[L].H0 + lv[L].H1;
        s[HPL_SYM_FH0] = 15 * lv[L].H0;
        s[HPL_SYM_IN0] = lv[L].n0;
        s[HPL_SYM_INP] = lv[L].n0 + lv[L].n1;
        s[HPL_SYM_FH1] = 15 * lv[L].H1;
    }
    return HPL_OK;
}

inline int64_t symv(const int64_t *sym, int id) { return id < 0 ? 0 : sym[id]; }

inline int64_t buf_bytes(int64_t rows, int cols) { return (rows * cols * 4 + 255) / 256 * 256; }

struct Runner {
    hpl_plan &pl;
    const hpl_level_tables *lv;
    int n_levels;
    const int64_t *sym;
    const float *pc[2];
    float *out;
    float *splitk;
    hipStream_t s;
    hplStream hs;
    const float *sf = nullptr;              // training: target flow (3, n0) and the loss scalar of HPL_OP_EPE3D
    float *loss = nullptr;
    hipStream_t main_s = nullptr, side_s = nullptr;
    bool side_busy = false;                 // side-stream work the main stream has not waited for yet
    float *amax_base = nullptr;             // AMAX_SLOTS scalars in the workspace (cleared when a run starts at op 0)
    const float *cur_a_amax = nullptr, *cur_b_amax = nullptr;      // of the op being issued (prepare_amax)
    float *cur_y_amax = nullptr;            // where the op being issued leaves the largest magnitude of what it writes (or null)
    bool cur_y_guarded = false;             // ... and the guard word of what it writes beside it

    // the largest magnitude of columns [0, cols) of a view, reduced on the MAIN stream unless this run already has it
    // (guard: the consumer is a guarded launch -- the reduction also leaves the view's guard word, hpl_amax_rows; a reduction kept from
    // an unguarded consumer is not reused for a guarded one)
    int amax_of(const hpl_ref &r, const View &v, int64_t rows, int cols, const float *&slot, bool guard) {
        const int64_t off = r.buf >= 0 ? symv(sym, r.row_off_sym) : 0;
        guard = guard && range_guard() && range_guard_mode() != 3;
        for (const auto &e : pl.amax)
            if (e.buf == r.buf && e.row_off == off && e.rows == rows && e.col_off == r.col_off && e.cols == cols && (e.guarded || !guard)) { slot = e.slot; return HPL_OK; }
        HPL_REQUIRE(pl.amax_used < AMAX_SLOTS, "hpl_plan_run: more than %d operand reductions in one run", AMAX_SLOTS);
        float *dst = amax_base + 2 * pl.amax_used++;
        const int rc = hpl_gc::amax_launch(v.p, v.ld, rows, cols, dst, main_s, guard ? reinterpret_cast<unsigned *>(dst + 1) : nullptr);
        if (rc) return rc;
        if (r.buf >= 0) pl.amax.push_back({r.buf, off, rows, r.col_off, cols, dst, guard});
        slot = dst;
        return HPL_OK;
    }
    // an op writes columns [col_off, col_off + cols) of a matrix: reductions over any of them are stale
    void amax_forget(const hpl_ref &r) {
        if (r.buf < 0) return;
        for (size_t i = 0; i < pl.amax.size();) {
            const auto &e = pl.amax[i];
            if (e.buf == r.buf && e.col_off < r.col_off + r.cols && r.col_off < e.col_off + e.cols) { pl.amax[i] = pl.amax.back(); pl.amax.pop_back(); }
            else ++i;
        }
    }
    // the op just issued left the largest magnitude of rows [off, off + rows) x its N columns of `out` in cur_y_amax
    void amax_produced(const hpl_op &op) {
        if (!cur_y_amax || op.out.buf < 0) return;
        pl.amax.push_back({op.out.buf, symv(sym, op.out.row_off_sym), symv(sym, op.m_sym), op.out.col_off, op.N, cur_y_amax, cur_y_guarded});
        cur_y_amax = nullptr;
    }
    // the range guard covers the forward: the data gradients of the training program carry HPL_FLAG_NOGUARD (include/hpl_bcl.h)
    static bool guarded(const hpl_op &op) { return range_guard() && !(op.flags & HPL_FLAG_NOGUARD); }
    float *amax_slot() { return pl.amax_used < AMAX_SLOTS ? amax_base + 2 * pl.amax_used++ : nullptr; }
    // wide launches in the fp16-pair mode scale their operands by their largest magnitudes: reduce them (main stream, before a
    // side-stream op is fenced) for the ops that can qualify (gconv_common.h split3_maybe / wgrad3.hip's test)
    int prepare_amax(const hpl_op &op, bool side) {
        cur_a_amax = cur_b_amax = nullptr;
        cur_y_amax = nullptr;
        cur_y_guarded = false;
        if (hpl_gc::split_planes() != 2) return HPL_OK;
        View A, B;
        int rc;
        if (op.kind == HPL_OP_LEAKY_BWD) {          // its result usually feeds a wide data / weight gradient: reduce it on the way
            if (!side && op.out.buf >= 0 && op.N >= 128 && symv(sym, op.m_sym) >= 1024) cur_y_amax = amax_slot();
            return HPL_OK;
        }
        if (op.kind == HPL_OP_GCONV) {
            if (op.weight < 0 || op.weight >= (int)pl.weights.size() || pl.weights[op.weight].wt3_planes != 2 || !pl.weights[op.weight].Wt3) return HPL_OK;
            const int64_t M = symv(sym, op.m_sym);
            if ((op.flags & HPL_FLAG_SCATTER) || !hpl_gc::split3_maybe(M, op.C, op.F > 15 ? 15 : op.F, op.N)) return HPL_OK;
            if ((rc = view(op.a, A, "gconv input"))) return rc;
            if ((rc = amax_of(op.a, A, A.rows, op.C, cur_a_amax, guarded(op)))) return rc;
            // a wide launch's result usually feeds the next wide launch (the 1x1 convs behind a blur conv): its epilogue reduces it
            if (!side && op.out.buf >= 0 && op.N >= 128) cur_y_amax = amax_slot();
            return HPL_OK;
        }
        if (op.kind == HPL_OP_WGRAD) {
            const int64_t M = symv(sym, op.m_sym);
            const bool taps = (op.flags & HPL_FLAG_TAPS) && op.table == HPL_TBL_BLUR0;
            if (!(op.N >= 256 && op.C >= 128 && M >= 8192 && (taps || (op.F == 1 && op.table == HPL_TBL_NONE)))) return HPL_OK;
            if ((rc = view(op.a, A, "wgrad input")) || (rc = view(op.b, B, "wgrad output gradient"))) return rc;
            if ((rc = amax_of(op.a, A, A.rows, op.C, cur_a_amax, false))) return rc;
            return amax_of(op.b, B, M, op.N, cur_b_amax, false);
        }
        return HPL_OK;
    }

    hipEvent_t fence_event() {
        if (pl.fence_used == pl.fence.size()) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            pl.fence.push_back(e);
        }
        return pl.fence[pl.fence_used++];
    }
    // the side stream starts behind everything enqueued on the main stream so far
    int to_side() {
        hipEvent_t e = fence_event();
        HPL_REQUIRE(e, "hpl_plan_run_range: hipEventCreate failed");
        if (hipEventRecord(e, main_s) != hipSuccess || hipStreamWaitEvent(side_s, e, 0) != hipSuccess) { set_error("hpl_plan_run_range: fence"); return HPL_EHIP; }
        s = side_s; hs = reinterpret_cast<hplStream>(side_s);
        side_busy = true;
        return HPL_OK;
    }
    void to_main() { s = main_s; hs = reinterpret_cast<hplStream>(main_s); }
    // the main stream waits for the side stream's work
    int join() {
        if (!side_busy) return HPL_OK;
        hipEvent_t e = fence_event();
        HPL_REQUIRE(e, "hpl_plan_run_range: hipEventCreate failed");
        if (hipEventRecord(e, side_s) != hipSuccess || hipStreamWaitEvent(main_s, e, 0) != hipSuccess) { set_error("hpl_plan_run_range: join"); return HPL_EHIP; }
        side_busy = false;
        return HPL_OK;
    }

    int view(const hpl_ref &r, View &v, const char *what) const {
        if (r.buf == HPL_BUF_OUT) {
            v.p = out + r.col_off;
            v.ld = 3;
            v.rows = sym[HPL_SYM_N0];
            v.cols = r.cols;
            HPL_REQUIRE(r.col_off + r.cols <= 3, "hpl_plan_run: %s exceeds the [N][3] output", what);
            return HPL_OK;
        }
        HPL_REQUIRE(r.buf >= 0 && r.buf < (int)pl.bufs.size(), "hpl_plan_run: %s refers to buffer %d", what, r.buf);
        const int64_t off = symv(sym, r.row_off_sym);
        const int64_t have = pl.rows[r.buf] - off;
        v.rows = r.rows_sym < 0 ? have : sym[r.rows_sym];
        HPL_REQUIRE(off >= 0 && v.rows >= 0 && v.rows <= have && r.col_off >= 0 &&
                        r.col_off + r.cols <= pl.bufs[r.buf].cols,
                    "hpl_plan_run: %s outside buffer %d (%lld x %d): rows %lld + %lld, columns %d + %d", what, r.buf,
                    (long long)pl.rows[r.buf], pl.bufs[r.buf].cols, (long long)off, (long long)v.rows, r.col_off, r.cols);
        HPL_REQUIRE(pl.base[r.buf], "hpl_plan_run: %s refers to buffer %d, which no active op of the program uses", what, r.buf);
        v.ld = pl.bufs[r.buf].cols;
        v.p = pl.base[r.buf] + off * v.ld + r.col_off;
        v.cols = r.cols;
        return HPL_OK;
    }

    void bracket(bool on, bool stop) {
        if (!on) return;
        if (pl.pool_used >= (size_t)1 << 16) return;          // a profile nobody reads: stop recording (pairs stay aligned: even cap)
        if (pl.pool_used == pl.pool.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            pl.pool.push_back(e);
        }
        (void)stop;
        (void)hipEventRecord(pl.pool[pl.pool_used++], s);
    }

    int gconv(const hpl_op &op) {
        View A, Y, R;
        int rc = view(op.a, A, "gconv input");
        if (rc) return rc;
        rc = view(op.out, Y, "gconv output");
        if (rc) return rc;
        const bool has_res = op.res.buf != -1;
        if (has_res && (rc = view(op.res, R, "gconv residual"))) return rc;
        View Y2 = {};
        const bool has_out2 = op.out2.buf != -1;
        int64_t rows2 = 0;
        if (has_out2) {
            if ((rc = view(op.out2, Y2, "gconv second output"))) return rc;
            rows2 = symv(sym, op.rows2_sym);
            HPL_REQUIRE(Y2.cols >= op.N && rows2 >= 0 && rows2 <= Y2.rows, "hpl_plan_run: second output (%lld rows of %lld, N=%d of %d)",
                        (long long)rows2, (long long)Y2.rows, op.N, Y2.cols);
        }
        HPL_REQUIRE(op.weight >= 0 && op.weight < (int)pl.weights.size(), "hpl_plan_run: weight image %d", op.weight);
        HPL_REQUIRE(op.level >= 0 && op.level < n_levels, "hpl_plan_run: op uses lattice level %d of %d", op.level, n_levels);
        const hpl_level_tables &t = lv[op.level];
        const hpl_weight &w = pl.weights[op.weight];
        const int64_t M = symv(sym, op.m_sym);
        const int32_t *nbr = nullptr, *perm = nullptr, *tidx = nullptr, *tmask = nullptr;
        int64_t stride = 0, reg = 0;
        int ngroups = 0;
        switch (op.table) {
        case HPL_TBL_NONE: break;
        case HPL_TBL_BLUR_PAIR:
            nbr = t.blur; stride = t.blur_stride;
            if (op.order != HPL_ORD_NONE) { perm = t.blur_perm; tidx = t.blur_perm_tidx; tmask = t.blur_perm_tmask; }
            break;
        case HPL_TBL_BLUR0:
            nbr = t.blur; stride = t.blur_stride;
            if (op.order == HPL_ORD_GROUPS && t.n_up_groups >= 2) ngroups = t.n_up_groups;
            else if (op.order != HPL_ORD_NONE) { perm = t.up_perm; tidx = t.up_perm_tidx; tmask = t.up_perm_tmask; }
            break;
        case HPL_TBL_CORR1:
            nbr = t.corr1; stride = t.corr1_stride;
            if (op.order != HPL_ORD_NONE) { perm = t.corr1_perm; tidx = t.corr1_perm_tidx; tmask = t.corr1_perm_tmask; }
            break;
        case HPL_TBL_CORR2: nbr = t.corr2; stride = 15 * t.H0; break;
        case HPL_TBL_REGULAR: reg = symv(sym, op.reg_stride_sym); break;
        default: HPL_REQUIRE(false, "hpl_plan_run: gconv with table kind %d", op.table);
        }
        HPL_REQUIRE(op.table == HPL_TBL_NONE || op.table == HPL_TBL_REGULAR || nbr, "hpl_plan_run: level %d lacks table %d",
                    op.level, op.table);
        HPL_REQUIRE(op.table != HPL_TBL_REGULAR || (reg > 0 && (op.F - 1) * reg + M <= A.rows),
                    "hpl_plan_run: regular stride %lld x %d taps outside the %lld input rows", (long long)reg, op.F,
                    (long long)A.rows);
        const bool accum = (op.flags & HPL_FLAG_ACCUM) != 0, scatter = (op.flags & HPL_FLAG_SCATTER) != 0;
        HPL_REQUIRE(!(accum && has_res) && !(scatter && (has_out2 || ngroups >= 2 || !t.corr2 || op.aux <= 0 || op.N % op.aux)),
                    "hpl_plan_run: accumulate / scatter flags on an op they do not fit");
        HPL_REQUIRE(A.cols >= op.C && (scatter || (Y.cols >= op.N && Y.rows >= M)), "hpl_plan_run: gconv shapes (C=%d of %d, N=%d of %d)",
                    op.C, A.cols, op.N, Y.cols);
        HPL_REQUIRE(!scatter || (Y.cols >= op.aux && Y.rows >= t.H1), "hpl_plan_run: scatter target too small");
        const bool prof = pl.profile_tag >= 0 && op.tag == pl.profile_tag;
        auto pass = [&](int f0, int F, const int32_t *row_perm, bool first, bool last, const int32_t *ti,
                        const int32_t *tm) -> int {
            hpl_gconv_desc d = {};
            d.A = A.p + (reg ? (int64_t)f0 * reg * A.ld : 0);
            d.lda = A.ld;
            d.rows_a = A.rows - (reg ? (int64_t)f0 * reg : 0);
            d.nbr = nbr ? nbr + (int64_t)f0 * stride : nullptr;
            d.nbr_stride = stride;
            d.reg_stride = reg;
            d.M = M; d.C = op.C; d.F = F;
            const int64_t wrows = w.rows - (int64_t)f0 * op.C;
            d.Wt = w.Wt + (int64_t)f0 * op.C * w.ldw;
            d.ldw = w.ldw;
            d.N = op.N;
            d.w_rows = (int32_t)imin(wrows, cdiv((int64_t)F * op.C, 32) * 32);
            d.act = last ? op.act : HPL_ACT_NONE;
            d.slope = op.slope;
            d.bias = (first && op.bias >= 0) ? pl.biases[op.bias] : nullptr;
            if (first) {
                if (has_res) { d.res = R.p; d.ldres = R.ld; d.res_mod = op.res_mod_sym >= 0 ? sym[op.res_mod_sym] : R.rows; }
                else if (accum && !scatter) { d.res = Y.p; d.ldres = Y.ld; d.res_mod = M; }
            } else {
                d.res = Y.p; d.ldres = Y.ld; d.res_mod = M;
            }
            d.Y = Y.p; d.ldy = Y.ld;
            if (last && has_out2) { d.Y2 = Y2.p; d.ldy2 = Y2.ld; d.rows2 = rows2; }
            if (last && cur_y_amax && !scatter) {
                d.y_amax = cur_y_amax;
                if (guarded(op) && range_guard_mode() < 3) { d.y_guard = reinterpret_cast<uint32_t *>(cur_y_amax + 1); cur_y_guarded = true; }
            }
            d.row_perm = row_perm;
            if (row_perm && ti && tm) { d.tile_idx = ti; d.tile_mask = tm; d.tile_bm = ngroups >= 2 ? t.group_tile_bm : t.tile_bm; }
            // split-operand image of the same rows (csrc/gconv3.hip takes the launch if it qualifies): k-blocks of 8 rows
            if (w.Wt3 && ((int64_t)f0 * op.C) % 8 == 0 && (w.wt3_planes != 2 || cur_a_amax)) {
                d.Wt3 = static_cast<const char *>(w.Wt3) + (int64_t)f0 * op.C / 8 * w.ldw * 16;
                d.wt3_plane_stride = w.wt3_plane_stride;
                d.wt3_planes = w.wt3_planes;
                d.a_amax = cur_a_amax; d.w_amax = w.w_amax;
                if (cur_a_amax && guarded(op) && range_guard_mode() != 2) { d.a_guard = reinterpret_cast<const uint32_t *>(cur_a_amax + 1); d.guard_trips = pl.guard_trips; }
            }
            if (prof) d.clock_probe = pl.clock_probe;
            if (scatter) { d.scat = t.corr2; d.scat_stride = 15 * t.H0; d.scat_c = op.aux; }
            else
            if (M * op.N <= SPLITK_ELEMS) { d.ws = splitk; d.ws_bytes = SPLITK_WS_BYTES; }
            bracket(prof, false);
            const int r = hpl_gconv_forward(&d, hs);
            bracket(prof, true);
            return r;
        };
        if (ngroups >= 2) {
            for (int g = 0; g < ngroups; ++g) {
                rc = pass(t.up_group_cut[g], t.up_group_cut[g + 1] - t.up_group_cut[g], t.up_group_perm[g], g == 0,
                          g == ngroups - 1, t.up_group_tidx[g], t.up_group_tmask[g]);
                if (rc) return rc;
            }
            return HPL_OK;
        }
        if ((nbr || reg) && op.F > 15) {            // radius-2 stencils: accumulating passes over tap ranges
            for (int f0 = 0; f0 < op.F; f0 += 15) {
                rc = pass(f0, (int)imin(15, op.F - f0), nullptr, f0 == 0, f0 + 15 >= op.F, nullptr, nullptr);
                if (rc) return rc;
            }
            return HPL_OK;
        }
        return pass(0, op.F, perm, true, true, tidx, tmask);
    }

    int run_op(const hpl_op &op) {
        const bool prof = pl.profile_tag >= 0 && op.tag == pl.profile_tag && op.kind != HPL_OP_GCONV;
        int rc = HPL_OK;
        View A, Y;
        switch (op.kind) {
        case HPL_OP_GCONV: return gconv(op);
        case HPL_OP_SPLAT: {
            if ((rc = view(op.a, A, "splat input")) || (rc = view(op.out, Y, "splat output"))) return rc;
            HPL_REQUIRE(op.level >= 0 && op.level < n_levels, "hpl_plan_run: splat at level %d", op.level);
            const hpl_level_tables &t = lv[op.level];
            const int64_t H = symv(sym, op.m_sym);
            HPL_REQUIRE(op.table == HPL_TBL_CSR_PAIR || op.table == HPL_TBL_CSR_C0, "hpl_plan_run: splat table kind %d", op.table);
            HPL_REQUIRE(H == (op.table == HPL_TBL_CSR_PAIR ? t.H0 + t.H1 : t.H0) && Y.rows >= H && Y.cols >= op.C &&
                            A.cols >= op.C && A.rows >= (op.table == HPL_TBL_CSR_PAIR ? t.n0 + t.n1 : t.n0),
                        "hpl_plan_run: splat shapes at level %d", op.level);
            bracket(prof, false);
            rc = ((op.flags & HPL_FLAG_ACCUM) ? hpl_splat_add : hpl_splat)(A.p, A.ld, op.C, t.csr_ptr, t.csr_pt, t.csr_w,
                                                                         op.use_norm ? t.csr_norm : nullptr, H, Y.p, Y.ld, hs);
            bracket(prof, true);
            return rc;
        }
        case HPL_OP_SLICE: {
            if ((rc = view(op.a, A, "slice input")) || (rc = view(op.out, Y, "slice output"))) return rc;
            HPL_REQUIRE(op.level >= 0 && op.level < n_levels && op.table == HPL_TBL_CLOUD0, "hpl_plan_run: slice at level %d", op.level);
            const hpl_level_tables &t = lv[op.level];
            const int64_t N = symv(sym, op.m_sym);
            HPL_REQUIRE(N == t.n0 && A.rows >= t.H0 && A.cols >= op.C && Y.cols >= op.C && Y.rows >= N,
                        "hpl_plan_run: slice shapes at level %d", op.level);
            bracket(prof, false);
            rc = hpl_slice(A.p, A.ld, op.C, t.bary0, t.off0, N, nullptr, op.bias >= 0 ? pl.biases[op.bias] : nullptr, Y.p,
                           Y.ld, hs);
            bracket(prof, true);
            return rc;
        }
        case HPL_OP_COPY: {
            if ((rc = view(op.out, Y, "copy output"))) return rc;
            const int64_t rows = symv(sym, op.m_sym);
            const float *src;
            int64_t lds;
            if (op.a.buf == -1) {                 // el_minus_gr of the level: external table [rows][4]
                HPL_REQUIRE(op.level >= 0 && op.level < n_levels && op.C == 4, "hpl_plan_run: emg copy at level %d", op.level);
                src = lv[op.level].emg_pair;
                lds = 4;
                HPL_REQUIRE(rows <= lv[op.level].n0 + lv[op.level].n1, "hpl_plan_run: emg copy of %lld rows", (long long)rows);
            } else {
                if ((rc = view(op.a, A, "copy input"))) return rc;
                HPL_REQUIRE(A.rows >= rows && A.cols >= op.C, "hpl_plan_run: copy input too small");
                src = A.p;
                lds = A.ld;
            }
            HPL_REQUIRE(Y.rows >= rows && Y.cols >= op.C, "hpl_plan_run: copy output too small");
            if (rows == 0) return HPL_OK;
            if (op.flags & HPL_FLAG_ACCUM) return add_cols(src, lds, Y.p, Y.ld, rows, op.C, s);
            if (op.C % 4 == 0 && lds % 4 == 0 && Y.ld % 4 == 0 && aligned16(src) && aligned16(Y.p)) {
                const int grid = (int)imin(cdiv(rows * (op.C / 4), 256), 4096);
                k_copy_cols4<<<grid, 256, 0, s>>>(reinterpret_cast<const float4 *>(src), lds / 4,
                                                  reinterpret_cast<float4 *>(Y.p), Y.ld / 4, rows, op.C / 4);
            } else {
                const int grid = (int)imin(cdiv(rows * op.C, 256), 4096);
                k_copy_cols<<<grid, 256, 0, s>>>(src, lds, Y.p, Y.ld, rows, op.C);
            }
            HPL_CHECK_LAUNCH("hpl_plan_run (copy)");
            return HPL_OK;
        }
        case HPL_OP_LOAD: {
            if ((rc = view(op.out, Y, "load output"))) return rc;
            const int64_t n = symv(sym, op.m_sym);
            HPL_REQUIRE((op.ext == 0 || op.ext == 1) && Y.rows >= n && Y.cols >= 3, "hpl_plan_run: load");
            return hpl_transpose(pc[op.ext], n, Y.p, Y.ld, 3, n, hs);
        }
        case HPL_OP_WGRAD: {
            View B;
            if ((rc = view(op.a, A, "wgrad input")) || (rc = view(op.b, B, "wgrad output gradient"))) return rc;
            HPL_REQUIRE(op.level >= 0 && op.level < n_levels, "hpl_plan_run: wgrad at level %d", op.level);
            HPL_REQUIRE(op.weight >= 0 && op.weight < (int)pl.weights.size() && op.bias < (int)pl.biases.size(), "hpl_plan_run: wgrad image %d", op.weight);
            const hpl_level_tables &t = lv[op.level];
            const int64_t M = symv(sym, op.m_sym);
            const int32_t *nbr = nullptr;
            int64_t stride = 0, reg = 0;
            switch (op.table) {
            case HPL_TBL_NONE: break;
            case HPL_TBL_BLUR_PAIR: case HPL_TBL_BLUR0: nbr = t.blur; stride = t.blur_stride; break;
            case HPL_TBL_CORR1: nbr = t.corr1; stride = t.corr1_stride; break;
            case HPL_TBL_CORR2: nbr = t.corr2; stride = 15 * t.H0; break;
            case HPL_TBL_REGULAR: reg = symv(sym, op.reg_stride_sym); break;
            default: HPL_REQUIRE(false, "hpl_plan_run: wgrad with table kind %d", op.table);
            }
            HPL_REQUIRE(op.table == HPL_TBL_NONE || op.table == HPL_TBL_REGULAR || nbr, "hpl_plan_run: level %d lacks table %d", op.level, op.table);
            HPL_REQUIRE(A.cols >= op.C && B.cols >= op.N && B.rows >= M, "hpl_plan_run: wgrad shapes");
            const hpl_weight &g = pl.weights[op.weight];
            const bool taps = (op.flags & HPL_FLAG_TAPS) && op.table == HPL_TBL_BLUR0 && t.up_tap_m && t.up_tap_row && t.up_tap_ptr;
            return hpl_gconv_wgrad_scaled(A.p, A.ld, A.rows, nbr, stride, reg, M, op.C, op.F, B.p, B.ld, op.N, const_cast<float *>(g.Wt), g.ldw,
                                          taps ? t.up_tap_m : nullptr, taps ? t.up_tap_row : nullptr, taps ? t.up_tap_ptr : nullptr,
                                          taps ? t.up_tap_max : 0, op.bias >= 0 ? const_cast<float *>(pl.biases[op.bias]) : nullptr,
                                          cur_a_amax, cur_b_amax, hs);
        }
        case HPL_OP_LEAKY_BWD: {
            View B;
            if ((rc = view(op.a, A, "leaky_bwd dY")) || (rc = view(op.b, B, "leaky_bwd Y")) || (rc = view(op.out, Y, "leaky_bwd dX"))) return rc;
            const int64_t M = symv(sym, op.m_sym);
            HPL_REQUIRE(A.rows >= M && B.rows >= M && Y.rows >= M && A.cols >= op.N && B.cols >= op.N && Y.cols >= op.N, "hpl_plan_run: leaky_bwd shapes");
            return hpl_leaky_bwd_amax(A.p, A.ld, B.p, B.ld, op.slope, Y.p, Y.ld, M, op.N, cur_y_amax, hs);
        }
        case HPL_OP_COLSUM: {
            if ((rc = view(op.a, A, "colsum input"))) return rc;
            const int64_t M = symv(sym, op.m_sym);
            HPL_REQUIRE(op.bias >= 0 && op.bias < (int)pl.biases.size() && A.rows >= M && A.cols >= op.C, "hpl_plan_run: colsum");
            return hpl_colsum(A.p, A.ld, M, op.C, const_cast<float *>(pl.biases[op.bias]), hs);
        }
        case HPL_OP_SPLAT_BWD: {
            if ((rc = view(op.a, A, "splat_bwd input")) || (rc = view(op.out, Y, "splat_bwd output"))) return rc;
            HPL_REQUIRE(op.level >= 0 && op.level < n_levels && (op.table == HPL_TBL_CSR_PAIR || op.table == HPL_TBL_CSR_C0),
                        "hpl_plan_run: splat_bwd at level %d table %d", op.level, op.table);
            const hpl_level_tables &t = lv[op.level];
            const bool pair = op.table == HPL_TBL_CSR_PAIR;
            HPL_REQUIRE(A.cols >= op.C && Y.cols >= op.C && A.rows >= (pair ? t.H0 + t.H1 : t.H0) && Y.rows >= (pair ? t.n0 + t.n1 : t.n0),
                        "hpl_plan_run: splat_bwd shapes at level %d", op.level);
            HPL_REQUIRE(!pair || (t.bary1 && t.off1), "hpl_plan_run: the lattice of a training step must carry cloud 2's barycentric tables");
            auto fn = (op.flags & HPL_FLAG_ACCUM) ? hpl_slice_add : hpl_slice;
            rc = fn(A.p, A.ld, op.C, t.bary0, t.off0, t.n0, op.use_norm ? t.csr_norm : nullptr, nullptr, Y.p, Y.ld, hs);
            if (rc || !pair) return rc;
            return fn(A.p + t.H0 * A.ld, A.ld, op.C, t.bary1, t.off1, t.n1, op.use_norm ? t.csr_norm + t.H0 : nullptr, nullptr,
                      Y.p + t.n0 * Y.ld, Y.ld, hs);
        }
        case HPL_OP_PSUM: {
            if ((rc = view(op.a, A, "psum input")) || (rc = view(op.out, Y, "psum output"))) return rc;
            const int64_t mod = symv(sym, op.m_sym);
            HPL_REQUIRE(mod > 0 && op.F > 0 && A.rows >= mod * op.F && Y.rows >= mod && A.cols >= op.N && Y.cols >= op.N, "hpl_plan_run: psum shapes");
            return hpl_psum(A.p, A.ld, mod * op.F, mod, op.N, Y.p, Y.ld, (op.flags & HPL_FLAG_ACCUM) ? 1 : 0, hs);
        }
        case HPL_OP_REGROUP: {
            if ((rc = view(op.a, A, "regroup input")) || (rc = view(op.out, Y, "regroup output"))) return rc;
            const int64_t M = symv(sym, op.m_sym);
            HPL_REQUIRE(A.rows >= M && A.cols >= op.F * op.C && Y.rows >= M * op.F && Y.cols >= op.C, "hpl_plan_run: regroup shapes");
            return hpl_regroup(A.p, A.ld, M, op.F, op.C, Y.p, Y.ld, (op.flags & HPL_FLAG_ACCUM) ? 1 : 0, hs);
        }
        case HPL_OP_ZERO: {
            if ((rc = view(op.out, Y, "zero output"))) return rc;
            return zero_cols(Y.p, Y.ld, op.m_sym >= 0 ? symv(sym, op.m_sym) : Y.rows, Y.cols, s);
        }
        case HPL_OP_EPE3D: {
            if ((rc = view(op.out, Y, "epe3d gradient"))) return rc;
            HPL_REQUIRE(sf && loss, "hpl_plan_run_range: the program holds a loss op but no target flow / loss pointer was given");
            HPL_REQUIRE(Y.ld == 3 && Y.rows >= sym[HPL_SYM_N0], "hpl_plan_run: the loss gradient is a dense [N0][3] matrix");
            return hpl_epe3d(out, sf, sym[HPL_SYM_N0], Y.p, loss, hs);
        }
        case HPL_OP_VCOPY:
            HPL_REQUIRE(op.weight >= 0 && op.weight < (int)pl.biases.size() && op.bias >= 0 && op.bias < (int)pl.biases.size() && op.N > 0,
                        "hpl_plan_run: vcopy");
            return vcopy(pl.biases[op.weight], const_cast<float *>(pl.biases[op.bias]), op.N, s);
        case HPL_OP_UNLAYOUT: {
            HPL_REQUIRE(pl.ul_jobs && op.aux >= 0 && op.aux + 1 < (int)pl.ul_first.size(), "hpl_plan_run: un-layout bucket %d (hpl_plan_set_unlayout)", op.aux);
            const int first = pl.ul_first[op.aux], n = pl.ul_first[op.aux + 1] - first;
            if (n <= 0) return HPL_OK;
            return hpl_weight_unlayout_batch(pl.ul_jobs + first, n, pl.ul_prefix + first, pl.ul_offset[op.aux + 1] - pl.ul_offset[op.aux],
                                             pl.ul_src + pl.ul_offset[op.aux], hs);
        }
        case HPL_OP_GSUM: {
            View R;
            if ((rc = view(op.a, A, "gather-sum input")) || (rc = view(op.out, Y, "gather-sum output"))) return rc;
            HPL_REQUIRE(op.level >= 0 && op.level < n_levels && op.F >= 1 && op.N > 0, "hpl_plan_run: gather-sum at level %d", op.level);
            const hpl_level_tables &t = lv[op.level];
            const int64_t M = symv(sym, op.m_sym);
            const bool has_res = op.res.buf != -1;
            if (has_res && (rc = view(op.res, R, "gather-sum residual"))) return rc;
            const float *bias = op.bias >= 0 ? pl.biases[op.bias] : nullptr;
            if (op.flags & HPL_FLAG_INVERSE) {
                View B;
                if ((rc = view(op.b, B, "inverse table"))) return rc;
                HPL_REQUIRE(B.ld == B.cols && B.rows * B.cols >= (int64_t)op.F * M && Y.ld == Y.cols && Y.rows * Y.cols >= M * op.N &&
                                A.cols >= op.N && !has_res, "hpl_plan_run: gather-sum through an inverse table: shapes");
                return hpl_gather_sum(A.p, A.ld, reinterpret_cast<const int32_t *>(B.p), M, M, op.F, op.N, 0, bias, nullptr, 0, 0, op.act,
                                      op.slope, Y.p, op.N, hs);
            }
            HPL_REQUIRE(op.table == HPL_TBL_CORR2 && t.corr2 && M == 15 * t.H0 && A.cols >= op.F * op.N && A.rows >= t.H1 && Y.rows >= M &&
                            Y.cols >= op.N, "hpl_plan_run: gather-sum shapes at level %d", op.level);
            return hpl_gather_sum(A.p, A.ld, t.corr2, 15 * t.H0, M, op.F, op.N, op.N, bias, has_res ? R.p : nullptr, has_res ? R.ld : 0,
                                  has_res ? (op.res_mod_sym >= 0 ? sym[op.res_mod_sym] : R.rows) : 0, op.act, op.slope, Y.p, Y.ld, hs);
        }
        case HPL_OP_INVERT: {
            if ((rc = view(op.out, Y, "inverse table"))) return rc;
            HPL_REQUIRE(op.level >= 0 && op.level < n_levels && lv[op.level].corr2, "hpl_plan_run: invert at level %d", op.level);
            const hpl_level_tables &t = lv[op.level];
            HPL_REQUIRE(Y.ld == Y.cols && Y.rows * Y.cols >= 225 * t.H1, "hpl_plan_run: inverse table buffer too small");
            return hpl_table_invert(t.corr2, 15 * t.H0, 15, t.H0, 15, t.H1, reinterpret_cast<int32_t *>(Y.p), hs);
        }
        default: HPL_REQUIRE(false, "hpl_plan_run: unknown op kind %d", op.kind);
        }
        return HPL_OK;
    }
};

}  // namespace

extern "C" hpl_plan *hpl_plan_create(const hpl_op *ops, int n_ops, const hpl_buf *bufs, int n_bufs,
                                     const hpl_weight *weights, int n_weights, const float *const *biases,
                                     int n_biases) {
    if (!ops || n_ops <= 0 || !bufs || n_bufs <= 0 || n_weights < 0 || n_biases < 0) {
        set_error("hpl_plan_create: bad arguments");
        return nullptr;
    }
    hpl_plan *p = new (std::nothrow) hpl_plan();
    if (!p) return nullptr;
    p->ops.assign(ops, ops + n_ops);
    p->bufs.assign(bufs, bufs + n_bufs);
    if (n_weights) p->weights.assign(weights, weights + n_weights);
    if (n_biases) p->biases.assign(biases, biases + n_biases);
    // the el_minus_gr copies are hoisted in front of the layers (one launch): only if their columns are theirs alone
    auto overlaps = [](const hpl_ref &a, const hpl_ref &b) {
        return a.buf >= 0 && a.buf == b.buf && a.col_off < b.col_off + b.cols && b.col_off < a.col_off + a.cols;
    };
    for (int i = 0; i < n_ops && p->hoist_emg; ++i) {
        if (!(ops[i].kind == HPL_OP_COPY && ops[i].a.buf == -1)) continue;
        for (int j = 0; j < n_ops; ++j)
            if (j != i && (overlaps(ops[i].out, ops[j].out) || (ops[j].kind == HPL_OP_GCONV && overlaps(ops[i].out, ops[j].out2)))) {
                // (the two orders of an Up layer never both run: same columns under opposite conditions are fine)
                const bool exclusive = ops[i].cond != HPL_COND_ALWAYS && ops[j].cond != HPL_COND_ALWAYS &&
                                       ops[i].cond != ops[j].cond && ops[i].cond_level == ops[j].cond_level;
                if (!exclusive) p->hoist_emg = false;
            }
    }
    for (const hpl_buf &b : p->bufs)
        if (b.cols <= 0 || b.rows_sym < 0 || b.rows_sym >= MAX_SYMS) {
            set_error("hpl_plan_create: bad buffer (rows symbol %d, %d columns)", b.rows_sym, b.cols);
            delete p;
            return nullptr;
        }
    return p;
}

extern "C" void hpl_plan_destroy(hpl_plan *plan) {
    if (!plan) return;
    for (hipEvent_t e : plan->pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : plan->fence) (void)hipEventDestroy(e);
    if (plan->guard_trips) (void)hipFree(plan->guard_trips);
    delete plan;
}

namespace {
// Workspace layout.  A matrix lives from the first op that touches it to the last one that does (ops whose per-pair condition is
// false do not count; what a side-stream op reads stays until the end of the program; the hoisted el_minus_gr copies run first).
// Matrices whose lives do not overlap share memory: lowest-offset first fit in order of first use.  Inference needs the live set
// of its widest layer (~0.35 GB at N = 8 192 instead of the 1.05 GB of all matrices), a training step every forward matrix plus the
// short-lived gradient matrices.  The split-K scratch sits in front (fixed offset).  Cached per set of row counts.
int64_t plan_layout(hpl_plan &pl, const hpl_level_tables *lv, int n_levels, const int64_t *sym) {
    if (pl.lay_sym.size() == (size_t)MAX_SYMS && std::equal(pl.lay_sym.begin(), pl.lay_sym.end(), sym)) return pl.lay_total;
    const int nb = (int)pl.bufs.size(), nops = (int)pl.ops.size();
    std::vector<int> first(nb, nops + 1), last(nb, -1);
    auto touch = [&](const hpl_ref &r, int at, bool to_end) {
        if (r.buf < 0 || r.buf >= nb) return;
        first[r.buf] = std::min(first[r.buf], at);
        last[r.buf] = std::max(last[r.buf], to_end ? nops : at);
    };
    for (int i = 0; i < nops; ++i) {
        const hpl_op &op = pl.ops[i];
        if (op.cond != HPL_COND_ALWAYS) {
            if (op.cond_level < 0 || op.cond_level >= n_levels) continue;
            const bool shrink = lv[op.cond_level].n0 < lv[op.cond_level].H0;
            if ((op.cond == HPL_COND_SHRINK) != shrink) continue;
        }
        const bool emg = op.kind == HPL_OP_COPY && op.a.buf == -1 && pl.hoist_emg;
        const bool side = (op.flags & HPL_FLAG_SIDE) != 0;
        const int at = emg ? 0 : i;
        touch(op.a, at, side); touch(op.b, at, side); touch(op.res, at, side);
        touch(op.out, at, false); touch(op.out2, at, false);
    }
    std::vector<int> order;
    for (int b = 0; b < nb; ++b) if (last[b] >= 0) order.push_back(b);
    std::vector<int64_t> size(nb, 0);
    for (int b : order) size[b] = buf_bytes(sym[pl.bufs[b].rows_sym], pl.bufs[b].cols);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return first[x] != first[y] ? first[x] < first[y] : size[x] > size[y]; });
    pl.lay_off.assign(nb, -1);
    int64_t total = HEAD_BYTES;
    std::vector<std::pair<int64_t, int64_t>> busy;          // (offset, end) of placed matrices alive together with the candidate
    std::vector<int> placed;
    for (int b : order) {
        busy.clear();
        for (int q : placed)
            if (first[q] <= last[b] && first[b] <= last[q] && size[q] > 0) busy.emplace_back(pl.lay_off[q], pl.lay_off[q] + size[q]);
        std::sort(busy.begin(), busy.end());
        int64_t at = HEAD_BYTES;
        for (const auto &iv : busy) {
            if (iv.first - at >= size[b]) break;
            at = std::max(at, iv.second);
        }
        pl.lay_off[b] = at;
        total = std::max(total, at + size[b]);
        placed.push_back(b);
    }
    pl.lay_sym.assign(sym, sym + MAX_SYMS);
    pl.lay_total = total + 256;
    return pl.lay_total;
}
}  // namespace

extern "C" int64_t hpl_plan_workspace_bytes(const hpl_plan *plan, const hpl_level_tables *levels, int n_levels) {
    int64_t sym[MAX_SYMS];
    if (!plan || resolve_syms(levels, n_levels, sym) != HPL_OK) return -1;
    return plan_layout(*const_cast<hpl_plan *>(plan), levels, n_levels, sym);
}

extern "C" int hpl_plan_run(hpl_plan *plan, const hpl_level_tables *levels, int n_levels, const float *pc1,
                            const float *pc2, float *out, void *workspace, int64_t workspace_bytes, hplStream stream) {
    HPL_REQUIRE(plan, "hpl_plan_run: null argument");
    return hpl_plan_run_range(plan, levels, n_levels, pc1, pc2, nullptr, out, nullptr, workspace, workspace_bytes, stream, nullptr, 0,
                              (int)plan->ops.size(), 1);
}

extern "C" int hpl_plan_set_unlayout(hpl_plan *plan, const hpl_relayout_job *jobs, const int64_t *prefix, const float *src,
                                     const int32_t *bucket_first, const int64_t *bucket_offset, int n_buckets) {
    HPL_REQUIRE(plan && jobs && prefix && src && bucket_first && bucket_offset && n_buckets > 0, "hpl_plan_set_unlayout: bad arguments");
    plan->ul_jobs = jobs; plan->ul_prefix = prefix; plan->ul_src = src;
    plan->ul_first.assign(bucket_first, bucket_first + n_buckets + 1);
    plan->ul_offset.assign(bucket_offset, bucket_offset + n_buckets + 1);
    return HPL_OK;
}

extern "C" int hpl_plan_run_range(hpl_plan *plan, const hpl_level_tables *levels, int n_levels, const float *pc1, const float *pc2,
                                  const float *sf, float *out, float *loss, void *workspace, int64_t workspace_bytes, hplStream stream,
                                  hplStream side_stream, int op_begin, int op_end, int join) {
    HPL_REQUIRE(plan && pc1 && pc2 && out && workspace, "hpl_plan_run: null argument");
    HPL_REQUIRE(op_begin >= 0 && op_begin <= op_end && op_end <= (int)plan->ops.size(), "hpl_plan_run_range: ops [%d, %d) of %d", op_begin,
                op_end, (int)plan->ops.size());
    int64_t sym[MAX_SYMS];
    int rc = resolve_syms(levels, n_levels, sym);
    if (rc) return rc;
    // carve the activation matrices and the split-K scratch out of the workspace
    char *w = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) / 256 * 256);
    char *const end = reinterpret_cast<char *>(workspace) + workspace_bytes;
    const int64_t need = plan_layout(*plan, levels, n_levels, sym);
    HPL_REQUIRE(w + need - 256 <= end, "hpl_plan_run: workspace of %lld bytes is too small (hpl_plan_workspace_bytes: %lld)",
                (long long)workspace_bytes, (long long)need);
    plan->base.resize(plan->bufs.size());
    plan->rows.resize(plan->bufs.size());
    for (size_t i = 0; i < plan->bufs.size(); ++i) {
        plan->rows[i] = sym[plan->bufs[i].rows_sym];
        plan->base[i] = plan->lay_off[i] >= 0 ? reinterpret_cast<float *>(w + plan->lay_off[i]) : nullptr;
    }
    float *splitk = reinterpret_cast<float *>(w);
    Runner r{*plan, levels, n_levels, sym, {pc1, pc2}, out, splitk, to_stream(stream), stream};
    r.sf = sf; r.loss = loss;
    r.main_s = to_stream(stream);
    r.side_s = side_stream ? to_stream(side_stream) : nullptr;
    r.amax_base = reinterpret_cast<float *>(w + SPLITK_WS_BYTES);
    plan->fence_used = 0;
    if (!plan->guard_trips && hpl_gc::split_planes() == 2) {
        if (hipMalloc(reinterpret_cast<void **>(&plan->guard_trips), 4) != hipSuccess || hipMemsetAsync(plan->guard_trips, 0, 4, r.main_s) != hipSuccess) {
            set_error("hpl_plan_run: allocating the range-guard counter failed");
            return HPL_EHIP;
        }
    }
    if (op_begin == 0) {          // a new pair: nothing reduced yet
        plan->amax.clear();
        plan->amax_used = 0;
        if (hpl_gc::split_planes() == 2 && hipMemsetAsync(r.amax_base, 0, AMAX_SLOTS * 8, r.main_s) != hipSuccess) { set_error("hpl_plan_run: hipMemsetAsync failed"); return HPL_EHIP; }
    }
    auto active = [&](const hpl_op &op, bool &run) -> int {
        run = true;
        if (op.cond != HPL_COND_ALWAYS) {
            HPL_REQUIRE(op.cond_level >= 0 && op.cond_level < n_levels, "hpl_plan_run: condition on level %d", op.cond_level);
            const bool shrink = levels[op.cond_level].n0 < levels[op.cond_level].H0;
            run = (op.cond == HPL_COND_SHRINK) == shrink;
        }
        return HPL_OK;
    };
    auto is_emg = [](const hpl_op &op) { return op.kind == HPL_OP_COPY && op.a.buf == -1; };
    // every el_minus_gr copy of the forward in one launch, ahead of the layers (hpl_plan_create checked that nothing
    // else writes those columns: plan->hoist_emg)
    constexpr int batch_emg = 1;
    const bool hoist = batch_emg && plan->hoist_emg;
    if (hoist && op_begin == 0) {
        EmgJobs jobs;
        jobs.n = 0;
        int64_t most = 0;
        auto flush = [&]() {
            if (!jobs.n) return;
            dim3 grid((unsigned)imin(cdiv(most, 256), 64), (unsigned)jobs.n);
            k_copy_emg_batch<<<grid, 256, 0, r.s>>>(jobs);
            jobs.n = 0;
            most = 0;
        };
        for (const hpl_op &op : plan->ops) {
            if (!is_emg(op)) continue;
            bool run;
            if ((rc = active(op, run))) return rc;
            if (!run) continue;
            View Y;
            if ((rc = r.view(op.out, Y, "copy output"))) return rc;
            const int64_t rows = symv(sym, op.m_sym);
            HPL_REQUIRE(op.level >= 0 && op.level < n_levels && op.C == 4 && Y.cols >= 4 && Y.rows >= rows &&
                            rows <= levels[op.level].n0 + levels[op.level].n1,
                        "hpl_plan_run: emg copy of %lld rows at level %d", (long long)rows, op.level);
            if (rows == 0) continue;
            jobs.src[jobs.n] = reinterpret_cast<const float4 *>(levels[op.level].emg_pair);
            jobs.dst[jobs.n] = Y.p;
            jobs.ldd[jobs.n] = Y.ld;
            jobs.rows[jobs.n] = rows;
            most = most > rows ? most : rows;
            if (++jobs.n == EMG_JOBS) flush();
        }
        flush();
        HPL_CHECK_LAUNCH("hpl_plan_run (emg copies)");
    }
    for (int i = op_begin; i < op_end; ++i) {
        const hpl_op &op = plan->ops[i];
        if (hoist && is_emg(op)) continue;
        bool run;
        if ((rc = active(op, run))) return rc;
        if (!run) continue;
        const bool side = r.side_s && (op.flags & HPL_FLAG_SIDE);
        if ((rc = r.prepare_amax(op, side))) return rc;
        r.amax_forget(op.out);
        r.amax_forget(op.out2);
        if (side) { if ((rc = r.to_side())) return rc; }
        else if (op.kind == HPL_OP_UNLAYOUT && (rc = r.join())) return rc;       // (it reads what the side-stream wgrads wrote)
        rc = r.run_op(op);
        if (side) r.to_main();
        if (rc) return rc;
        r.amax_produced(op);
    }
    if (!join) return HPL_OK;
    r.side_busy = r.side_s != nullptr;         // (earlier ranges of the same step may have left work there)
    return r.join();
}

extern "C" int hpl_plan_guard_trips(hpl_plan *plan, int64_t *count) {
    HPL_REQUIRE(plan && count, "hpl_plan_guard_trips: null argument");
    *count = 0;
    if (!plan->guard_trips) return HPL_OK;
    int32_t v = 0;
    if (hipMemcpy(&v, plan->guard_trips, 4, hipMemcpyDeviceToHost) != hipSuccess) { set_error("hpl_plan_guard_trips: read-back failed"); return HPL_EHIP; }
    *count = v;
    return HPL_OK;
}

extern "C" int hpl_plan_profile(hpl_plan *plan, int tag) {
    HPL_REQUIRE(plan, "hpl_plan_profile: null plan");
    plan->profile_tag = tag;
    return HPL_OK;
}

extern "C" int hpl_plan_clock_probe(hpl_plan *plan, int64_t *clock_probe) {
    HPL_REQUIRE(plan, "hpl_plan_clock_probe: null plan");
    plan->clock_probe = clock_probe;
    return HPL_OK;
}

extern "C" int hpl_plan_profile_read(hpl_plan *plan, int *launches, float *total_ms) {
    HPL_REQUIRE(plan && launches && total_ms, "hpl_plan_profile_read: null argument");
    *launches = 0;
    *total_ms = 0.f;
    for (size_t i = 0; i + 1 < plan->pool_used; i += 2) {
        if (hipEventSynchronize(plan->pool[i + 1]) != hipSuccess) { set_error("hpl_plan_profile_read: event wait failed"); return HPL_EHIP; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, plan->pool[i], plan->pool[i + 1]) == hipSuccess) { *total_ms += ms; ++*launches; }
    }
    plan->pool_used = 0;
    return HPL_OK;
}
