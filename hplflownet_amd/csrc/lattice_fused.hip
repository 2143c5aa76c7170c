// lattice_fused.hip -- the multi-scale permutohedral lattice of a pair (transforms/transforms.py:358-485: keys and
// barycentric weights :300-353, hash + first-appearance vertex ids :171-207, neighbour tables :209-255) built WITHOUT
// the host in the loop.
//
// The staged builder (lattice.hip + lattice_builder.hip) reads the two vertex counts of every level back to the host
// because they size the next arrays and grids: 7 round trips and ~260 small dependent launches per pair.  Here
//   * every array is sized by a BOUND (vertices <= min(4 x input points, row cap)) and laid out with the EXACT counts,
//     which live in a device block (`dims`) every kernel reads: the tables are bit-identical to the staged builder's;
//   * a launch is a set of TASKS, each a grid-stride loop over device-side counts, so its grid does not depend on
//     them either; the whole build of a pair is enqueued at once and the host reads the counts back ONCE, at the end
//     (the forward needs them on the host: grids, workspaces);
//   * a level is 9 dependent phases (keys -> insert -> flags -> ids -> tables -> 4 phases of CSR / row-order work);
//     phases 5-9 of level L do not feed level L+1, so they share launches with phases 1-5 of the next levels:
//     4 launches per level + 5 = 33 launches for the 7 levels of HPLFlowNet (staged: ~260).
// A level whose vertex count exceeds its bound raises the overflow flag; the caller then rebuilds the pair with the
// staged builder (exact sizes).  All of it is HBM-bound integer work on a few hundred KB: what matters is the number
// of dependent launches, not bandwidth.
#include "lattice_common.h"
#include "lattice_fused.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

using namespace hpl;
using namespace hpl::lat;
using namespace hpl::fused;

namespace {

constexpr int SORT_CHUNK = 2048;        // rows per workgroup pass of the row sort (4 waves x 8 rounds x 64 lanes)
constexpr int SCAN_CHUNK = 1024;        // elements per workgroup of the scans

enum TaskKind {
    T_KEYS = 1, T_INSERT, T_FLAGS, T_IDS, T_OFF, T_BLUR, T_CORR2, T_CSR_SUMS, T_SORT1, T_CSR_SCAN, T_SORT2, T_FILL, T_TILE,
    T_CSR_RANK, T_TILE_RANK, T_HIST2
};

struct Task {
    int32_t kind, level, job, blk0, nblk;
};
constexpr int MAX_TASKS = 16;
struct Launch {
    int32_t n;
    Task t[MAX_TASKS];
};

struct Off15 {
    int v[15 * 4];      // neighbour offsets of radius 1 (transforms.py:112-130)
};

__device__ __forceinline__ int npts(const Level &L, int c) { return L.prev_dims ? L.prev_dims[D_H0 + c] : L.n_host[c]; }
__device__ __forceinline__ int dev_pow2(int x) { int p = 64; while (p < x) p <<= 1; return p; }
__device__ __forceinline__ int dcdiv(int a, int b) { return (a + b - 1) / b; }
// hash slot {key lo, key hi, first, id}
__device__ __forceinline__ int64_t slot_key(const int4 &s) { return (int64_t)(((uint64_t)(uint32_t)s.y << 32) | (uint32_t)s.x); }

// rows of a sort job for this pair, 0 if the job does not run (the same rules as lattice_builder.hip level_tail)
__device__ __forceinline__ int job_rows(const Level &L, const SortJob &J) {
    const int H0 = L.dims[D_H0], H1 = L.dims[D_H1];
    if (J.role == 0) return (H0 + H1 >= L.perm_min_rows) ? H0 + H1 : 0;
    const int n0 = npts(L, 0);
    const bool sparse = (double)H0 / (double)n0 >= (double)L.min_sparsity;
    const bool grouped = L.wide != 0 && L.n_groups >= 2 && sparse && H0 >= L.groups_min_rows;
    if (J.role >= 2) return grouped ? H0 : 0;
    return (H0 >= L.perm_min_rows && (!(grouped && L.wide == 1) || L.has_corr)) ? H0 : 0;
}

// exclusive scan of one int per thread over the 256 threads of the workgroup (scr: 8 ints of LDS); *total = the sum
__device__ __forceinline__ int block_scan_excl(int v, int *scr, int *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    __syncthreads();                      // scr may still be read from the previous use
    if (lane == 63) scr[w] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) base += (i < w) ? scr[i] : 0;
    if (total) *total = scr[0] + scr[1] + scr[2] + scr[3];
    return base + inc - v;
}

__device__ __forceinline__ int block_sum(int v, int *scr) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scr[threadIdx.x >> 6] = v;
    __syncthreads();
    return scr[0] + scr[1] + scr[2] + scr[3];
}

// ------------------------------------------------------------------------------------------------ phase 1: keys
// keys + barycentric + el_minus_gr of both clouds (transforms.py:300-353), the joint key range (:384-385), and the
// clearing of everything the later phases of this level accumulate into
__device__ void task_keys(const Level &L, int b, int nblk, const Elev &E, int *scr) {
    const int n0 = npts(L, 0), n1 = npts(L, 1);
    const int t0 = b * 256 + threadIdx.x, nth = nblk * 256;
    // the points go to the first few workgroups (4 per lane): every workgroup that has points ends in 8 atomics on the
    // SAME 8 words of the dims block, and a thousand of them would be a chain longer than the rest of the phase
    const int pblk = min(nblk, max(1, dcdiv(n0 + n1, 1024)));
    if (b < pblk) {
        int lo[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX}, hi[4] = {INT_MIN, INT_MIN, INT_MIN, INT_MIN};
        for (int i = t0; i < n0 + n1; i += pblk * 256) {
            const int c = i >= n0 ? 1 : 0;
            const int p = c ? i - n0 : i, N = c ? n1 : n0;
            float q[3];
            if (L.pc[0]) {
                const float *pc = L.pc[c];
                q[0] = pc[p]; q[1] = pc[N + p]; q[2] = pc[2 * N + p];
            } else {        // the points are the vertices of the level above (transforms.py:461-467)
                const int32_t *vk = L.prev_vk[c];
                const int64_t vs = L.prev_vstride[c];
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (float)vk[j * vs + p] / L.prev_div;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float acc = E.e[0 * 3 + k] * v[0];
                    acc = fmaf(E.e[1 * 3 + k], v[1], acc);
                    acc = fmaf(E.e[2 * 3 + k], v[2], acc);
                    acc = fmaf(E.e[3 * 3 + k], v[3], acc);
                    q[k] = acc;
                }
            }
            lattice_point(q[0], q[1], q[2], p, N, L.scale, E, L.keys[c], L.bary[c], L.emg + (c ? 4 * (int64_t)n0 : 0), 4, lo, hi);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {       // (lanes without points hold the neutral values)
            int l = lo[j], h = hi[j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                l = min(l, __shfl_xor(l, o));
                h = max(h, __shfl_xor(h, o));
            }
            if ((threadIdx.x & 63) == 0) { scr[(threadIdx.x >> 6) * 8 + j] = l; scr[(threadIdx.x >> 6) * 8 + 4 + j] = h; }
        }
        __syncthreads();
        if (threadIdx.x < 8) {
            const int j = threadIdx.x;
            int v = scr[j];
            for (int w = 1; w < 4; ++w) v = j < 4 ? min(v, scr[w * 8 + j]) : max(v, scr[w * 8 + j]);
            if (j < 4) { if (v != INT_MAX) atomicMin(&L.dims[D_MM + j], v); }
            else if (v != INT_MIN) atomicMax(&L.dims[D_MM + j], v);
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int cap = dev_pow2(8 * (c ? n1 : n0));
        int4 *ts = L.tslot[c];
        for (int i = t0; i < cap; i += nth) ts[i] = make_int4(-1, -1, INT_MAX, -1);          // (EMPTY key, no first entry, no id)
    }
    const int ne = 4 * (n0 + n1);
    for (int i = t0; i <= ne; i += nth) { L.cnt[i] = 0; L.cursor[i] = 0; }
    for (int q = 0; q < L.n_jobs; ++q) {
        const SortJob &J = L.job[q];
        const int ch = min(J.chunks_b, dcdiv(ne, SORT_CHUNK)) * 256;
        for (int i = t0; i < ch; i += nth) J.hist1[i] = 0;
    }
}

// ------------------------------------------------------------------------------------------------ phase 2: insert
// One lane per entry j = 4 * point + remainder: CAS of the packed key into the cloud's open-addressing table, atomicMin of
// j into the slot (the owner = first appearance).  The staged builder deduplicates 1024 keys in LDS first, which saves
// global atomics; here the build is bound by the LENGTH of its dependency chains, not by atomic throughput (<= 2^18 keys),
// and the direct form is two dependent global operations per lane instead of a workgroup-serial LDS round.
__device__ void task_insert(const Level &L, int b, int nblk) {
    const int n0 = npts(L, 0), n1 = npts(L, 1);
    const int32_t *mm = L.dims + D_MM;
    const int nth = nblk * 256;
    const uint64_t mask0 = (uint64_t)dev_pow2(8 * n0) - 1, mask1 = (uint64_t)dev_pow2(8 * n1) - 1;
    for (int i = b * 256 + threadIdx.x; i < 4 * (n0 + n1); i += nth) {
        const int c = i >= 4 * n0 ? 1 : 0;
        const int j = c ? i - 4 * n0 : i, n = c ? n1 : n0;
        const int32_t *__restrict__ keys = L.keys[c];
        int4 *__restrict__ ts = L.tslot[c];
        const uint64_t mask = c ? mask1 : mask0;
        const int p = j >> 2, r = j & 3;
        int k[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) k[x] = keys[((int64_t)x * n + p) * 4 + r];
        const unsigned long long packed = (unsigned long long)pack_key(k, mm);
        uint64_t sl = mix64(packed) & mask;
        while (true) {
            const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long *>(&ts[sl]),
                                                      (unsigned long long)EMPTY, packed);
            if (prev == (unsigned long long)EMPTY || prev == packed) break;
            sl = (sl + 1) & mask;
        }
        atomicMin(&reinterpret_cast<int32_t *>(&ts[sl])[2], j);
        L.slot[c][j] = (int)sl;
    }
}

// ------------------------------------------------------------------------------------------------ phases 3, 4: ids
// An entry owns its vertex if it is the first to name it (tfirst[slot] == j); the vertex id is the owner's rank among
// owners, i.e. first-appearance order (transforms.py:183-192).  Phase 3 counts owners per 1024-entry chunk, phase 4
// turns the counts of the chunks before it + a scan inside the chunk into ids.
__device__ void task_flags(const Level &L, int b, int nblk, int *scr) {
    const int n0 = npts(L, 0), n1 = npts(L, 1);
    const int ch0 = dcdiv(4 * n0, SCAN_CHUNK), ch1 = dcdiv(4 * n1, SCAN_CHUNK);
    for (int q = b; q < ch0 + ch1; q += nblk) {
        const int c = q >= ch0 ? 1 : 0;
        const int qc = c ? q - ch0 : q, E = 4 * (c ? n1 : n0);
        const int32_t *__restrict__ slot = L.slot[c];
        const int4 *__restrict__ ts = L.tslot[c];
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = qc * SCAN_CHUNK + threadIdx.x * 4 + k;
            cnt += (j < E && reinterpret_cast<const int32_t *>(&ts[slot[j]])[2] == j) ? 1 : 0;
        }
        const int tot = block_sum(cnt, scr);
        if (threadIdx.x == 0) L.bsum[c][qc] = tot;
    }
}

__device__ void task_ids(const Level &L, int b, int nblk, int *scr) {
    const int n0 = npts(L, 0), n1 = npts(L, 1);
    const int ch0 = dcdiv(4 * n0, SCAN_CHUNK), ch1 = dcdiv(4 * n1, SCAN_CHUNK);
    for (int q = b; q < ch0 + ch1; q += nblk) {
        const int c = q >= ch0 ? 1 : 0;
        const int qc = c ? q - ch0 : q, n = c ? n1 : n0, E = 4 * n, nch = c ? ch1 : ch0;
        const int32_t *__restrict__ slot = L.slot[c];
        int4 *__restrict__ ts = L.tslot[c];
        const int32_t *__restrict__ keys = L.keys[c];
        int part = 0;
        for (int i = threadIdx.x; i < qc; i += 256) part += L.bsum[c][i];
        const int before = block_sum(part, scr);
        int fl[4], s = 0, sl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = qc * SCAN_CHUNK + threadIdx.x * 4 + k;
            sl[k] = j < E ? slot[j] : 0;
            fl[k] = (j < E && reinterpret_cast<const int32_t *>(&ts[sl[k]])[2] == j) ? 1 : 0;
            s += fl[k];
        }
        int tot;
        int id = before + block_scan_excl(s, scr, &tot);
        const int Hb = L.Hb[c];
        const int64_t vs = L.vstride[c];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!fl[k]) continue;
            if (id < Hb) {
                const int j = qc * SCAN_CHUNK + threadIdx.x * 4 + k;
                reinterpret_cast<int32_t *>(&ts[sl[k]])[3] = id;
                const int p = j >> 2, r = j & 3;
#pragma unroll
                for (int x = 0; x < 4; ++x) L.vk[c][x * vs + id] = keys[((int64_t)x * n + p) * 4 + r];
            }
            ++id;
        }
        if (qc == nch - 1 && threadIdx.x == 0) {
            const int H = before + tot;
            L.dims[D_H0 + c] = H;
            if (H > Hb) L.hdr[HDR_OVERFLOW] = 1;
        }
    }
}

// ------------------------------------------------------------------------------------------------ phase 5: tables
// lattice_offset of every point (transforms.py:194-207) + the per-vertex entry counts of the splat CSR
__device__ void task_off(const Level &L, int b, int nblk) {
    const int n0 = npts(L, 0), n1 = npts(L, 1), H0 = L.dims[D_H0];
    const int nth = nblk * 256;
    for (int i = b * 256 + threadIdx.x; i < 4 * (n0 + n1); i += nth) {
        const int c = i >= 4 * n0 ? 1 : 0;
        const int j = c ? i - 4 * n0 : i, n = c ? n1 : n0;
        const int v = reinterpret_cast<const int32_t *>(&L.tslot[c][L.slot[c][j]])[3];
        L.off[c][(j & 3) * n + (j >> 2)] = v;
        atomicAdd(&L.cnt[v + (c ? H0 : 0)], 1);          // integer atomics: deterministic counts
    }
}

__device__ __forceinline__ uint32_t gray_rank(uint32_t m) {        // position of m in the reflected Gray sequence
    m ^= m >> 1; m ^= m >> 2; m ^= m >> 4; m ^= m >> 8;
    return m;
}

// blur table of the pair [15][H0 + H1] (cloud 2's vertices numbered behind cloud 1's; transforms.py:209-221), one lane
// per vertex: its 15 probes give the tap-presence mask, hence the sort keys of the row orders and their first digit counts
__device__ void task_blur(const Level &L, int b, int nblk, const Off15 &o, int *hist) {
    const int H0 = L.dims[D_H0], H1 = L.dims[D_H1], Hp = H0 + H1;
    const int n0 = npts(L, 0), n1 = npts(L, 1);
    const int32_t *mm = L.dims + D_MM;
    int rows[MAX_JOBS];
#pragma unroll
    for (int q = 0; q < MAX_JOBS; ++q) rows[q] = q < L.n_jobs ? job_rows(L, L.job[q]) : 0;
    const int nth = nblk * 256;
    for (int h0 = b * 256; h0 < Hp; h0 += nth) {            // workgroup-uniform trip count (the barriers below)
        const int h = h0 + threadIdx.x;
        const bool valid = h < Hp;
        uint32_t bits = 0;
        // the digit counts of the row orders: the workgroup's 256 rows lie in ONE 2048-row chunk, so they are counted in LDS and
        // flushed with one atomic per digit that occurs (round 5; a loop over the distinct keys of every wave with one global
        // atomic each -- tap masks differ from row to row on the fine levels -- was 17-19 us of this task's 55 at levels 0 and 1)
#pragma unroll
        for (int q = 0; q < MAX_JOBS; ++q)
            if (rows[q]) hist[q * 256 + threadIdx.x] = 0;
        __syncthreads();
        if (valid) {
            const int c = h >= H0 ? 1 : 0;
            const int hh = c ? h - H0 : h;
            const int32_t *vk = L.vk[c];
            const int64_t vs = L.vstride[c];
            const int4 *ts = L.tslot[c];
            const uint64_t mask = (uint64_t)dev_pow2(8 * (c ? n1 : n0)) - 1;
            const int shift = c ? H0 : 0;
            int kv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) kv[j] = vk[j * vs + hh];
            // The probes of a vertex are independent: hash, load the first two slots of each (ONE 16-byte word per slot: key +
            // id), decide; only a run of two collisions (rare at load <= 0.5) walks on.  Two batches (8 + 7 probes): one batch's
            // slots are 64 registers -- with all 15 in flight the kernel needed 195 registers (2 waves per SIMD for EVERY task of
            // the build, and no room beside a 128 x 256 tile of a forward); with 8 it fits 128 (4 waves per SIMD).
            auto probe = [&](auto f0_tag, auto n_tag) {
                constexpr int F0 = decltype(f0_tag)::value, NF = decltype(n_tag)::value;
                int64_t pk[NF];
                uint32_t sl[NF];
#pragma unroll
                for (int u = 0; u < NF; ++u) {
                    int k[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) k[j] = kv[j] + o.v[(F0 + u) * 4 + j];
                    pk[u] = pack_key(k, mm);
                    sl[u] = (uint32_t)(mix64((uint64_t)pk[u]) & mask);
                }
                int4 s0[NF], s1[NF];
#pragma unroll
                for (int u = 0; u < NF; ++u) { s0[u] = ts[sl[u]]; s1[u] = ts[(sl[u] + 1) & (uint32_t)mask]; }
#pragma unroll
                for (int u = 0; u < NF; ++u) {
                    const int64_t k0 = slot_key(s0[u]), k1 = slot_key(s1[u]);
                    int32_t id;
                    if (pk[u] < 0 || k0 == EMPTY) id = -1;                    // (keys of real vertices are >= 0)
                    else if (k0 == pk[u]) id = s0[u].w;
                    else if (k1 == EMPTY) id = -1;
                    else if (k1 == pk[u]) id = s1[u].w;
                    else {
                        uint64_t x = ((uint64_t)sl[u] + 2) & mask;
                        while (true) {
                            const int4 sx = ts[x];
                            const int64_t kk = slot_key(sx);
                            if (kk == pk[u]) { id = sx.w; break; }
                            if (kk == EMPTY) { id = -1; break; }
                            x = (x + 1) & mask;
                        }
                    }
                    L.blur[(int64_t)(F0 + u) * Hp + h] = id >= 0 ? id + shift : -1;
                    bits |= id >= 0 ? (1u << (F0 + u)) : 0u;
                }
            };
            probe(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
            probe(std::integral_constant<int, 8>{}, std::integral_constant<int, 7>{});
        }
#pragma unroll
        for (int q = 0; q < MAX_JOBS; ++q) {
            if (rows[q] == 0) continue;                     // (uniform)
            const SortJob &J = L.job[q];
            const bool mine = valid && h < rows[q];
            const uint32_t key = gray_rank((bits >> J.f0) & ((1u << J.F) - 1u));
            if (mine) {
                J.key[h] = key;
                atomicAdd(&hist[q * 256 + (int)(key & 255u)], 1);
            }
        }
        __syncthreads();
        static_assert(SORT_CHUNK % 256 == 0, "a workgroup's rows in one chunk");
#pragma unroll
        for (int q = 0; q < MAX_JOBS; ++q) {
            if (rows[q] == 0) continue;
            const int c = hist[q * 256 + threadIdx.x];
            if (c) atomicAdd(&L.job[q].hist1[(h0 / SORT_CHUNK) * 256 + (int)threadIdx.x], c);
        }
        __syncthreads();                                   // (the next round clears the counters)
    }
}

// pc2_corr_indices in the kernel-ready layout [15][15 * H0] (transforms.py:223-241)
__device__ void task_corr2(const Level &L, int b, int nblk, const Off15 &o) {
    const int H0 = L.dims[D_H0];
    const int n1 = npts(L, 1);
    const int32_t *mm = L.dims + D_MM;
    const uint64_t mask = (uint64_t)dev_pow2(8 * n1) - 1;
    const int chh = dcdiv(H0, 256);
    const int32_t *vk = L.vk[0];
    const int64_t vs = L.vstride[0];
    for (int q = b; q < 225 * chh; q += nblk) {
        const int kf = q / chh, h = (q - kf * chh) * 256 + threadIdx.x;
        if (h >= H0) continue;
        const int kc = kf / 15, f = kf - kc * 15;
        int k[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = vk[j * vs + h] + o.v[kc * 4 + j] + o.v[f * 4 + j];
        // (one 16-byte word per probed slot: key and id together)
        const int64_t packed = pack_key(k, mm);
        int32_t id = -1;
        if (packed >= 0) {
            uint64_t x = mix64((uint64_t)packed) & mask;
            while (true) {
                const int4 sx = L.tslot[1][x];
                const int64_t kk = slot_key(sx);
                if (kk == packed) { id = sx.w; break; }
                if (kk == EMPTY) break;
                x = (x + 1) & mask;
            }
        }
        L.corr2[(int64_t)kf * H0 + h] = id;
    }
}

// ------------------------------------------------------------------------------------------------ splat CSR
__device__ void task_csr_sums(const Level &L, int b, int nblk, int *scr) {
    const int Hp = L.dims[D_H0] + L.dims[D_H1];
    const int nch = dcdiv(Hp, SCAN_CHUNK);
    for (int q = b; q < nch; q += nblk) {
        int s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = q * SCAN_CHUNK + threadIdx.x * 4 + k;
            s += i < Hp ? L.cnt[i] : 0;
        }
        const int tot = block_sum(s, scr);
        if (threadIdx.x == 0) L.csum[q] = tot;
    }
}

__device__ void task_csr_scan(const Level &L, int b, int nblk, int *scr) {
    const int Hp = L.dims[D_H0] + L.dims[D_H1];
    const int nch = dcdiv(Hp, SCAN_CHUNK);
    for (int q = b; q < nch; q += nblk) {
        int part = 0;
        for (int i = threadIdx.x; i < q; i += 256) part += L.csum[i];
        const int before = block_sum(part, scr);
        int v[4], s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = q * SCAN_CHUNK + threadIdx.x * 4 + k;
            v[k] = i < Hp ? L.cnt[i] : 0;
            s += v[k];
        }
        int tot;
        int run = before + block_scan_excl(s, scr, &tot);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = q * SCAN_CHUNK + threadIdx.x * 4 + k;
            if (i < Hp) L.csr_ptr[i] = run;
            run += v[k];
        }
        if (q == nch - 1 && threadIdx.x == 0) L.csr_ptr[Hp] = before + tot;
    }
}

// entry e = r * n + p over cloud 1's tables, then cloud 2's (vertices shifted by H0): index_ops.hip Seg2
__device__ void task_fill(const Level &L, int b, int nblk) {
    const int n0 = npts(L, 0), n1 = npts(L, 1), H0 = L.dims[D_H0];
    const int ne0 = 4 * n0, ne = 4 * (n0 + n1);
    const int nth = nblk * 256;
    for (int e = b * 256 + threadIdx.x; e < ne; e += nth) {
        const int v = e < ne0 ? L.off[0][e] : L.off[1][e - ne0] + H0;
        const int s = atomicAdd(&L.cursor[v], 1);
        L.ent[L.csr_ptr[v] + s] = e;
    }
}

// a 16-lane group per vertex ranks the entries of its segment (ascending e: the summation order of the splat) and emits
// (point, weight); then the density normaliser 1 / (sum + 1e-5) in that order (models/bilateralNN.py:168-183).  Segments of
// <= 16 entries (nearly all) live in the group's registers: ranks by shuffles, the ordered weights through LDS.
__device__ void task_csr_rank(const Level &L, int b, int nblk, float *wbuf) {
    const int n0 = npts(L, 0), n1 = npts(L, 1), Hp = L.dims[D_H0] + L.dims[D_H1];
    const int ne0 = 4 * n0;
    const int lg = threadIdx.x & 15, grp = threadIdx.x >> 4, lane0 = threadIdx.x & 48;
    const int ngroups = nblk * 16;
    for (int v0 = 0; v0 < Hp; v0 += ngroups) {
        const int v = v0 + b * 16 + grp;
        int bb = 0, ee = 0;
        if (v < Hp) { bb = L.csr_ptr[v]; ee = L.csr_ptr[v + 1]; }
        const int len = ee - bb;
        if (len <= 16) {
            const int x = lg < len ? L.ent[bb + lg] : INT_MAX;
            int rank = 0;
#pragma unroll
            for (int o = 0; o < 16; ++o) rank += (__shfl(x, lane0 + o, 64) < x) ? 1 : 0;
            if (lg < len) {
                const float w = x < ne0 ? L.bary[0][x] : L.bary[1][x - ne0];
                L.csr_pt[bb + rank] = x < ne0 ? x % n0 : n0 + (x - ne0) % n1;
                L.csr_w[bb + rank] = w;
                wbuf[grp * 16 + rank] = w;
            }
            if (lg == 0 && v < Hp) {            // (same wave: the LDS stores above are ordered before these loads)
                float s = 0.f;
                for (int i = 0; i < len; ++i) s += wbuf[grp * 16 + i];
                L.norm[v] = 1.0f / (s + 1e-5f);
            }
        } else if (len <= 64) {
            // (round 5) segments of 17 .. 64 entries -- most vertices of level 2 (14.6 contributors on average) and of the coarse
            // levels -- are staged in LDS by their 16-lane group and ranked from there: the general path below reads the segment
            // from memory once per entry (47.6 us for level 2, 15-21 us for levels of a few hundred vertices)
            int *ents = reinterpret_cast<int *>(wbuf) + 256 + grp * 64;
            float *ws = wbuf + 256 + 1024 + grp * 64;
            for (int i = lg; i < len; i += 16) ents[i] = L.ent[bb + i];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (same wave: its LDS stores are done before its loads below)
            __builtin_amdgcn_wave_barrier();
            for (int i = lg; i < len; i += 16) {
                const int x = ents[i];
                int rank = 0;
                for (int j = 0; j < len; ++j) rank += (ents[j] < x) ? 1 : 0;
                const float w = x < ne0 ? L.bary[0][x] : L.bary[1][x - ne0];
                L.csr_pt[bb + rank] = x < ne0 ? x % n0 : n0 + (x - ne0) % n1;
                L.csr_w[bb + rank] = w;
                ws[rank] = w;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (lg == 0) {
                float s = 0.f;
                for (int i = 0; i < len; ++i) s += ws[i];
                L.norm[v] = 1.0f / (s + 1e-5f);
            }
        } else {
            for (int i = bb + lg; i < ee; i += 16) {
                const int x = L.ent[i];
                int rank = 0;
                for (int j = bb; j < ee; ++j) rank += (L.ent[j] < x) ? 1 : 0;
                L.csr_pt[bb + rank] = x < ne0 ? x % n0 : n0 + (x - ne0) % n1;
                L.csr_w[bb + rank] = x < ne0 ? L.bary[0][x] : L.bary[1][x - ne0];
            }
            __threadfence_block();            // the group's stores, then its lane 0 reads them back
            if (lg == 0) {
                float s = 0.f;
                for (int i = bb; i < ee; ++i) s += *reinterpret_cast<volatile float *>(&L.csr_w[i]);
                L.norm[v] = 1.0f / (s + 1e-5f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ row orders
// Stable LSD radix sort of the rows by the Gray rank of their tap mask (<= 15 bits: two 8-bit digits; tap groups of
// <= 8 taps: one), values = ascending row ids -- the order hpl_tap_order produces (row_order.hip).  A pass: the digit
// counts of every 2048-row chunk exist already (pass 1: counted while the keys were made; pass 2: counted by pass 1's
// scatter); a workgroup derives the base of each digit for its chunk from them, ranks its rows per digit with wave
// ballots (rows in lane order inside a round, rounds and waves in order: stable) and scatters.
__device__ void task_sort(const Level &L, const SortJob &J, int pass, int b, int nblk, char *smem) {
    const int M = job_rows(L, J);
    if (M == 0 || (pass == 2 && !J.two_pass)) return;
    int *base = reinterpret_cast<int *>(smem);     // [256]
    int *whist = base + 256;                       // [4][256]
    int *scr = whist + 1024;                       // [8]
    const int nch = dcdiv(M, SORT_CHUNK);
    const int32_t *hist = pass == 1 ? J.hist1 : J.hist2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, d = threadIdx.x;
    const bool to_tmp = pass == 1 && J.two_pass;
    for (int q = b; q < nch; q += nblk) {
        int before = 0, tot = 0;
        for (int b0 = 0; b0 < nch; b0 += 8) {          // 8 independent loads in flight
            int v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = b0 + u < nch ? hist[(b0 + u) * 256 + d] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                tot += v[u];
                before += b0 + u < q ? v[u] : 0;
            }
        }
        const int ex = block_scan_excl(tot, scr, nullptr);
        base[d] = ex + before;
#pragma unroll
        for (int i = 0; i < 4; ++i) whist[i * 256 + d] = 0;
        __syncthreads();
        uint32_t key[8];
        int val[8], rw[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int row = q * SORT_CHUNK + w * 512 + s * 64 + lane;
            const bool valid = row < M;
            key[s] = valid ? (pass == 1 ? J.key[row] : J.tkey[row]) : 0u;
            val[s] = valid ? (pass == 1 ? row : J.tval[row]) : 0;
            const uint32_t dg = pass == 1 ? (key[s] & 255u) : ((key[s] >> 8) & 255u);
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool one = (dg >> bit) & 1u;
                const unsigned long long bal = __ballot(one);
                peers &= one ? bal : ~bal;
            }
            rw[s] = 0;
            if (valid) {
                const int prefix = whist[w * 256 + dg];
                rw[s] = prefix + __popcll(peers & ((1ull << lane) - 1ull));
                if (lane == 63 - __clzll(peers)) whist[w * 256 + dg] = prefix + __popcll(peers);
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int row = q * SORT_CHUNK + w * 512 + s * 64 + lane;
            const bool valid = row < M;
            const uint32_t dg = pass == 1 ? (key[s] & 255u) : ((key[s] >> 8) & 255u);
            int pos = base[dg] + rw[s];
            for (int i = 0; i < w; ++i) pos += whist[i * 256 + dg];
            if (!valid) continue;
            if (to_tmp) { J.tkey[pos] = key[s]; J.tval[pos] = val[s]; }
            else J.perm[pos] = val[s];
        }
        __syncthreads();
    }
}

// digit counts of pass 2 per 2048-row chunk of pass 1's output (a phase of its own: counting them from pass 1's scatter
// would be one atomic per row on a few hot words)
__device__ void task_hist2(const Level &L, const SortJob &J, int b, int nblk, int *lh) {
    const int M = job_rows(L, J);
    if (M == 0 || !J.two_pass) return;
    const int nch = dcdiv(M, SORT_CHUNK);
    const int lane = threadIdx.x & 63;
    for (int q = b; q < nch; q += nblk) {
        lh[threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int row = q * SORT_CHUNK + s * 256 + threadIdx.x;
            const bool valid = row < M;
            const uint32_t dg = valid ? ((J.tkey[row] >> 8) & 255u) : 0u;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool one = (dg >> bit) & 1u;
                const unsigned long long bal = __ballot(one);
                peers &= one ? bal : ~bal;
            }
            if (valid && lane == 63 - __clzll(peers)) atomicAdd(&lh[dg], (int)__popcll(peers));
        }
        __syncthreads();
        J.hist2[q * 256 + threadIdx.x] = lh[threadIdx.x];
        __syncthreads();
    }
}

// per-tile gather indices + tap masks of a row order (gconv.hip k_tile_index)
__device__ void task_tile(const Level &L, const SortJob &J, int b, int nblk, int *masks) {
    const int M = job_rows(L, J);
    if (M == 0) return;
    const int Hp = L.dims[D_H0] + L.dims[D_H1];
    const int BM = J.bm, F = J.F;
    const int32_t *nbr = L.blur + (int64_t)J.f0 * Hp;
    const int tiles = dcdiv(M, BM);
    const int t = threadIdx.x;
    const int r = t % BM;
    for (int tile = b; tile < tiles; tile += nblk) {
        if (t < 8) masks[t] = 0;
        __syncthreads();
        const int m = tile * BM + r;
        const int v = m < M ? J.perm[m] : -1;
        int32_t *out = J.tidx + (int64_t)tile * F * BM;
        int mybits = 0;
        for (int f = t / BM; f < F; f += 256 / BM) {
            const int row = v >= 0 ? nbr[(int64_t)f * Hp + v] : -1;
            out[f * BM + r] = row;
            mybits |= row >= 0 ? (1 << f) : 0;
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) mybits |= __shfl_xor(mybits, o, 64);     // OR over the 32 rows of a block
        if ((t & 31) == 0 && mybits) {
            atomicOr(&masks[0], mybits);
            atomicOr(&masks[2 + (r >> 5)], mybits);
        }
        __syncthreads();
        if (t < 8) J.tmask[(int64_t)tile * 8 + t] = masks[t];
        __syncthreads();
    }
}

// schedule of the tiles, most taps first (stable counting sort by tap count: gconv.hip k_tile_rank); one workgroup
__device__ void task_tile_rank(const Level &L, const SortJob &J, int *sm) {
    const int M = job_rows(L, J);
    if (M == 0) return;
    const int tiles = dcdiv(M, J.bm);
    int *hist = sm, *base = sm + 16, *running = sm + 32, *wcnt = sm + 48;       // wcnt [4][16]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t < 16) { hist[t] = 0; running[t] = 0; }
    __syncthreads();
    for (int j = t; j < tiles; j += 256) atomicAdd(&hist[__popc(J.tmask[(int64_t)j * 8] & 0x7fff)], 1);
    __syncthreads();
    if (t == 0) {
        int off = 0;
        for (int c = 15; c >= 0; --c) { base[c] = off; off += hist[c]; }
    }
    __syncthreads();
    for (int j0 = 0; j0 < tiles; j0 += 256) {
        const int j = j0 + t;
        const int c = j < tiles ? __popc(J.tmask[(int64_t)j * 8] & 0x7fff) : -1;
        int mine = 0;
#pragma unroll
        for (int cls = 0; cls < 16; ++cls) {
            const unsigned long long bal = __ballot(c == cls);
            if (c == cls) mine = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wcnt[wave * 16 + cls] = __popcll(bal);
        }
        __syncthreads();
        if (c >= 0) {
            int before = 0;
            for (int w = 0; w < wave; ++w) before += wcnt[w * 16 + c];
            J.tmask[(int64_t)(base[c] + running[c] + before + mine) * 8 + 6] = j;
        }
        __syncthreads();
        if (t < 16) running[t] += wcnt[t] + wcnt[16 + t] + wcnt[32 + t] + wcnt[48 + t];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ the kernel
__global__ void __launch_bounds__(256) k_lattice_fused(const Level *__restrict__ levels, const Launch l, const Elev E,
                                                       const Off15 o) {
    __shared__ __attribute__((aligned(16))) char smem[9216];        // the sort's digit bases and per-wave counts; scan scratch; csr_rank's segments
    int ti = 0;
    for (int i = 1; i < l.n; ++i) ti = ((int)blockIdx.x >= l.t[i].blk0) ? i : ti;
    const Task t = l.t[ti];
    const Level &L = levels[t.level];
    if (L.hdr[HDR_OVERFLOW]) return;
    const int b = (int)blockIdx.x - t.blk0;
    int *ism = reinterpret_cast<int *>(smem);
    switch (t.kind) {
    case T_KEYS: task_keys(L, b, t.nblk, E, ism); break;
    case T_INSERT: task_insert(L, b, t.nblk); break;
    case T_FLAGS: task_flags(L, b, t.nblk, ism); break;
    case T_IDS: task_ids(L, b, t.nblk, ism); break;
    case T_OFF: task_off(L, b, t.nblk); break;
    case T_BLUR: task_blur(L, b, t.nblk, o, ism); break;
    case T_CORR2: task_corr2(L, b, t.nblk, o); break;
    case T_CSR_SUMS: task_csr_sums(L, b, t.nblk, ism); break;
    case T_SORT1: task_sort(L, L.job[t.job], 1, b, t.nblk, smem); break;
    case T_CSR_SCAN: task_csr_scan(L, b, t.nblk, ism); break;
    case T_SORT2: task_sort(L, L.job[t.job], 2, b, t.nblk, smem); break;
    case T_FILL: task_fill(L, b, t.nblk); break;
    case T_TILE: task_tile(L, L.job[t.job], b, t.nblk, ism); break;
    case T_CSR_RANK: task_csr_rank(L, b, t.nblk, reinterpret_cast<float *>(smem)); break;
    case T_TILE_RANK: task_tile_rank(L, L.job[t.job], ism); break;
    case T_HIST2: task_hist2(L, L.job[t.job], b, t.nblk, ism); break;
    default: break;
    }
}

__global__ void k_fused_begin(int32_t *dims, int n_levels) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1 + n_levels) * DIM_INTS) return;
    const int rec = i / DIM_INTS, k = i - rec * DIM_INTS;
    int v = 0;
    if (rec >= 1 && k >= D_MM && k < D_MM + 4) v = INT_MAX;
    if (rec >= 1 && k >= D_MM + 4 && k < D_MM + 8) v = INT_MIN;
    dims[i] = v;
}

inline int64_t align256(int64_t b) { return (b + 255) / 256 * 256; }
inline int filter_size(int r) { return (r + 1) * (r + 1) * (r + 1) * (r + 1) - r * r * r * r; }
inline int clampi(int64_t v, int lo, int hi) { return (int)(v < lo ? lo : (v > hi ? hi : v)); }

}  // namespace

namespace hpl {
namespace fused {

int64_t default_row_cap(int64_t n0, int64_t n1) { return 16 * imax(n0, n1); }

bool supported(const hpl_lattice_spec &sp) {
    if (sp.n_levels < 1 || sp.n_levels > HPL_MAX_LEVELS || sp.n_groups > 4) return false;
    for (int L = 0; L < sp.n_levels; ++L) {
        const int bcn = sp.bcn_radius[L], cf = sp.corr_filter_radius[L], cc = sp.corr_corr_radius[L];
        if (bcn != 1) return false;                         // every level of the shipped configs blurs with radius 1
        if ((cf == -1) != (cc == -1)) return false;
        if (cf != -1 && (cf != 1 || cc != 1)) return false; // corr1 then IS the cloud-1 blur table
    }
    if (sp.n_groups >= 2)
        for (int g = 0; g < sp.n_groups; ++g)
            if (sp.group_cut[g + 1] <= sp.group_cut[g] || sp.group_cut[g + 1] > 15) return false;
    return true;
}

int64_t layout(const hpl_lattice_spec &sp, int64_t n0, int64_t n1, const int64_t *bounds, const float *pc1,
               const float *pc2, char *arena, Plan &plan) {
    int64_t used = 0;
    auto take = [&](int64_t bytes) -> char * {
        char *r = arena ? arena + used : nullptr;
        used += align256(bytes);
        return r;
    };
    plan.n_levels = sp.n_levels;
    plan.d_levels = reinterpret_cast<Level *>(take(sizeof(Level) * HPL_MAX_LEVELS));
    plan.d_dims = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * DIM_INTS * (1 + HPL_MAX_LEVELS)));
    const int64_t cap = default_row_cap(n0, n1);
    int64_t nb[2] = {n0, n1};
    for (int Li = 0; Li < sp.n_levels; ++Li) {
        Level &L = plan.lv[Li];
        memset(&L, 0, sizeof(L));
        L.index = Li; L.n_levels = sp.n_levels;
        L.n_host[0] = (int32_t)n0; L.n_host[1] = (int32_t)n1;
        int64_t Hb[2];
        for (int c = 0; c < 2; ++c) {
            Hb[c] = imin(4 * nb[c], cap);
            if (bounds && bounds[Li] > 0) Hb[c] = imin(4 * nb[c], bounds[Li]);
            L.nb[c] = (int32_t)nb[c];
            L.Hb[c] = (int32_t)Hb[c];
        }
        if (4 * (nb[0] + nb[1]) >= (int64_t)INT32_MAX / 64) return -1;
        const int64_t Nbp = nb[0] + nb[1], Hbp = Hb[0] + Hb[1];
        L.scale = sp.scale[Li];
        L.prev_div = Li ? sp.next_divisor[Li - 1] : 1.f;
        L.has_blur = 1;
        L.has_corr = sp.corr_filter_radius[Li] != -1;
        L.wide = sp.wide_up[Li];
        L.n_groups = sp.n_groups;
        L.perm_min_rows = (int32_t)imin(sp.perm_min_rows, INT32_MAX);
        L.groups_min_rows = (int32_t)imin(sp.groups_min_rows > 0 ? sp.groups_min_rows : sp.perm_min_rows, INT32_MAX);
        L.min_sparsity = sp.groups_min_sparsity;
        L.hdr = plan.d_dims;
        L.dims = plan.d_dims ? plan.d_dims + DIM_INTS * (1 + Li) : nullptr;
        L.prev_dims = (Li && plan.d_dims) ? plan.d_dims + DIM_INTS * Li : nullptr;
        if (Li == 0) { L.pc[0] = pc1; L.pc[1] = pc2; }
        else {
            const Level &P = plan.lv[Li - 1];
            L.prev_vk[0] = P.vk[0]; L.prev_vk[1] = P.vk[1];
            L.prev_vstride[0] = P.vstride[0]; L.prev_vstride[1] = P.vstride[1];
        }
        L.emg = reinterpret_cast<float *>(take(Nbp * 16));
        for (int c = 0; c < 2; ++c) {
            L.keys[c] = reinterpret_cast<int32_t *>(take(nb[c] * 64));
            L.bary[c] = reinterpret_cast<float *>(take(nb[c] * 16));
            L.off[c] = reinterpret_cast<int32_t *>(take(nb[c] * 16));
            L.vstride[c] = (int32_t)Hb[c];
            L.vk[c] = reinterpret_cast<int32_t *>(take(Hb[c] * 16));
            const int64_t capb = pow2_at_least(8 * nb[c]);
            L.tslot[c] = reinterpret_cast<int4 *>(take(capb * 16));
            L.slot[c] = reinterpret_cast<int32_t *>(take(nb[c] * 16));
            L.bsum[c] = reinterpret_cast<int32_t *>(take(cdiv(4 * nb[c], SCAN_CHUNK) * 4 + 4));
        }
        L.blur = reinterpret_cast<int32_t *>(take(15 * Hbp * 4));
        if (L.has_corr) L.corr2 = reinterpret_cast<int32_t *>(take(225 * Hb[0] * 4));
        const int64_t ne = 4 * Nbp;
        L.cnt = reinterpret_cast<int32_t *>(take((ne + 1) * 4));
        L.cursor = reinterpret_cast<int32_t *>(take((ne + 1) * 4));
        L.csum = reinterpret_cast<int32_t *>(take(cdiv(Hbp, SCAN_CHUNK) * 4 + 4));
        L.ent = reinterpret_cast<int32_t *>(take(ne * 4));
        L.csr_ptr = reinterpret_cast<int32_t *>(take((Hbp + 1) * 4));
        L.csr_pt = reinterpret_cast<int32_t *>(take(ne * 4));
        L.csr_w = reinterpret_cast<float *>(take(ne * 4));
        L.norm = reinterpret_cast<float *>(take(Hbp * 4));
        // row orders that can exist under the bounds
        int nj = 0;
        auto add_job = [&](int kind, int role, int f0, int F, int bm, int64_t Mb) {
            SortJob &J = L.job[nj++];
            J.kind = kind; J.role = role; J.f0 = f0; J.F = F; J.bm = bm; J.two_pass = F > 8;
            J.chunks_b = (int32_t)cdiv(Mb, SORT_CHUNK);
            J.key = reinterpret_cast<uint32_t *>(take(Mb * 4));
            J.hist1 = reinterpret_cast<int32_t *>(take((int64_t)J.chunks_b * 256 * 4));
            J.hist2 = reinterpret_cast<int32_t *>(take((int64_t)J.chunks_b * 256 * 4));
            if (J.two_pass) {
                J.tkey = reinterpret_cast<uint32_t *>(take(Mb * 4));
                J.tval = reinterpret_cast<int32_t *>(take(Mb * 4));
            }
            J.perm = reinterpret_cast<int32_t *>(take(Mb * 4));
            const int64_t tiles = cdiv(Mb, bm);
            J.tidx = reinterpret_cast<int32_t *>(take(tiles * F * bm * 4));
            J.tmask = reinterpret_cast<int32_t *>(take(tiles * 8 * 4));
        };
        const int gbm = sp.group_tile_bm == 128 ? 128 : 64;
        if (Hbp >= sp.perm_min_rows) add_job(1, 0, 0, 15, 64, Hbp);
        if (L.wide != 0 && sp.n_groups >= 2 && Hb[0] >= L.groups_min_rows)
            for (int g = 0; g < sp.n_groups; ++g)
                add_job(2, 2 + g, sp.group_cut[g], sp.group_cut[g + 1] - sp.group_cut[g], gbm, Hb[0]);
        if (Hb[0] >= sp.perm_min_rows) add_job(2, 1, 0, 15, 64, Hb[0]);
        L.n_jobs = nj;
        nb[0] = Hb[0]; nb[1] = Hb[1];
    }
    plan.bytes = used;
    return used;
}

int enqueue(Plan &plan, Level *lv_stage, int32_t *dims_host, hipEvent_t counts_ev, hipStream_t s) {
    const int nlev = plan.n_levels;
    memcpy(lv_stage, plan.lv, sizeof(Level) * nlev);
    if (hipMemcpyAsync(plan.d_levels, lv_stage, sizeof(Level) * nlev, hipMemcpyHostToDevice, s) != hipSuccess) {
        set_error("hpl_lattice (fused): copy of the level descriptors failed");
        return HPL_EHIP;
    }
    k_fused_begin<<<cdiv((1 + nlev) * DIM_INTS, 256), 256, 0, s>>>(plan.d_dims, nlev);
    const Elev E = make_elev();
    const Offsets full = make_offsets(1);
    Off15 o;
    for (int i = 0; i < 60; ++i) o.v[i] = full.v[i];
    const int n_launches = 4 * (nlev - 1) + 10;
    plan.launches = 1;
    for (int t = 0; t < n_launches; ++t) {
        Launch l;
        l.n = 0;
        int blk = 0;
        // grids are sized by BOUNDS, i.e. mostly idle workgroups -- and an idle workgroup still needs a CU slot that a tile of
        // a forward kernel running beside the build cannot use meanwhile (under a kernel trace of the pipelined bench the
        // lattice launches showed up with 20 % of the GPU's busy time).  384 workgroups per task keep every grid-stride loop
        // short at N = 8 192 (one build alone: 0.76 ms, uncapped 0.78) and take the build's cost in the pipeline to ~0
        // (pairs/s with the build ~ without it; 1-9 % over the uncapped grids depending on the box).
        constexpr int64_t max_blocks = 384;
        bool dropped = false;
        auto add = [&](int kind, int level, int job, int64_t nblk) {
            if (l.n >= MAX_TASKS) { dropped = true; return; }
            nblk = imin(nblk, max_blocks);          // (round 5: 1 024 for the CSR ranking alone, eleven rounds -> four at levels 0-1: no gain)
            Task &k = l.t[l.n++];
            k.kind = kind; k.level = level; k.job = job; k.blk0 = blk; k.nblk = (int)nblk;
            blk += (int)nblk;
        };
        for (int Li = 0; Li < nlev; ++Li) {
            const int k = t - 4 * Li;
            if (k < 0 || k > 9) continue;
            const Level &L = plan.lv[Li];
            const int64_t Nbp = (int64_t)L.nb[0] + L.nb[1], Hbp = (int64_t)L.Hb[0] + L.Hb[1];
            const int64_t capb = pow2_at_least(8 * (int64_t)L.nb[0]) + pow2_at_least(8 * (int64_t)L.nb[1]);
            switch (k) {
            case 0: add(T_KEYS, Li, 0, clampi(cdiv(capb, 512), 1, 1024)); break;
            case 1: add(T_INSERT, Li, 0, clampi(cdiv(4 * Nbp, 256), 1, 2048)); break;
            case 2: add(T_FLAGS, Li, 0, clampi(cdiv(4 * Nbp, SCAN_CHUNK) + 1, 1, 512)); break;
            case 3: add(T_IDS, Li, 0, clampi(cdiv(4 * Nbp, SCAN_CHUNK) + 1, 1, 512)); break;
            case 4:
                add(T_OFF, Li, 0, clampi(cdiv(4 * Nbp, 1024), 1, 256));
                add(T_BLUR, Li, 0, clampi(cdiv(Hbp, 256), 1, 1024));
                if (L.has_corr) add(T_CORR2, Li, 0, clampi(225 * cdiv(L.Hb[0], 256) / 4, 1, 2048));
                break;
            case 5:
                add(T_CSR_SUMS, Li, 0, clampi(cdiv(Hbp, SCAN_CHUNK), 1, 256));
                for (int q = 0; q < L.n_jobs; ++q) add(T_SORT1, Li, q, clampi(L.job[q].chunks_b, 1, 256));
                break;
            case 6:
                add(T_CSR_SCAN, Li, 0, clampi(cdiv(Hbp, SCAN_CHUNK), 1, 256));
                for (int q = 0; q < L.n_jobs; ++q) {
                    if (L.job[q].two_pass) add(T_HIST2, Li, q, clampi(L.job[q].chunks_b, 1, 256));
                    else add(T_TILE, Li, q, clampi(cdiv((int64_t)L.job[q].chunks_b * SORT_CHUNK, L.job[q].bm), 1, 1024));
                }
                break;
            case 7:
                add(T_FILL, Li, 0, clampi(cdiv(4 * Nbp, 1024), 1, 256));
                for (int q = 0; q < L.n_jobs; ++q) {
                    if (L.job[q].two_pass) add(T_SORT2, Li, q, clampi(L.job[q].chunks_b, 1, 256));
                    else add(T_TILE_RANK, Li, q, 1);
                }
                break;
            case 8:
                add(T_CSR_RANK, Li, 0, clampi(cdiv(Hbp, 16), 1, 1024));
                for (int q = 0; q < L.n_jobs; ++q)
                    if (L.job[q].two_pass) add(T_TILE, Li, q, clampi(cdiv((int64_t)L.job[q].chunks_b * SORT_CHUNK, L.job[q].bm), 1, 1024));
                break;
            case 9:
                for (int q = 0; q < L.n_jobs; ++q)
                    if (L.job[q].two_pass) add(T_TILE_RANK, Li, q, 1);
                break;
            }
        }
        // (shipped specs peak at 14 tasks per launch: a spec that needs more must fail here, not produce incomplete tables)
        HPL_REQUIRE(!dropped, "hpl_lattice_begin: launch %d of the fused build needs more than %d tasks", t, MAX_TASKS);
        if (l.n == 0) continue;
        static const int split = getenv("HPL_FUSED_SPLIT") ? atoi(getenv("HPL_FUSED_SPLIT")) : 0;
        if (split) {        // diagnostic: every task as a launch of its own, so that a kernel trace times the tasks
            static const char *names[] = {"", "keys", "insert", "flags", "ids", "off", "blur", "corr2", "csr_sums", "sort1", "csr_scan",
                                          "sort2", "fill", "tile", "csr_rank", "tile_rank", "hist2"};
            for (int i = 0; i < l.n; ++i) {
                Launch one;
                one.n = 1;
                one.t[0] = l.t[i];
                one.t[0].blk0 = 0;
                k_lattice_fused<<<l.t[i].nblk, 256, 0, s>>>(plan.d_levels, one, E, o);
                if (split > 1) fprintf(stderr, "launch %d task %s level %d job %d blocks %d\n", t, names[l.t[i].kind], l.t[i].level, l.t[i].job, l.t[i].nblk);
            }
            ++plan.launches;
            goto after_launch;
        }
        k_lattice_fused<<<blk, 256, 0, s>>>(plan.d_levels, l, E, o);
        ++plan.launches;
    after_launch:
        if (t == 4 * (nlev - 1) + 3) {        // every level's vertex counts exist: start the one read-back of the build
            if (hipMemcpyAsync(dims_host, plan.d_dims, sizeof(int32_t) * DIM_INTS * (1 + nlev), hipMemcpyDeviceToHost, s) !=
                    hipSuccess ||
                hipEventRecord(counts_ev, s) != hipSuccess) {
                set_error("hpl_lattice (fused): read-back of the vertex counts failed");
                return HPL_EHIP;
            }
        }
    }
    HPL_CHECK_LAUNCH("hpl_lattice (fused)");
    return HPL_OK;
}

}  // namespace fused
}  // namespace hpl
