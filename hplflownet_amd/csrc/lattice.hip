// lattice.hip -- permutohedral lattice construction on gfx950.
//
// Replaces the reference's CPU half of the hot path: torch-CPU float ops
// (transforms/transforms.py:300-353), a Python `set` (:387-391), the Numba loops of
// build_unsymmetric (:133-261) and the klib khash behind CFFI (models/khash_int2int.h).
//
//  * keys kernel: one lane per point, the exact op sequence of oracle/lattice_oracle.c
//    (k-ordered fmaf chain, v_rndne, stable descending rank) -> bit-identical keys,
//    barycentric weights and el_minus_gr;
//  * hash build: the table is a global open-addressing array of 64-bit packed keys, but a
//    workgroup never sends duplicate keys to it: its 256 points x 4 simplex vertices are
//    first inserted into a 2048-slot LDS table (ds_cmpst_rtn_b64 + ds_min), where
//    neighbouring points collapse onto shared vertices, and only the distinct keys of the
//    group CAS into global memory, each carrying the minimum entry index seen -- so the
//    global atomics are per distinct (group, vertex) instead of per (point, remainder);
//  * vertex ids must reproduce the reference's first-appearance numbering: a slot keeps
//    the minimum entry index j = 4*point + remainder, the entries that own their slot
//    (first[slot] == j) are flagged, and an exclusive scan of the flags is the id;
//  * neighbour tables are pure lookups (15 + 15 + 225 per pc1 vertex), one lane per probe.
//
// All HBM-bound integer work.  Algorithmic bytes per level: keys 12N in, 80N out; hash
// 64N key bytes + table; neighbours 16H in, 4*(15+15+225)*H1 + 60*H2 out.
#include "lattice_common.h"

using namespace hpl;

namespace hpl {
// defined in index_ops.hip
int exclusive_scan_i32(const int32_t *cnt, int64_t n, int32_t *ptr, int32_t *tmp, hipStream_t s);
}  // namespace hpl

namespace {
using namespace hpl::lat;

__global__ void k_lattice_keys(const float *__restrict__ pc, int64_t N, float scale, const Elev E,
                               int32_t *__restrict__ keys, float *__restrict__ bary,
                               float *__restrict__ emg, int64_t emg_ld) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    lattice_point(pc[n], pc[N + n], pc[2 * N + n], n, N, scale, E, keys, bary, emg, emg_ld);
}

// Both clouds in one launch (blockIdx.z).  The points are either given ((3, N) arrays) or are the
// vertices of the previous level, computed on the fly from their integer keys exactly as
// hpl_lattice_next_points does (transforms.py:461-467) -- the (3, H) arrays are never written.
struct KeysPair {
    const float *pc[2];
    const int32_t *vk[2];
    int64_t vstride[2], n[2];
    int32_t *keys[2];
    float *bary[2], *emg[2];
};

__global__ void k_lattice_keys_pair(const KeysPair a, float divisor, float scale, const Elev E, int64_t emg_ld) {
    const int c = blockIdx.z;
    const int64_t N = a.n[c];
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float q[3];
    if (a.pc[c]) {
        q[0] = a.pc[c][n]; q[1] = a.pc[c][N + n]; q[2] = a.pc[c][2 * N + n];
    } else {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (float)a.vk[c][(int64_t)j * a.vstride[c] + n] / divisor;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float acc = E.e[0 * 3 + i] * v[0];
            acc = fmaf(E.e[1 * 3 + i], v[1], acc);
            acc = fmaf(E.e[2 * 3 + i], v[2], acc);
            acc = fmaf(E.e[3 * 3 + i], v[3], acc);
            q[i] = acc;
        }
    }
    lattice_point(q[0], q[1], q[2], n, N, scale, E, a.keys[c], a.bary[c], a.emg[c], emg_ld);
}

// ---------------------------------------------------------------- workspace layout
struct CloudWS {
    int64_t *tkeys;    // [cap]
    int32_t *tfirst;   // [cap]  min entry index
    int32_t *tid;      // [cap]  vertex id
    int32_t *slot;     // [E]    table slot of every entry
    int32_t *flag;     // [E+1]  ownership flags, then their exclusive scan (in place copy)
    int32_t *scan;     // [E+1]
    int32_t *scan_tmp; // [1026] block sums of the scan
    uint64_t mask;     // cap - 1
    int64_t n, E, cap;
};
struct WS {
    int32_t *mm;       // [8] mins[4], maxs[4]
    CloudWS c[2];
    int64_t bytes;
};

WS carve(void *base, int64_t n1, int64_t n2) {
    WS w;
    char *p = reinterpret_cast<char *>(base);
    auto take = [&](int64_t bytes) { char *r = p; p += (bytes + 255) / 256 * 256; return r; };
    w.mm = reinterpret_cast<int32_t *>(take(8 * sizeof(int32_t)));
    const int64_t ns[2] = {n1, n2};
    for (int c = 0; c < 2; ++c) {
        CloudWS &q = w.c[c];
        q.n = ns[c]; q.E = 4 * ns[c]; q.cap = pow2_at_least(2 * q.E); q.mask = (uint64_t)q.cap - 1;
        q.tkeys = reinterpret_cast<int64_t *>(take(q.cap * 8));
        q.tfirst = reinterpret_cast<int32_t *>(take(q.cap * 4));
        q.tid = reinterpret_cast<int32_t *>(take(q.cap * 4));
        q.slot = reinterpret_cast<int32_t *>(take(q.E * 4));
        q.flag = reinterpret_cast<int32_t *>(take((q.E + 1) * 4));
        q.scan = reinterpret_cast<int32_t *>(take((q.E + 1) * 4));
        q.scan_tmp = reinterpret_cast<int32_t *>(take(1026 * 4));
    }
    w.bytes = p - reinterpret_cast<char *>(base);
    return w;
}

__global__ void k_init_ws(int32_t *mm, int64_t *tk1, int32_t *tf1, int64_t cap1, int64_t *tk2, int32_t *tf2,
                          int64_t cap2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4) mm[i] = INT32_MAX;
    else if (i < 8) mm[i] = INT32_MIN;
    if (i < cap1) { tk1[i] = EMPTY; tf1[i] = INT32_MAX; }
    if (i < cap2) { tk2[i] = EMPTY; tf2[i] = INT32_MAX; }
}

// per-coordinate min/max over keys [4][n][4] of one cloud (transforms.py:384-385)
// Both clouds of a level go through every stage in ONE launch: blockIdx.z selects the cloud.
struct Two {
    const int32_t *keys[2];
    int64_t n[2];
};

__global__ void k_minmax(const Two in, int32_t *mm) {
    const int j = blockIdx.y;   // coordinate
    const int32_t *keys = in.keys[blockIdx.z];
    const int64_t n = in.n[blockIdx.z];
    const int4 *p = reinterpret_cast<const int4 *>(keys + (int64_t)j * n * 4);
    int lo = INT32_MAX, hi = INT32_MIN;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int4 v = p[i];
        lo = min(min(lo, v.x), min(v.y, min(v.z, v.w)));
        hi = max(max(hi, v.x), max(v.y, max(v.z, v.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o));
        hi = max(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mm[j], lo);
        atomicMax(&mm[4 + j], hi);
    }
}

// Hash build with LDS staging.  256 points per workgroup = 1024 entries.
constexpr int LSLOTS = 2048;
struct HashArgs {
    const int32_t *keys[2];
    int64_t n[2];
    int64_t *tkeys[2];
    int32_t *tfirst[2];
    uint64_t mask[2];
    int32_t *slot[2];
    int32_t *flag[2];
    int32_t *scan[2];
    int32_t *tid[2];
    int32_t *vkeys[2];
    int32_t *off[2];
    int32_t *counts;
};

__global__ void __launch_bounds__(256) k_hash_insert(const HashArgs a, const int32_t *__restrict__ mm) {
    const int c = blockIdx.z;
    const int32_t *__restrict__ keys = a.keys[c];
    const int64_t n = a.n[c];
    if ((int64_t)blockIdx.x * 256 >= n) return;          // grid sized for the larger cloud
    int64_t *__restrict__ tkeys = a.tkeys[c];
    int32_t *__restrict__ tfirst = a.tfirst[c];
    const uint64_t mask = a.mask[c];
    int32_t *__restrict__ slot_of = a.slot[c];
    __shared__ unsigned long long lkeys[LSLOTS];
    __shared__ int lmin[LSLOTS];
    __shared__ int lglob[LSLOTS];
    for (int i = threadIdx.x; i < LSLOTS; i += 256) { lkeys[i] = (unsigned long long)EMPTY; lmin[i] = INT32_MAX; }
    __syncthreads();
    const int64_t pnt = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int ls[4] = {-1, -1, -1, -1};
    if (pnt < n) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int k[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) k[j] = keys[((int64_t)j * n + pnt) * 4 + r];
            const unsigned long long packed = (unsigned long long)pack_key(k, mm);
            int s = (int)(mix64(packed) & (LSLOTS - 1));
            while (true) {
                const unsigned long long prev = atomicCAS(&lkeys[s], (unsigned long long)EMPTY, packed);
                if (prev == (unsigned long long)EMPTY || prev == packed) break;
                s = (s + 1) & (LSLOTS - 1);
            }
            atomicMin(&lmin[s], (int)(pnt * 4 + r));
            ls[r] = s;
        }
    }
    __syncthreads();
    // distinct keys of this workgroup -> global table
    for (int i = threadIdx.x; i < LSLOTS; i += 256) {
        const unsigned long long packed = lkeys[i];
        if (packed == (unsigned long long)EMPTY) continue;
        uint64_t s = mix64(packed) & mask;
        while (true) {
            const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long *>(&tkeys[s]),
                                                      (unsigned long long)EMPTY, packed);
            if (prev == (unsigned long long)EMPTY || prev == packed) break;
            s = (s + 1) & mask;
        }
        atomicMin(&tfirst[s], lmin[i]);
        lglob[i] = (int)s;
    }
    __syncthreads();
    if (pnt < n) {
#pragma unroll
        for (int r = 0; r < 4; ++r) slot_of[pnt * 4 + r] = lglob[ls[r]];
    }
}

__global__ void k_flags(const HashArgs a) {
    const int c = blockIdx.z;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < 4 * a.n[c]) a.flag[c][j] = (a.tfirst[c][a.slot[c][j]] == (int32_t)j) ? 1 : 0;
}

// owners publish their id (= rank among owners, i.e. first-appearance order) and vertex key
__global__ void k_assign_ids(const HashArgs a) {
    const int c = blockIdx.z;
    const int32_t *__restrict__ keys = a.keys[c];
    const int64_t n = a.n[c];
    const int32_t *__restrict__ slot_of = a.slot[c];
    const int32_t *__restrict__ flag = a.flag[c];
    const int32_t *__restrict__ scan = a.scan[c];
    int32_t *__restrict__ tid = a.tid[c];
    int32_t *__restrict__ vkeys = a.vkeys[c];
    const int64_t E = 4 * n, vstride = E;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) a.counts[c] = scan[E];
    if (j >= E || !flag[j]) return;
    const int32_t id = scan[j];
    tid[slot_of[j]] = id;
    const int64_t p = j >> 2;
    const int r = (int)(j & 3);
#pragma unroll
    for (int c = 0; c < 4; ++c) vkeys[(int64_t)c * vstride + id] = keys[((int64_t)c * n + p) * 4 + r];
}

__global__ void k_offsets(const HashArgs a) {
    const int c = blockIdx.z;
    const int64_t n = a.n[c];
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= 4 * n) return;
    a.off[c][(j & 3) * n + (j >> 2)] = a.tid[c][a.slot[c][j]];
}

// out[f * ostride + h] = id of vertex (key_h + off_f) in `table`, -1 if absent
__global__ void k_neighbors(const int32_t *__restrict__ vkeys, int64_t vstride, int64_t H, const Offsets offs,
                            const int32_t *__restrict__ mm, const int64_t *__restrict__ tkeys,
                            const int32_t *__restrict__ tid, uint64_t mask, int32_t *__restrict__ out,
                            int64_t ostride, int32_t shift) {
    const int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (h >= H) return;
    int k[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) k[c] = vkeys[(int64_t)c * vstride + h] + offs.v[f * 4 + c];
    const int32_t id = lookup(tkeys, tid, mask, pack_key(k, mm));
    out[(int64_t)f * ostride + h] = id >= 0 ? id + shift : -1;
}

// blur tables of both clouds in one launch (blockIdx.z = cloud), each against its own hash table
struct NbrArgs {
    const int32_t *vkeys[2];
    int64_t vstride[2], H[2];
    const int64_t *tkeys[2];
    const int32_t *tid[2];
    uint64_t mask[2];
    int32_t *out[2];
    int64_t ostride[2];
    int32_t shift[2];
};

__global__ void k_neighbors2(const NbrArgs a, const Offsets offs, const int32_t *__restrict__ mm) {
    const int c = blockIdx.z;
    const int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (h >= a.H[c]) return;
    int k[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) k[j] = a.vkeys[c][(int64_t)j * a.vstride[c] + h] + offs.v[f * 4 + j];
    const int32_t id = lookup(a.tkeys[c], a.tid[c], a.mask[c], pack_key(k, mm));
    a.out[c][(int64_t)f * a.ostride[c] + h] = id >= 0 ? id + a.shift[c] : -1;
}

// corr2p[k][f*H1 + h] = id in table 2 of (key1_h + coff_k + foff_f)   (transforms.py:223-241)
__global__ void k_neighbors_corr2(const int32_t *__restrict__ vkeys, int64_t vstride, int64_t H1,
                                  const Offsets coff, const Offsets foff, const int32_t *__restrict__ mm,
                                  const int64_t *__restrict__ tkeys2, const int32_t *__restrict__ tid2,
                                  uint64_t mask2, int32_t *__restrict__ out) {
    const int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int kf = blockIdx.y;
    const int kc = kf / foff.n, f = kf - kc * foff.n;
    if (h >= H1) return;
    int k[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) k[c] = vkeys[(int64_t)c * vstride + h] + coff.v[kc * 4 + c] + foff.v[f * 4 + c];
    out[((int64_t)kc * foff.n + f) * H1 + h] = lookup(tkeys2, tid2, mask2, pack_key(k, mm));
}

__global__ void k_next_points(const int32_t *__restrict__ vkeys, int64_t vstride, int64_t H, float divisor,
                              const Elev E, float *__restrict__ out) {
    const int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (float)vkeys[(int64_t)j * vstride + h] / divisor;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float acc = E.e[0 * 3 + i] * v[0];
        acc = fmaf(E.e[1 * 3 + i], v[1], acc);
        acc = fmaf(E.e[2 * 3 + i], v[2], acc);
        acc = fmaf(E.e[3 * 3 + i], v[3], acc);
        out[(int64_t)i * H + h] = acc;
    }
}

}  // namespace

extern "C" int hpl_lattice_keys(const float *pc, int64_t N, float scale, int32_t *keys, float *bary,
                                float *emg, int64_t emg_ld, hplStream stream) {
    HPL_REQUIRE(pc && keys && bary && emg && N > 0 && (emg_ld == 0 || emg_ld >= 4), "hpl_lattice_keys: bad arguments");
    HPL_REQUIRE(aligned16(keys), "hpl_lattice_keys: keys must be 16-byte aligned");
    k_lattice_keys<<<(int)cdiv(N, 256), 256, 0, to_stream(stream)>>>(pc, N, scale, make_elev(), keys, bary, emg, emg_ld);
    HPL_CHECK_LAUNCH("hpl_lattice_keys");
    return HPL_OK;
}

extern "C" int hpl_lattice_keys_pair(const float *pc1, const float *pc2, const int32_t *vk1, const int32_t *vk2,
                                     int64_t vstride1, int64_t vstride2, float divisor, int64_t n1, int64_t n2,
                                     float scale, int32_t *keys1, int32_t *keys2, float *bary1, float *bary2,
                                     float *emg1, float *emg2, int64_t emg_ld, hplStream stream) {
    HPL_REQUIRE((pc1 && pc2) || (vk1 && vk2 && divisor != 0.f && vstride1 >= n1 && vstride2 >= n2),
                "hpl_lattice_keys_pair: give the points or the previous level's vertex keys");
    HPL_REQUIRE(keys1 && keys2 && bary1 && bary2 && emg1 && emg2 && n1 > 0 && n2 > 0 && (emg_ld == 0 || emg_ld >= 4),
                "hpl_lattice_keys_pair: bad arguments");
    HPL_REQUIRE(aligned16(keys1) && aligned16(keys2), "hpl_lattice_keys_pair: keys must be 16-byte aligned");
    KeysPair a;
    a.pc[0] = pc1; a.pc[1] = pc2; a.vk[0] = pc1 ? nullptr : vk1; a.vk[1] = pc2 ? nullptr : vk2;
    a.vstride[0] = vstride1; a.vstride[1] = vstride2; a.n[0] = n1; a.n[1] = n2;
    a.keys[0] = keys1; a.keys[1] = keys2; a.bary[0] = bary1; a.bary[1] = bary2; a.emg[0] = emg1; a.emg[1] = emg2;
    k_lattice_keys_pair<<<dim3((unsigned)cdiv(imax(n1, n2), 256), 1, 2), 256, 0, to_stream(stream)>>>(
        a, divisor, scale, make_elev(), emg_ld);
    HPL_CHECK_LAUNCH("hpl_lattice_keys_pair");
    return HPL_OK;
}

extern "C" int64_t hpl_lattice_workspace_bytes(int64_t n1, int64_t n2) {
    if (n1 <= 0 || n2 <= 0) return 0;
    return carve(nullptr, n1, n2).bytes;
}

extern "C" int hpl_lattice_hash(const int32_t *keys1, int64_t n1, const int32_t *keys2, int64_t n2,
                                int32_t *off1, int32_t *off2, int32_t *vkeys1, int32_t *vkeys2,
                                int32_t *counts, void *workspace, int64_t workspace_bytes, hplStream stream) {
    HPL_REQUIRE(keys1 && keys2 && off1 && off2 && vkeys1 && vkeys2 && counts && workspace,
                "hpl_lattice_hash: null pointer");
    HPL_REQUIRE(n1 > 0 && n2 > 0 && 4 * n1 < INT32_MAX && 4 * n2 < INT32_MAX, "hpl_lattice_hash: bad sizes");
    HPL_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, "hpl_lattice_hash: workspace must be 256-byte aligned");
    WS w = carve(workspace, n1, n2);
    HPL_REQUIRE(workspace_bytes >= w.bytes, "hpl_lattice_hash: workspace too small (%lld < %lld)",
                (long long)workspace_bytes, (long long)w.bytes);
    hipStream_t s = to_stream(stream);
    const int64_t capmax = imax(w.c[0].cap, w.c[1].cap);
    k_init_ws<<<(int)cdiv(capmax, 256), 256, 0, s>>>(w.mm, w.c[0].tkeys, w.c[0].tfirst, w.c[0].cap, w.c[1].tkeys,
                                                      w.c[1].tfirst, w.c[1].cap);
    HashArgs a;
    const int32_t *keys[2] = {keys1, keys2};
    int32_t *offs[2] = {off1, off2};
    int32_t *vk[2] = {vkeys1, vkeys2};
    for (int c = 0; c < 2; ++c) {
        const CloudWS &q = w.c[c];
        a.keys[c] = keys[c]; a.n[c] = q.n; a.tkeys[c] = q.tkeys; a.tfirst[c] = q.tfirst; a.mask[c] = q.mask;
        a.slot[c] = q.slot; a.flag[c] = q.flag; a.scan[c] = q.scan; a.tid[c] = q.tid; a.vkeys[c] = vk[c];
        a.off[c] = offs[c];
    }
    a.counts = counts;
    const int64_t nmax = imax(n1, n2), emax = 4 * nmax;
    Two two;
    two.keys[0] = keys1; two.keys[1] = keys2; two.n[0] = n1; two.n[1] = n2;
    k_minmax<<<dim3((unsigned)imin(cdiv(nmax, 1024), 16), 4, 2), 256, 0, s>>>(two, w.mm);   // few groups: 8 contended atomics per wave
    k_hash_insert<<<dim3((unsigned)cdiv(nmax, 256), 1, 2), 256, 0, s>>>(a, w.mm);
    k_flags<<<dim3((unsigned)cdiv(emax, 256), 1, 2), 256, 0, s>>>(a);
    for (int c = 0; c < 2; ++c) {
        const CloudWS &q = w.c[c];
        int rc = exclusive_scan_i32(q.flag, q.E, q.scan, q.scan_tmp, s);
        if (rc != HPL_OK) return rc;
    }
    k_assign_ids<<<dim3((unsigned)cdiv(emax, 256), 1, 2), 256, 0, s>>>(a);
    k_offsets<<<dim3((unsigned)cdiv(emax, 256), 1, 2), 256, 0, s>>>(a);
    HPL_CHECK_LAUNCH("hpl_lattice_hash");
    return HPL_OK;
}

extern "C" int hpl_lattice_neighbors(const void *workspace, int64_t n1, int64_t n2, const int32_t *vkeys1,
                                     const int32_t *vkeys2, int64_t H1, int64_t H2, int bcn_radius,
                                     int corr_filter_radius, int corr_corr_radius, int32_t *blur1,
                                     int32_t *blur2, int64_t blur_stride, int64_t blur2_shift, int32_t *corr1,
                                     int32_t *corr2, hplStream stream) {
    HPL_REQUIRE(workspace && vkeys1 && vkeys2 && n1 > 0 && n2 > 0, "hpl_lattice_neighbors: bad arguments");
    HPL_REQUIRE(H1 > 0 && H2 > 0 && H1 <= 4 * n1 && H2 <= 4 * n2, "hpl_lattice_neighbors: bad vertex counts");
    HPL_REQUIRE(bcn_radius <= 2 && corr_filter_radius <= 2 && corr_corr_radius <= 2,
                "hpl_lattice_neighbors: radius > 2 not supported");
    HPL_REQUIRE((corr_filter_radius == -1) == (corr_corr_radius == -1), "hpl_lattice_neighbors: corr radii must both be set or both -1");
    WS w = carve(const_cast<void *>(workspace), n1, n2);
    hipStream_t s = to_stream(stream);
    if (bcn_radius != -1) {
        HPL_REQUIRE(blur1 && blur2, "hpl_lattice_neighbors: null blur table");
        HPL_REQUIRE(blur_stride == 0 || blur_stride >= imax(H1, H2), "hpl_lattice_neighbors: blur_stride too small");
        HPL_REQUIRE(blur2_shift >= 0 && blur2_shift + H2 < (int64_t)INT32_MAX, "hpl_lattice_neighbors: bad blur2_shift");
        const Offsets o = make_offsets(bcn_radius);
        NbrArgs a;
        const int32_t *vk[2] = {vkeys1, vkeys2};
        int32_t *outs[2] = {blur1, blur2};
        const int64_t Hs[2] = {H1, H2}, ns[2] = {n1, n2};
        for (int c = 0; c < 2; ++c) {
            a.vkeys[c] = vk[c]; a.vstride[c] = 4 * ns[c]; a.H[c] = Hs[c]; a.tkeys[c] = w.c[c].tkeys;
            a.tid[c] = w.c[c].tid; a.mask[c] = w.c[c].mask; a.out[c] = outs[c];
            a.ostride[c] = blur_stride ? blur_stride : Hs[c];
            a.shift[c] = c ? (int32_t)blur2_shift : 0;
        }
        k_neighbors2<<<dim3((unsigned)cdiv(imax(H1, H2), 256), o.n, 2), 256, 0, s>>>(a, o, w.mm);
    }
    if (corr_filter_radius != -1) {
        HPL_REQUIRE(corr2, "hpl_lattice_neighbors: null corr table");
        const Offsets co = make_offsets(corr_corr_radius), fo = make_offsets(corr_filter_radius);
        if (corr1)      // NULL: the caller reuses blur1 (same radius, same table)
            k_neighbors<<<dim3((unsigned)cdiv(H1, 256), co.n), 256, 0, s>>>(vkeys1, 4 * n1, H1, co, w.mm, w.c[0].tkeys,
                                                                            w.c[0].tid, w.c[0].mask, corr1, H1, 0);
        k_neighbors_corr2<<<dim3((unsigned)cdiv(H1, 256), co.n * fo.n), 256, 0, s>>>(
            vkeys1, 4 * n1, H1, co, fo, w.mm, w.c[1].tkeys, w.c[1].tid, w.c[1].mask, corr2);
    }
    HPL_CHECK_LAUNCH("hpl_lattice_neighbors");
    return HPL_OK;
}

extern "C" int hpl_lattice_next_points(const int32_t *vkeys, int64_t vstride, int64_t H, float divisor,
                                       float *out, hplStream stream) {
    HPL_REQUIRE(vkeys && out && H > 0 && vstride >= H && divisor != 0.f, "hpl_lattice_next_points: bad arguments");
    k_next_points<<<(int)cdiv(H, 256), 256, 0, to_stream(stream)>>>(vkeys, vstride, H, divisor, make_elev(), out);
    HPL_CHECK_LAUNCH("hpl_lattice_next_points");
    return HPL_OK;
}
