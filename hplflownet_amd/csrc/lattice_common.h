// lattice_common.h -- device / host helpers shared by the staged (lattice.hip) and the fused (lattice_fused.hip)
// lattice builders: the float part of transforms/transforms.py:300-353, key packing (:70-86), the 64-bit
// open-addressing probe and the neighbour offsets of Traverse (:112-130).
#pragma once
#include "common.h"

#include <math.h>

namespace hpl {
namespace lat {

constexpr int64_t EMPTY = -1;   // packed keys of real vertices are >= 0

struct Elev {
    float e[12];   // (4,3) row-major elevation matrix, transforms.py:271-276
    float stdf;    // float32((d+1) * sqrt(2/3)), transforms.py:275
};

inline Elev make_elev() {
    Elev E;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j) {
            float left = (j >= i) ? 1.0f : 0.0f;
            if (i >= 1 && j == i - 1) left += -(float)i;
            const float prod = (float)(j + 1) * (float)(j + 2);
            const float right = 1.0f / sqrtf(prod);
            E.e[i * 3 + j] = left * right;
        }
    E.stdf = (float)(4.0 * sqrt(2.0 / 3.0));
    return E;
}

__device__ __forceinline__ int canonical(int i, int j) { return (j < 4 - i) ? j : j - 4; }

// transforms/transforms.py:300-353, same statement order as oracle hpl_keys_and_barycentric
__device__ __forceinline__ void lattice_point(float q0, float q1, float q2, int64_t n, int64_t N, float scale,
                                              const Elev &E, int32_t *__restrict__ keys, float *__restrict__ bary,
                                              float *__restrict__ emg, int64_t emg_ld, int *lo = nullptr,
                                              int *hi = nullptr) {
    const float p0 = q0 * scale, p1 = q1 * scale, p2 = q2 * scale;
    float el[4], gr[4], res[4];
    int rank[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float acc = E.e[j * 3 + 0] * p0;
        acc = fmaf(E.e[j * 3 + 1], p1, acc);
        acc = fmaf(E.e[j * 3 + 2], p2, acc);
        el[j] = acc * E.stdf;
        gr[j] = rintf(el[j] / 4.0f) * 4.0f;      // round half to even
        res[j] = el[j] - gr[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int r = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (res[k] > res[j] || (res[k] == res[j] && k < j)) r++;
        rank[j] = r;
    }
    float sum = ((gr[0] + gr[1]) + (gr[2] + gr[3])) / 4.0f;
    const int s = (int)sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (s > 0 && (float)rank[j] >= 4.0f - sum) { gr[j] -= 4.0f; rank[j] -= 4; }
        else if (s < 0 && (float)rank[j] < -sum) { gr[j] += 4.0f; rank[j] += 4; }
        rank[j] += s;
    }
    float b[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        res[j] = el[j] - gr[j];
        if (emg_ld) emg[n * emg_ld + j] = res[j];     // point-major (channel-last model input)
        else emg[(int64_t)j * N + n] = res[j];         // (4, N), the reference layout
    }
    // rank is a permutation of 0..3: resolve the dynamic index with selects (no scratch)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 5; ++t)
            if (t == 3 - rank[j]) b[t] += res[j];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 5; ++t)
            if (t == 4 - rank[j]) b[t] -= res[j];
#pragma unroll
    for (int t = 0; t < 5; ++t) b[t] /= 4.0f;
    b[0] += 1.0f + b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bary[(int64_t)t * N + n] = b[t];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int4 k;
        const int g = (int)gr[j];
        k.x = g + canonical(rank[j], 0);
        k.y = g + canonical(rank[j], 1);
        k.z = g + canonical(rank[j], 2);
        k.w = g + canonical(rank[j], 3);
        *reinterpret_cast<int4 *>(keys + ((int64_t)j * N + n) * 4) = k;
        if (lo) {      // per-coordinate key range of this point (the fused builder reduces it on the fly, transforms.py:384-385)
            lo[j] = min(lo[j], min(min(k.x, k.y), min(k.z, k.w)));
            hi[j] = max(hi[j], max(max(k.x, k.y), max(k.z, k.w)));
        }
    }
}

inline int64_t pow2_at_least(int64_t x) { int64_t p = 64; while (p < x) p <<= 1; return p; }

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}

// key2int, transforms.py:70-86 (no range check, on purpose)
__device__ __forceinline__ int64_t pack_key(const int k[4], const int32_t *__restrict__ mm) {
    int64_t res = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        res += (int64_t)k[i] - mm[i];
        res *= (int64_t)mm[4 + i + 1] - mm[i + 1] + 1;
    }
    return res + ((int64_t)k[3] - mm[3]);
}

__device__ __forceinline__ int32_t lookup(const int64_t *__restrict__ tkeys, const int32_t *__restrict__ tid,
                                          uint64_t mask, int64_t packed) {
    if (packed < 0) return -1;   // never inserted (all stored keys are >= 0)
    uint64_t s = mix64((uint64_t)packed) & mask;
    while (true) {
        const int64_t k = tkeys[s];
        if (k == packed) return tid[s];
        if (k == EMPTY) return -1;
        s = (s + 1) & mask;
    }
}

struct Offsets {
    int n;
    int v[65 * 4];   // radius <= 2
};

inline void walk(int radius, int axis, int has_zero, const int *start, Offsets &o) {   // transforms.py:112-130
    if (axis > 3) {
        for (int j = 0; j < 4; ++j) o.v[o.n * 4 + j] = start[j];
        o.n++;
        return;
    }
    int cur[4] = {start[0], start[1], start[2], start[3]};
    const int steps = (has_zero || axis < 3) ? radius + 1 : 1;
    for (int i = 0; i < steps; ++i) {
        walk(radius, axis + 1, has_zero || (i == 0), cur, o);
        for (int j = 0; j < 4; ++j) cur[j] -= 1;
        cur[axis] += 4;
    }
}

inline Offsets make_offsets(int radius) {
    Offsets o;
    o.n = 0;
    const int zero[4] = {0, 0, 0, 0};
    walk(radius, 0, 0, zero, o);
    return o;
}

}  // namespace lat
}  // namespace hpl
