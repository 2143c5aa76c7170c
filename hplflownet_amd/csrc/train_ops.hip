// train_ops.hip -- the small HBM-bound pieces of the native training step (include/hpl_bcl.h "Training"): the gradient of
// a broadcast residual (hpl_psum), the row regrouping behind the displacement filter's data gradient (hpl_regroup), the
// EPE3D loss with its gradient (hpl_epe3d), and the strided zero / add / vector-copy helpers csrc/executor.hip issues.
// Reference: autograd over models/bnn_flow.py:189-208 (expand + cat of the pc1 half, Conv2d((15,1)) displacement filter),
// models/epe3d_loss.py:9-10, main.py:213-214.  All of them move each byte once; none is worth more than full lines.
#include "common.h"
#include <math.h>

using namespace hpl;

namespace {

// out[h, n] (+)= sum_j X[(j*mod + h), n]: one lane per (h, 4 columns), the j blocks in ascending order (deterministic)
template <typename V>
__global__ void __launch_bounds__(256) k_psum(const float *__restrict__ X, int64_t ldx, int periods, int64_t mod, int NV,
                                              float *__restrict__ out, int64_t ldo, int accumulate) {
    constexpr int VW = sizeof(V) / 4;
    const int64_t total = mod * NV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t h = i / NV;
        const int c = (int)(i - h * NV) * VW;
        float acc[VW];
#pragma unroll
        for (int u = 0; u < VW; ++u) acc[u] = 0.f;
        for (int j = 0; j < periods; ++j) {
            const V v = *reinterpret_cast<const V *>(X + ((int64_t)j * mod + h) * ldx + c);
            const float *e = reinterpret_cast<const float *>(&v);
#pragma unroll
            for (int u = 0; u < VW; ++u) acc[u] += e[u];
        }
        float *d = out + h * ldo + c;
#pragma unroll
        for (int u = 0; u < VW; ++u) d[u] = accumulate ? d[u] + acc[u] : acc[u];
    }
}

// out[(f*M + m), c] (+)= G[m, f*C + c]: consecutive lanes walk a row of G (coalesced reads, C-float runs written)
template <typename V>
__global__ void __launch_bounds__(256) k_regroup(const float *__restrict__ G, int64_t ldg, int64_t M, int F, int CV,
                                                 float *__restrict__ out, int64_t ldo, int accumulate) {
    constexpr int VW = sizeof(V) / 4;
    const int64_t row = (int64_t)F * CV, total = M * row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / row;
        const int x = (int)(i - m * row), f = x / CV, c = (x - f * CV) * VW;
        const V v = *reinterpret_cast<const V *>(G + m * ldg + (int64_t)f * CV * VW + c);
        V *d = reinterpret_cast<V *>(out + ((int64_t)f * M + m) * ldo + c);
        if (accumulate) {
            V o = *d;
            float *oe = reinterpret_cast<float *>(&o);
            const float *ve = reinterpret_cast<const float *>(&v);
#pragma unroll
            for (int u = 0; u < VW; ++u) oe[u] += ve[u];
            *d = o;
        } else {
            *d = v;
        }
    }
}

// One workgroup: thread t owns points t, t + 1024, ...; per-thread partial sums in ascending point order, then a fixed
// tree over the 1024 threads -- the loss is the same number on every run.
__global__ void __launch_bounds__(1024) k_epe3d(const float *__restrict__ pred, const float *__restrict__ sf, int64_t N,
                                                float *__restrict__ grad, float *__restrict__ loss) {
    __shared__ float part[1024];
    const float inv_n = 1.0f / (float)N;
    float acc = 0.f;
    for (int64_t n = threadIdx.x; n < N; n += 1024) {
        const float dx = pred[n * 3 + 0] - sf[n], dy = pred[n * 3 + 1] - sf[N + n], dz = pred[n * 3 + 2] - sf[2 * N + n];
        const float r = sqrtf(dx * dx + dy * dy + dz * dz);
        acc += r;
        const float s = r > 0.f ? inv_n / r : 0.f;
        if (grad) {
            grad[n * 3 + 0] = dx * s;
            grad[n * 3 + 1] = dy * s;
            grad[n * 3 + 2] = dz * s;
        }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss) *loss = part[0] * inv_n;
}

__global__ void __launch_bounds__(256) k_zero_cols(float *__restrict__ dst, int64_t ldd, int64_t rows, int cols) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols;
        dst[r * ldd + (i - r * cols)] = 0.f;
    }
}

__global__ void __launch_bounds__(256) k_add_cols(const float *__restrict__ src, int64_t lds, float *__restrict__ dst, int64_t ldd,
                                                  int64_t rows, int cols) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        dst[r * ldd + c] += src[r * lds + c];
    }
}

__global__ void __launch_bounds__(256) k_add_cols4(const float4 *__restrict__ src, int64_t lds4, float4 *__restrict__ dst,
                                                   int64_t ldd4, int64_t rows, int cols4) {
    const int64_t total = rows * cols4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols4;
        const int c = (int)(i - r * cols4);
        const float4 a = src[r * lds4 + c];
        float4 d = dst[r * ldd4 + c];
        d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
        dst[r * ldd4 + c] = d;
    }
}

__global__ void k_vcopy(const float *__restrict__ src, float *__restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

}  // namespace

namespace hpl {

int zero_cols(float *dst, int64_t ldd, int64_t rows, int cols, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return HPL_OK;
    if (ldd == cols) {
        if (hipMemsetAsync(dst, 0, (size_t)rows * cols * 4, s) != hipSuccess) { set_error("zero: hipMemsetAsync failed"); return HPL_EHIP; }
        return HPL_OK;
    }
    k_zero_cols<<<(int)imin(cdiv(rows * cols, 256), 4096), 256, 0, s>>>(dst, ldd, rows, cols);
    HPL_CHECK_LAUNCH("zero_cols");
    return HPL_OK;
}

int add_cols(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t rows, int cols, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return HPL_OK;
    if (cols % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && aligned16(src) && aligned16(dst))
        k_add_cols4<<<(int)imin(cdiv(rows * (cols / 4), 256), 4096), 256, 0, s>>>(reinterpret_cast<const float4 *>(src), lds / 4,
                                                                              reinterpret_cast<float4 *>(dst), ldd / 4, rows, cols / 4);
    else
        k_add_cols<<<(int)imin(cdiv(rows * cols, 256), 4096), 256, 0, s>>>(src, lds, dst, ldd, rows, cols);
    HPL_CHECK_LAUNCH("add_cols");
    return HPL_OK;
}

int vcopy(const float *src, float *dst, int n, hipStream_t s) {
    if (n <= 0) return HPL_OK;
    k_vcopy<<<(n + 255) / 256, 256, 0, s>>>(src, dst, n);
    HPL_CHECK_LAUNCH("vcopy");
    return HPL_OK;
}

}  // namespace hpl

namespace {
// Adam (torch.optim.Adam, weight_decay = 0, amsgrad off: main.py:138-140) over flat fp32 arrays: one pass, 16 B per element read
// and 12 written.  The arithmetic follows torch's fused kernel (ATen fused_adam_utils.cuh adam_math) operation by operation; the bias
// corrections and 1 - beta are computed on the host in double (as torch does) and arrive rounded once: step_size = lr / (1 - beta1^t),
// bc2_sqrt = sqrt(1 - beta2^t), w1 = 1 - beta1, w2 = 1 - beta2.
struct AdamK { float step_size, w1, b2, w2, eps, bc2_sqrt; };
__device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, const AdamK &k) {
    const float step_size = k.step_size, eps = k.eps, bc2_sqrt = k.bc2_sqrt;
    m = m + k.w1 * (g - m);                             // lerp(m, g, 1 - beta1)
    v = k.b2 * v + k.w2 * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * m / denom;
}
__global__ void __launch_bounds__(256) k_adam_flat(float *__restrict__ P, const float *__restrict__ G, float *__restrict__ M, float *__restrict__ V,
                                                   int64_t n, const AdamK k) {
    const int64_t n4 = n / 4, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 p = reinterpret_cast<float4 *>(P)[i], m = reinterpret_cast<float4 *>(M)[i], v = reinterpret_cast<float4 *>(V)[i];
        const float4 g = reinterpret_cast<const float4 *>(G)[i];
        adam1(p.x, g.x, m.x, v.x, k);
        adam1(p.y, g.y, m.y, v.y, k);
        adam1(p.z, g.z, m.z, v.z, k);
        adam1(p.w, g.w, m.w, v.w, k);
        reinterpret_cast<float4 *>(P)[i] = p;
        reinterpret_cast<float4 *>(M)[i] = m;
        reinterpret_cast<float4 *>(V)[i] = v;
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) adam1(P[i], G[i], M[i], V[i], k);
}
}  // namespace

extern "C" int hpl_adam_flat(float *p, const float *g, float *m, float *v, int64_t n, double lr, double beta1, double beta2, double eps,
                             int64_t step, hplStream stream) {
    HPL_REQUIRE(p && g && m && v && n >= 0 && aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v) && step >= 1 && beta1 >= 0.0 &&
                    beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0,
                "hpl_adam_flat: null / unaligned pointer or bad arguments (n=%lld step=%lld)", (long long)n, (long long)step);
    if (n == 0) return HPL_OK;
    AdamK k;
    k.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
    k.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    k.w1 = (float)(1.0 - beta1); k.b2 = (float)beta2; k.w2 = (float)(1.0 - beta2); k.eps = (float)eps;
    const int grid = (int)imin(cdiv(cdiv(n, 4), 256), 256 * 16);
    k_adam_flat<<<grid, 256, 0, to_stream(stream)>>>(p, g, m, v, n, k);
    HPL_CHECK_LAUNCH("hpl_adam_flat");
    return HPL_OK;
}

extern "C" int hpl_psum(const float *X, int64_t ldx, int64_t rows, int64_t mod, int N, float *out, int64_t ldo, int accumulate,
                        hplStream stream) {
    HPL_REQUIRE(X && out && N > 0 && mod > 0 && rows >= 0 && rows % mod == 0 && ldx >= N && ldo >= N,
                "hpl_psum: bad arguments (rows=%lld mod=%lld N=%d)", (long long)rows, (long long)mod, N);
    if (rows == 0) return HPL_OK;
    const int periods = (int)(rows / mod);
    hipStream_t s = to_stream(stream);
    const bool vec = N % 4 == 0 && ldx % 4 == 0 && aligned16(X);
    const int nv = vec ? N / 4 : N;
    const int grid = (int)imin(cdiv(mod * nv, 256), 8192);
    if (vec) k_psum<float4><<<grid, 256, 0, s>>>(X, ldx, periods, mod, nv, out, ldo, accumulate);
    else k_psum<float><<<grid, 256, 0, s>>>(X, ldx, periods, mod, nv, out, ldo, accumulate);
    HPL_CHECK_LAUNCH("hpl_psum");
    return HPL_OK;
}

extern "C" int hpl_regroup(const float *G, int64_t ldg, int64_t M, int F, int C, float *out, int64_t ldo, int accumulate,
                           hplStream stream) {
    HPL_REQUIRE(G && out && M >= 0 && F > 0 && C > 0 && ldg >= (int64_t)F * C && ldo >= C, "hpl_regroup: bad arguments");
    if (M == 0) return HPL_OK;
    hipStream_t s = to_stream(stream);
    const bool vec = C % 4 == 0 && ldg % 4 == 0 && ldo % 4 == 0 && aligned16(G) && aligned16(out);
    const int cv = vec ? C / 4 : C;
    const int grid = (int)imin(cdiv(M * F * cv, 256), 8192);
    if (vec) k_regroup<float4><<<grid, 256, 0, s>>>(G, ldg, M, F, cv, out, ldo, accumulate);
    else k_regroup<float><<<grid, 256, 0, s>>>(G, ldg, M, F, cv, out, ldo, accumulate);
    HPL_CHECK_LAUNCH("hpl_regroup");
    return HPL_OK;
}

extern "C" int hpl_epe3d(const float *pred, const float *sf, int64_t N, float *grad, float *loss, hplStream stream) {
    HPL_REQUIRE(pred && sf && N > 0 && (grad || loss), "hpl_epe3d: bad arguments");
    k_epe3d<<<1, 1024, 0, to_stream(stream)>>>(pred, sf, N, grad, loss);
    HPL_CHECK_LAUNCH("hpl_epe3d");
    return HPL_OK;
}
