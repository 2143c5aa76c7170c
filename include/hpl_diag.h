/*
 * hpl_diag.h -- libhplbcl_diag.so: measurement helpers beside the product library (include/hpl_bcl.h).  Nothing of the hot path
 * links or loads this library; bench.py and tools/ do.  stream: a hipStream_t as void*.  Returns 0, -1 (bad argument), -2 (HIP).
 */
#ifndef HPL_DIAG_H
#define HPL_DIAG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic: `blocks` workgroups of 4 waves each issue iters*64 v_mfma_f32_32x32x2_f32 per wave
 * with no memory traffic; out needs blocks*256 floats.  flops = blocks*4*|iters|*64*4096.
 * iters > 0: four independent accumulators per wave; iters < 0: ONE accumulator (every MFMA
 * depends on the previous one, as in the 32x32-per-wave tiles of the gather-GEMM). */
int hpl_mfma_probe(float *out, int blocks, int iters, void *stream);
/* The same instruction stream (four accumulators per wave) on operands that change with every MFMA: mode 1 = eight
 * pseudo-random register values per lane and operand, rotated; mode 2 = operands read from LDS with two ds_read_b32
 * per MFMA, as in the gather-GEMM loop.  The chip clocks to its power budget: real data toggles the multipliers and
 * lowers the sustained clock below what hpl_mfma_probe's constant operands reach.  clk (DEVICE, optional):
 * clk[0] = shader cycles, clk[1] = 100 MHz wall ticks spent by workgroup 0 in the MFMA loop. */
int hpl_mfma_probe_data(float *out, int blocks, int iters, int mode, long long *clk, void *stream);

/* A/B of the splat as scatter-adds (reference: models/bilateralNN.py:24-29 sparse_sum; BASELINE.json north star: "LDS-staged ...
 * wavefront-reduced atomicAdd") against the product's CSR segmented reduction (hpl_splat).  out[off[r][n], :] += bary[r][n] *
 * norm[off[r][n]] * feat[n, :]; out [H][ldo] is cleared by the call (counted in its time).  mode 0: global atomics per element;
 * mode 1: per-workgroup LDS open-address table of the block's vertices, one global atomic per element of an occupied slot.
 * C % 4 == 0.  The sums depend on the order the atomics land in: a measurement, not a product path. */
int hpl_diag_splat_atomic(const float *feat, int64_t ldf, int C, const float *bary, const int32_t *off, int64_t N,
                          const float *norm, int64_t H, float *out, int64_t ldo, int mode, void *stream);

/* A chain of `steps` small dependent steps (workgroup b reads the `words` floats workgroup b + 1 wrote in the step before, adds 1,
 * writes its own block; grid <= 256 workgroups of 256 threads, one per CU) issued as mode 0: `steps` kernel launches; mode 1: ONE
 * persistent launch with a monotonic-counter grid barrier between steps; mode 2: the same with an XCD-hierarchical barrier.
 * a, b: grid * words floats each (a initialised by the caller); bar: 320 words of DEVICE scratch (modes 1, 2).  After the call
 * the result of step steps - 1 is in (steps odd ? b : a): every element = its start value + steps.  tools/bench_chain.py. */
int hpl_diag_chain(float *a, float *b, int grid, int words, int steps, unsigned *bar, int mode, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HPL_DIAG_H */
