// tools/mfma_probe.hip -- what the f32 matrix pipe of an MI355X sustains in practice: a register-only loop of
// v_mfma_f32_32x32x2_f32 (the instruction of k_eval_scores_mfma) on 1/2/4 independent accumulator tiles per
// wavefront, 1 or 2 wavefronts per SIMD, every CU busy.  Prints TFLOP/s against the 157.3 TFLOP/s nominal peak
// (256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz) and the shader clock the run really had (s_memtime ticks / wall time),
// so that the evaluator's roofline fraction can be read against the clock the chip holds under matrix load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); exit(1); } } while (0)

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(int iters, float seed, float *sink, unsigned long long *ticks) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
  // operands: constants (seed >= 0) or eight different pseudo-random values per lane (seed < 0) -- the data a real
  // kernel feeds toggles far more bits than a constant does, and the clock the chip holds depends on it
  float a[8], b[8];
  unsigned h = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    h = h * 1664525u + 1013904223u;
    a[r] = seed >= 0.f ? seed + threadIdx.x * 1e-6f : ((int)(h >> 8) - (1 << 23)) * (1.f / (1 << 23));
    h = h * 1664525u + 1013904223u;
    b[r] = seed >= 0.f ? seed * 0.5f : ((int)(h >> 8) - (1 << 23)) * (1.f / (1 << 23));
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[(r + i) & 7], acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) s += acc[i][e];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int NACC>
static void run(int wgs_per_cu, int cus, int iters, float *sink, unsigned long long *ticks, float seed = 1.0f) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = cus * wgs_per_cu;
  hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, iters / 10, seed, sink, ticks);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, iters, seed, sink, ticks);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long t = 0;
  CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
  const double flop = (double)grid * 4 * iters * 8.0 * NACC * (32.0 * 32 * 2 * 2);
  const double cyc_per_mfma = (double)t / ((double)iters * 8 * NACC);
  printf("%s acc tiles/wave %d  waves/SIMD %d : %8.2f TFLOP/s = %.3f of 157.3 | %.1f ms | counter ticks per MFMA %.2f, ticks/us %.1f\n",
         seed >= 0.f ? "constant operands" : "random operands  ", NACC, wgs_per_cu, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3, ms, cyc_per_mfma, (double)t / (ms * 1e3));
}

int main(int argc, char **argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 20000;
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("%s: %d CUs, clockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  float *sink; unsigned long long *ticks;
  CK(hipMalloc(&sink, 64)); CK(hipMalloc(&ticks, 64));
  for (int w = 1; w <= 2; w++) {
    run<1>(w, p.multiProcessorCount, iters, sink, ticks);
    run<2>(w, p.multiProcessorCount, iters, sink, ticks);
    run<4>(w, p.multiProcessorCount, iters, sink, ticks);
  }
  // long run: does the clock sag under sustained matrix load?
  run<4>(2, p.multiProcessorCount, iters * 8, sink, ticks);
  run<4>(3, p.multiProcessorCount, iters, sink, ticks);
  run<4>(4, p.multiProcessorCount, iters, sink, ticks);
  // the same with operands that toggle
  run<4>(1, p.multiProcessorCount, iters, sink, ticks, -1.f);
  run<4>(2, p.multiProcessorCount, iters, sink, ticks, -1.f);
  run<4>(2, p.multiProcessorCount, iters * 8, sink, ticks, -1.f);
  return 0;
}
