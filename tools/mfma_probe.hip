// tools/mfma_probe.hip -- what the f32 matrix pipe of an MI355X sustains in practice: a register-only loop of
// v_mfma_f32_32x32x2_f32 (the instruction of k_eval_scores_mfma) on 1/2/4 independent accumulator tiles per
// wavefront, 1 or 2 wavefronts per SIMD, every CU busy.  Prints TFLOP/s against the 157.3 TFLOP/s nominal peak
// (256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz) and the shader clock the run really had (s_memtime ticks / wall time),
// so that the evaluator's roofline fraction can be read against the clock the chip holds under matrix load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); exit(1); } } while (0)

// FILL ordinary vector instructions (FKIND 0: v_add_u32 on private registers; 1: ds_read_b32; 2: ds_write_b32) after every
// MFMA: what do the non-matrix instructions of a real loop cost while the matrix pipe is saturated?
template <int NACC, int FILL, int FKIND, int EVERY = 1>
__global__ void __launch_bounds__(256) k_mfma_fill(int iters, float seed, float *sink, unsigned long long *ticks, const float *gmem) {
  __shared__ float lds[4096];
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
  float a = seed + threadIdx.x * 1e-6f, b = seed * 0.5f;
  unsigned f[4] = {threadIdx.x, threadIdx.x + 1, threadIdx.x + 2, threadIdx.x + 3};
  float lv[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 l4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  f32x2 l2[2] = {{0.f, 0.f}, {0.f, 0.f}};
  const float *gp = gmem + (size_t)blockIdx.x * 4096 + threadIdx.x * 4;
  const float *gbase = gmem + (size_t)blockIdx.x * 4096;
  const unsigned goff = threadIdx.x * 16;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)gbase, 0, 16384, 0x27000);
  const unsigned m0v = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * 1024);
  lds[threadIdx.x] = seed;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
      for (int i = 0; i < NACC; i++) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < (((r * NACC + i) % EVERY == 0) ? FILL : 0); k++) {
          const unsigned la = ((threadIdx.x + (r * NACC + i) * 64 + k * 256) & 1023) * 4;       // LDS byte address, dword access
          const unsigned la4 = ((threadIdx.x + (r * NACC + i) * 16 + k * 64) & 255) * 16;          // 16-byte access
          if (FKIND == 0) asm volatile("v_add_u32 %0, %0, 1" : "+v"(f[k & 3]));
          if (FKIND == 1) asm volatile("ds_read_b32 %0, %1" : "=v"(lv[k & 3]) : "v"(la));
          if (FKIND == 2) asm volatile("ds_write_b32 %0, %1" : : "v"(la), "v"(a) : "memory");
          if (FKIND == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(l4[k & 1]) : "v"(la4));
          if (FKIND == 4) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(l4[k & 1]) : "v"(gp + ((r * NACC + i) & 3) * 256));
          if (FKIND == 6) asm volatile("ds_read2_b32 %0, %1 offset1:32" : "=v"(l2[k & 1]) : "v"(la));
          if (FKIND == 7) asm volatile("ds_write2_b32 %0, %1, %2 offset1:32" : : "v"(la), "v"(a), "v"(b) : "memory");
          if (FKIND == 8) asm volatile("ds_write_b128 %0, %1" : : "v"(la4), "v"(l4[0]) : "memory");
          if (FKIND == 9) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(l4[k & 1]) : "v"(goff + ((r * NACC + i) & 3) * 1024), "s"(gbase));
          if (FKIND == 10) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(l4[k & 1]) : "v"(goff + ((r * NACC + i) & 3) * 1024), "s"(rsrc));
          if (FKIND == 11) asm volatile("global_load_dword %0, %1, off" : "=v"(lv[k & 3]) : "v"(gp + ((r * NACC + i) & 3) * 256));
          if (FKIND == 5) {
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0v), "v"(gp + ((r * NACC + i) & 3) * 256) : "memory");
          }
        }
      }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = lv[0] + lv[1] + lv[2] + lv[3] + (float)(f[0] + f[1] + f[2] + f[3]) + l4[0][0] + l4[1][3] + l2[0][0] + l2[1][1];
#pragma unroll
  for (int i = 0; i < NACC; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) s += acc[i][e];
  if (s == 12345.678f) sink[0] = s + lds[threadIdx.x ^ 5];
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int FILL, int FKIND, int EVERY = 1>
static void run_fill(int wgs_per_cu, int cus, int iters, float *sink, unsigned long long *ticks, const float *gmem) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = cus * wgs_per_cu;
  hipLaunchKernelGGL((k_mfma_fill<4, FILL, FKIND, EVERY>), dim3(grid), dim3(256), 0, 0, iters / 10, 1.0f, sink, ticks, gmem);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_mfma_fill<4, FILL, FKIND, EVERY>), dim3(grid), dim3(256), 0, 0, iters, 1.0f, sink, ticks, gmem);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = (double)grid * 4 * iters * 8.0 * 4 * (32.0 * 32 * 2 * 2);
  static const char *kind[] = {"v_add_u32", "ds_read_b32", "ds_write_b32", "ds_read_b128", "global_load_dwordx4", "global_load_lds_dwordx4", "ds_read2_b32", "ds_write2_b32", "ds_write_b128", "global_load_dwordx4 saddr", "buffer_load_dwordx4 offen", "global_load_dword"};
  printf("waves/SIMD %d, %d x %-26s per %d MFMA : %8.2f TFLOP/s = %.3f of 157.3\n", wgs_per_cu, FILL, kind[FKIND], EVERY, flop / ms * 1e-9,
         flop / ms * 1e-9 / 157.3);
}

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
  float *gmem;
  CK(hipMalloc(&gmem, (size_t)p.multiProcessorCount * 4 * 4096 * 4 + 65536));
  CK(hipMemset(gmem, 0, (size_t)p.multiProcessorCount * 4 * 4096 * 4 + 65536));
  for (int w = 1; w <= 3; w += 2) {
    run_fill<1, 0>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<2, 0>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<4, 0>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<8, 0>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<1, 1>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<2, 1>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<1, 2>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<2, 2>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<1, 3>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<1, 6>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<1, 7>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<1, 8>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<1, 4, 8>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<2, 4, 8>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<1, 5, 8>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<1, 9, 8>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<1, 10, 8>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<1, 11, 8>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<1, 4, 32>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
    run_fill<2, 5, 8>(w, p.multiProcessorCount, iters / 2, sink, ticks, gmem);
  }
  // the same with operands that toggle
  run<4>(1, p.multiProcessorCount, iters, sink, ticks, -1.f);
  run<4>(2, p.multiProcessorCount, iters, sink, ticks, -1.f);
  run<4>(2, p.multiProcessorCount, iters * 8, sink, ticks, -1.f);
  return 0;
}
