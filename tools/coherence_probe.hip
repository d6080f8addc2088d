// tools/coherence_probe.hip -- how do concurrent workgroups on DIFFERENT XCDs see each other's
// non-atomic read-modify-write updates of the same embedding row?  (Hogwild across 8 non-coherent
// L2s.)  Each of G workgroups performs N times: load row (3200 B), add 1, store row -- with
// plain / nt / sc1 / sc0sc1 accesses.  "retention" = final value / (G*N): 1.0 = no update lost.
// Also times a streaming variant for the bandwidth cost of each flavour.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int AUX>
__global__ void rmw(float *row, int dim, int iters) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)row, 0, dim * 4, 0x27000);
  const int off = threadIdx.x * 16;
  for (int i = 0; i < iters; i++) {
    u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX);
    t.x = __float_as_uint(__uint_as_float(t.x) + 1.f);
    t.y = __float_as_uint(__uint_as_float(t.y) + 1.f);
    t.z = __float_as_uint(__uint_as_float(t.z) + 1.f);
    t.w = __float_as_uint(__uint_as_float(t.w) + 1.f);
    __builtin_amdgcn_raw_buffer_store_b128(t, r, off, 0, AUX);
  }
}
// streaming gather/scatter: each WG walks pseudo-random rows of a big table, r+w each row
template <int AUX>
__global__ void stream_rows(float *tab, long long nrows, int dim, int per_wg) {
  unsigned long long s = blockIdx.x * 0x9E3779B97F4A7C15ull + 12345;
  const int off = threadIdx.x * 16;
  for (int i = 0; i < per_wg; i += 8) {
    u32x4 t[8];
    __amdgpu_buffer_rsrc_t r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      s = s * 25214903917ull + 11;
      long long row = (long long)((s >> 16) % (unsigned long long)nrows);
      r[j] = __builtin_amdgcn_make_buffer_rsrc((void *)(tab + row * dim), 0, dim * 4, 0x27000);
      t[j] = __builtin_amdgcn_raw_buffer_load_b128(r[j], off, 0, AUX);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      t[j].x = __float_as_uint(__uint_as_float(t[j].x) + 1.f);
      __builtin_amdgcn_raw_buffer_store_b128(t[j], r[j], off, 0, AUX);
    }
  }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
template <int AUX> int run(const char *name, float *row, float *tab, long long nrows) {
  const int dim = 800, G = 64, N = 2000;
  CK(hipMemset(row, 0, dim * 4));
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(rmw<AUX>, dim3(G), dim3(256), 0, 0, row, dim, N);
  CK(hipDeviceSynchronize());
  std::vector<float> h(dim);
  CK(hipMemcpy(h.data(), row, dim * 4, hipMemcpyDeviceToHost));
  double mn = 1e30, mx = 0;
  for (float x : h) { if (x < mn) mn = x; if (x > mx) mx = x; }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int wgs = 256 * 8, per = 512;
  hipLaunchKernelGGL(stream_rows<AUX>, dim3(wgs), dim3(256), 0, 0, tab, nrows, dim, per);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(stream_rows<AUX>, dim3(wgs), dim3(256), 0, 0, tab, nrows, dim, per);
  CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double bytes = (double)wgs * per * dim * 4 * 2;
  printf("%-8s retention min %.4f max %.4f (of %d)   stream r+w %.2f TB/s (%.2f ms)\n", name, mn / (G * N),
         mx / (G * N), G * N, bytes / ms / 1e9, ms);
  return 0;
}
int main() {
  float *row, *tab;
  const long long nrows = 800000;   // 2.56 GB: well past L2 (32 MB) and Infinity Cache (256 MB)
  CK(hipMalloc(&row, 800 * 4));
  CK(hipMalloc(&tab, nrows * 800 * 4));
  CK(hipMemset(tab, 0, nrows * 800 * 4));
  run<0>("plain", row, tab, nrows);
  run<2>("nt", row, tab, nrows);
  run<16>("sc1", row, tab, nrows);
  run<17>("sc0sc1", row, tab, nrows);
  run<1>("sc0", row, tab, nrows);
  return 0;
}
