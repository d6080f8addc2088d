// tools/coherence_probe2.hip -- does a workgroup see the row updates of workgroups on OTHER XCDs
// while the kernel is still running?  G workgroups (block b lands on XCD b%8) each do N sparse
// read-modify-write passes over ONE shared 3200-byte row (load, +1, store) separated by a random
// sleep, so genuine RMW races are rare (duty cycle ~1/64 per workgroup).  With memory that is
// coherent between XCDs the final value is close to G*N; if every XCD's L2 keeps a private dirty
// copy until the kernel ends, the final value is about (G/8)*N/... = one XCD's share.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int AUX, int SAUX = AUX>
__global__ void sparse_rmw(float *row, int dim, int iters, int sleep_units) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)row, 0, dim * 4, 0x27000);
  const int off = threadIdx.x * 16;
  unsigned long long s = blockIdx.x * 0x9E3779B97F4A7C15ull + 777;
  for (int i = 0; i < iters; i++) {
    s = s * 25214903917ull + 11;
    const int naps = (int)((s >> 33) % (2 * sleep_units + 1));     // wave-uniform (depends on blockIdx only)
    for (int k = 0; k < naps; k++) __builtin_amdgcn_s_sleep(127);  // ~127*64 cycles ~ 3.4 us each
    __syncthreads();
    u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX);
    t.x = __float_as_uint(__uint_as_float(t.x) + 1.f);
    t.y = __float_as_uint(__uint_as_float(t.y) + 1.f);
    t.z = __float_as_uint(__uint_as_float(t.z) + 1.f);
    t.w = __float_as_uint(__uint_as_float(t.w) + 1.f);
    __builtin_amdgcn_raw_buffer_store_b128(t, r, off, 0, SAUX);
    __syncthreads();
  }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
template <int AUX, int SAUX = AUX> int run(const char *name, float *row, int G, int N, int sleep_units) {
  const int dim = 800;
  CK(hipMemset(row, 0, dim * 4));
  CK(hipDeviceSynchronize());
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((sparse_rmw<AUX, SAUX>), dim3(G), dim3(256), 0, 0, row, dim, N, sleep_units);
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  std::vector<float> h(dim);
  CK(hipMemcpy(h.data(), row, dim * 4, hipMemcpyDeviceToHost));
  double mn = 1e30, mx = 0;
  for (float x : h) { if (x < mn) mn = x; if (x > mx) mx = x; }
  printf("%-7s G=%3d N=%4d sleep=%3d: retention min %.4f max %.4f  (%.2f ms)\n", name, G, N, sleep_units,
         mn / ((double)G * N), mx / ((double)G * N), ms);
  return 0;
}
int main() {
  float *row;
  CK(hipMalloc(&row, 800 * 4));
  for (int G : {8, 64}) {
    for (int sl : {8, 64}) {
      run<0>("plain", row, G, 400, sl);
      run<2>("nt", row, G, 400, sl);
      run<16>("sc1", row, G, 400, sl);
      run<17>("sc0sc1", row, G, 400, sl);
      run<2, 16>("nt+sc1", row, G, 400, sl);     // L1-bypassing L2-cached loads, write-through stores
      run<0, 16>("pl+sc1", row, G, 400, sl);     // plain loads, write-through stores
      run<16, 0>("sc1+pl", row, G, 400, sl);
      run<2, 0>("nt+pl", row, G, 400, sl);       // round 6: L1-bypassing loads, plain write-back stores (what an XCD's copies of the hot rows use)
    }
  }
  // fine-grained / uncached allocations with plain accesses
  float *fg = nullptr;
  if (hipExtMallocWithFlags((void **)&fg, 800 * 4, hipDeviceMallocFinegrained) == hipSuccess) {
    run<0>("fg+pln", fg, 64, 400, 8);
    run<0>("fg+pln", fg, 64, 400, 64);
  } else printf("fine-grained alloc failed\n");
  float *uc = nullptr;
  if (hipExtMallocWithFlags((void **)&uc, 800 * 4, hipDeviceMallocUncached) == hipSuccess) {
    run<0>("uc+pln", uc, 64, 400, 8);
    run<0>("uc+pln", uc, 64, 400, 64);
  } else printf("uncached alloc failed\n");
  return 0;
}
