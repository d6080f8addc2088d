// tools/atomic_probe.hip -- fp32 atomic adds to embedding rows on MI355X.
//  (1) COHERENCE: G workgroups (spread over all XCDs) each add 1.0f N times to every element of ONE 3200-byte row with
//      buffer_atomic_add_f32, cache policy aux = 0 and aux = 16 (sc1, agent scope).  Lossless means every element ends
//      at exactly G * N.  The round-3 advisor asked for this: the training kernels issued their row atomics with aux = 0
//      while every other coherent row access carries sc1.  A second pass interleaves sc1 loads of the same row by the
//      same workgroups (the non-atomic readers of the training kernels) and checks that they never see a value above
//      the final sum or a decreasing sequence.
//  (2) THROUGHPUT of random-row adds over a 1.28 GB table (uniform rows) and over a small hot set, in two lane layouts:
//        strided    lane l adds its own 16-byte column, element by element: 4 instructions, each touching 8 cache lines
//                   of the wavefront's 1 KiB segment with 8 dwords per line (what add_col did in round 3)
//        contiguous the wavefront's 256 deltas are transposed through LDS so that instruction e adds dwords [64 e, 64 e + 64):
//                   4 instructions, each touching 2 cache lines with 32 dwords per line
//      next to plain sc1 16-byte stores of the same rows (the non-atomic update).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); exit(1); } } while (0)

template <int AUX, bool READERS>
__global__ void __launch_bounds__(256) k_coherence(float *row, int dim, int n, unsigned *bad) {
  const int tid = threadIdx.x, col0 = tid * 4;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)row, 0, dim * 4, 0x27000);
  if (col0 >= dim) return;
  float last = 0.f;
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int e = 0; e < 4; e++) (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(1.0f, r, (col0 + e) * 4, 0, AUX);
    if (READERS && (i & 7) == 0) {
      const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, col0 * 4, 0, 16));
      if (v < last) atomicAdd(bad, 1u);       // a reader must never see the row go backwards
      last = v;
    }
  }
}

// LAYOUT 0 strided, 1 contiguous (LDS transpose), 2 = plain sc1 b128 store of the delta (baseline, not an add)
template <int LAYOUT, int AUX>
__global__ void __launch_bounds__(256) k_add_rows(float *tab, unsigned nrows, unsigned hot, int dim, int iters, int rows_per_iter) {
  __shared__ float tr[4][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col0 = tid * 4;
  const unsigned rowb = (unsigned)dim * 4u;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, (int)(nrows * rowb), 0x27000);
  unsigned long long s = (blockIdx.x + 1) * 0x9E3779B97F4A7C15ull;
  const bool act = col0 < dim;
  float d[4] = {1.f, 2.f, 3.f, 4.f};
  for (int it = 0; it < iters; it++) {
    for (int k = 0; k < rows_per_iter; k++) {
      s = s * 25214903917ull + 11;
      const unsigned row = __builtin_amdgcn_readfirstlane((unsigned)((s >> 20) % (hot ? hot : nrows)));
      const int soff = (int)(row * rowb);
      if (LAYOUT == 2) {
        u32x4 t; t.x = __float_as_uint(d[0]); t.y = __float_as_uint(d[1]); t.z = __float_as_uint(d[2]); t.w = __float_as_uint(d[3]);
        if (act) __builtin_amdgcn_raw_buffer_store_b128(t, r, col0 * 4, soff, 16);
      } else if (LAYOUT == 0) {
        if (act) {
#pragma unroll
          for (int e = 0; e < 4; e++) (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(d[e], r, (col0 + e) * 4, soff, AUX);
        }
      } else if (LAYOUT == 3) {
        // quad-transposed: instruction e, lane 4 q + j adds dword 16 q + 4 e + j -- every quad of lanes covers 16 contiguous
        // bytes, 8 dwords per cache line and instruction as in the strided layout but in two 16-byte pieces instead of eight
        // 4-byte ones (reachable with DPP quad permutes alone, no LDS)
        *reinterpret_cast<float4 *>(&tr[wave][lane * 4]) = make_float4(d[0], d[1], d[2], d[3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int w = 16 * (lane >> 2) + 4 * e + (lane & 3);
          const int c = wave * 256 + w;
          const float v = tr[wave][w];
          if (c < dim) (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, r, c * 4, soff, AUX);
        }
        __builtin_amdgcn_wave_barrier();
      } else {
        // this wavefront's 256 floats: lane l wrote [4 l, 4 l + 4); instruction e takes [64 e + l]
        *reinterpret_cast<float4 *>(&tr[wave][lane * 4]) = make_float4(d[0], d[1], d[2], d[3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int c = wave * 256 + e * 64 + lane;      // element index inside the row
          const float v = tr[wave][e * 64 + lane];
          if (c < dim) (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, r, c * 4, soff, AUX);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

template <int LAYOUT, int AUX>
static void tput(const char *name, float *tab, unsigned nrows, unsigned hot, int dim) {
  const int grid = 256 * 4, iters = 200, rpi = 8;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_add_rows<LAYOUT, AUX>), dim3(grid), dim3(256), 0, 0, tab, nrows, hot, dim, iters, rpi);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  const double rows = (double)grid * iters * rpi;
  printf("%-44s rows %-8s (%u): %8.3f ms  %7.1f M row-updates/s  %6.2f TB/s of row bytes\n", name, hot ? "hot set" : "uniform", hot, best,
         rows / best / 1e3, rows * dim * 4 / best / 1e9);
  fflush(stdout);
}

int main() {
  const int dim = 800;
  const unsigned nrows = 400000;
  float *tab; unsigned *bad;
  CK(hipMalloc(&tab, (size_t)nrows * dim * 4)); CK(hipMalloc(&bad, 4));
  // ---- (1) coherence
  const int G = 64, N = 2000;
  std::vector<float> h(dim);
  auto check = [&](const char *name) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), tab, dim * 4, hipMemcpyDeviceToHost));
    unsigned hb = 0; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    int wrong = 0; float lo = 1e30f, hi = -1e30f;
    for (int i = 0; i < dim; i++) { wrong += h[i] != (float)(G * N); lo = h[i] < lo ? h[i] : lo; hi = h[i] > hi ? h[i] : hi; }
    printf("coherence %-36s: expected %d everywhere, got [%g, %g], %d of %d elements wrong, %u backwards reads -> %s\n", name, G * N, lo, hi,
           wrong, dim, hb, wrong == 0 && hb == 0 ? "LOSSLESS" : "LOST UPDATES");
  };
  CK(hipMemset(tab, 0, dim * 4)); CK(hipMemset(bad, 0, 4));
  hipLaunchKernelGGL((k_coherence<0, false>), dim3(G), dim3(256), 0, 0, tab, dim, N, bad);
  check("aux=0 (no scope bits)");
  CK(hipMemset(tab, 0, dim * 4)); CK(hipMemset(bad, 0, 4));
  hipLaunchKernelGGL((k_coherence<16, false>), dim3(G), dim3(256), 0, 0, tab, dim, N, bad);
  check("aux=16 (sc1)");
  CK(hipMemset(tab, 0, dim * 4)); CK(hipMemset(bad, 0, 4));
  hipLaunchKernelGGL((k_coherence<0, true>), dim3(G), dim3(256), 0, 0, tab, dim, N, bad);
  check("aux=0 + sc1 readers");
  CK(hipMemset(tab, 0, dim * 4)); CK(hipMemset(bad, 0, 4));
  hipLaunchKernelGGL((k_coherence<16, true>), dim3(G), dim3(256), 0, 0, tab, dim, N, bad);
  check("aux=16 + sc1 readers");
  // ---- (2) throughput
  CK(hipMemset(tab, 0, (size_t)nrows * dim * 4));
  for (unsigned hot : {0u, 256u, 8u, 1u}) {      // uniform rows; then ever smaller sets: how many adds per second does ONE row take?
    tput<2, 16>("sc1 16-byte stores (no add; baseline)", tab, nrows, hot, dim);
    tput<0, 0>("atomic add, strided lanes, aux=0", tab, nrows, hot, dim);
    tput<0, 16>("atomic add, strided lanes, sc1", tab, nrows, hot, dim);
    tput<1, 0>("atomic add, contiguous (LDS transpose), aux=0", tab, nrows, hot, dim);
    tput<1, 16>("atomic add, contiguous (LDS transpose), sc1", tab, nrows, hot, dim);
    tput<3, 16>("atomic add, quad-transposed (16-B pieces), sc1", tab, nrows, hot, dim);
  }
  return 0;
}
