// w2b_kernels_misc.hip -- InitNet, export, replica-sync elementwise kernels, launch geometry helpers.
#include "w2b_device.hpp"

namespace {
// ------------------------------------------------------------------------------------ small kernels
// InitNet (ref :343-361): the low 16 bits of the LCG have period 65536, so the init values are a
// 65536-entry periodic pattern; v is filled first, then u.
__global__ void k_init_net(float *u, float *v, long long n, const float *__restrict__ lut) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    v[i] = lut[i & 65535];
    u[i] = lut[(n + i) & 65535];
  }
}

// save loop value quantize(u+v) (ref :549-550,568-569)
template <int QM>
__global__ void k_export(const float *__restrict__ u, const float *__restrict__ v, float *__restrict__ out,
                         long long n, QParam qp) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = quant<QM>(u[i] + v[i], qp);
}

__global__ void k_sub(float *w, const float *__restrict__ base, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) w[i] -= base[i];
}
__global__ void k_add_snap(float *w, float *base, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float x = w[i] + base[i];
    w[i] = x;
    base[i] = x;
  }
}
__global__ void k_scale_snap(float *w, float *base, float s, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float x = w[i] * s;
    w[i] = x;
    if (base) base[i] = x;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------ launchers
// ------------------------------------------------------------------------------------ launchers
int w2b_block_threads(int dim, int *vec_out) {
  const int vec = (dim % 4 == 0) ? 4 : 1;
  const int cols = dim / vec;
  const int threads = ((cols + 63) / 64) * 64;
  if (vec_out) *vec_out = vec;
  return threads;   // caller rejects > 1024
}

size_t w2b_lds_bytes(int dim, int window, int negative, bool worker_form, bool exact) {
  const int maxc = (2 * window + 1 + 3) & ~3, maxt = (negative + 1 + 3) & ~3;
  int vec;
  const int threads = w2b_block_threads(dim, &vec);
  size_t ints = (size_t)W2B_STASH * threads * vec + 2 * W2B_T * W2B_MAXW + 2 * maxc + 3 * maxt;
  if (worker_form) ints += ((W2B_MAX_SEN + 3) & ~3) + 4 + (sizeof(WorkerLds) + 3) / 4 + 4;
  else ints += 4;
  if (exact) ints += (size_t)W2B_T * (W2B_EXACT_COLS + 1) + 4;
  return ints * 4;
}

// grid == 0: as many workgroups as are resident at once (occupancy query for the exact

hipError_t w2b_launch_init_net(float *u, float *v, long long n, const float *lut, hipStream_t s) {
  hipLaunchKernelGGL(k_init_net, dim3(2048), dim3(256), 0, s, u, v, n, lut);
  return hipGetLastError();
}

hipError_t w2b_launch_export(const float *u, const float *v, float *out, long long n, int bitlevel,
                             hipStream_t s) {
  QParam qp;
  qp.bitlevel = bitlevel;
  qp.steps_i = (bitlevel >= 4) ? (1 << (bitlevel - 1)) : 1;
  qp.steps_f = (float)qp.steps_i;
  return dispatch_q(bitlevel, [&](auto qm) -> hipError_t {
    constexpr int QM = decltype(qm)::value;
    hipLaunchKernelGGL((k_export<QM>), dim3(2048), dim3(256), 0, s, u, v, out, n, qp);
    return hipGetLastError();
  });
}

hipError_t w2b_launch_sub(float *w, const float *base, long long n, hipStream_t s) {
  hipLaunchKernelGGL(k_sub, dim3(2048), dim3(256), 0, s, w, base, n);
  return hipGetLastError();
}
hipError_t w2b_launch_add_snap(float *w, float *base, long long n, hipStream_t s) {
  hipLaunchKernelGGL(k_add_snap, dim3(2048), dim3(256), 0, s, w, base, n);
  return hipGetLastError();
}
hipError_t w2b_launch_scale_snap(float *w, float *base, float sc, long long n, hipStream_t s) {
  hipLaunchKernelGGL(k_scale_snap, dim3(2048), dim3(256), 0, s, w, base, sc, n);
  return hipGetLastError();
}

