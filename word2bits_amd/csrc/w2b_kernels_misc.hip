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
  typedef float f4 __attribute__((ext_vector_type(4)));
  const long long stride = (long long)gridDim.x * blockDim.x;
  const bool aligned = ((((size_t)u) | ((size_t)v) | ((size_t)out)) & 15) == 0;
  const long long n4 = aligned ? (n >> 2) : 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f4 a = reinterpret_cast<const f4 *>(u)[i], b = reinterpret_cast<const f4 *>(v)[i];
    f4 o;
    o.x = quant<QM>(a.x + b.x, qp); o.y = quant<QM>(a.y + b.y, qp);
    o.z = quant<QM>(a.z + b.z, qp); o.w = quant<QM>(a.w + b.w, qp);
    reinterpret_cast<f4 *>(out)[i] = o;
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = quant<QM>(u[i] + v[i], qp);
}

// Bit-packed form of the same values (SURVEY 8 f2 "optional bit-packed output"; include/word2bits_corpus.h has the
// layout): one wavefront per block of 64 columns of a row, a coalesced 256-byte read of each table, the bits of the 64
// quantized values gathered with ballots -- bitlevel 1: one 64-bit word of signs; bitlevel 2: the signs, then the
// magnitudes (1 = the outer level).  Columns beyond the row contribute zero bits.
template <int QM>
__global__ void k_export_packed(const float *__restrict__ u, const float *__restrict__ v, unsigned long long *__restrict__ out,
                                long long rows, int dim, int blocks_per_row) {
  const QParam qp{QM, 1, 1.f};
  const int lane = (int)threadIdx.x & 63;
  const long long stride = (long long)gridDim.x * (blockDim.x >> 6), total = rows * blocks_per_row;
  for (long long w = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); w < total; w += stride) {   // wave-uniform
    const long long row = w / blocks_per_row;
    const int col = (int)(w % blocks_per_row) * 64 + lane;
    const bool in = col < dim;
    float q = 1.f;
    if (in) q = quant<QM>(u[row * dim + col] + v[row * dim + col], qp);
    const unsigned long long sign = __ballot(in && q < 0.f);
    if (QM == 1) {
      if (lane == 0) out[w] = sign;
    } else {
      const unsigned long long mag = __ballot(in && __builtin_fabsf(q) > .5f);
      if (lane == 0) { out[2 * w] = sign; out[2 * w + 1] = mag; }
    }
  }
}

// ---- replica exchange (w2b_trainer.cpp, "multi-GPU"): one CHUNK of [u || v] at a time, 16 bytes per lane, on the
// exchange streams WHILE the training kernels keep updating the same rows.  The model is therefore read and written at
// agent scope (sc1 buffer accesses, like the training kernels' own row accesses); base / d / s belong to the exchange.
typedef float w2b_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ w2b_f4 xchg_ld_sc1(__amdgpu_buffer_rsrc_t r, long long i) {
  const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(i * 16), 0, 16);
  w2b_f4 o;
  o.x = __uint_as_float(t.x); o.y = __uint_as_float(t.y); o.z = __uint_as_float(t.z); o.w = __uint_as_float(t.w);
  return o;
}
__device__ __forceinline__ void xchg_st_sc1(__amdgpu_buffer_rsrc_t r, long long i, const w2b_f4 &v) {
  u32x4 t;
  t.x = __float_as_uint(v.x); t.y = __float_as_uint(v.y); t.z = __float_as_uint(v.z); t.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(t, r, (int)(i * 16), 0, 16);
}
// d = s = w - base: what this replica has added to the chunk since the last exchange (n = floats, a multiple of 4,
// n * 4 < 2^31; every pointer 16-byte aligned).  s is what the collective then sums in place over all replicas.
__global__ void k_xchg_delta(float *w, const float *__restrict__ base, float *__restrict__ d, float *__restrict__ s, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x, n4 = n >> 2;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, (int)(n * 4), 0x27000);
  const w2b_f4 *b4 = reinterpret_cast<const w2b_f4 *>(base);
  w2b_f4 *d4 = reinterpret_cast<w2b_f4 *>(d), *s4 = reinterpret_cast<w2b_f4 *>(s);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const w2b_f4 x = xchg_ld_sc1(rw, i) - b4[i];
    d4[i] = x;
    s4[i] = x;
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {   // (n % 4 floats)
    const float x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, (int)(i * 4), 0, 16)) - base[i];
    d[i] = x;
    s[i] = x;
  }
}
// s now holds the sum over all replicas.  What the OTHER replicas added, comb - d, goes on top of the rows as they are NOW --
// whatever this replica has trained since the delta was taken stays -- and base becomes the common state base + comb.  Elements
// nobody else touched are not written.  comb:
//   fac == nullptr:  a * s                      (a = 1: delta-sum, a = 1 / replicas: average of the deltas)
//   fac != nullptr:  per ROW of [u || v] a factor on the summed delta (k_xchg_factor below: 1 for a row that only one replica
//                    changed, towards 1 / contributors for a row that every replica has saturated): safe = a * fac[row] * s;
//   ... and CELL (round 6, mode 2's default): per ELEMENT, where the whole sum a * s lands in the same quantization cell as the
//                    safe step -- the same sign at one bit; quantize(base + a s) == quantize(base + safe), ref :73-108 -- the
//                    whole sum is taken.  The forward values (all any dot product ever sees, ref :439,:464) are then exactly the
//                    safe rule's, while the fp32 master keeps the inertia that ONE shared model would have accumulated from the
//                    same updates: with the safe step alone the masters of frequent rows grow c times too slowly and their signs
//                    flip c times too easily (8 replicas: -7.6 % of the single replica's epoch loss; with the cells: -2.9 % at
//                    131 K words per replica between exchanges, -0.5 % on the literal configs[1] stream at 1 M).
// first = index of w[0] in [u || v].  base is identical on every replica and so is s: every replica computes the same comb.
template <int QM, bool CELL>
__global__ void k_xchg_apply(float *w, float *__restrict__ base, const float *__restrict__ d, const float *__restrict__ s,
                             float a, long long n, const float *__restrict__ fac, long long first, int dim, QParam qp) {
  const long long stride = (long long)gridDim.x * blockDim.x, n4 = n >> 2;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, (int)(n * 4), 0x27000);
  w2b_f4 *b4 = reinterpret_cast<w2b_f4 *>(base);
  const w2b_f4 *d4 = reinterpret_cast<const w2b_f4 *>(d), *s4 = reinterpret_cast<const w2b_f4 *>(s);
  auto comb = [&](long long i, float sum, float b) -> float {      // i: float index inside this chunk
    const float big = sum * a;
    if (!fac) return big;
    const float safe = sum * (a * fac[(first + i) / dim]);
    if (!CELL) return safe;
    return __float_as_uint(quant<QM>(b + safe, qp)) == __float_as_uint(quant<QM>(b + big, qp)) ? big : safe;
  };
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const w2b_f4 sum = s4[i], b = b4[i];
    w2b_f4 c;
    c.x = comb(4 * i, sum.x, b.x); c.y = comb(4 * i + 1, sum.y, b.y); c.z = comb(4 * i + 2, sum.z, b.z); c.w = comb(4 * i + 3, sum.w, b.w);
    const w2b_f4 others = c - d4[i];
    b4[i] = b + c;
    if (others.x != 0.f || others.y != 0.f || others.z != 0.f || others.w != 0.f) xchg_st_sc1(rw, i, xchg_ld_sc1(rw, i) + others);
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float c = comb(i, s[i], base[i]), others = c - d[i];
    base[i] += c;
    if (others != 0.f) {
      const float x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, (int)(i * 4), 0, 16)) + others;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), rw, (int)(i * 4), 0, 16);
    }
  }
}
// The combination rule of mode 2, per ROW of [u || v]: cnt[g] (in) = number of replicas that changed row g since the last
// exchange, (out) = the factor k on the SUM of their deltas (w2b_trainer.cpp "the combination rule of mode 2" has what was measured).
//   rules 0 and 2 (round 6; 0 = the default, with the per-element cells of k_xchg_apply on top; 2 = without) -- exponential
//     saturation.  A row that has received n updates in a replica since the last exchange
//     has contracted towards where those updates pull it by rho = 1 - exp(-n / tau); c replicas' updates applied one after the
//     other -- what the reference's threads do to one shared row, ref :489-491,:500-502 -- would have contracted it by
//     1 - (1 - rho)^c, and the sum of the c deltas is c * rho, so
//         k = (1 - exp(-c n / tau)) / (c (1 - exp(-n / tau)))       -> 1 for n << tau (the sum), -> 1 / c for n >> tau (the mean)
//     with n = rate[g] * words (expected updates of the row per trained centre word, from the word counts, times the centre
//     words a replica has trained since the last exchange) and tau = tau_u for rows of u, tau_v for rows of v.
//   rule 1 (rounds 4-5) -- hard threshold: k = 1 / c for rows 1..sat_u of u / 1..sat_v of v (n >= 32), 1 otherwise.  On a
//     Zipf vocabulary the rows within a factor of a few of ANY threshold carry the same share of all updates whatever the
//     interval, and on one side of it they move c times too far.
__global__ void k_xchg_factor(float *__restrict__ cnt, const float *__restrict__ rate, float words, float tau_u, float tau_v,
                              long long V, int rule, int sat_u, int sat_v) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < 2 * V; g += stride) {
    const float c = cnt[g];
    float k = 1.f;
    if (c > 1.f) {
      const bool is_v = g >= V;
      if (rule == 1) {
        const long long r = is_v ? g - V : g;
        if (r >= 1 && r <= (is_v ? sat_v : sat_u)) k = 1.f / c;
      } else if (rate) {
        const float x = rate[g] * words / (is_v ? tau_v : tau_u);
        if (x > 1e-6f) k = expm1f(-c * x) / (c * expm1f(-x));
        k = fminf(1.f, fmaxf(k, 1.f / c));
      }
    }
    cnt[g] = k;
  }
}
// cnt[r] = 1 if row r of [u || v] differs from base (this replica has trained it since the last exchange), else 0.
// One wavefront per row at a time.
__global__ void k_xchg_touched(const float *w, const float *__restrict__ base, float *__restrict__ cnt, long long rows, int dim) {
  const int lane = threadIdx.x & 63;
  const long long nw = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < rows; r += nw) {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)(w + r * dim), 0, dim * 4, 0x27000);
    bool diff = false;
    for (int c = lane; c < dim; c += 64)
      diff = diff || __builtin_amdgcn_raw_buffer_load_b32(rw, c * 4, 0, 16) != __float_as_uint(base[r * dim + c]);
    const bool any = __ballot(diff) != 0ull;
    if (lane == 0) cnt[r] = any ? 1.f : 0.f;
  }
}
// ---- XCD-local copies of the hot rows (XHot in w2b_device.hpp): all eight copies of hot row k meet the master row k + 1,
// with the merge rule of xhot_merge_row (the copy itself where the master still holds the copy's entry, else a step of
// weight xhot_w towards the copy), XCD after XCD; afterwards master == every copy == every entry.  One workgroup per hot
// row (u rows first), 16 bytes per thread.  Runs before and after every training launch: idempotent, and a master that
// was changed in between (w2b_set_model, a replica exchange) is simply adopted.
__global__ void k_xhot_fold(const W2bParams P) {
  const int nu = P.xhot_u, nv = P.xhot_v, dim = P.dim;
  const int r = blockIdx.x;
  const bool is_u = r < nu;
  const int k = is_u ? r : r - nu;
  // the master row is read and written at agent scope (sc1), like every other access to it: a replica exchange may be
  // applying the other replicas' contribution to the same row on its own stream (k_xchg_apply), and a plain access could
  // work on a stale L2 line of this XCD
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void *)((is_u ? P.u : P.v) + (long long)(k + 1) * dim), 0, dim * 4, 0x27000);
  const long long per_xcd = 2ll * (nu + nv) * dim + (long long)(nu + nv) * W2B_MAXW,      // copies, entries, merge locks
                  copy_off = (long long)(is_u ? k : nu + k) * dim, entry_off = copy_off + (long long)(nu + nv) * dim;
  if (threadIdx.x < W2B_MAXW)                      // (no merge is in progress between launches)
    for (int x = 0; x < W2B_NXCD; x++)
      reinterpret_cast<unsigned *>(P.xhot + x * per_xcd + 2ll * (nu + nv) * dim)[(is_u ? k : nu + k) * W2B_MAXW + threadIdx.x] = 0u;
  for (int c = threadIdx.x; c < dim / 4; c += blockDim.x) {
    w2b_f4 m = xchg_ld_sc1(rm, c);
    for (int x = 0; x < W2B_NXCD; x++) {
      const w2b_f4 cv = reinterpret_cast<const w2b_f4 *>(P.xhot + x * per_xcd + copy_off)[c];
      const w2b_f4 ev = reinterpret_cast<const w2b_f4 *>(P.xhot + x * per_xcd + entry_off)[c];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const bool ce = __float_as_uint(cv[i]) == __float_as_uint(ev[i]);
        const bool me = __float_as_uint(m[i]) == __float_as_uint(ev[i]);
        m[i] = me ? cv[i] : (ce ? m[i] : m[i] + P.xhot_w * (cv[i] - m[i]));
      }
    }
    xchg_st_sc1(rm, c, m);
    for (int x = 0; x < W2B_NXCD; x++) {
      reinterpret_cast<w2b_f4 *>(P.xhot + x * per_xcd + copy_off)[c] = m;
      reinterpret_cast<w2b_f4 *>(P.xhot + x * per_xcd + entry_off)[c] = m;
    }
  }
}
// progress counters of the replicas (alpha schedule on the GLOBAL word count, ref :391; see W2bShared)
__global__ void k_wca_pack(const W2bShared *sh, unsigned long long *buf) { buf[0] = sh->word_count_actual; }
__global__ void k_wca_unpack(W2bShared *sh, const unsigned long long *buf) {
  sh->wca_others = buf[1] - buf[0];       // buf[1] = sum over all replicas
  sh->wca_at_sync = buf[0];
}

}  // namespace

// ------------------------------------------------------------------------------------ launchers
int w2b_block_threads(int dim, int *vec_out, int *wide_out) {
  int vec = (dim % 4 == 0) ? 4 : 1;
  const int cols = dim / vec;
  int threads = ((cols + 63) / 64) * 64;
  int wide = 0;
  if (threads > 1024) {       // more columns than a workgroup has threads: every thread owns several 4-byte columns
    threads = 1024;
    vec = 1;
    wide = 1;
  }
  if (vec_out) *vec_out = vec;
  if (wide_out) *wide_out = wide;
  return threads;
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

// bitlevel 1 or 2 only (the caller checks); out: rows x ceil(dim / 64) x bitlevel words
hipError_t w2b_launch_export_packed(const float *u, const float *v, unsigned long long *out, long long rows, int dim,
                                    int bitlevel, hipStream_t s) {
  const int bpr = (dim + 63) / 64;
  if (bitlevel == 1) hipLaunchKernelGGL((k_export_packed<1>), dim3(2048), dim3(256), 0, s, u, v, out, rows, dim, bpr);
  else hipLaunchKernelGGL((k_export_packed<2>), dim3(2048), dim3(256), 0, s, u, v, out, rows, dim, bpr);
  return hipGetLastError();
}

hipError_t w2b_launch_xhot_fold(const W2bParams &p, hipStream_t s) {
  if (!p.xhot || p.xhot_u + p.xhot_v <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_xhot_fold, dim3(p.xhot_u + p.xhot_v), dim3(256), 0, s, p);
  return hipGetLastError();
}
hipError_t w2b_launch_wca_pack(const W2bShared *sh, unsigned long long *buf, hipStream_t s) {
  hipLaunchKernelGGL(k_wca_pack, dim3(1), dim3(1), 0, s, sh, buf);
  return hipGetLastError();
}
hipError_t w2b_launch_wca_unpack(W2bShared *sh, const unsigned long long *buf, hipStream_t s) {
  hipLaunchKernelGGL(k_wca_unpack, dim3(1), dim3(1), 0, s, sh, buf);
  return hipGetLastError();
}
hipError_t w2b_launch_xchg_delta(float *w, const float *base, float *d, float *s_, long long n, hipStream_t s) {
  hipLaunchKernelGGL(k_xchg_delta, dim3(1024), dim3(256), 0, s, w, base, d, s_, n);
  return hipGetLastError();
}
hipError_t w2b_launch_xchg_apply(float *w, float *base, const float *d, const float *s_, float a, long long n,
                                 const float *fac, long long first, int dim, int bitlevel, int cells, hipStream_t s) {
  QParam qp;
  qp.bitlevel = bitlevel;
  qp.steps_i = (bitlevel >= 4) ? (1 << (bitlevel - 1)) : 1;
  qp.steps_f = (float)qp.steps_i;
  if (!fac || !cells || bitlevel == 0) {      // (no quantization: every value is a cell of its own -- the safe step)
    hipLaunchKernelGGL((k_xchg_apply<0, false>), dim3(1024), dim3(256), 0, s, w, base, d, s_, a, n, fac, first, dim, qp);
    return hipGetLastError();
  }
  return dispatch_q(bitlevel, [&](auto qm) -> hipError_t {
    constexpr int QM = decltype(qm)::value;
    hipLaunchKernelGGL((k_xchg_apply<QM, true>), dim3(1024), dim3(256), 0, s, w, base, d, s_, a, n, fac, first, dim, qp);
    return hipGetLastError();
  });
}
hipError_t w2b_launch_xchg_factor(float *cnt, const float *rate, float words, float tau_u, float tau_v, long long V, int rule,
                                  int sat_u, int sat_v, hipStream_t s) {
  hipLaunchKernelGGL(k_xchg_factor, dim3(512), dim3(256), 0, s, cnt, rate, words, tau_u, tau_v, V, rule, sat_u, sat_v);
  return hipGetLastError();
}
hipError_t w2b_launch_xchg_touched(const float *w, const float *base, float *cnt, long long rows, int dim, hipStream_t s) {
  hipLaunchKernelGGL(k_xchg_touched, dim3(2048), dim3(256), 0, s, w, base, cnt, rows, dim);
  return hipGetLastError();
}
