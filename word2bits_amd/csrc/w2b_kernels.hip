// w2b_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for the Word2Bits training hot path.
//
// Reference math: TrainModelThread, ref src/word2bits.cpp:426-503 (phases A/B/C of one centre word),
// quantize ref :73-108, sigmoid table ref :473-475, LCG ref :428,455, sentence reader ref :394-413.
//
// Decomposition (MI355X-first, not a translation of the CPU loops):
//   * one WORKGROUP per centre word, one THREAD per 16-byte column of an embedding row
//     (dim=800 -> 200 active lanes of a 256-thread workgroup).  A wavefront's load of a row is one
//     contiguous, fully coalesced 1 KiB segment; every thread issues the loads of up to 8 context
//     rows and W2B_T target rows back-to-back, so one workgroup keeps tens of 3.2 KB rows in
//     flight -- the kernel is a pure HBM gather/scatter stream (0.6 flop/byte, no MFMA).
//   * per-thread partial dot products are reduced with wavefront shuffles, then across the
//     wavefronts through LDS; lane i of every wavefront computes the gradient scalar g of target i
//     (sigmoid-table lookup) and v_readlane broadcasts it.
//   * because a thread owns the same column of every row, the context sum (ref :439-441), the
//     error accumulation (ref :486-488) and the duplicate-row updates (ref :494-503) are executed
//     in exactly the reference's order per element; only the dot product f (ref :464-466) is
//     re-associated (tree instead of serial chain) -- this is the one source of fp32 deviation.
//   * duplicate target rows inside one centre word are serialised by cutting the chunk at the
//     duplicate (the later occurrence re-reads the row the earlier one wrote, as the CPU does).
//   * quantisation is on READ (straight-through): masters stay fp32 (SURVEY finding 3).
//
// Compiled with -ffp-contract=off so that a*b+c is two roundings, as in the bit-reference build.
#include "w2b_internal.h"
#include <type_traits>

#ifndef W2B_T
#define W2B_T 9    // target rows kept in registers per chunk (negative=24 -> 25 targets = 9 + 9 + 7)
#endif
#ifndef W2B_CA
#define W2B_CA 8   // context rows loaded per sub-chunk
#endif
#ifndef W2B_STASH
#define W2B_STASH 8 // context rows whose raw fp32 columns stay in LDS between phase A and phase C
#endif
#ifndef W2B_MINWAVES
#define W2B_MINWAVES 4  // waves per SIMD the 256-thread kernels are register-allocated for (= workgroups per CU)
#endif
#ifndef W2B_MEMMODE
#define W2B_MEMMODE 0   // cache policy of row accesses: 0 default (L2 write-back), 1 sc1 = agent scope
                        // (coherent between the 8 XCD L2s), 2 nontemporal
#endif

namespace {

// ------------------------------------------------------------------------------------ quantizer
// QM: 0 identity, 1 one bit, 2 two bits, 3 generic run-time bitlevel (>=3)
struct QParam { int bitlevel; int steps_i; float steps_f; };

template <int QM>
__device__ __forceinline__ float quant(float x, const QParam &q) {
  if (QM == 0) return x;
  const float sgn = (x < 0.f) ? -1.f : 1.f;          // +0, -0, NaN -> +1 (ref :80)
  if (QM == 1) return sgn / 3.f;                      // ref :85-87
  const float mag = x * sgn;
  if (QM == 2) {                                       // ref :91-94
    const float lvl = (mag >= 0.f && mag <= .5f) ? .25f : .75f;
    return sgn * lvl;
  }
  float lvl = 0.f;                                     // bitlevel 3 falls through to +-0
  if (q.bitlevel >= 4) {                               // ref :99-104
    int k = (int)(mag * q.steps_f + .5f);              // v_cvt saturates where x86 yields INT_MIN
    k = k > q.steps_i ? q.steps_i : k;
    lvl = (float)k / q.steps_f;
  }
  return sgn * lvl;
}

// ------------------------------------------------------------------------------------ row access
template <int VEC> struct Col { float e[VEC]; };

// buffer ops with an explicit cache policy: aux bit4 = sc1 (agent scope), bit1 = nt
// Rows are addressed through a buffer resource whose base is the (workgroup-uniform) row start and
// whose size is the row length: the row base lives in SGPRs, every lane contributes one 32-bit
// offset, and lanes beyond the row are dropped by the hardware bounds check.
#if W2B_MEMMODE == 0
#define W2B_AUX 0
#elif W2B_MEMMODE == 1
#define W2B_AUX 16
#else
#define W2B_AUX 2
#endif
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int VEC>
__device__ __forceinline__ Col<VEC> load_col(const float *tab, long long row, int dim, int col0) {
  Col<VEC> c;
  const float *rowp = tab + __builtin_amdgcn_readfirstlane((int)row) * (long long)dim;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)rowp, 0, dim * 4, 0x27000);
  if (VEC == 4) {
    u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, col0 * 4, 0, W2B_AUX);
    c.e[0] = __uint_as_float(t.x); c.e[1 % VEC] = __uint_as_float(t.y);
    c.e[2 % VEC] = __uint_as_float(t.z); c.e[3 % VEC] = __uint_as_float(t.w);
  } else {
    c.e[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, col0 * 4, 0, W2B_AUX));
  }
  return c;
}
template <int VEC>
__device__ __forceinline__ void store_col(float *tab, long long row, int dim, int col0, const Col<VEC> &c) {
  float *rowp = tab + __builtin_amdgcn_readfirstlane((int)row) * (long long)dim;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)rowp, 0, dim * 4, 0x27000);
  if (VEC == 4) {
    u32x4 t;
    t.x = __float_as_uint(c.e[0]); t.y = __float_as_uint(c.e[1 % VEC]);
    t.z = __float_as_uint(c.e[2 % VEC]); t.w = __float_as_uint(c.e[3 % VEC]);
    __builtin_amdgcn_raw_buffer_store_b128(t, r, col0 * 4, 0, W2B_AUX);
  } else {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(c.e[0]), r, col0 * 4, 0, W2B_AUX);
  }
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
__device__ __forceinline__ double wave_sum_d(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
// LDS traffic between lanes of ONE wavefront is executed in order by the hardware; this only stops
// the compiler from moving LDS accesses across the point.
#define W2B_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
__device__ __forceinline__ unsigned long long lane_lt_mask(int lane) {
  return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

// ------------------------------------------------------------------------------------ LDS carving
// All regions are 4-byte typed; the carve keeps the float region 16-byte aligned.
struct WordLds {
  int *ctx;     // [maxc]  context rows of u, window order (ref :431-436)
  int *umult;   // [maxc]  multiplicity at the first occurrence of a row, 0 at later duplicates
  int *tgt;     // [maxt]  target rows of v: [0] = centre word (label 1), then kept negatives (label 0)
  int *prev;    // [maxt]  index of the previous occurrence of the same target row, or -1
  float *red;   // [2][W2B_T][W2B_MAXW] cross-wave partial dot products (double buffered)
  float *stash; // [W2B_STASH][blockDim][VEC] raw u columns of the first context rows, private to the
                // owning thread: phase C updates them without a second trip to memory
};

__device__ __forceinline__ int round4(int x) { return (x + 3) & ~3; }

__device__ __forceinline__ WordLds carve_word_lds(int *base, int window, int negative, int vec) {
  const int maxc = round4(2 * window + 1), maxt = round4(negative + 1);
  WordLds L;
  L.stash = reinterpret_cast<float *>(base);
  base += W2B_STASH * blockDim.x * vec;
  L.red = reinterpret_cast<float *>(base);
  int *p = base + 2 * W2B_T * W2B_MAXW;
  L.ctx = p; p += maxc;
  L.umult = p; p += maxc;
  L.tgt = p; p += maxt;
  L.prev = p; p += maxt;
  return L;
}

// ------------------------------------------------------------------------------------ one centre word
// Preconditions: L.ctx[0..cw), L.tgt[0..nt) published by a __syncthreads(); cw >= 1, nt >= 1.
// Ends with a __syncthreads() (lists may be overwritten afterwards).
template <int QM, int VEC, bool LOSS>
__device__ __forceinline__ void process_word(const W2bParams &P, const WordLds &L, const QParam &qp,
                                             const int cw, const int nt, const float alpha,
                                             double &loss_acc) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int dim = P.dim, col0 = tid * VEC;
  const bool active = col0 < dim;
  const float ar2 = (2.f * alpha) * P.reg;                      // 2*alpha*reg (ref :490,:501)

  // ---- duplicate bookkeeping (tiny, O(n^2) over <= 2*window / negative+1 entries)
  for (int i = tid; i < nt; i += blockDim.x) {
    const int me = L.tgt[i];
    int pd = -1;
    for (int j = 0; j < i; j++) pd = (L.tgt[j] == me) ? j : pd;
    L.prev[i] = pd;
  }
  for (int i = tid; i < cw; i += blockDim.x) {
    const int me = L.ctx[i];
    bool first = true;
    for (int j = 0; j < i; j++) first = first && (L.ctx[j] != me);
    int mult = 0;
    if (first) for (int j = i; j < cw; j++) mult += (L.ctx[j] == me);
    L.umult[i] = mult;
  }
  __syncthreads();

  auto chunk_end = [&](int start) {
    int end = start + 1;
    while (end < nt && end - start < W2B_T && L.prev[end] < start) end++;
    return end;
  };

  Col<VEC> x[W2B_T];
  int start = 0, end = chunk_end(0);
  // issue the first chunk of target-row loads before the context phase so both gathers overlap
#pragma unroll
  for (int i = 0; i < W2B_T; i++) {
#pragma unroll
    for (int e = 0; e < VEC; e++) x[i].e[e] = 0.f;
    if (active && start + i < end) x[i] = load_col<VEC>(P.v, L.tgt[start + i], dim, col0);
  }

  // ---- phase A: context_avg = (1/cw) * sum_j quantize(u[ctx_j])   (ref :431-449)
  Col<VEC> avg;
  float regsq = 0.f;
#pragma unroll
  for (int e = 0; e < VEC; e++) avg.e[e] = 0.f;
  for (int j0 = 0; j0 < cw; j0 += W2B_CA) {
    Col<VEC> r[W2B_CA];
#pragma unroll
    for (int jj = 0; jj < W2B_CA; jj++)
      if (active && j0 + jj < cw) r[jj] = load_col<VEC>(P.u, L.ctx[j0 + jj], dim, col0);
#pragma unroll
    for (int jj = 0; jj < W2B_CA; jj++)
      if (active && j0 + jj < cw) {
        if (j0 + jj < W2B_STASH) {
#pragma unroll
          for (int e = 0; e < VEC; e++) L.stash[((j0 + jj) * blockDim.x + tid) * VEC + e] = r[jj].e[e];
        }
#pragma unroll
        for (int e = 0; e < VEC; e++) {
          const float q = quant<QM>(r[jj].e[e], qp);
          avg.e[e] += q;
          if (LOSS) regsq += q * q;
        }
      }
  }
  {
    const float cwf = (float)cw;
#pragma unroll
    for (int e = 0; e < VEC; e++) avg.e[e] = active ? avg.e[e] / cwf : 0.f;   // ref :449
  }

  // ---- phase B: targets (ref :450-492)
  Col<VEC> err;
#pragma unroll
  for (int e = 0; e < VEC; e++) err.e[e] = 0.f;
  int par = 0;
  for (;;) {
    const int n = end - start;
    float p[W2B_T], p2[W2B_T];
#pragma unroll
    for (int i = 0; i < W2B_T; i++) {
      float s = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        const float q = quant<QM>(x[i].e[e], qp);
        s += avg.e[e] * q;                                      // ref :466 (re-associated)
        if (LOSS) s2 += q * q;
      }
      p[i] = active ? s : 0.f;
      p2[i] = active ? s2 : 0.f;
    }
    float *red = L.red + par * (W2B_T * W2B_MAXW);
#pragma unroll
    for (int i = 0; i < W2B_T; i++) {
      if (i < n) {
        const float s = wave_sum(p[i]);
        if (lane == 0) red[i * W2B_MAXW + wave] = s;
      }
    }
    __syncthreads();
    // lane i of every wavefront: f_i, then g_i (ref :473-475)
    float gl = 0.f;
    if (lane < n) {
      float f = 0.f;
      for (int w = 0; w < nwaves; w++) f += red[lane * W2B_MAXW + w];
      const float label = (start + lane == 0) ? 1.f : 0.f;      // target 0 is the centre word
      float g;
      if (f > 6.f) g = (label - 1.f) * alpha;
      else if (f < -6.f) g = label * alpha;
      else g = (label - P.exp_table[(int)((f + 6.f) * 83.f)]) * alpha;
      gl = g;
      if (LOSS && wave == 0) {                                  // ref :480-483
        const float dp = (label != 0.f) ? f : -f;
        float sg;
        if (dp > 6.f) sg = 1.f;
        else if (dp < -6.f) sg = 1e-9f;
        else sg = 1.f / (1.f + expf(-dp));
        loss_acc += (double)logf(sg);
      }
    }
    if (LOSS && P.reg != 0.f) {                                 // reg * sum q^2 of every target row
#pragma unroll
      for (int i = 0; i < W2B_T; i++)
        if (i < n) {
          const float s2 = wave_sum(p2[i]);
          if (lane == 0) loss_acc -= (double)(P.reg * s2);
        }
    }
    // error accumulation + row update, in target order (ref :486-491)
#pragma unroll
    for (int i = 0; i < W2B_T; i++) {
      if (i < n) {
        const float g = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gl), i));
        if (active) {
#pragma unroll
          for (int e = 0; e < VEC; e++) {
            const float xv = x[i].e[e];
            err.e[e] += g * quant<QM>(xv, qp);
            x[i].e[e] = xv + (g * avg.e[e] - ar2 * xv);
          }
          store_col<VEC>(P.v, L.tgt[start + i], dim, col0, x[i]);
        }
      }
    }
    start = end;
    if (start >= nt) break;
    end = chunk_end(start);
    par ^= 1;
#pragma unroll
    for (int i = 0; i < W2B_T; i++)
      if (active && start + i < end) x[i] = load_col<VEC>(P.v, L.tgt[start + i], dim, col0);
  }

  // ---- phase C: u[ctx_j] += context_avge - 2*alpha*reg*u[ctx_j]   (ref :494-503)
  for (int j0 = 0; j0 < cw; j0 += W2B_CA) {
    Col<VEC> r[W2B_CA];
#pragma unroll
    for (int jj = 0; jj < W2B_CA; jj++)
      if (active && j0 + jj < cw && L.umult[j0 + jj] > 0) {
        if (j0 + jj < W2B_STASH) {
#pragma unroll
          for (int e = 0; e < VEC; e++) r[jj].e[e] = L.stash[((j0 + jj) * blockDim.x + tid) * VEC + e];
        } else {
          r[jj] = load_col<VEC>(P.u, L.ctx[j0 + jj], dim, col0);
        }
      }
#pragma unroll
    for (int jj = 0; jj < W2B_CA; jj++)
      if (active && j0 + jj < cw) {
        const int m = L.umult[j0 + jj];
        if (m > 0) {
          for (int k = 0; k < m; k++) {      // a row that occurs m times in the window is updated m times
#pragma unroll
            for (int e = 0; e < VEC; e++) r[jj].e[e] = r[jj].e[e] + (err.e[e] - ar2 * r[jj].e[e]);
          }
          store_col<VEC>(P.u, L.ctx[j0 + jj], dim, col0, r[jj]);
        }
      }
  }
  if (LOSS && P.reg != 0.f) {
    const float s = wave_sum(regsq);
    if (lane == 0) loss_acc -= (double)(P.reg * s);             // ref :437-445 (summed over the window)
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------ form (ii): tuples
template <int QM, int VEC, bool LOSS, int MAXTHREADS>
__global__ void __launch_bounds__(MAXTHREADS, (MAXTHREADS <= 256 ? W2B_MINWAVES : 1)) k_train_tuples(const W2bParams P, const long long n,
                                                      const int32_t *__restrict__ center,
                                                      const int32_t *__restrict__ ctx_off,
                                                      const int32_t *__restrict__ ctx,
                                                      const int32_t *__restrict__ neg,
                                                      const float alpha) {
  extern __shared__ int smem[];
  const WordLds L = carve_word_lds(smem, P.window, P.negative, VEC);
  int *s_cnt = L.prev + round4(P.negative + 1);   // [0] cw, [1] nt
  const int tid = threadIdx.x, lane = tid & 63;
  QParam qp;
  qp.bitlevel = P.bitlevel;
  qp.steps_i = (P.bitlevel >= 4) ? (1 << (P.bitlevel - 1)) : 1;
  qp.steps_f = (float)qp.steps_i;
  double loss_acc = 0.0;
  const int K = P.negative;
  for (long long i = blockIdx.x; i < n; i += gridDim.x) {
    if (tid < 64) {
      const int c0 = ctx_off[i], cw = ctx_off[i + 1] - c0;
      for (int j = lane; j < cw; j += 64) L.ctx[j] = ctx[c0 + j];
      const int word = center[i];
      int cnt = 0;
      for (int d0 = 0; d0 < K; d0 += 64) {
        const int d = d0 + lane;
        const int t = (d < K) ? neg[i * K + d] : -1;
        const bool keep = (t >= 0) && (t != word);             // skipped draw, ref :458
        const unsigned long long m = __ballot(keep);
        if (keep) L.tgt[1 + cnt + __popcll(m & lane_lt_mask(lane))] = t;
        cnt += __popcll(m);
      }
      if (lane == 0) { L.tgt[0] = word; s_cnt[0] = cw; s_cnt[1] = 1 + cnt; }
    }
    __syncthreads();
    const int cw = s_cnt[0], nt = s_cnt[1];
    if (cw > 0) process_word<QM, VEC, LOSS>(P, L, qp, cw, nt, alpha, loss_acc);
    else __syncthreads();
  }
  if (LOSS) {
    if (tid < 64) {
      const double s = wave_sum_d(loss_acc);
      if (lane == 0) atomicAdd(&P.shared->loss_tuples, s);
    } else if ((tid & 63) == 0 && loss_acc != 0.0) {
      atomicAdd(&P.shared->loss_tuples, loss_acc);   // reg terms booked by lane 0 of the other waves
    }
  }
}

// ------------------------------------------------------------------------------------ form (i): workers
struct WorkerLds {            // scalars of one worker, owned by wavefront 0
  unsigned long long rng;
  long long cursor, wc, last_wc;
  int sen_len, sen_pos, override_, eof, done, cw, nt, pad;
  float alpha;
};

__device__ __forceinline__ unsigned long long lcg_jump(const W2bParams &P, unsigned long long x, int k) {
  return P.jump_a[k] * x + P.jump_c[k];
}

// The sentence reader of ref :394-413, executed by wavefront 0 (64 tokens per trip).
// All scalars are wave-uniform.
__device__ __forceinline__ void read_sentence(const W2bParams &P, int *s_sen, unsigned long long &rng,
                                              long long &cursor, long long &wc, int &ovr, int &eof,
                                              int &len_out, const int lane) {
  int len = 0;
  bool stop = false;
  const bool sub = (P.sample > 0.f);
  if (ovr != -2) {                       // truncated first word of the shard (mid-word fseek, ref :377)
    const int w = ovr;
    ovr = -2;
    if (w != -1) {
      wc++;
      if (w == 0) stop = true;
      else {
        bool kept = true;
        if (sub) {
          rng = rng * W2B_LCG_A + W2B_LCG_C;
          kept = !(P.keep[w] < (float)(rng & 0xFFFF) / 65536.f);
        }
        if (kept) { if (lane == 0) s_sen[0] = w; len = 1; }
      }
    }
  }
  while (!stop) {
    const long long i = cursor + lane;
    const bool in = i < P.n_tokens;
    const int tok = in ? P.corpus[i] : 0;
    const bool isw = in && tok != 0;
    const unsigned long long mw = __ballot(isw);
    const unsigned long long lt = lane_lt_mask(lane);
    bool kept = isw;
    if (sub && isw) {
      const unsigned long long x = lcg_jump(P, rng, __popcll(mw & lt) + 1);
      kept = !(P.keep[tok] < (float)(x & 0xFFFF) / 65536.f);    // ref :403-406
    }
    const unsigned long long mk = __ballot(kept);
    const int kpos = __popcll(mk & lt);
    const bool lim = kept && (len + kpos + 1 >= W2B_MAX_SEN);    // ref :410
    const unsigned long long mt = __ballot(!in || (in && tok == 0) || lim);
    const unsigned long long min_ = __ballot(in);
    const int e = mt ? (__ffsll((long long)mt) - 1) : 64;
    const int ncons = e + ((e < 64 && ((min_ >> e) & 1ull)) ? 1 : 0);
    const unsigned long long cmask = (ncons >= 64) ? ~0ull : ((1ull << ncons) - 1ull);
    if (kept && lane < ncons) s_sen[len + kpos] = tok;
    len += __popcll(mk & cmask);
    wc += ncons;
    cursor += ncons;
    if (sub) rng = lcg_jump(P, rng, __popcll(mw & cmask));
    if (e < 64) {
      stop = true;
      if (!((min_ >> e) & 1ull)) eof = 1;
    }
  }
  len_out = len;
}

template <int QM, int VEC, bool LOSS, int MAXTHREADS>
__global__ void __launch_bounds__(MAXTHREADS, (MAXTHREADS <= 256 ? W2B_MINWAVES : 1)) k_train_workers(const W2bParams P, const long long max_positions) {
  extern __shared__ int smem[];
  const WordLds L = carve_word_lds(smem, P.window, P.negative, VEC);
  int *s_sen = L.prev + round4(P.negative + 1);
  WorkerLds *S = reinterpret_cast<WorkerLds *>(s_sen + round4(W2B_MAX_SEN) + 4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wid = blockIdx.x;
  if (wid >= P.num_threads) return;
  W2bWorker *G = P.workers + wid;
  if (G->done) return;
  QParam qp;
  qp.bitlevel = P.bitlevel;
  qp.steps_i = (P.bitlevel >= 4) ? (1 << (P.bitlevel - 1)) : 1;
  qp.steps_f = (float)qp.steps_i;
  // restore the worker
  for (int i = tid; i < G->sen_len; i += blockDim.x) s_sen[i] = G->sen[i];
  if (tid == 0) {
    S->rng = G->rng; S->cursor = G->cursor; S->wc = G->word_count; S->last_wc = G->last_word_count;
    S->sen_len = G->sen_len; S->sen_pos = G->sen_pos; S->override_ = G->first_override;
    S->eof = 0; S->done = 0; S->cw = 0; S->nt = 0; S->alpha = 0.f;
  }
  __syncthreads();
  double loss_acc = 0.0;
  const int W = P.window, K = P.negative;
  for (long long it = 0; it < max_positions; ++it) {
    if (wave == 0) {
      unsigned long long rng = S->rng;
      long long cursor = S->cursor, wc = S->wc, last_wc = S->last_wc;
      int sen_len = S->sen_len, sen_pos = S->sen_pos, ovr = S->override_, eof = S->eof;
      int done = 0, cw = 0, nt = 0;
      float alpha = 0.f;
      if (wc - last_wc > 10000) {                                    // ref :379-393
        if (lane == 0) {
          const unsigned long long d = (unsigned long long)(wc - last_wc);
          const unsigned long long wca = atomicAdd(&P.shared->word_count_actual, d) + d;
          // other replicas are assumed to progress at the same pace (exact for a single replica)
          const long long wca_all = (long long)wca * (P.total_threads / P.num_threads);
          float a = P.starting_alpha * (1.f - (float)wca_all / (float)(P.iter * P.train_words + 1));
          if ((double)a < (double)P.starting_alpha * 0.0001) a = (float)((double)P.starting_alpha * 0.0001);
          __hip_atomic_store(&P.shared->alpha, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        last_wc = wc;
      }
      if (sen_len == 0) {                                            // ref :394-413
        read_sentence(P, s_sen, rng, cursor, wc, ovr, eof, sen_len, lane);
        sen_pos = 0;
        W2B_WAVE_SYNC();
      }
      if (eof || wc > P.train_words / P.total_threads) {              // ref :414-423 (local_iter == 1)
        if (lane == 0)
          atomicAdd(&P.shared->word_count_actual, (unsigned long long)(wc - last_wc));
        last_wc = wc;
        done = 1;
      } else {
        const int word = (sen_len > 0) ? s_sen[sen_pos] : 0;          // ref :424
        rng = rng * W2B_LCG_A + W2B_LCG_C;                            // ref :428-429
        const int b = (int)(rng % (unsigned long long)W);
        const int hi = 2 * W + 1 - b;
        for (int a0 = b; a0 < hi; a0 += 64) {                         // ref :431-436
          const int a = a0 + lane;
          const int c = sen_pos - W + a;
          const bool ok = (a < hi) && (a != W) && (c >= 0) && (c < sen_len);
          const unsigned long long m = __ballot(ok);
          if (ok) L.ctx[cw + __popcll(m & lane_lt_mask(lane))] = s_sen[c];
          cw += __popcll(m);
        }
        if (cw > 0) {                                                 // ref :450-460
          int cnt = 0;
          for (int d0 = 1; d0 <= K; d0 += 64) {
            const int d = d0 + lane;
            bool keep = false;
            int t = 0;
            if (d <= K) {
              const unsigned long long x = lcg_jump(P, rng, d);
              t = P.table[(x >> 16) % (unsigned long long)P.table_size];
              if (t == 0) t = (int)(x % (unsigned long long)(P.vocab_size - 1)) + 1;
              keep = (t != word);
            }
            const unsigned long long m = __ballot(keep);
            if (keep) L.tgt[1 + cnt + __popcll(m & lane_lt_mask(lane))] = t;
            cnt += __popcll(m);
          }
          if (lane == 0) L.tgt[0] = word;
          nt = 1 + cnt;
          rng = lcg_jump(P, rng, K);
          alpha = __hip_atomic_load(&P.shared->alpha, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        sen_pos++;                                                    // ref :505-509
        if (sen_pos >= sen_len) sen_len = 0;
      }
      if (lane == 0) {
        S->rng = rng; S->cursor = cursor; S->wc = wc; S->last_wc = last_wc;
        S->sen_len = sen_len; S->sen_pos = sen_pos; S->override_ = ovr; S->eof = eof;
        S->done = done; S->cw = cw; S->nt = nt; S->alpha = alpha;
      }
    }
    __syncthreads();
    if (S->done) break;
    const int cw = S->cw, nt = S->nt;
    const float alpha = S->alpha;
    if (cw > 0) process_word<QM, VEC, LOSS>(P, L, qp, cw, nt, alpha, loss_acc);
    else __syncthreads();
  }
  // save the worker
  __syncthreads();
  const int sl = S->sen_len;
  for (int i = tid; i < sl; i += blockDim.x) G->sen[i] = s_sen[i];
  double lsum = 0.0;
  if (LOSS) {
    // wave 0 holds the log-sigmoid terms on its lanes; lane 0 of every wave holds reg terms
    if (wave == 0) lsum = wave_sum_d(loss_acc);
    else if (lane == 0 && loss_acc != 0.0) atomicAdd(&G->loss, loss_acc);
  }
  if (tid == 0) {
    G->rng = S->rng; G->cursor = S->cursor; G->word_count = S->wc; G->last_word_count = S->last_wc;
    G->sen_len = S->sen_len; G->sen_pos = S->sen_pos; G->first_override = S->override_;
    if (LOSS) atomicAdd(&G->loss, lsum);
    if (S->done) { G->done = 1; atomicAdd(&P.shared->workers_done, 1); }
  }
}

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

template <typename F>
hipError_t dispatch_q(int bitlevel, F &&f) {
  switch (bitlevel) {
    case 0: return f(std::integral_constant<int, 0>());
    case 1: return f(std::integral_constant<int, 1>());
    case 2: return f(std::integral_constant<int, 2>());
    default: return f(std::integral_constant<int, 3>());
  }
}

}  // namespace

// ------------------------------------------------------------------------------------ launchers
int w2b_block_threads(int dim, int *vec_out) {
  const int vec = (dim % 4 == 0) ? 4 : 1;
  const int cols = dim / vec;
  const int threads = ((cols + 63) / 64) * 64;
  if (vec_out) *vec_out = vec;
  return threads;   // caller rejects > 1024
}

size_t w2b_lds_bytes(int dim, int window, int negative, bool worker_form) {
  const int maxc = (2 * window + 1 + 3) & ~3, maxt = (negative + 1 + 3) & ~3;
  int vec;
  const int threads = w2b_block_threads(dim, &vec);
  size_t ints = (size_t)W2B_STASH * threads * vec + 2 * W2B_T * W2B_MAXW + 2 * maxc + 2 * maxt;
  if (worker_form) ints += ((W2B_MAX_SEN + 3) & ~3) + 4 + (sizeof(WorkerLds) + 3) / 4 + 4;
  else ints += 4;
  return ints * 4;
}

// grid == 0: as many workgroups as are resident at once (occupancy query for the exact
// instantiation), so the grid-stride loop has no tail of late-starting workgroups.
template <typename KernelT>
static int auto_grid(KernelT kernel, int threads, size_t lds, int num_cus, int per_cu_override, long long n) {
  int nb = 0;
  if (per_cu_override > 0) nb = per_cu_override;
  else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, lds) != hipSuccess || nb < 1) nb = 2;
  long long g = (long long)num_cus * nb;
  if (g > n) g = n;
  return (int)(g < 1 ? 1 : g);
}

hipError_t w2b_launch_tuples(const W2bParams &p, long long n, const int32_t *center, const int32_t *ctx_off,
                             const int32_t *ctx, const int32_t *neg, float alpha, int grid, int num_cus,
                             int per_cu_override, bool loss, hipStream_t s) {
  int vec;
  const int threads = w2b_block_threads(p.dim, &vec);
  const size_t lds = w2b_lds_bytes(p.dim, p.window, p.negative, false);
  return dispatch_q(p.bitlevel, [&](auto qm) -> hipError_t {
    constexpr int QM = decltype(qm)::value;
#define W2B_LAUNCH_T2(VEC, LOSS, MAXT)                                                                      \
    do {                                                                                                     \
      auto kern = k_train_tuples<QM, VEC, LOSS, MAXT>;                                                       \
      const int g = grid > 0 ? grid : auto_grid(kern, threads, lds, num_cus, per_cu_override, n);            \
      hipLaunchKernelGGL(kern, dim3(g), dim3(threads), lds, s, p, n, center, ctx_off, ctx, neg, alpha);      \
    } while (0)
#define W2B_LAUNCH_T(VEC, LOSS) \
    do { if (threads <= 256) W2B_LAUNCH_T2(VEC, LOSS, 256); else W2B_LAUNCH_T2(VEC, LOSS, 1024); } while (0)
    if (vec == 4) { if (loss) W2B_LAUNCH_T(4, true); else W2B_LAUNCH_T(4, false); }
    else { if (loss) W2B_LAUNCH_T(1, true); else W2B_LAUNCH_T(1, false); }
#undef W2B_LAUNCH_T
#undef W2B_LAUNCH_T2
    return hipGetLastError();
  });
}

hipError_t w2b_launch_workers(const W2bParams &p, long long max_positions, bool loss, hipStream_t s) {
  int vec;
  const int threads = w2b_block_threads(p.dim, &vec);
  const size_t lds = w2b_lds_bytes(p.dim, p.window, p.negative, true);
  return dispatch_q(p.bitlevel, [&](auto qm) -> hipError_t {
    constexpr int QM = decltype(qm)::value;
#define W2B_LAUNCH_W(VEC, LOSS) \
    do { if (threads <= 256) hipLaunchKernelGGL((k_train_workers<QM, VEC, LOSS, 256>), dim3(p.num_threads), dim3(threads), lds, s, p, max_positions); \
         else hipLaunchKernelGGL((k_train_workers<QM, VEC, LOSS, 1024>), dim3(p.num_threads), dim3(threads), lds, s, p, max_positions); } while (0)
    if (vec == 4) { if (loss) W2B_LAUNCH_W(4, true); else W2B_LAUNCH_W(4, false); }
    else { if (loss) W2B_LAUNCH_W(1, true); else W2B_LAUNCH_W(1, false); }
#undef W2B_LAUNCH_W
    return hipGetLastError();
  });
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
