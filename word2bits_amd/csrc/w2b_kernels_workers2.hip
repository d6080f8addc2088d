// w2b_kernels_workers2.hip -- form (i), SENTENCE-RESIDENT variant of the worker kernel.
//
// Consecutive sentence positions share 2*window-1 of their context rows (ref src/word2bits.cpp:431-436
// walks sen[p-window .. p+window]).  The plain worker kernel (w2b_kernels_workers.hip) reads and writes
// every context row of every position from/to HBM: 2*cw of the 2*(cw+K+1) row transfers per centre
// word.  Here a workgroup keeps the fp32 rows of the sliding window [p-R, p+R] RESIDENT IN LDS:
//   * a row enters the window once (one read) and leaves it once (one read-modify-write), however
//     many centre words use it in between;  phase A (ref :431-449) and phase C (ref :494-503) become
//     LDS traffic;
//   * a word that occurs at several window positions shares ONE slot (reference semantics: a row
//     that occurs twice is updated twice, in order);
//   * next to the value every slot keeps the fp16 sum of what this worker added to it.  When the row
//     leaves the window and nobody else changed it meanwhile (wavefront checksum of the row bits), the
//     exact fp32 value is stored -- a single worker stays bit-identical to the plain kernel; otherwise
//     the worker's delta is added to the CURRENT row (merge), so concurrent Hogwild workers do not
//     erase each other's updates;
//   * rows are thread-private columns of LDS (a thread only ever touches its own 8 bytes of every
//     slot): no barriers, no bank conflicts.
// The radius R is window when the window fits in LDS next to a second workgroup, else window-1 with
// the two outermost context rows held in registers for the step; otherwise the launcher falls back
// to the plain kernel.
#include "w2b_device.hpp"

#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>

// target rows per chunk (negative=24 -> 13 + 12).  A workgroup is up to 7 data wavefronts (one thread per 8-byte
// column) + 1 producer wavefront; two workgroups per CU put four wavefronts on every SIMD, so the kernel is held
// to 128 VGPRs (__launch_bounds__(512, 4)).
#define W2B_T2MAX 25
#define W2B_NDWMAX 8   // data wavefronts per worker (stride of the per-wavefront LDS tables)
template <int VEC> struct T2For { static constexpr int value = 13; };
#define W2B_RCH 2   // window rows moved per trip when many enter/leave at once (sentence boundaries)

namespace {

struct Win2Lds {          // scalars owned by the producer wavefront (extends WorkerLds)
  WorkerLds w;
  int clo, chi;           // sentence positions currently resident (empty when chi < clo)
};

// What the producer wavefront hands to the data wavefronts for ONE step (double buffered in LDS)
struct Step2 {
  int stop;               // 1: this pass only empties the window (epoch finished, or end of the launch)
  int cw, nt, uc_n, n_ret, n_adm;
  int next_row;           // row of the position that enters the window at the NEXT step (-1: unknown / none)
  int nck;                // number of target chunks = number of workgroup barriers inside the data phase
  float alpha;
  int pad[3];
};

// Explicit LDS address space on every pointer of the kernel's LDS record: dereferences compile to ds_*
// instructions.  (With generic pointers the two step buffers were selected through a struct reference and
// address-space inference gave up: 250 flat_load/flat_store per step, each tied to vmcnt AND lgkmcnt, so
// every window access also waited for the target rows in flight.)
#define W2B_LDS __attribute__((address_space(3)))

struct Win2 {
  W2B_LDS float *win;             // [S][dim]   current fp32 value of the resident rows
  W2B_LDS __half *dlt;            // [S][dim]   what this worker added since the row entered (fp16)
  W2B_LDS unsigned *csum;         // [S][4]     per-wavefront xor checksum of the row bits at entry
  W2B_LDS float *red;             // [2][W2B_T2MAX][4]
  W2B_LDS int *slot_row, *slot_ref, *pos_slot;          // [S]
  W2B_LDS int *ret_slot, *ret_row, *adm_slot, *adm_row; // [S+2]
  W2B_LDS int *cslot;             // [maxc] slot of every context position; -1-k = k-th register-held row
  W2B_LDS int *uc_row;            // [2]   rows of the (at most two) context positions outside the radius
  W2B_LDS int *tgt, *prev, *cend; // [maxt]
  W2B_LDS int *sen;               // [1000]
  W2B_LDS unsigned long long *ja, *jc;   // [nj] LCG jump-ahead table (copy of P.jump_a / P.jump_c)
  W2B_LDS Win2Lds *S;
  W2B_LDS Step2 *St;              // per-step scalars (this struct exists twice: one per step buffer)
};

__host__ __device__ inline int w2_round4(int x) { return (x + 3) & ~3; }

// LDS bytes of the sentence-resident kernel for radius R
__host__ __device__ inline size_t win2_lds_bytes(int dim, int window, int negative, int R) {
  const int S = 2 * R + 1, maxc = w2_round4(2 * window + 1), maxt = w2_round4(negative + 1);
  size_t b = (size_t)S * dim * 4 + (size_t)S * dim * 2;            // win + dlt
  b = (b + 15) & ~(size_t)15;
  b += (size_t)(S + 2) * W2B_NDWMAX * 4;                           // csum (window slots + 2 hot target rows)
  b += 2 * W2B_T2MAX * W2B_NDWMAX * 4;                             // red
  b += (size_t)(3 * w2_round4(S) + maxt + w2_round4(W2B_MAX_SEN)) * 4;               // slot tables, prev, sen
  b += 2 * ((size_t)(4 * w2_round4(S + 2) + maxc + 4 + 2 * maxt) * 4 + sizeof(Step2));  // step lists x 2
  b += sizeof(Win2Lds) + 16;
  b += (size_t)2 * 8 * (negative + 2 > 66 ? negative + 2 : 66) + 16;   // LCG jump tables
  return b;
}

__device__ __forceinline__ Win2 carve_win2(W2B_LDS int *base, int dim, int window, int negative, int R, int buf) {
  const int S = 2 * R + 1, maxc = w2_round4(2 * window + 1), maxt = w2_round4(negative + 1);
  Win2 L;
  W2B_LDS char *p = (W2B_LDS char *)base;
  L.win = (W2B_LDS float *)p; p += (size_t)S * dim * 4;
  L.dlt = (W2B_LDS __half *)p; p += (size_t)S * dim * 2;
  p = (W2B_LDS char *)(((unsigned)(size_t)p + 15u) & ~15u);
  L.csum = (W2B_LDS unsigned *)p; p += (size_t)(S + 2) * W2B_NDWMAX * 4;
  L.red = (W2B_LDS float *)p; p += 2 * W2B_T2MAX * W2B_NDWMAX * 4;
  W2B_LDS int *q = (W2B_LDS int *)p;
  L.slot_row = q; q += w2_round4(S);
  L.slot_ref = q; q += w2_round4(S);
  L.pos_slot = q; q += w2_round4(S);
  L.prev = q; q += maxt;
  L.sen = q; q += w2_round4(W2B_MAX_SEN);
  const int per_buf = 4 * w2_round4(S + 2) + maxc + 4 + 2 * maxt + (int)(sizeof(Step2) / 4);
  q += buf * per_buf;                                  // the per-step lists exist twice
  L.ret_slot = q; q += w2_round4(S + 2);
  L.ret_row = q; q += w2_round4(S + 2);
  L.adm_slot = q; q += w2_round4(S + 2);
  L.adm_row = q; q += w2_round4(S + 2);
  L.cslot = q; q += maxc;
  L.uc_row = q; q += 4;
  L.tgt = q; q += maxt;
  L.cend = q; q += maxt;
  L.St = (W2B_LDS Step2 *)q; q += sizeof(Step2) / 4;
  q += (1 - buf) * per_buf;
  L.S = (W2B_LDS Win2Lds *)(((unsigned)(size_t)q + 15u) & ~15u);
  const int nj = negative + 2 > 66 ? negative + 2 : 66;
  L.ja = (W2B_LDS unsigned long long *)(((unsigned)(size_t)(L.S + 1) + 15u) & ~15u);
  L.jc = L.ja + nj;
  return L;
}

template <int VEC>
__device__ __forceinline__ unsigned col_bits(const Col<VEC> &c) {
  unsigned h = 0;
#pragma unroll
  for (int e = 0; e < VEC; e++) h ^= __float_as_uint(c.e[e]) * (2u * e + 3u);
  return h;
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
// LDS accesses of a thread's own column of a window row: VEC floats (8 or 16 bytes) and VEC fp16 deltas
template <int VEC> __device__ __forceinline__ Col<VEC> lds_ld(const W2B_LDS float *p);
template <> __device__ __forceinline__ Col<4> lds_ld<4>(const W2B_LDS float *p) {
  const f32x4_t t = *(const W2B_LDS f32x4_t *)p;
  Col<4> c;
  c.e[0] = t.x; c.e[1] = t.y; c.e[2] = t.z; c.e[3] = t.w;
  return c;
}
template <> __device__ __forceinline__ Col<2> lds_ld<2>(const W2B_LDS float *p) {
  const f32x2_t t = *(const W2B_LDS f32x2_t *)p;
  Col<2> c;
  c.e[0] = t.x; c.e[1] = t.y;
  return c;
}
__device__ __forceinline__ void lds_st(W2B_LDS float *p, const Col<4> &c) {
  f32x4_t t;
  t.x = c.e[0]; t.y = c.e[1]; t.z = c.e[2]; t.w = c.e[3];
  *(W2B_LDS f32x4_t *)p = t;
}
__device__ __forceinline__ void lds_st(W2B_LDS float *p, const Col<2> &c) {
  f32x2_t t;
  t.x = c.e[0]; t.y = c.e[1];
  *(W2B_LDS f32x2_t *)p = t;
}
template <int VEC> __device__ __forceinline__ Col<VEC> lds_ldh(const W2B_LDS __half *p);
template <> __device__ __forceinline__ Col<4> lds_ldh<4>(const W2B_LDS __half *p) {
  const u32x2_t t = *(const W2B_LDS u32x2_t *)p;
  const unsigned tx = t.x, ty = t.y;
  const __half2 a = *reinterpret_cast<const __half2 *>(&tx), b = *reinterpret_cast<const __half2 *>(&ty);
  Col<4> c;
  c.e[0] = __low2float(a); c.e[1] = __high2float(a); c.e[2] = __low2float(b); c.e[3] = __high2float(b);
  return c;
}
template <> __device__ __forceinline__ Col<2> lds_ldh<2>(const W2B_LDS __half *p) {
  const unsigned raw = *(const W2B_LDS unsigned *)p;
  const __half2 a = *reinterpret_cast<const __half2 *>(&raw);
  Col<2> c;
  c.e[0] = __low2float(a); c.e[1] = __high2float(a);
  return c;
}
__device__ __forceinline__ void lds_sth(W2B_LDS __half *p, const Col<4> &c) {
  const __half2 a = __floats2half2_rn(c.e[0], c.e[1]), b = __floats2half2_rn(c.e[2], c.e[3]);
  u32x2_t t;
  t.x = *reinterpret_cast<const unsigned *>(&a);
  t.y = *reinterpret_cast<const unsigned *>(&b);
  *(W2B_LDS u32x2_t *)p = t;
}
__device__ __forceinline__ void lds_sth(W2B_LDS __half *p, const Col<2> &c) {
  const __half2 a = __floats2half2_rn(c.e[0], c.e[1]);
  *(W2B_LDS unsigned *)p = *reinterpret_cast<const unsigned *>(&a);
}

// ---- rows leaving the window: store them (exact value when untouched by others, else merge the delta)
template <int VEC, int MM>
__device__ __forceinline__ void window_retire(const W2bParams &P, const Win2 &L, int n_ret, bool active,
                                              int col0, int lane, int wave) {
  const int dim = P.dim;
  for (int i0 = 0; i0 < n_ret; i0 += W2B_RCH) {
    Col<VEC> rw[W2B_RCH], rd[W2B_RCH], g[W2B_RCH];
#pragma unroll
    for (int i = 0; i < W2B_RCH; i++)
      if (i0 + i < n_ret) {
        const int s = L.ret_slot[i0 + i];
#pragma unroll
        for (int e = 0; e < VEC; e++) { rw[i].e[e] = 0.f; rd[i].e[e] = 0.f; g[i].e[e] = 0.f; }
        if (active) {
          rw[i] = lds_ld<VEC>(L.win + s * dim + col0);
          rd[i] = lds_ldh<VEC>(L.dlt + s * dim + col0);
          g[i] = load_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.u, L.ret_row[i0 + i], dim, col0, P.tab_bytes);
        }
      }
#pragma unroll
    for (int i = 0; i < W2B_RCH; i++)
      if (i0 + i < n_ret) {
        const int s = L.ret_slot[i0 + i];
        const unsigned now = wave_xor(active ? col_bits(g[i]) : 0u);
        const bool untouched = (now == L.csum[s * W2B_NDWMAX + wave]);     // wave-uniform
        if (active) {
          Col<VEC> o;
#pragma unroll
          for (int e = 0; e < VEC; e++) o.e[e] = untouched ? rw[i].e[e] : g[i].e[e] + rd[i].e[e];
          store_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.u, L.ret_row[i0 + i], dim, col0, o, P.tab_bytes);
        }
      }
  }
}

// ---- rows entering the window
template <int VEC, int MM>
__device__ __forceinline__ void window_admit(const W2bParams &P, const Win2 &L, int n_adm, bool active,
                                             int col0, int lane, int wave) {
  const int dim = P.dim;
  for (int i0 = 0; i0 < n_adm; i0 += W2B_RCH) {
    Col<VEC> a[W2B_RCH];
#pragma unroll
    for (int i = 0; i < W2B_RCH; i++)
      if (i0 + i < n_adm) {
#pragma unroll
        for (int e = 0; e < VEC; e++) a[i].e[e] = 0.f;
        if (active) a[i] = load_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.u, L.adm_row[i0 + i], dim, col0, P.tab_bytes);
      }
#pragma unroll
    for (int i = 0; i < W2B_RCH; i++)
      if (i0 + i < n_adm) {
        const int s = L.adm_slot[i0 + i];
        const unsigned cs = wave_xor(active ? col_bits(a[i]) : 0u);
        if (lane == 0) L.csum[s * W2B_NDWMAX + wave] = cs;
        if (active) {
          Col<VEC> z;
#pragma unroll
          for (int e = 0; e < VEC; e++) z.e[e] = 0.f;
          lds_st(L.win + s * dim + col0, a[i]);
          lds_sth(L.dlt + s * dim + col0, z);
        }
      }
  }
}

// The hottest target rows (rows 1 and 2 of v: the vocabulary is sorted by frequency) live in REGISTERS of the
// worker, value + accumulated delta, and are merged with memory every W2B_HOT_PERIOD steps with the same
// exact-or-merge rule as the window rows.  Coherent accesses to one embedding row serialise at its memory
// line (about 7 M read-modify-writes per second); on Zipf-distributed ids the most frequent word alone is a
// target of 0.3 centre words in every position, which capped the whole GPU at 23 M words/s.  The producer
// places these rows at slots 0 / 1 of a chunk (prep_lists, hot_first), so only those two slots test for them.
#define W2B_HOT_PERIOD 32   // default of W2bParams::hot_period (a power of two; W2B_HOT_PERIOD in the environment overrides)
template <int VEC> struct HotV { Col<VEC> v0, v1, d0, d1; int on; };

template <int VEC, int MM>
__device__ __forceinline__ void hot_merge(const W2bParams &P, const Win2 &L, HotV<VEC> &H, int NS, bool active,
                                          int col0, int lane, int wave) {
  if (!H.on) return;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    if (k + 1 >= P.vocab_size) break;
    Col<VEC> &val = k ? H.v1 : H.v0;
    Col<VEC> &del = k ? H.d1 : H.d0;
    Col<VEC> g;
#pragma unroll
    for (int e = 0; e < VEC; e++) g.e[e] = 0.f;
    if (active) g = load_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.v, k + 1, P.dim, col0, P.tab_bytes);
    const unsigned now = wave_xor(active ? col_bits(g) : 0u);
    bool mine = false;                         // did this worker add anything to these columns since the last merge?
#pragma unroll
    for (int e = 0; e < VEC; e++) mine = mine || (del.e[e] != 0.f);
    if (__ballot(active && mine) == 0ull) {
      // nothing of ours to publish: adopt the current row and do NOT write (a read-modify-write of a row we did
      // not change could only overwrite somebody else's newer value)
      if (active) val = g;
      if (lane == 0) L.csum[(NS + k) * W2B_NDWMAX + wave] = now;
      continue;
    }
    const bool untouched = (now == L.csum[(NS + k) * W2B_NDWMAX + wave]);
    if (active) {
#pragma unroll
      for (int e = 0; e < VEC; e++) { val.e[e] = untouched ? val.e[e] : g.e[e] + del.e[e]; del.e[e] = 0.f; }
      store_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.v, k + 1, P.dim, col0, val, P.tab_bytes);
    }
    const unsigned cs = wave_xor(active ? col_bits(val) : 0u);
    if (lane == 0) L.csum[(NS + k) * W2B_NDWMAX + wave] = cs;
  }
}

#ifdef W2B_PHASE_TIMERS
#define W2B_TICK2(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); \
    atomicAdd(&P.shared->dbg[k], n_ - t2_); t2_ = n_; } } while (0)
#else
#define W2B_TICK2(k) do { } while (0)
#endif
// ---- write-back of one leaving row: exact value if nobody else changed the row, else merge our delta
template <int VEC, int MM>
__device__ __forceinline__ void retire_finish(const W2bParams &P, int row, unsigned csum_at_entry, const Col<VEC> &g,
                                              const Col<VEC> &rw, const Col<VEC> &rd, bool active, int col0) {
  const unsigned now = wave_xor(active ? col_bits(g) : 0u);
  const bool untouched = (now == csum_at_entry);                          // wave-uniform
  if (active) {
    Col<VEC> o;
#pragma unroll
    for (int e = 0; e < VEC; e++) o.e[e] = untouched ? rw.e[e] : g.e[e] + rd.e[e];
    store_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.u, row, P.dim, col0, o, P.tab_bytes);
  }
}

// ---- steady state: at most W2B_RCH rows leave and enter per step: one memory round trip for both
template <int VEC, int MM>
__device__ __forceinline__ void window_exchange(const W2bParams &P, const Win2 &L, int n_ret, int n_adm, bool active,
                                                int col0, int lane, int wave) {
  const int dim = P.dim;
  Col<VEC> rw[W2B_RCH], rd[W2B_RCH], g[W2B_RCH], a[W2B_RCH];
#pragma unroll
  for (int i = 0; i < W2B_RCH; i++) {
#pragma unroll
    for (int e = 0; e < VEC; e++) { rw[i].e[e] = 0.f; rd[i].e[e] = 0.f; g[i].e[e] = 0.f; a[i].e[e] = 0.f; }
    if (active && i < n_ret) {
      const int s = L.ret_slot[i];
      rw[i] = lds_ld<VEC>(L.win + s * dim + col0);
      rd[i] = lds_ldh<VEC>(L.dlt + s * dim + col0);
      g[i] = load_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.u, L.ret_row[i], dim, col0, P.tab_bytes);
    }
  }
#pragma unroll
  for (int i = 0; i < W2B_RCH; i++)
    if (active && i < n_adm) a[i] = load_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.u, L.adm_row[i], dim, col0, P.tab_bytes);
#pragma unroll
  for (int i = 0; i < W2B_RCH; i++)
    if (i < n_ret) {
      const int s = L.ret_slot[i];
      const unsigned now = wave_xor(active ? col_bits(g[i]) : 0u);
      const bool untouched = (now == L.csum[s * W2B_NDWMAX + wave]);
      if (active) {
        Col<VEC> o;
#pragma unroll
        for (int e = 0; e < VEC; e++) o.e[e] = untouched ? rw[i].e[e] : g[i].e[e] + rd[i].e[e];
        store_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.u, L.ret_row[i], dim, col0, o, P.tab_bytes);
      }
    }
#pragma unroll
  for (int i = 0; i < W2B_RCH; i++)
    if (i < n_adm) {
      const int s = L.adm_slot[i];
      const unsigned cs = wave_xor(active ? col_bits(a[i]) : 0u);
      if (lane == 0) L.csum[s * W2B_NDWMAX + wave] = cs;
      if (active) {
        Col<VEC> z;
#pragma unroll
        for (int e = 0; e < VEC; e++) z.e[e] = 0.f;
        lds_st(L.win + s * dim + col0, a[i]);
        lds_sth(L.dlt + s * dim + col0, z);
      }
    }
}

// ---- one centre word on the resident window.  cslot[j] >= 0: LDS slot; -1-k: register-held row k.
template <int QM, int VEC, bool LOSS, int MM>
__device__ __forceinline__ void process_word2(const W2bParams &P, const Win2 &L, const QParam &qp, const int ndw,
                                              const int cw,
                                              const int nt, const int uc_n, const float alpha, double &loss_acc,
                                              HotV<VEC> &H) {
  constexpr int W2B_T2 = T2For<VEC>::value;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = ndw;   // data wavefronts 0..ndw-1
  const int dim = P.dim, col0 = tid * VEC;
  const bool active = col0 < dim;
  const float ar2 = (2.f * alpha) * P.reg;

#ifdef W2B_PHASE_TIMERS
  unsigned long long t2_ = wall_clock64();
#endif
  // Two register buffers of target rows: while chunk k is reduced and updated, the rows of chunk k+1 are
  // already in flight (unless chunk k+1 repeats a row of an earlier chunk: then it is loaded after the stores)
  Col<VEC> xa[W2B_T2], xb[W2B_T2];
  int ra[W2B_T2], rb[W2B_T2];
  auto load_chunk = [&](Col<VEC> (&X)[W2B_T2], int (&Rw)[W2B_T2], const int s, const int e) {
    const int mine = L.tgt[min(s + lane, nt - 1)] & 0x3fffffff;    // bit 30 = label, see prep_lists
#pragma unroll
    for (int i = 0; i < W2B_T2; i++) Rw[i] = __builtin_amdgcn_readlane(mine, i);
#pragma unroll
    for (int i = 0; i < W2B_T2; i++)
#pragma unroll
      for (int ee = 0; ee < VEC; ee++) X[i].e[ee] = 0.f;
    if (active) {      // one exec mask for the whole chunk, not one per row
#pragma unroll
      for (int i = 0; i < W2B_T2; i++) {
        if (s + i < e) {
          // the register-resident hot rows can only sit at slot 0 (row 1 or 2) or slot 1 (row 2)
          if (i == 0 && H.on && Rw[i] == 1) X[i] = H.v0;
          else if (i <= 1 && H.on && Rw[i] == 2) X[i] = H.v1;
          else X[i] = load_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.v, Rw[i], dim, col0, P.tab_bytes);
        }
      }
    }
  };
  int start = 0, chunk = 0, end = L.cend[0] & 0xffff;
  load_chunk(xa, ra, start, end);
  // the (at most two) context rows outside the radius live in registers for this step
  Col<VEC> ur0, ur1;
#pragma unroll
  for (int e = 0; e < VEC; e++) { ur0.e[e] = 0.f; ur1.e[e] = 0.f; }
  if (active && uc_n > 0) ur0 = load_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.u, L.uc_row[0], dim, col0, P.tab_bytes);
  if (active && uc_n > 1) ur1 = load_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.u, L.uc_row[1], dim, col0, P.tab_bytes);

  W2B_TICK2(6);
  // ---- phase A from LDS (ref :431-449), window order
  Col<VEC> avg;
  float regsq = 0.f;
#pragma unroll
  for (int e = 0; e < VEC; e++) avg.e[e] = 0.f;
  if (active) {
    for (int j = 0; j < cw; j++) {
      const int s = L.cslot[j];
      Col<VEC> r;
      if (s >= 0) {
        r = lds_ld<VEC>(L.win + s * dim + col0);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; e++) r.e[e] = (s == -1) ? ur0.e[e] : ur1.e[e];
      }
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        const float q = quant<QM>(r.e[e], qp);
        avg.e[e] += q;
        if (LOSS) regsq += q * q;
      }
    }
    const float cwf = (float)cw;
#pragma unroll
    for (int e = 0; e < VEC; e++) avg.e[e] = avg.e[e] / cwf;
  }

  W2B_TICK2(7);
  // ---- phase B (ref :450-492): identical to process_word, W2B_T2 rows per chunk
  Col<VEC> err;
#pragma unroll
  for (int e = 0; e < VEC; e++) err.e[e] = 0.f;
  int par = 0;
  auto process_chunk = [&](Col<VEC> (&x)[W2B_T2], int (&rows)[W2B_T2], const int start, const int end) {
    const int n = end - start;
    float p[W2B_T2], p2[W2B_T2];
#pragma unroll
    for (int i = 0; i < W2B_T2; i++) {
      float t[VEC], s2 = 0.f;
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        const float q = quant<QM>(x[i].e[e], qp);
        t[e] = avg.e[e] * q;
        if (LOSS) s2 += q * q;
      }
      const float s = (VEC == 4) ? (t[0] + t[1 % VEC]) + (t[2 % VEC] + t[3 % VEC]) : t[0] + t[1 % VEC];
      p[i] = active ? s : 0.f;
      p2[i] = active ? s2 : 0.f;
    }
    W2B_LDS float *red = L.red + par * (W2B_T2 * W2B_NDWMAX);
#pragma unroll
    for (int i = 0; i < W2B_T2; i++) p[i] = wave_sum(p[i]);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < W2B_T2; i++)
        if (i < n) red[i * W2B_NDWMAX + wave] = p[i];
    }
    __syncthreads();
    float gl = 0.f;
    if (lane < n) {
      float f = 0.f;
      // pairs of 8-byte-column wavefronts first: the same binary tree over the elements as the plain kernel's
      // 16-byte-column wavefronts, so that a single worker reproduces the plain kernel bit for bit
      if (VEC == 4) {      // 16-byte columns: a wavefront covers the same 256 columns as in the plain kernel
        for (int w = 0; w < nwaves; w++) f += red[lane * W2B_NDWMAX + w];
      } else {
        for (int w = 0; w < nwaves; w += 2)
          f += red[lane * W2B_NDWMAX + w] + ((w + 1 < nwaves) ? red[lane * W2B_NDWMAX + w + 1] : 0.f);
      }
      // the centre word: entry bit 30 when the producer reorders chunks (hot rows first), else list index 0
      const float label = (H.on ? ((L.tgt[start + lane] >> 30) & 1) : (start + lane == 0)) ? 1.f : 0.f;
      float g;
      if (f > 6.f) g = (label - 1.f) * alpha;
      else if (f < -6.f) g = label * alpha;
      else g = (label - P.exp_table[(int)((f + 6.f) * 83.f)]) * alpha;
      gl = g;
      if (LOSS && wave == 0) {
        const float dp = (label != 0.f) ? f : -f;
        float sg;
        if (dp > 6.f) sg = 1.f;
        else if (dp < -6.f) sg = 1e-9f;
        else sg = 1.f / (1.f + expf(-dp));
        loss_acc += (double)logf(sg);
      }
    }
    if (LOSS && P.reg != 0.f) {
#pragma unroll
      for (int i = 0; i < W2B_T2; i++)
        if (i < n) {
          const float s2 = wave_sum(p2[i]);
          if (lane == 0) loss_acc -= (double)(P.reg * s2);
        }
    }
    if (active) {        // one exec mask for the whole update section, not one per row
#pragma unroll
      for (int i = 0; i < W2B_T2; i++) {
        if (i < n) {
          const float g = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gl), i));
          float dd[VEC];
#pragma unroll
          for (int e = 0; e < VEC; e++) {
            float xv = x[i].e[e];
            // opaque copy: re-derive the quantized value here instead of keeping VEC extra registers per
            // row alive since the dot product (halves the register footprint of a chunk)
            if (QM != 0) asm volatile("" : "+v"(xv));
            err.e[e] += g * quant<QM>(xv, qp);
            dd[e] = g * avg.e[e] - ar2 * xv;
            x[i].e[e] = xv + dd[e];
          }
          if (i == 0 && H.on && rows[i] == 1) {
            H.v0 = x[i];
#pragma unroll
            for (int e = 0; e < VEC; e++) H.d0.e[e] += dd[e];
          } else if (i <= 1 && H.on && rows[i] == 2) {
            H.v1 = x[i];
#pragma unroll
            for (int e = 0; e < VEC; e++) H.d1.e[e] += dd[e];
          } else {
            store_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.v, rows[i], dim, col0, x[i], P.tab_bytes);
          }
        }
      }
    }
    par ^= 1;
  };
  for (;;) {
    {   // buffer A holds chunk `chunk`
      const bool more = end < nt;
      const int raw = more ? L.cend[chunk + 1] : 0;
      const int nend = raw & 0xffff, dep = raw >> 16;
      if (more && !dep) load_chunk(xb, rb, end, nend);
      process_chunk(xa, ra, start, end);
      if (!more) break;
      if (dep) load_chunk(xb, rb, end, nend);
      start = end; end = nend; chunk++;
    }
    {   // buffer B holds chunk `chunk`
      const bool more = end < nt;
      const int raw = more ? L.cend[chunk + 1] : 0;
      const int nend = raw & 0xffff, dep = raw >> 16;
      if (more && !dep) load_chunk(xa, ra, end, nend);
      process_chunk(xb, rb, start, end);
      if (!more) break;
      if (dep) load_chunk(xa, ra, end, nend);
      start = end; end = nend; chunk++;
    }
  }

  W2B_TICK2(8);
  // ---- phase C on the resident rows (ref :494-503), window order; duplicates hit the same slot twice
  if (active) {
    for (int j = 0; j < cw; j++) {
      const int s = L.cslot[j];
      if (s >= 0) {
        Col<VEC> w0 = lds_ld<VEC>(L.win + s * dim + col0), dl = lds_ldh<VEC>(L.dlt + s * dim + col0);
#pragma unroll
        for (int e = 0; e < VEC; e++) {
          const float d = err.e[e] - ar2 * w0.e[e];
          w0.e[e] = w0.e[e] + d;
          dl.e[e] = dl.e[e] + d;
        }
        lds_st(L.win + s * dim + col0, w0);
        lds_sth(L.dlt + s * dim + col0, dl);
      } else if (s == -1) {
#pragma unroll
        for (int e = 0; e < VEC; e++) ur0.e[e] = ur0.e[e] + (err.e[e] - ar2 * ur0.e[e]);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; e++) ur1.e[e] = ur1.e[e] + (err.e[e] - ar2 * ur1.e[e]);
      }
    }
    if (uc_n > 0) store_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.u, L.uc_row[0], dim, col0, ur0, P.tab_bytes);
    if (uc_n > 1) store_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.u, L.uc_row[1], dim, col0, ur1, P.tab_bytes);
  }
  W2B_TICK2(9);
  if (LOSS && P.reg != 0.f) {
    const float s = wave_sum(regsq);
    if (lane == 0) loss_acc -= (double)(P.reg * s);
  }
}

// --------------------------------------------------------------------------------------------------
// NDW + 1 wavefronts per worker: wavefronts 0..NDW-1 own the embedding columns (data phase); the last one is the
// PRODUCER: it walks the sentence, the LCG ledger, the window bookkeeping and the negative draws ONE STEP
// AHEAD and hands the lists over through a double-buffered LDS record.  The data wavefronts never wait for
// the scalar work of a step (it was 25-30 % of the step time when wavefront 0 did both).
// Barrier discipline: every wavefront executes the same s_barrier sequence per step: nck barriers inside
// the data phase (one per target chunk; the producer executes them after its own work) + one at the end.
// Register budget: 8-byte columns = up to 7 + 1 wavefronts per worker, two workers per CU -> 4 wavefronts per SIMD
// (128 VGPRs); 16-byte columns = up to 4 + 1 wavefronts per worker, two workers per CU -> at most 3 per SIMD (168).
template <int QM, int VEC, bool LOSS, int MM>
__global__ void __launch_bounds__(VEC == 4 ? 320 : 512, VEC == 4 ? 3 : 4) k_train_workers2(const W2bParams P, const long long max_positions,
                                                           const int R, const int NDW) {
  extern __shared__ int smem[];
  W2B_LDS int *const smem_lds = (W2B_LDS int *)smem;
  const Win2 L0 = carve_win2(smem_lds, P.dim, P.window, P.negative, R, 0);
  const Win2 L1 = carve_win2(smem_lds, P.dim, P.window, P.negative, R, 1);
  const Win2 &L = L0;                                   // everything that is not double buffered
  W2B_LDS WorkerLds *S = &L.S->w;
  W2B_LDS int *s_sen = L.sen;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = (wave == NDW);
  const int wid = blockIdx.x;
  if (wid >= P.num_threads) return;
  W2bWorker *G = P.workers + wid;
  if (G->done) return;
  QParam qp;
  qp.bitlevel = P.bitlevel;
  qp.steps_i = (P.bitlevel >= 4) ? (1 << (P.bitlevel - 1)) : 1;
  qp.steps_f = (float)qp.steps_i;
  const int NS = 2 * R + 1;
  const int col0 = tid * VEC;
  const bool active = !producer && col0 < P.dim;
  for (int i = tid; i < G->sen_len; i += blockDim.x) s_sen[i] = G->sen[i];
  for (int i = tid; i < NS; i += blockDim.x) { L.slot_row[i] = -1; L.slot_ref[i] = 0; L.pos_slot[i] = 0; }
  for (int i = tid; i < (P.negative + 2 > 66 ? P.negative + 2 : 66); i += blockDim.x) { L.ja[i] = P.jump_a[i]; L.jc[i] = P.jump_c[i]; }
  if (tid == 0) {
    S->rng = G->rng; S->cursor = G->cursor; S->wc = G->word_count; S->last_wc = G->last_word_count;
    S->sen_len = G->sen_len; S->sen_pos = G->sen_pos; S->override_ = G->first_override;
    S->eof = 0; S->done = 0; S->cw = 0; S->nt = 0; S->alpha = 0.f;
    L.S->clo = 0; L.S->chi = -1;
  }
  __syncthreads();
  double loss_acc = 0.0;
  const int W = P.window, K = P.negative;
#ifdef W2B_PHASE_TIMERS
#define W2B_TICK(k) do { if (blockIdx.x == 0 && tid == 0) { const unsigned long long n_ = wall_clock64(); \
    atomicAdd(&P.shared->dbg[k], n_ - tick_); tick_ = n_; } } while (0)
#define W2B_TICKP(k) do { if (blockIdx.x == 0 && tid == NDW * 64) { const unsigned long long n_ = wall_clock64(); \
    atomicAdd(&P.shared->dbg[k], n_ - tick_); tick_ = n_; } } while (0)
  unsigned long long tick_ = wall_clock64();
#else
#define W2B_TICK(k) do { } while (0)
#define W2B_TICKP(k) do { } while (0)
#endif
  // producer registers: the unigram-table gather and the alpha load of the NEXT step are issued at the end
  // of a preparation, so that their latency is not on the producer's critical path either
  int t_pref = 0;
  bool pref_ok = false;
  float alpha_pref = P.starting_alpha;
  bool alpha_pref_ok = false;
  // data-wavefront registers: the row that enters the window at the next step, loaded one step early
  Col<VEC> apre;
#pragma unroll
  for (int e = 0; e < VEC; e++) apre.e[e] = 0.f;
  int apre_row = -1;
  // data-wavefront registers: the two hottest target rows (value + delta)
  HotV<VEC> H;
  H.on = P.hot_rows;
#pragma unroll
  for (int e = 0; e < VEC; e++) { H.v0.e[e] = 0.f; H.v1.e[e] = 0.f; H.d0.e[e] = 0.f; H.d1.e[e] = 0.f; }
  if (!producer && H.on) {
    if (active) H.v0 = load_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.v, 1, P.dim, col0, P.tab_bytes);
    if (active && P.vocab_size > 2) H.v1 = load_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.v, 2, P.dim, col0, P.tab_bytes);
    const unsigned c0 = wave_xor(active ? col_bits(H.v0) : 0u), c1 = wave_xor(active ? col_bits(H.v1) : 0u);
    if (lane == 0) { L.csum[(NS + 0) * W2B_NDWMAX + wave] = c0; L.csum[(NS + 1) * W2B_NDWMAX + wave] = c1; }
  }

  // ---- the preparation of one pass (producer wavefront only; all 64 lanes, wave-uniform control flow)
  auto prepare = [&](const Win2 &O, const bool last) {
      unsigned long long rng = S->rng;
      long long cursor = S->cursor, wc = S->wc, last_wc = S->last_wc;
      int sen_len = S->sen_len, sen_pos = S->sen_pos, ovr = S->override_, eof = S->eof;
      int done = 0, cw = 0, nt = 0, uc_n = 0, nck = 0, next_row = -1;
      float alpha = 0.f, alpha_own = 0.f;
      bool new_sentence = false, alpha_set = false;
      int lo = 0, hi = -1;                               // window wanted for this step (empty = flush)
      int p = 0, b = 0, word = 0;
      bool train = false;
      if (!last) {
        if (wc - last_wc > 10000) {                                    // ref :379-393
          if (lane == 0) {
            const unsigned long long d = (unsigned long long)(wc - last_wc);
            const unsigned long long wca = atomicAdd(&P.shared->word_count_actual, d) + d;
            const long long wca_all = (long long)wca * (P.total_threads / P.num_threads);
            float a = P.starting_alpha * (1.f - (float)wca_all / (float)(P.iter * P.train_words + 1));
            if ((double)a < (double)P.starting_alpha * 0.0001) a = (float)((double)P.starting_alpha * 0.0001);
            __hip_atomic_store(&P.shared->alpha, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            alpha_own = a;
          }
          alpha_own = __shfl(alpha_own, 0, 64);
          alpha_set = true;                              // this worker's own write is the newest value it may see
          last_wc = wc;
        }
        if (sen_len == 0) {                                            // ref :394-413
          read_sentence(P, (int *)s_sen, rng, cursor, wc, ovr, eof, sen_len, lane);
          sen_pos = 0;
          new_sentence = true;
          W2B_WAVE_SYNC();
        }
        if (eof || wc > P.train_words / P.total_threads) {            // ref :414-423
          if (lane == 0) atomicAdd(&P.shared->word_count_actual, (unsigned long long)(wc - last_wc));
          last_wc = wc;
          done = 1;
        } else {
          train = true;
          p = sen_pos;
          word = (sen_len > 0) ? s_sen[p] : 0;                          // ref :424
          rng = rng * W2B_LCG_A + W2B_LCG_C;                            // ref :428-429
          b = (int)fast_mod(rng, (unsigned long long)W, P.window_magic);
          if (sen_len > 0) { lo = max(0, p - R); hi = min(sen_len - 1, p + R); }
        }
      }
      // ---------------- window bookkeeping: make the resident range [lo, hi]
      int clo = L.S->clo, chi = L.S->chi, n_ret = 0, n_adm = 0;
      if (new_sentence || !train) {
        if (lane < NS) L.slot_ref[lane] = 0;              // every resident position belonged to the old sentence
        clo = 0; chi = -1;
        W2B_WAVE_SYNC();
      } else {
        for (int q = clo; q <= chi; q++) {                // positions that leave: [clo, lo) and (hi, chi]
          if (q >= lo && q <= hi) { q = hi; continue; }
          if (lane == 0) L.slot_ref[L.pos_slot[q % NS]]--;
          W2B_WAVE_SYNC();
        }
      }
      // Positions entering the window.  Pass 1 re-uses rows that are still resident (also rows of
      // positions that just left: they are revived instead of being written back and re-read).  Pass 2
      // gives the remaining words a slot; only then may a leaving row's slot be recycled, so a row that is
      // wanted again in this very step is never reloaded before its write-back.
      for (int pass = 0; pass < 2; pass++) {
        for (int q = lo; q <= hi; q++) {
          if (q >= clo && q <= chi) { q = chi; continue; }                 // already resident
          if (pass == 1 && L.pos_slot[q % NS] >= 0) continue;             // resolved in pass 1
          const int w = s_sen[q];
          const bool match = (lane < NS) && (L.slot_row[lane] == w);
          const unsigned long long mm = __ballot(match);
          int s = -1;
          if (mm) {                                                       // the word is resident: share its slot
            s = __ffsll((long long)mm) - 1;
            if (lane == 0) L.slot_ref[s]++;
          } else if (pass == 1) {
            const unsigned long long fr = __ballot((lane < NS) && (L.slot_row[lane] == -1));
            if (fr) s = __ffsll((long long)fr) - 1;
            else {                                                         // recycle the slot of a leaving row
              const unsigned long long pend = __ballot((lane < NS) && (L.slot_ref[lane] == 0));
              s = __ffsll((long long)pend) - 1;
              if (lane == 0) { O.ret_slot[n_ret] = s; O.ret_row[n_ret] = L.slot_row[s]; }
              n_ret++;
            }
            if (lane == 0) {
              L.slot_row[s] = w; L.slot_ref[s] = 1;
              O.adm_slot[n_adm] = s; O.adm_row[n_adm] = w;
            }
            n_adm++;
          }
          if (lane == 0) L.pos_slot[q % NS] = s;
          W2B_WAVE_SYNC();
        }
      }
      {                                                                  // whatever is unreferenced leaves
        const bool leaving = (lane < NS) && (L.slot_row[lane] != -1) && (L.slot_ref[lane] == 0);
        const unsigned long long ml = __ballot(leaving);
        if (leaving) {
          const int k = n_ret + __popcll(ml & lane_lt_mask(lane));
          O.ret_slot[k] = lane; O.ret_row[k] = L.slot_row[lane];
          L.slot_row[lane] = -1;
        }
        n_ret += __popcll(ml);
        W2B_WAVE_SYNC();
      }
      if (train) {
        const int hiA = 2 * W + 1 - b;
        for (int a0 = b; a0 < hiA; a0 += 64) {                          // ref :431-436
          const int a = a0 + lane;
          const int c = p - W + a;
          const bool ok = (a < hiA) && (a != W) && (c >= 0) && (c < sen_len);
          const unsigned long long m = __ballot(ok);
          int slot = 0;
          if (ok) {
            if (c >= lo && c <= hi) slot = L.pos_slot[c % NS];
            else {                                     // outside the radius (|c - p| == window): resident anyway?
              const int w = s_sen[c];
              slot = (c < p) ? -1 : -2;                // provisional: register-held row (left / right)
              for (int s2 = 0; s2 < NS; s2++) slot = (L.slot_row[s2] == w) ? s2 : slot;
            }
          }
          if (ok) O.cslot[cw + __popcll(m & lane_lt_mask(lane))] = slot;
          cw += __popcll(m);
        }
        W2B_WAVE_SYNC();
        if (cw > 0 && R < W) {
          // The radius is window-1: the two outermost context positions (only present when b == 0) are
          // not resident.  They are the first / last entry of the context list; each one that is not
          // resident through another position becomes a register-held row of this step.
          const int first = O.cslot[0], lastc = O.cslot[cw - 1];
          int wl = -1;
          if (first == -1) { wl = s_sen[p - W]; if (lane == 0) O.uc_row[0] = wl; uc_n = 1; }
          if (lastc == -2) {
            const int wr = s_sen[p + W];
            if (uc_n == 1 && wr == wl) { if (lane == 0) O.cslot[cw - 1] = -1; }       // same word on both ends
            else {
              if (lane == 0) { O.uc_row[uc_n] = wr; O.cslot[cw - 1] = -1 - uc_n; }
              uc_n++;
            }
          }
          W2B_WAVE_SYNC();
        }
        if (cw > 0) {                                                    // ref :450-460
          int cnt = 0;
          for (int d0 = 1; d0 <= K; d0 += 64) {
            const int d = d0 + lane;
            bool keep = false;
            int t = 0;
            if (d <= K) {
              const unsigned long long x = (L.ja[d] * rng + L.jc[d]);
              t = (pref_ok && d0 == 1) ? t_pref : P.table[fast_mod(x >> 16, (unsigned long long)P.table_size, P.table_magic)];
              if (t == 0) t = (int)(x % (unsigned long long)(P.vocab_size - 1)) + 1;
              keep = (t != word);
            }
            const unsigned long long m = __ballot(keep);
            if (keep) O.tgt[1 + cnt + __popcll(m & lane_lt_mask(lane))] = t;
            cnt += __popcll(m);
          }
          if (lane == 0) O.tgt[0] = word;
          nt = 1 + cnt;
          rng = (L.ja[K] * rng + L.jc[K]);
          alpha = alpha_set ? alpha_own
                            : (alpha_pref_ok ? alpha_pref
                                             : __hip_atomic_load(&P.shared->alpha, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT));
          nck = prep_lists<T2For<VEC>::value, W2B_LDS int *>(O.tgt, L.prev, O.cend, nt, (W2B_LDS int *)nullptr, (W2B_LDS int *)nullptr, 0, lane, true, P.hot_rows != 0);
        }
        const int nq = p + 1 + R;                                        // enters the window at the next step
        next_row = (p + 1 < sen_len && nq < sen_len) ? s_sen[nq] : -1;
        sen_pos++;                                                       // ref :505-509
        if (sen_pos >= sen_len) sen_len = 0;
        // ---- prefetch for the next step (valid unless the next step starts with a sentence read, whose
        // sub-sampling draws come first in the LCG ledger)
        pref_ok = (sen_len != 0);
        if (pref_ok && lane < K) {                                       // lane l serves draw d = l + 1
          const unsigned long long xb = rng * W2B_LCG_A + W2B_LCG_C;     // the next step's window draw
          const unsigned long long x = (L.ja[lane + 1] * xb + L.jc[lane + 1]);
          t_pref = P.table[fast_mod(x >> 16, (unsigned long long)P.table_size, P.table_magic)];
        }
        alpha_pref = __hip_atomic_load(&P.shared->alpha, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        alpha_pref_ok = true;
      } else {
        pref_ok = false;
      }
      if (lane == 0) {
        S->rng = rng; S->cursor = cursor; S->wc = wc; S->last_wc = last_wc;
        S->sen_len = sen_len; S->sen_pos = sen_pos; S->override_ = ovr; S->eof = eof;
        L.S->clo = lo; L.S->chi = hi;
        O.St->stop = (done || last) ? 1 : 0; O.St->cw = cw; O.St->nt = nt; O.St->uc_n = uc_n;
        O.St->n_ret = n_ret; O.St->n_adm = n_adm; O.St->next_row = next_row; O.St->nck = (cw > 0) ? nck : 0;
        O.St->alpha = alpha;
        if (done) S->done = 1;
      }
  };

  if (producer) prepare(L0, max_positions == 0);
  __syncthreads();
  for (long long it = 0;; ++it) {
    const Win2 &I = (it & 1) ? L1 : L0;                 // this step's lists
    const bool stop = I.St->stop != 0;
    const int nck = I.St->nck;
    if (producer) {
      W2B_TICKP(10);
      if (!stop) prepare((it & 1) ? L0 : L1, it + 1 == max_positions);
      W2B_TICKP(11);
      for (int i = 0; i < nck; i++) __syncthreads();
    } else {
      W2B_TICK(0);
      // ---------------- data phase
      const int n_ret = I.St->n_ret, n_adm = I.St->n_adm;
      bool deferred = false;                    // steady state: the leaving row is merged back AFTER the step
      int d_row = -1;
      unsigned d_csum = 0;
      Col<VEC> d_g, d_rw, d_rd;
#pragma unroll
      for (int e = 0; e < VEC; e++) { d_g.e[e] = 0.f; d_rw.e[e] = 0.f; d_rd.e[e] = 0.f; }
      if (n_ret <= 1 && n_adm <= 1) {
        const int uc_n = I.St->uc_n;
        if (n_ret == 1) {
          const int s = I.ret_slot[0];
          d_row = I.ret_row[0];
          d_csum = L.csum[s * W2B_NDWMAX + wave];
          if (active) {
            d_rw = lds_ld<VEC>(L.win + s * P.dim + col0);
            d_rd = lds_ldh<VEC>(L.dlt + s * P.dim + col0);
            d_g = load_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.u, d_row, P.dim, col0, P.tab_bytes);               // consumed after the step: no stall
          }
          deferred = true;
        }
        if (n_adm == 1) {
          const int s = I.adm_slot[0], row = I.adm_row[0];
          Col<VEC> a = apre;                                                   // loaded during the previous step
          if (row != apre_row) {
#pragma unroll
            for (int e = 0; e < VEC; e++) a.e[e] = 0.f;
            if (active) a = load_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.u, row, P.dim, col0, P.tab_bytes);
          }
          const unsigned cs = wave_xor(active ? col_bits(a) : 0u);
          if (lane == 0) L.csum[s * W2B_NDWMAX + wave] = cs;
          if (active) {
            Col<VEC> z;
#pragma unroll
            for (int e = 0; e < VEC; e++) z.e[e] = 0.f;
            lds_st(L.win + s * P.dim + col0, a);
            lds_sth(L.dlt + s * P.dim + col0, z);
          }
        }
        // a register-held outer row of this step that is the row leaving right now must see the merge
        if (deferred && ((uc_n > 0 && I.uc_row[0] == d_row) || (uc_n > 1 && I.uc_row[1] == d_row))) {
          retire_finish<VEC, MM>(P, d_row, d_csum, d_g, d_rw, d_rd, active, col0);
          deferred = false;
        }
        // prefetch the row that enters at the next step (never one whose store is still ahead of us)
        const int nr = I.St->next_row;
        const bool is_uc = (uc_n > 0 && I.uc_row[0] == nr) || (uc_n > 1 && I.uc_row[1] == nr);
        apre_row = (nr >= 0 && !(deferred && nr == d_row) && !is_uc) ? nr : -1;
        if (apre_row >= 0 && active) apre = load_col<VEC, (MM & 7), ((MM >> 3) & 1)>(P.u, apre_row, P.dim, col0, P.tab_bytes);
      } else {
        if (n_ret <= W2B_RCH && n_adm <= W2B_RCH) {
          window_exchange<VEC, MM>(P, I, n_ret, n_adm, active, col0, lane, wave);
        } else {
          if (n_ret) window_retire<VEC, MM>(P, I, n_ret, active, col0, lane, wave);
          if (n_adm) window_admit<VEC, MM>(P, I, n_adm, active, col0, lane, wave);
        }
        apre_row = -1;
      }
      W2B_TICK(4);
      if (!stop && I.St->cw > 0)
        process_word2<QM, VEC, LOSS, MM>(P, I, qp, NDW, I.St->cw, I.St->nt, I.St->uc_n, I.St->alpha, loss_acc, H);
      if (stop || (it & (P.hot_period - 1)) == P.hot_period - 1) hot_merge<VEC, MM>(P, L, H, NS, active, col0, lane, wave);
      if (deferred) retire_finish<VEC, MM>(P, d_row, d_csum, d_g, d_rw, d_rd, active, col0);
      W2B_TICK(5);
    }
    __syncthreads();                                     // lists of the next step are published; this step is done
    if (stop) break;
  }
  const int sl = S->sen_len;
  for (int i = tid; i < sl; i += blockDim.x) G->sen[i] = s_sen[i];
  double lsum = 0.0;
  if (LOSS) {
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

}  // namespace

// Column width of the sentence-resident kernel.  16 bytes per lane whenever the row allows it: measured with
// tools/row_probe.hip, random 3200-byte rows move at 5.8 TB/s with 16-byte lanes against 4.2 TB/s with 8-byte lanes
// at the same number of bytes in flight (the texture path is bound by lane-accesses, not bytes).
static int win2_vec(int dim) {
  if (const char *e = getenv("W2B_WIN2_VEC")) { const int v = atoi(e); if (v == 2 && dim % 2 == 0) return 2; }
  return (dim % 4 == 0 && dim / 4 <= 4 * 64) ? 4 : 2;
}
int w2b_workers2_vec(int dim) { return win2_vec(dim); }
static int win2_threads(int dim) { const int vec = win2_vec(dim); return (((dim / vec) + 63) / 64 + 1) * 64; }

// Radius for which the sentence-resident kernel can run with two workgroups per CU (-1: use the plain kernel)
int w2b_window_radius(int dim, int window, int negative) {
  if (dim % 2 != 0 || dim > 2 * 64 * 7) return -1;     // 8- or 16-byte columns, at most 7 data wavefronts (+1 producer)
  const size_t budget = 80 * 1024;                 // two workgroups per 160 KiB CU
  if (win2_lds_bytes(dim, window, negative, window) <= budget) return window;
  if (window >= 2 && win2_lds_bytes(dim, window, negative, window - 1) <= budget) return window - 1;
  return -1;
}

// workgroups of the sentence-resident kernel that are resident per CU (occupancy query of the instantiation
// that would run)
int w2b_workers2_per_cu(const W2bParams &p, int R, bool loss) {
  const size_t lds = win2_lds_bytes(p.dim, p.window, p.negative, R);
  int nb = 0;
  (void)dispatch_mm(p.mem_mode, [&](auto mm) -> hipError_t {
    constexpr int MM = decltype(mm)::value;
    return dispatch_q(p.bitlevel, [&](auto qm) -> hipError_t {
      constexpr int QM = decltype(qm)::value;
      if (win2_vec(p.dim) == 4) {
        if (loss) return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_train_workers2<QM, 4, true, MM>, win2_threads(p.dim), lds);
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_train_workers2<QM, 4, false, MM>, win2_threads(p.dim), lds);
      }
      if (loss) return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_train_workers2<QM, 2, true, MM>, win2_threads(p.dim), lds);
      return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_train_workers2<QM, 2, false, MM>, win2_threads(p.dim), lds);
    });
  });
  return nb > 0 ? nb : 1;
}

hipError_t w2b_launch_workers2(const W2bParams &p, long long max_positions, int R, bool loss, hipStream_t s) {
  const int threads = win2_threads(p.dim);   // data wavefronts (one thread per 8-byte column) + 1 producer wavefront
  const int NDW = threads / 64 - 1;
  const size_t lds = win2_lds_bytes(p.dim, p.window, p.negative, R);
  static bool reported = false;
  if (!reported && getenv("W2B_DEBUG")) {
    reported = true;
    int nb = -1;
    if (win2_vec(p.dim) == 4) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_train_workers2<1, 4, false, 0>, threads, lds);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_train_workers2<1, 2, false, 0>, threads, lds);
    fprintf(stderr, "w2b debug: sentence-resident kernel R=%d lds=%zu B, resident workgroups/CU=%d\n", R, lds, nb);
  }
  return dispatch_mm(p.mem_mode, [&](auto mm) -> hipError_t {
    constexpr int MM = decltype(mm)::value;
    return dispatch_q(p.bitlevel, [&](auto qm) -> hipError_t {
      constexpr int QM = decltype(qm)::value;
      // template MM carries the memory mode in bits 0-2 and "tables >= 2 GiB" (per-row descriptors) in bit 3
#define W2B_LAUNCH_W2(VEC, LOSS, MMV) hipLaunchKernelGGL((k_train_workers2<QM, VEC, LOSS, MMV>), dim3(p.num_threads), dim3(threads), lds, s, p, max_positions, R, NDW)
      const int vec = win2_vec(p.dim);
      if (p.tab_bytes) {
        if (vec == 4) { if (loss) W2B_LAUNCH_W2(4, true, MM); else W2B_LAUNCH_W2(4, false, MM); }
        else { if (loss) W2B_LAUNCH_W2(2, true, MM); else W2B_LAUNCH_W2(2, false, MM); }
      } else {
        if (vec == 4) { if (loss) W2B_LAUNCH_W2(4, true, MM + 8); else W2B_LAUNCH_W2(4, false, MM + 8); }
        else { if (loss) W2B_LAUNCH_W2(2, true, MM + 8); else W2B_LAUNCH_W2(2, false, MM + 8); }
      }
#undef W2B_LAUNCH_W2
      return hipGetLastError();
    });
  });
}
