// w2b_kernels_resident.hip -- form (i), SENTENCE-RESIDENT variant of the worker kernel.
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
//   * when a row enters the window its value is also written to a per-worker SCRATCH row in global memory
//     ("what it looked like when it entered").  When the row leaves and nobody else changed it meanwhile
//     (wavefront checksum of the row bits), the exact fp32 value is stored -- a single worker stays
//     bit-identical to the plain kernel; otherwise the worker's own contribution (value - entry, in fp32) is
//     added to the CURRENT row (merge), so concurrent Hogwild workers do not erase each other's updates.  The
//     scratch row is read only in that conflict case, and LDS holds nothing but the fp32 values -- the whole
//     window (R = window) fits next to a second workgroup up to D = 1000;
//   * the most frequent target rows of v (rows 1..hot_n: the vocabulary is sorted by frequency; hot_n is chosen
//     by the host from the word counts and the number of workers, 0 on flat distributions) are read and written at
//     this XCD's copy (XHot in w2b_device.hpp: nt accesses served by the XCD's L2, shared by all workers of the XCD)
//     and every hot_period steps a worker brings one of the copies up to date with its master row.  Coherent
//     accesses to one embedding row serialise at its memory line (about 7 M read-modify-writes per second); on
//     Zipf-distributed ids the most frequent word alone is a target of 0.3 centre words in every position;
//   * rows are thread-private 16-byte columns of LDS (a thread only ever touches its own 16 bytes of every
//     slot): no barriers, no bank conflicts; memory is accessed 16 bytes per lane (tools/row_probe.hip: random
//     3200-byte rows move at 5.8 TB/s with 16-byte lanes against 4.2 TB/s with 8-byte lanes at the same number of
//     bytes in flight).
// The radius R is window when the window fits in LDS next to a second workgroup, else window-1 with
// the two outermost context rows held in registers for the step; otherwise the launcher falls back
// to the plain kernel.
#include "w2b_device.hpp"

#include <cstdio>
#include <cstdlib>

// Register budget and workgroup shape.  A worker is up to 4 data wavefronts + 1 producer wavefront.  The hardware
// places wavefront i of a workgroup on SIMD i mod 4, so a worker that is a workgroup of its own puts TWO of its five
// wavefronts on SIMD 0, and a second such workgroup fits on the CU only within 128 VGPRs (measured: at 168 VGPRs 256
// and 512 one-worker workgroups run at the same speed although the occupancy API answers 2 per CU).  128 VGPRs hold
// 13 target rows: two memory round trips per centre word at negative = 24.
// Default: W2B_WPG = 1 worker per workgroup, W2B_RT = 13, 128 VGPRs, s_barrier.
// Alternative (measured, kept selectable at build time): W2B_WPG = 2 INDEPENDENT workers per workgroup -- ten
// wavefronts land 3/3/2/2 on the SIMDs, 168 VGPRs each, W2B_RT = 25: all target rows of a centre word in registers, one
// round trip per word AND two workers per CU; the workers synchronise their own wavefronts through LDS counters
// (worker_barrier) because s_barrier would couple them.  On MI355X it is +2 % at negative = 24 and -10 % at negative =
// 12 (25-row code for 13-row chunks), and the counter barrier itself costs 2 % against s_barrier: not the default.
#ifndef W2B_WPG
#define W2B_WPG 1       // workers per workgroup
#endif
#ifndef W2B_RT
#define W2B_RT (W2B_WPG == 1 ? 13 : 25)       // target rows per chunk (x 4 VGPRs)
#endif
#ifndef W2B_RES_WAVES
#define W2B_RES_WAVES (W2B_WPG == 1 ? 4 : 3)  // wavefronts per SIMD the kernel is register-allocated for
#endif
#define W2B_RB 5        // rows whose partial dot products are formed and reduced together (bounds the live temporaries)
#define W2B_NDWMAX 4    // data wavefronts per worker (one thread per 16-byte column: D <= 1024) + 1 producer wavefront
#define W2B_RCH 2       // window rows moved per trip when many enter/leave at once (sentence boundaries)

namespace {

// Phase timers (builds with -DW2B_PHASE_TIMERS; worker 0 only; shader clocks; printed by w2b_trainer_destroy under
// W2B_DEBUG):  producer wavefront: [0] prepare, [1] waiting at the end-of-step barrier, [2] loss bookkeeping
//   data wavefront 0  : [4] loads issued -> window exchange + phase A done, [5] partial dots (+ wait for the rows),
//                       [6] chunk barrier wait, [7] g + updates + stores, [8] phase C + hot merge,
//                       [9] end-of-step barrier wait, [10] steps
#ifdef W2B_PHASE_TIMERS
#define W2B_TICK(k) do { if (timing_) { const unsigned long long n_ = __builtin_readcyclecounter(); \
    atomicAdd(&P.shared->dbg[k], n_ - tick_); tick_ = n_; } } while (0)
#define W2B_COUNT(k) do { if (timing_) atomicAdd(&P.shared->dbg[k], 1ull); } while (0)
#else
#define W2B_TICK(k) do { } while (0)
#define W2B_COUNT(k) do { } while (0)
#endif

// Explicit LDS address space on every pointer of the kernel's LDS record: dereferences compile to ds_*
// instructions.  (With generic pointers the two step buffers were selected through a struct reference and
// address-space inference gave up: 250 flat_load/flat_store per step, each tied to vmcnt AND lgkmcnt, so
// every window access also waited for the target rows in flight.)
#define W2B_LDS __attribute__((address_space(3)))

struct Win2Lds {          // scalars owned by the producer wavefront (extends WorkerLds)
  WorkerLds w;
  int clo, chi;           // sentence positions currently resident (empty when chi < clo)
  unsigned bar_step, bar_chunk;   // arrival counters of the worker's two barriers (monotonic within a launch)
  double loss_reg;                // regularisation terms of the loss booked by the data wavefronts (reg != 0 only)
};

// A barrier among the wavefronts of ONE worker (W2B_WPG > 1; with one worker per workgroup it is s_barrier).
// Monotonic LDS counter: every arriving wavefront adds 1, everybody
// polls until `target` arrivals have been counted.  LDS operations of a wavefront execute in order and the CU's LDS
// serves all of them in arrival order, so what a wavefront wrote to LDS before its arrival is visible to whoever has
// seen the count -- no s_waitcnt on outstanding GLOBAL memory operations is involved (the row stores of a step stay
// in flight across the barrier; __syncthreads() would drain them).  The wavefront-scope fences only stop the compiler
// from moving LDS accesses across the barrier.
__device__ __forceinline__ void worker_barrier(W2B_LDS unsigned *cnt, unsigned target, int lane, bool arrive = true) {
#if W2B_WPG == 1           // the workgroup IS the worker: the hardware barrier (the producer executes the chunk barriers too)
  __syncthreads();
  return;
#endif
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  if (arrive && lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - target) < 0)
    __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// What the producer wavefront hands to the data wavefronts for ONE step (double buffered in LDS)
struct Step2 {
  int stop;               // 1: this pass only empties the window (epoch finished, or end of the launch)
  int cw, nt, uc_n, n_ret, n_adm;
  int next_row;           // row of the position that enters the window at the NEXT step (-1: unknown / none)
  int nck;                // number of target chunks = number of workgroup barriers inside the data phase
  float alpha;
  int cdup;               // 1: a window slot occurs more than once in this step's context list (phase C must run in order)
  int refresh;            // 1: the (single) row of the retire list does not leave: it is merged with memory and stays resident
  int pad[1];
};


// LDS record of one worker: every list has a compile-time capacity, so every field is a CONSTANT offset from the
// worker's LDS base (round 2 carved run-time sized arrays: 40-odd LDS pointers in SGPRs, 119 of them spilled to VGPR
// lanes in the benchmarked instantiation).  The kernel therefore runs for window <= W2B_WMAX and negative < W2B_TMAX;
// wider shapes use the plain worker kernel.
#define W2B_WMAX 16                     // radius <= window <= 16: at most 33 window slots
#define W2B_SMAX (2 * W2B_WMAX + 1 + 3) // window slots / retire + admit lists (S + 2 entries), padded
#define W2B_TMAX 64                     // negative + 1 targets
#define W2B_NJ 66                       // LCG jump-ahead entries (read_sentence jumps up to 64 draws; negative + 2)
struct StepRec {          // what the producer wavefront hands to the data wavefronts for ONE step (double buffered)
  Step2 st;
  int ret_slot[W2B_SMAX], ret_row[W2B_SMAX], ret_gen[W2B_SMAX];
  int adm_slot[W2B_SMAX], adm_row[W2B_SMAX], adm_gen[W2B_SMAX];
  int cslot[W2B_SMAX];    // slot of every context position; -1-k = k-th register-held row
  int uc_row[4];          // rows of the (at most two) context positions outside the radius
  int tgt[W2B_TMAX], cend[W2B_TMAX];
  float lossf[W2B_TMAX];  // dot products f of this step's targets, for the loss bookkeeping
};
struct WorkRec {
  Win2Lds S;
  int slot_row[W2B_SMAX], slot_ref[W2B_SMAX], pos_slot[W2B_SMAX], slot_gen[W2B_SMAX];
  int slot_born[W2B_SMAX];       // step at which the slot's row was last read from / merged with memory
  int prev[W2B_TMAX];
  float red[2][W2B_RT][W2B_NDWMAX];
  unsigned csum[W2B_SMAX][W2B_NDWMAX];   // per-wavefront xor checksum of the row bits at entry
  unsigned long long ja[W2B_NJ], jc[W2B_NJ];   // LCG jump-ahead table (copy of P.jump_a / P.jump_c)
  int sen[(W2B_MAX_SEN + 3) & ~3];
  StepRec step[2];
};
// view of a worker's LDS: the record, one of its two step buffers, and the window rows behind the record
struct Win2 {
  W2B_LDS WorkRec *w;
  W2B_LDS StepRec *s;
  W2B_LDS float *win;     // [S][dim] current fp32 value of the resident rows (thread-private 16-byte columns)
};

__host__ __device__ inline size_t win2_rec_bytes() { return (sizeof(WorkRec) + 15) & ~(size_t)15; }
// LDS bytes of the sentence-resident kernel for radius R
__host__ __device__ inline size_t win2_lds_bytes(int dim, int window, int negative, int R) {
  (void)window; (void)negative;
  return win2_rec_bytes() + (size_t)(2 * R + 1) * dim * 4;
}

__device__ __forceinline__ Win2 carve_win2(W2B_LDS int *base, int buf) {
  Win2 L;
  L.w = (W2B_LDS WorkRec *)base;
  L.s = &L.w->step[buf];
  L.win = (W2B_LDS float *)((W2B_LDS char *)base + win2_rec_bytes());
  return L;
}

typedef Col<4> Col4;
__device__ __forceinline__ unsigned col_bits(const Col4 &c) {
  unsigned h = 0;
#pragma unroll
  for (int e = 0; e < 4; e++) h ^= __float_as_uint(c.e[e]) * (2u * e + 3u);
  return h;
}
__device__ __forceinline__ Col4 col_zero() {
  Col4 c;
#pragma unroll
  for (int e = 0; e < 4; e++) c.e[e] = 0.f;
  return c;
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));
// LDS accesses of a thread's own 16-byte column of a resident row
__device__ __forceinline__ Col4 lds_ld(const W2B_LDS float *p) {
  const f32x4_t t = *(const W2B_LDS f32x4_t *)p;
  Col4 c;
  c.e[0] = t.x; c.e[1] = t.y; c.e[2] = t.z; c.e[3] = t.w;
  return c;
}
__device__ __forceinline__ void lds_st(W2B_LDS float *p, const Col4 &c) {
  f32x4_t t;
  t.x = c.e[0]; t.y = c.e[1]; t.z = c.e[2]; t.w = c.e[3];
  *(W2B_LDS f32x4_t *)p = t;
}

// Everything a data thread needs to address the tables, its scratch rows and its columns
template <int MM>
struct Rows {
  static constexpr int M = MM & 7, TB = (MM >> 3) & 1;     // memory mode, table form (1 = per-row descriptors)
  static constexpr bool ATOMIC = ((MM >> 4) & 1) != 0;     // rows 1..atomic_rank of v are updated with atomic adds
  const W2bParams &P;
  long long scratch0;      // first scratch row of this worker
  int nsh;                 // scratch rows per generation = window slots
  int dim, col0;
  bool active;
  float *hot_c, *hot_e;    // this XCD's copies of the hottest rows of v and their entry values (XHot)
  unsigned *hot_l;         // their merge locks
  int nh;
  __device__ __forceinline__ Col4 ld_u(int row) const { return load_col<4, M, TB>(P.u, row, dim, col0, P.tab_bytes); }
  __device__ __forceinline__ Col4 ld_v(int row) const { return load_col<4, M, TB>(P.v, row, dim, col0, P.tab_bytes); }
  __device__ __forceinline__ void st_u(int row, const Col4 &c) const { store_col<4, M, TB>(P.u, row, dim, col0, c, P.tab_bytes); }
  __device__ __forceinline__ void st_v(int row, const Col4 &c) const { store_col<4, M, TB>(P.v, row, dim, col0, c, P.tab_bytes); }
  __device__ __forceinline__ Col4 ld_hot(int k) const { return xhot_ld(hot_c, k, nh, dim, col0); }
  __device__ __forceinline__ void st_hot(int k, const Col4 &c) const { xhot_st(hot_c, k, nh, dim, col0, c); }
  __device__ __forceinline__ void add_v1(int row, int e, float d) const {         // v[row][col0 + e] += d, atomically
    Col<1> c;
    c.e[0] = d;
    add_col<1, TB>(P.v, row, dim, col0 + e, c, P.tab_bytes);
  }
  // scratch ("entry") rows: written with plain stores, read back (rarely) past the L1
  __device__ __forceinline__ Col4 ld_entry(int gen, int slot) const {
    return load_col<4, 0, 1>(P.entry, scratch0 + (long long)gen * nsh + slot, dim, col0, 0u);
  }
  __device__ __forceinline__ void st_entry(int gen, int slot, const Col4 &c) const {
    store_col<4, 1, 1>(P.entry, scratch0 + (long long)gen * nsh + slot, dim, col0, c, 0u);
  }
};

// ---- write-back of one leaving row: exact value if nobody else changed the row, else merge our contribution.
// A row is "ours alone" for the ordinary word: delta-sum, g + (value - entry), is then exactly what the reference's
// shared row would hold.  The most frequent context words (rows 1..P.uavg_rank; chosen like the hot target rows) are
// resident in EVERY worker's window most of the time; hundreds of workers adding their whole private progress to such
// a row over-shoots (every one of those deltas was trained against the same stale row), so for them the row moves a
// step of weight P.xhot_w from its current value towards ours -- the rule of the hot target rows (XHot).
// `keep`: the row stays resident (age-limited residency of those rows, Step2::refresh): the merged value also becomes
// the slot's value, its new entry and checksum.
template <int MM>
__device__ __forceinline__ void retire_finish(const Rows<MM> &A, const Win2 &L, int row, int gen, int slot, unsigned csum_at_entry,
                                              const Col4 &g, const Col4 &rw, bool keep, int lane, int wave) {
  const unsigned now = wave_xor(A.active ? col_bits(g) : 0u);
  const bool untouched = (now == csum_at_entry);                          // wave-uniform
  Col4 o = rw;
  if (!untouched && A.active) {
    if (row <= A.P.uavg_rank) {
#pragma unroll
      for (int e = 0; e < 4; e++) o.e[e] = g.e[e] + A.P.xhot_w * (rw.e[e] - g.e[e]);
    } else {
      const Col4 en = A.ld_entry(gen, slot);
#pragma unroll
      for (int e = 0; e < 4; e++) o.e[e] = g.e[e] + (rw.e[e] - en.e[e]);
    }
  }
  if (A.active) A.st_u(row, o);
  if (keep) {
    const unsigned cs = wave_xor(A.active ? col_bits(o) : 0u);
    if (lane == 0) L.w->csum[slot][wave] = cs;
    if (A.active) {
      if (!untouched) lds_st(L.win + slot * A.dim + A.col0, o);
      A.st_entry(gen, slot, o);
    }
  }
}

// ---- rows leaving the window (sentence boundaries: many at once)
// keep0: entry 0 of the list is a REFRESH (Step2::refresh: the row stays resident) that happens to coincide with several
// admissions, so that the step takes this generic path: the merged value must become the slot's value, entry and checksum
// exactly as on the steady-state path (round 3 passed keep = false here: the row was then always treated as touched by
// others at its real retirement)
template <int MM>
__device__ __forceinline__ void window_retire(const Rows<MM> &A, const Win2 &L, int n_ret, int lane, int wave, bool keep0) {
  for (int i0 = 0; i0 < n_ret; i0 += W2B_RCH) {
    Col4 rw[W2B_RCH], g[W2B_RCH];
#pragma unroll
    for (int i = 0; i < W2B_RCH; i++) {
      rw[i] = col_zero(); g[i] = col_zero();
      if (i0 + i < n_ret && A.active) {
        rw[i] = lds_ld(L.win + L.s->ret_slot[i0 + i] * A.dim + A.col0);
        g[i] = A.ld_u(L.s->ret_row[i0 + i]);
      }
    }
#pragma unroll
    for (int i = 0; i < W2B_RCH; i++)
      if (i0 + i < n_ret) {
        const int s = L.s->ret_slot[i0 + i];
        retire_finish<MM>(A, L, L.s->ret_row[i0 + i], L.s->ret_gen[i0 + i], s, L.w->csum[s][wave], g[i], rw[i], keep0 && i0 + i == 0, lane, wave);
      }
  }
}

// ---- rows entering the window
template <int MM>
__device__ __forceinline__ void window_admit(const Rows<MM> &A, const Win2 &L, int n_adm, int lane, int wave) {
  for (int i0 = 0; i0 < n_adm; i0 += W2B_RCH) {
    Col4 a[W2B_RCH];
#pragma unroll
    for (int i = 0; i < W2B_RCH; i++) {
      a[i] = col_zero();
      if (i0 + i < n_adm && A.active) a[i] = A.ld_u(L.s->adm_row[i0 + i]);
    }
#pragma unroll
    for (int i = 0; i < W2B_RCH; i++)
      if (i0 + i < n_adm) {
        const int s = L.s->adm_slot[i0 + i];
        const unsigned cs = wave_xor(A.active ? col_bits(a[i]) : 0u);
        if (lane == 0) L.w->csum[s][wave] = cs;
        if (A.active) {
          lds_st(L.win + s * A.dim + A.col0, a[i]);
          A.st_entry(L.s->adm_gen[i0 + i], s, a[i]);
        }
      }
  }
}

// --------------------------------------------------------------------------------------------------
// NDW + 1 wavefronts per worker: wavefronts 0..NDW-1 own the embedding columns (data phase); the last one is the
// PRODUCER: it walks the sentence, the LCG ledger, the window bookkeeping and the negative draws ONE STEP
// AHEAD and hands the lists over through a double-buffered LDS record.  The data wavefronts never wait for
// the scalar work of a step (it was 25-30 % of the step time when wavefront 0 did both).
// Barrier discipline (worker_barrier): per step the data wavefronts meet once per target chunk (the cross-wavefront
// sum of the dot products; with s_barrier the producer has to execute those too, after its own work) and all
// wavefronts of the worker meet once at the end of the step.
// UC: the radius is window-1 (the two outermost context rows of a step are register-held).
template <int QM, bool LOSS, int MM, bool UC>
__global__ void __launch_bounds__(W2B_WPG * 64 * (W2B_NDWMAX + 1), W2B_RES_WAVES)
k_train_resident(const W2bParams P, const long long max_positions, const int R, const int NDW,
                 const int lds_ints_per_worker) {
  extern __shared__ int smem[];
  const int WPT = (NDW + 1) * 64;                        // threads per worker
  // which worker of this workgroup: the same for all lanes of a wavefront (WPT is a multiple of 64) -- said so with
  // readfirstlane, or every LDS address and with it the whole control flow would count as divergent
  const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x / WPT);
  W2B_LDS int *const smem_lds = (W2B_LDS int *)smem + half * lds_ints_per_worker;
  const Win2 L0 = carve_win2(smem_lds, 0);
  const Win2 L1 = carve_win2(smem_lds, 1);
  const Win2 &L = L0;                                   // everything that is not double buffered
  W2B_LDS WorkerLds *S = &L.w->S.w;
  W2B_LDS int *s_sen = L.w->sen;
  const int tid = (int)threadIdx.x - half * WPT, lane = tid & 63, wave = tid >> 6;
  const bool producer = (wave == NDW);
  const int wid = (int)blockIdx.x * W2B_WPG + half;
  W2bWorker *G = P.workers + (wid < P.num_threads ? wid : 0);
  const bool valid = wid < P.num_threads && !G->done;
  QParam qp;
  qp.bitlevel = P.bitlevel;
  qp.steps_i = (P.bitlevel >= 4) ? (1 << (P.bitlevel - 1)) : 1;
  qp.steps_f = (float)qp.steps_i;
  const int NS = 2 * R + 1;
  const XHot XH = xhot_here(P);
  const int NH = XH.nv;                     // leading rows of v that live in this XCD's copies
  const Rows<MM> A{P, (long long)wid * 2 * NS, NS, P.dim, tid * 4, !producer && tid * 4 < P.dim, XH.cv, XH.ev, XH.lv, NH};
  const bool active = A.active;
  if (valid) {
    for (int i = tid; i < G->sen_len; i += WPT) s_sen[i] = G->sen[i];
    for (int i = tid; i < NS; i += WPT) { L.w->slot_row[i] = -1; L.w->slot_ref[i] = 0; L.w->pos_slot[i] = 0; L.w->slot_gen[i] = 0; }
    for (int i = tid; i < W2B_NJ; i += WPT) { L.w->ja[i] = P.jump_a[i]; L.w->jc[i] = P.jump_c[i]; }
    if (tid == 0) {
      S->rng = G->rng; S->cursor = G->cursor; S->wc = G->word_count; S->last_wc = G->last_word_count;
      S->sen_len = G->sen_len; S->sen_pos = G->sen_pos; S->override_ = G->first_override;
      S->eof = 0; S->done = 0; S->cw = 0; S->nt = 0; S->alpha = 0.f;
      L.w->S.clo = 0; L.w->S.chi = -1;
      L.w->S.bar_step = 0u; L.w->S.bar_chunk = 0u;
      L.w->S.loss_reg = 0.0;
    }
  }
  __syncthreads();            // the only workgroup-wide barrier: every wavefront of both workers is still here
  if (!valid) return;
  W2B_LDS unsigned *const bar_step = &L.w->S.bar_step, *const bar_chunk = &L.w->S.bar_chunk;
  unsigned n_step = 0u, n_chunk = 0u;       // barriers passed so far (identical in every wavefront of the worker)
#ifdef W2B_PHASE_TIMERS
  const bool timing_ = (wid == 0) && (wave == 0 || wave == NDW) && lane == 0;
  unsigned long long tick_ = __builtin_readcyclecounter();
#endif
  double loss_acc = 0.0;     // producer wavefront only: log-sigmoid terms (the data wavefronts carry nothing around their loop)
  const int W = P.window, K = P.negative;
  // producer registers: the unigram-table gather and the alpha load of the NEXT step are issued at the end
  // of a preparation, so that their latency is not on the producer's critical path either
  int t_pref = 0;
  int pstep = 0;                 // steps prepared so far in this launch (ages of the window slots)
  float alpha_pref = P.starting_alpha;
  // (whether the two prefetched values are valid is DERIVED inside prepare() from pstep and the sentence length; as
  // two flags carried from one preparation to the next they ended up in scratch memory: two loads through the vector
  // memory pipe per step, each behind a wait for everything the producer had in flight)
  // data-wavefront registers: the row that enters the window at the next step, loaded one step early
  Col4 apre = col_zero();
  int apre_row = -1;
  // data wavefronts: which of the XCD's hot rows this worker brings up to date next (workers take turns: workgroup b
  // runs on XCD b % 8, so the workers of one XCD start at different rows)
  int merge_cursor = (wid >> 3) * P.xhot_m;

  // ---- the preparation of one pass (producer wavefront only; all 64 lanes, wave-uniform control flow)
  auto prepare = [&](const Win2 &O, const bool last) {
      unsigned long long rng = S->rng;
      long long cursor = S->cursor, wc = S->wc, last_wc = S->last_wc;
      int sen_len = S->sen_len, sen_pos = S->sen_pos, ovr = S->override_, eof = S->eof;
      // Every preparation after the first one of a launch follows a preparation that trained a position (a stop pass is
      // never followed by another preparation), and that one has requested alpha and -- when its sentence went on, which
      // is what a non-zero length at entry says -- the table entries of this step's negative draws.
      const bool alpha_pref_ok = pstep > 0, pref_ok = pstep > 0 && sen_len != 0;
      int done = 0, cw = 0, nt = 0, uc_n = 0, nck = 0, next_row = -1, cdup = 0;
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
            const long long wca_all = w2b_global_progress(P, (long long)wca);
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
      int clo = L.w->S.clo, chi = L.w->S.chi, n_ret = 0, n_adm = 0;
      if (new_sentence || !train) {
        if (lane < NS) L.w->slot_ref[lane] = 0;              // every resident position belonged to the old sentence
        clo = 0; chi = -1;
        W2B_WAVE_SYNC();
      } else {
        for (int q = clo; q <= chi; q++) {                // positions that leave: [clo, lo) and (hi, chi]
          if (q >= lo && q <= hi) { q = hi; continue; }
          if (lane == 0) L.w->slot_ref[L.w->pos_slot[q % NS]]--;
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
          if (pass == 1 && L.w->pos_slot[q % NS] >= 0) continue;             // resolved in pass 1
          const int w = s_sen[q];
          const bool match = (lane < NS) && (L.w->slot_row[lane] == w);
          const unsigned long long mm = __ballot(match);
          int s = -1;
          if (mm) {                                                       // the word is resident: share its slot
            s = __ffsll((long long)mm) - 1;
            if (lane == 0) L.w->slot_ref[s]++;
          } else if (pass == 1) {
            const unsigned long long fr = __ballot((lane < NS) && (L.w->slot_row[lane] == -1));
            if (fr) s = __ffsll((long long)fr) - 1;
            else {                                                         // recycle the slot of a leaving row
              const unsigned long long pend = __ballot((lane < NS) && (L.w->slot_ref[lane] == 0));
              s = __ffsll((long long)pend) - 1;
              if (lane == 0) { O.s->ret_slot[n_ret] = s; O.s->ret_row[n_ret] = L.w->slot_row[s]; O.s->ret_gen[n_ret] = L.w->slot_gen[s]; }
              n_ret++;
            }
            if (lane == 0) {
              const int gen = L.w->slot_gen[s] ^ 1;          // the scratch row of the previous tenant stays readable
              L.w->slot_gen[s] = gen;
              L.w->slot_row[s] = w; L.w->slot_ref[s] = 1;
              L.w->slot_born[s] = pstep;
              O.s->adm_slot[n_adm] = s; O.s->adm_row[n_adm] = w; O.s->adm_gen[n_adm] = gen;
            }
            n_adm++;
          }
          if (lane == 0) L.w->pos_slot[q % NS] = s;
          W2B_WAVE_SYNC();
        }
      }
      {                                                                  // whatever is unreferenced leaves
        const bool leaving = (lane < NS) && (L.w->slot_row[lane] != -1) && (L.w->slot_ref[lane] == 0);
        const unsigned long long ml = __ballot(leaving);
        if (leaving) {
          const int k = n_ret + __popcll(ml & lane_lt_mask(lane));
          O.s->ret_slot[k] = lane; O.s->ret_row[k] = L.w->slot_row[lane]; O.s->ret_gen[k] = L.w->slot_gen[lane];
          L.w->slot_row[lane] = -1;
        }
        n_ret += __popcll(ml);
        W2B_WAVE_SYNC();
      }
      // Age-limited residency of the most frequent context words (rows 1..uavg_rank): such a word is in the window most
      // of the time and would otherwise stay private to this worker for hundreds of steps.  When nothing else leaves in
      // this step, the oldest one past win_refresh steps is merged with memory in place (retire_finish with `keep`).
      int refresh = 0;
      if (train && n_ret == 0 && P.win_refresh > 0 && P.uavg_rank > 0) {
        const bool cand = (lane < NS) && (L.w->slot_row[lane] > 0) && (L.w->slot_row[lane] <= P.uavg_rank) &&
                          (L.w->slot_ref[lane] > 0) && (pstep - L.w->slot_born[lane] >= P.win_refresh);
        const unsigned long long mc = __ballot(cand);
        if (mc) {
          const int sr = __ffsll((long long)mc) - 1;
          if (lane == 0) {
            O.s->ret_slot[0] = sr; O.s->ret_row[0] = L.w->slot_row[sr]; O.s->ret_gen[0] = L.w->slot_gen[sr];
            L.w->slot_born[sr] = pstep;
          }
          n_ret = 1;
          refresh = 1;
          W2B_WAVE_SYNC();
        }
      }
      pstep++;
      if (train) {
        const int hiA = 2 * W + 1 - b;
        for (int a0 = b; a0 < hiA; a0 += 64) {                          // ref :431-436
          const int a = a0 + lane;
          const int c = p - W + a;
          const bool ok = (a < hiA) && (a != W) && (c >= 0) && (c < sen_len);
          const unsigned long long m = __ballot(ok);
          int slot = 0;
          if (ok) {
            if (c >= lo && c <= hi) slot = L.w->pos_slot[c % NS];
            else {                                     // outside the radius (|c - p| == window): resident anyway?
              const int w = s_sen[c];
              slot = (c < p) ? -1 : -2;                // provisional: register-held row (left / right)
              for (int s2 = 0; s2 < NS; s2++) slot = (L.w->slot_row[s2] == w) ? s2 : slot;
            }
          }
          if (ok) O.s->cslot[cw + __popcll(m & lane_lt_mask(lane))] = slot;
          cw += __popcll(m);
        }
        W2B_WAVE_SYNC();
        if (cw > 0 && R < W) {
          // The radius is window-1: the two outermost context positions (only present when b == 0) are
          // not resident.  They are the first / last entry of the context list; each one that is not
          // resident through another position becomes a register-held row of this step.
          const int first = O.s->cslot[0], lastc = O.s->cslot[cw - 1];
          int wl = -1;
          if (first == -1) { wl = s_sen[p - W]; if (lane == 0) O.s->uc_row[0] = wl; uc_n = 1; }
          if (lastc == -2) {
            const int wr = s_sen[p + W];
            if (uc_n == 1 && wr == wl) { if (lane == 0) O.s->cslot[cw - 1] = -1; }       // same word on both ends
            else {
              if (lane == 0) { O.s->uc_row[uc_n] = wr; O.s->cslot[cw - 1] = -1 - uc_n; }
              uc_n++;
            }
          }
          W2B_WAVE_SYNC();
        }
        if (cw > 0) {
          // does a slot occur twice in the context list (a word at two window positions)?  Then phase C has to apply
          // its updates strictly in order; otherwise it may keep several LDS rows in flight.
          int dupf = 0;
          for (int j0 = 0; j0 < cw; j0 += 64) {
            const int mine = (j0 + lane < cw) ? O.s->cslot[j0 + lane] : -100 - lane;
            for (int j = 0; j < min(64, cw - j0); j++) {
              const int sj = __builtin_amdgcn_readlane(mine, j);
              dupf |= (__ballot(lane > j && mine == sj) != 0ull) ? 1 : 0;
            }
            if (cw > 64) dupf = 1;                   // (window > 32: not worth a second pass)
          }
          cdup = dupf;
        }
        if (cw > 0) {                                                    // ref :450-460
          int cnt = 0;
          for (int d0 = 1; d0 <= K; d0 += 64) {
            const int d = d0 + lane;
            bool keep = false;
            int t = 0;
            if (d <= K) {
              const unsigned long long x = (L.w->ja[d] * rng + L.w->jc[d]);
              t = (pref_ok && d0 == 1) ? t_pref : P.table[fast_mod(x >> 16, (unsigned long long)P.table_size, P.table_magic)];
              if (t == 0) t = (int)(x % (unsigned long long)(P.vocab_size - 1)) + 1;
              keep = (t != word);
            }
            const unsigned long long m = __ballot(keep);
            if (keep) O.s->tgt[1 + cnt + __popcll(m & lane_lt_mask(lane))] = t;
            cnt += __popcll(m);
          }
          if (lane == 0) O.s->tgt[0] = word;
          nt = 1 + cnt;
          rng = (L.w->ja[K] * rng + L.w->jc[K]);
          alpha = alpha_set ? alpha_own
                            : (alpha_pref_ok ? alpha_pref
                                             : __hip_atomic_load(&P.shared->alpha, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT));
          nck = prep_lists<W2B_RT, W2B_LDS int *>(O.s->tgt, L.w->prev, O.s->cend, nt, (W2B_LDS int *)nullptr, (W2B_LDS int *)nullptr, 0, lane);
        }
        const int nq = p + 1 + R;                                        // enters the window at the next step
        next_row = (p + 1 < sen_len && nq < sen_len) ? s_sen[nq] : -1;
        sen_pos++;                                                       // ref :505-509
        if (sen_pos >= sen_len) sen_len = 0;
        // ---- prefetch for the next step (valid unless the next step starts with a sentence read, whose
        // sub-sampling draws come first in the LCG ledger)
        if (sen_len != 0 && lane < K) {                                       // lane l serves draw d = l + 1
          const unsigned long long xb = rng * W2B_LCG_A + W2B_LCG_C;     // the next step's window draw
          const unsigned long long x = (L.w->ja[lane + 1] * xb + L.w->jc[lane + 1]);
          t_pref = P.table[fast_mod(x >> 16, (unsigned long long)P.table_size, P.table_magic)];
        }
        alpha_pref = __hip_atomic_load(&P.shared->alpha, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (lane == 0) {
        S->rng = rng; S->cursor = cursor; S->wc = wc; S->last_wc = last_wc;
        S->sen_len = sen_len; S->sen_pos = sen_pos; S->override_ = ovr; S->eof = eof;
        L.w->S.clo = lo; L.w->S.chi = hi;
        O.s->st.stop = (done || last) ? 1 : 0; O.s->st.cw = cw; O.s->st.nt = nt; O.s->st.uc_n = uc_n;
        O.s->st.n_ret = n_ret; O.s->st.n_adm = n_adm; O.s->st.next_row = next_row; O.s->st.nck = (cw > 0) ? nck : 0;
        O.s->st.alpha = alpha;
        O.s->st.cdup = cdup;
        O.s->st.refresh = refresh;
        if (done) S->done = 1;
      }
  };

  // Two role-specific step loops instead of one loop with a role branch inside: a wavefront has ONE register
  // allocation, and inside a common loop whatever one role carries around the loop is live in the other role's
  // branch too (the producer's bookkeeping cost the data path 20 VGPRs and vice versa).  Both loops execute the
  // same s_barrier sequence per step.
  if (producer) {
    prepare(L0, max_positions == 0);
    worker_barrier(bar_step, (unsigned)(NDW + 1) * ++n_step, lane);
    for (long long it = 0;; ++it) {
      const Win2 &I = (it & 1) ? L1 : L0;                 // this step's lists
      const bool stop = I.s->st.stop != 0;
      W2B_TICK(1);
      if (LOSS && it > 0) {
        // The log-sigmoid bookkeeping (ref :480-483) of the PREVIOUS step happens here, off the data wavefronts'
        // registers (expf/logf cost them ~40 VGPRs): wavefront 0 left the dot products f of that step's targets in the
        // step buffer, which nobody writes again before this wavefront has passed the next end-of-step barrier.
        const Win2 &Q = (it & 1) ? L0 : L1;               // the previous step's lists
        const int ntp = Q.s->st.cw > 0 ? Q.s->st.nt : 0;
        for (int i = lane; i < ntp; i += 64) {
          const float f = Q.s->lossf[i];
          const float dp = (i == 0) ? f : -f;                           // target 0 is the centre word (label 1)
          float sg;
          if (dp > 6.f) sg = 1.f;
          else if (dp < -6.f) sg = 1e-9f;
          else sg = 1.f / (1.f + expf(-dp));
          loss_acc += (double)logf(sg);
        }
      }
      W2B_TICK(2);
      if (!stop) prepare((it & 1) ? L0 : L1, it + 1 == max_positions);
      W2B_TICK(0);
#if W2B_WPG == 1
      for (int i = 0; i < I.s->st.nck; i++) __syncthreads();               // s_barrier counts every wavefront: the chunk barriers too
#endif
      worker_barrier(bar_step, (unsigned)(NDW + 1) * ++n_step, lane);   // lists of the next step are published; this step is done
      if (stop) break;                                    // (a stop pass trains nothing: no loss terms are left over)
    }
  } else {
    worker_barrier(bar_step, (unsigned)(NDW + 1) * ++n_step, lane);
    for (long long it = 0;; ++it) {
      const Win2 &I = (it & 1) ? L1 : L0;                 // this step's lists
      const bool stop = I.s->st.stop != 0;
      W2B_TICK(9);
      W2B_COUNT(10);
      // ---------------- data phase.  cslot[j] >= 0: LDS slot; -1-k: register-held row k.
      const W2B_LDS int *const tgt = I.s->tgt, *const cend = I.s->cend, *const cslot = I.s->cslot;
      const bool word_step = !stop && I.s->st.cw > 0;
      const int nt = I.s->st.nt, cw = I.s->st.cw, dim = P.dim, col0 = A.col0;
      const int uc_n = UC ? I.s->st.uc_n : 0;
      const float alpha = I.s->st.alpha;
      const float ar2 = (2.f * alpha) * P.reg;
      const int n_ret = I.s->st.n_ret, n_adm = I.s->st.n_adm;
      bool deferred = false;                    // steady state: the leaving row is merged back AFTER the step
      int d_row = -1, d_gen = 0, d_slot = 0;
      unsigned d_csum = 0;
      Col4 d_g = col_zero(), d_rw = col_zero();
      Col4 avg = col_zero(), err = col_zero(), ur0 = col_zero(), ur1 = col_zero();
      float regsq = 0.f;
      int start = 0, chunk = 0, end = word_step ? cend[0] : 0, par = 0;
      // The target rows of the current chunk.  Deliberately "defined" by an empty asm: the registers hold whatever
      // they held (slots beyond a chunk are masked out below), there is no instruction in front of the loads that
      // could make them wait (a zero-fill would be a write-after-write hazard against outstanding memory
      // operations), and -- not being carried around the step loop -- they cost the producer wavefront's branch
      // nothing (one register allocation serves both roles).
      Col4 x[W2B_RT];
      int rows[W2B_RT];
#pragma unroll
      for (int i = 0; i < W2B_RT; i++) {
        rows[i] = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) asm volatile("" : "=v"(x[i].e[e]));
      }
      // ONE loop over the target chunks with ONE load site (the rows of a chunk live in the same registers every
      // time); what happens once per step -- window exchange, phase A -- sits behind the first chunk's loads, which
      // have the longest way to go.
      for (;;) {
        if (word_step) {
          // a row repeated from an earlier chunk is re-read after that chunk's store (program order)
          const int mine = tgt[min(start + lane, nt - 1)];
#pragma unroll
          for (int i = 0; i < W2B_RT; i++) rows[i] = __builtin_amdgcn_readlane(mine, i);
          if (active) {
#pragma unroll
            for (int i = 0; i < W2B_RT; i++) {
              const unsigned hk = (unsigned)(rows[i] - 1);      // hot rows at this XCD's copy, the others at their master row
              if (start + i < end) {
                if (hk < (unsigned)NH) x[i] = A.ld_hot((int)hk);
                else x[i] = A.ld_v(rows[i]);
              }
            }
          }
        }
        int first = chunk;
        asm volatile("" : "+s"(first));          // opaque: keeps the compiler from peeling the first trip (two load sites)
        if (first == 0) {
          // ---- window exchange
          if (n_ret <= 1 && n_adm <= 1) {
            if (n_ret == 1) {
              d_slot = I.s->ret_slot[0];
              d_row = I.s->ret_row[0];
              d_gen = I.s->ret_gen[0];
              d_csum = L.w->csum[d_slot][wave];
              if (active) {
                d_rw = lds_ld(L.win + d_slot * dim + col0);
                d_g = A.ld_u(d_row);                                               // consumed after the step: no stall
              }
              deferred = true;
            }
            if (n_adm == 1) {
              const int s = I.s->adm_slot[0], row = I.s->adm_row[0];
              Col4 a = apre;                                                       // loaded during the previous step
              if (row != apre_row) {
                a = col_zero();
                if (active) a = A.ld_u(row);
              }
              const unsigned cs = wave_xor(active ? col_bits(a) : 0u);
              if (lane == 0) L.w->csum[s][wave] = cs;
              if (active) {
                lds_st(L.win + s * dim + col0, a);
                A.st_entry(I.s->adm_gen[0], s, a);
              }
            }
            // a register-held outer row of this step that is the row leaving right now must see the merge
            if (UC && deferred && ((uc_n > 0 && I.s->uc_row[0] == d_row) || (uc_n > 1 && I.s->uc_row[1] == d_row))) {
              retire_finish<MM>(A, L, d_row, d_gen, d_slot, d_csum, d_g, d_rw, I.s->st.refresh != 0, lane, wave);
              deferred = false;
            }
            // prefetch the row that enters at the next step (never one whose store is still ahead of us)
            const int nr = I.s->st.next_row;
            const bool is_uc = UC && ((uc_n > 0 && I.s->uc_row[0] == nr) || (uc_n > 1 && I.s->uc_row[1] == nr));
            apre_row = (nr >= 0 && !(deferred && nr == d_row) && !is_uc) ? nr : -1;
            if (apre_row >= 0 && active) apre = A.ld_u(apre_row);
          } else {
            if (n_ret) window_retire<MM>(A, I, n_ret, lane, wave, I.s->st.refresh != 0);
            if (n_adm) window_admit<MM>(A, I, n_adm, lane, wave);
            apre_row = -1;
          }
          if (word_step) {
            // the (at most two) context rows outside the radius live in registers for this step
            if (UC && active && uc_n > 0) ur0 = A.ld_u(I.s->uc_row[0]);
            if (UC && active && uc_n > 1) ur1 = A.ld_u(I.s->uc_row[1]);
            // ---- phase A from LDS (ref :431-449), window order
            if (active) {
              for (int j0 = 0; j0 < cw; j0 += 4) {          // four rows requested together, summed in window order
                Col4 r[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                  const int s = cslot[min(j0 + k, cw - 1)];
                  if (!UC || s >= 0) {
                    r[k] = lds_ld(L.win + s * dim + col0);
                  } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) r[k].e[e] = (s == -1) ? ur0.e[e] : ur1.e[e];
                  }
                }
#pragma unroll
                for (int k = 0; k < 4; k++)
                  if (j0 + k < cw) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                      const float q = quant<QM>(r[k].e[e], qp);
                      avg.e[e] += q;
                      if (LOSS) regsq += q * q;
                    }
                  }
              }
              const float cwf = (float)cw;
#pragma unroll
              for (int e = 0; e < 4; e++) avg.e[e] = avg.e[e] / cwf;
            }
          }
          // The row that left the window is merged back here: its current value was requested before the admit
          // above and has had phase A to arrive (memory returns in order, the target rows are needed next anyway),
          // and its registers are free before the dot products need room.
          if (deferred) retire_finish<MM>(A, L, d_row, d_gen, d_slot, d_csum, d_g, d_rw, I.s->st.refresh != 0, lane, wave);
          deferred = false;
        }
        if (!word_step) break;
        W2B_TICK(4);

        // ---- phase B (ref :450-492): the W2B_RT rows of this chunk are in registers
        const int n = end - start;
        W2B_LDS float *red = &L.w->red[par][0][0];
        // partial dot products, W2B_RB rows at a time: W2B_RB independent reduction chains interleave, and the
        // scheduling barrier keeps the compiler from forming all W2B_RT x 4 products first (that is 100 live VGPRs)
#pragma unroll
        for (int i0 = 0; i0 < W2B_RT; i0 += W2B_RB) {
          float p[W2B_RB];
#pragma unroll
          for (int k = 0; k < W2B_RB; k++) {
            const int i = i0 + k < W2B_RT ? i0 + k : W2B_RT - 1;
            float t[4];
#pragma unroll
            for (int e = 0; e < 4; e++) t[e] = avg.e[e] * quant<QM>(x[i].e[e], qp);       // ref :466, re-associated as a tree
            const float s = (t[0] + t[1]) + (t[2] + t[3]);
            p[k] = (active && i < n) ? s : 0.f;    // slots beyond the chunk hold stale rows: never written (no register
                                                   // hazard in front of the next loads), masked here
          }
#pragma unroll
          for (int k = 0; k < W2B_RB; k++) p[k] = wave_sum(p[k]);
          if (lane == 0) {
#pragma unroll
            for (int k = 0; k < W2B_RB; k++)
              if (i0 + k < n && i0 + k < W2B_RT) red[(i0 + k) * W2B_NDWMAX + wave] = p[k];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // The row prefetched for the next step's admit has arrived by now (memory returns in order and the target
        // rows were requested later).  Saying so here -- an empty asm that "redefines" it -- keeps the wait for it out
        // of the next step's window exchange, where it would be a wait for every load issued in between: that
        // step's whole chunk of target rows, before phase A instead of after it.
#pragma unroll
        for (int e = 0; e < 4; e++) asm volatile("" : "+v"(apre.e[e]));
        W2B_TICK(5);
        worker_barrier(bar_chunk, (unsigned)NDW * ++n_chunk, lane);
        W2B_TICK(6);
        float gl = 0.f;
        if (lane < n) {
          float f = 0.f;
          for (int w = 0; w < NDW; w++) f += red[lane * W2B_NDWMAX + w];       // the plain kernel's order over wavefronts
          const float label = (start + lane == 0) ? 1.f : 0.f;                 // target 0 is the centre word
          float g;
          if (f > 6.f) g = (label - 1.f) * alpha;
          else if (f < -6.f) g = label * alpha;
          else g = (label - P.exp_table[(int)((f + 6.f) * 83.f)]) * alpha;
          gl = g;
          if (LOSS && wave == 0) I.s->lossf[start + lane] = f;     // the producer wavefront books log(sigmoid) next step
        }
        if (LOSS && P.reg != 0.f) {            // reg * sum q^2 over the chunk's target rows (ref :469,481), re-derived
          float s2 = 0.f;                      // from the rows in registers: one running sum, one reduction per chunk
#pragma unroll
          for (int i = 0; i < W2B_RT; i++)
            if (i < n) {
#pragma unroll
              for (int e = 0; e < 4; e++) {
                float xv = x[i].e[e];
                if (QM >= 2) asm volatile("" : "+v"(xv));     // opaque: do not keep the dot product's quantized rows alive
                const float q = quant<QM>(xv, qp);
                s2 += q * q;
              }
            }
          s2 = wave_sum(active ? s2 : 0.f);
          if (lane == 0) __hip_atomic_fetch_add(&L.w->S.loss_reg, -(double)(P.reg * s2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // error accumulation + row update, in target order (ref :486-491)
#pragma unroll
        for (int i = 0; i < W2B_RT; i++) {
          if (i < n) {
            const float g = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gl), i));
            const unsigned hk = (unsigned)(rows[i] - 1);
            if (active) {
              if (Rows<MM>::ATOMIC && !(hk < (unsigned)NH) && rows[i] <= P.atomic_rank) {
                // a frequent row below the hot ones (w2b_tuning.atomic_rank): its delta is added atomically to the row
#pragma unroll
                for (int e = 0; e < 4; e++) {
                  float xv = x[i].e[e];
                  if (QM != 0) asm volatile("" : "+v"(xv));
                  err.e[e] += g * quant<QM>(xv, qp);
                  A.add_v1(rows[i], e, g * avg.e[e] - ar2 * xv);
                }
              } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                  float xv = x[i].e[e];
                  // opaque copy: re-derive the quantized value here instead of keeping 4 extra registers per row alive
                  if (QM != 0) asm volatile("" : "+v"(xv));
                  err.e[e] += g * quant<QM>(xv, qp);
                  x[i].e[e] = xv + (g * avg.e[e] - ar2 * xv);
                }
                if (hk < (unsigned)NH) A.st_hot((int)hk, x[i]);       // a hot row: this XCD's copy
                else A.st_v(rows[i], x[i]);
              }
            }
          }
        }
        W2B_TICK(7);
        start = end;
        if (start >= nt) break;
        end = cend[++chunk];
        par ^= 1;
      }

      if (word_step) {
        // ---- phase C on the resident rows (ref :494-503), window order; duplicates hit the same slot twice
        if (active && !UC && !I.s->st.cdup) {
          // no slot twice: the read-modify-writes are independent, four rows in flight
          for (int j0 = 0; j0 < cw; j0 += 4) {
            Col4 w[4];
#pragma unroll
            for (int k = 0; k < 4; k++) w[k] = lds_ld(L.win + cslot[min(j0 + k, cw - 1)] * dim + col0);
#pragma unroll
            for (int k = 0; k < 4; k++)
              if (j0 + k < cw) {
#pragma unroll
                for (int e = 0; e < 4; e++) w[k].e[e] = w[k].e[e] + (err.e[e] - ar2 * w[k].e[e]);
                lds_st(L.win + cslot[j0 + k] * dim + col0, w[k]);
              }
          }
        } else if (active) {
          for (int j = 0; j < cw; j++) {
            const int s = cslot[j];
            if (!UC || s >= 0) {
              Col4 w0 = lds_ld(L.win + s * dim + col0);
#pragma unroll
              for (int e = 0; e < 4; e++) w0.e[e] = w0.e[e] + (err.e[e] - ar2 * w0.e[e]);
              lds_st(L.win + s * dim + col0, w0);
            } else if (s == -1) {
#pragma unroll
              for (int e = 0; e < 4; e++) ur0.e[e] = ur0.e[e] + (err.e[e] - ar2 * ur0.e[e]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; e++) ur1.e[e] = ur1.e[e] + (err.e[e] - ar2 * ur1.e[e]);
            }
          }
          if (UC && uc_n > 0) A.st_u(I.s->uc_row[0], ur0);
          if (UC && uc_n > 1) A.st_u(I.s->uc_row[1], ur1);
        }
        if (LOSS && P.reg != 0.f) {
          const float s = wave_sum(regsq);
          if (lane == 0)                                              // ref :437-445 (summed over the window)
            __hip_atomic_fetch_add(&L.w->S.loss_reg, -(double)(P.reg * s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      if (NH > 0 && !stop && (it & (P.hot_period - 1)) == P.hot_period - 1) {
        // this worker's turn: xhot_m of the XCD's hot rows meet their master rows (the copies of a launch are folded
        // into the masters by k_xhot_fold afterwards, so a stop pass has nothing to do)
        for (int j = 0; j < P.xhot_m && j < NH; j++)
          xhot_merge_row<Rows<MM>::M, Rows<MM>::TB>(P.v, A.hot_c, A.hot_e, A.hot_l, (merge_cursor + j) % NH, NH, dim, col0, active,
                                                    wave, lane, P.tab_bytes, P.xhot_w);
        merge_cursor += P.xhot_m;
      }
      W2B_TICK(8);
      worker_barrier(bar_step, (unsigned)(NDW + 1) * ++n_step, lane);   // lists of the next step are published; this step is done
      if (stop) break;
    }
  }
  const int sl = S->sen_len;
  for (int i = tid; i < sl; i += WPT) G->sen[i] = s_sen[i];
  if (LOSS && producer) {       // producer lanes hold the log-sigmoid terms; the regularisation terms were summed in LDS
    const double lsum = wave_sum_d(loss_acc);
    if (lane == 0) {
      atomicAdd(&G->loss, lsum + L.w->S.loss_reg);
      atomicAdd(&P.shared->loss_epoch, lsum + L.w->S.loss_reg);       // what w2b_epoch_poll reports without a per-worker copy
    }
  }
  if (tid == 0) {
    G->rng = S->rng; G->cursor = S->cursor; G->word_count = S->wc; G->last_word_count = S->last_wc;
    G->sen_len = S->sen_len; G->sen_pos = S->sen_pos; G->first_override = S->override_;
    if (S->done) { G->done = 1; atomicAdd(&P.shared->workers_done, 1); }
  }
}

}  // namespace

static int win2_threads(int dim) { return (((dim / 4) + 63) / 64 + 1) * 64; }

// Geometry of the sentence-resident kernel for a shape: radius (-1: use the plain kernel).  Two workers (one workgroup
// each) share the 160 KiB of a CU.
int w2b_resident_plan(int dim, int window, int negative) {
  if (dim % 4 != 0 || dim > 4 * 64 * W2B_NDWMAX) return -1;     // 16-byte columns, at most 4 data wavefronts
  if (window > W2B_WMAX || negative + 1 > W2B_TMAX) return -1;   // capacities of the worker's LDS record (WorkRec / StepRec)
  const size_t budget = 80 * 1024;
  if (win2_lds_bytes(dim, window, negative, window) <= budget) return window;
  if (window >= 2 && win2_lds_bytes(dim, window, negative, window - 1) <= budget) return window - 1;
  return -1;
}

// can the sentence-resident kernel update rows 1..atomic_rank atomically for this launch?  (instantiated for the
// small-table form with radius == window only; otherwise the plain kernel, which does it at run time, takes over)
bool w2b_resident_atomic_ok(const W2bParams &p, int R) { return p.tab_bytes != 0 && R == p.window; }

// rows of scratch ("entry") memory per worker
long long w2b_resident_scratch_rows(int R) { return 2ll * (2 * R + 1); }

// workgroups of the sentence-resident kernel that are resident per CU (occupancy query of the instantiation
// that would run)
static size_t win2_lds_per_worker(const W2bParams &p, int R) {
  return (win2_lds_bytes(p.dim, p.window, p.negative, R) + 15) & ~(size_t)15;
}

// WORKERS of the sentence-resident kernel that are resident per CU (occupancy query of the instantiation that would
// run: workgroups per CU x W2B_WPG workers per workgroup)
int w2b_resident_per_cu(const W2bParams &p, int R, bool loss) {
  const size_t lds = W2B_WPG * win2_lds_per_worker(p, R);
  const int threads = W2B_WPG * win2_threads(p.dim);
  int nb = 0;
  (void)dispatch_q(p.bitlevel, [&](auto qm) -> hipError_t {
    constexpr int QM = decltype(qm)::value;
    // (the table-form / radius variants share the register budget)
    if (loss) return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_train_resident<QM, true, 0, false>, threads, lds);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_train_resident<QM, false, 0, false>, threads, lds);
  });
  return (nb > 0 ? nb : 1) * W2B_WPG;
}

// Coherent rows only (memory mode 0): with relaxed rows the launcher of the trainer picks the plain kernel.
hipError_t w2b_launch_resident(const W2bParams &p, long long max_positions, int R, bool loss, hipStream_t s, bool debug) {
  const int wthreads = win2_threads(p.dim);  // data wavefronts (one thread per 16-byte column) + 1 producer wavefront
  const int NDW = wthreads / 64 - 1;
  const size_t wlds = win2_lds_per_worker(p, R);
  const int threads = W2B_WPG * wthreads, grid = (p.num_threads + W2B_WPG - 1) / W2B_WPG;
  const size_t lds = W2B_WPG * wlds;
  const int lds_ints = (int)(wlds / 4);
  static bool reported = false;
  if (!reported && debug) {
    reported = true;
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_train_resident<1, false, 0, false>, threads, lds);
    fprintf(stderr, "w2b debug: sentence-resident kernel R=%d hot=%d lds=%zu B/worker threads=%d/worker, %d workers per workgroup, resident workgroups/CU=%d\n", R, p.xhot ? p.xhot_v : 0, wlds, wthreads, W2B_WPG, nb);
  }
  return dispatch_q(p.bitlevel, [&](auto qm) -> hipError_t {
    constexpr int QM = decltype(qm)::value;
    // template MM carries the memory mode in bits 0-2 (0: agent-scope rows), "tables >= 2 GiB" (per-row descriptors) in bit 3
    // and "atomic adds for rows 1..atomic_rank" in bit 4 (small-table form and radius == window only: w2b_resident_atomic_ok)
#define W2B_LAUNCH_R(LOSS, MMV, UCV) hipLaunchKernelGGL((k_train_resident<QM, LOSS, MMV, UCV>), dim3(grid), dim3(threads), lds, s, p, max_positions, R, NDW, lds_ints)
#define W2B_LAUNCH_R2(MMV, UCV) do { if (loss) W2B_LAUNCH_R(true, MMV, UCV); else W2B_LAUNCH_R(false, MMV, UCV); } while (0)
    if (R < p.window) { if (p.tab_bytes) W2B_LAUNCH_R2(0, true); else W2B_LAUNCH_R2(8, true); }
    else if (p.atomic_rank > 0 && p.tab_bytes) W2B_LAUNCH_R2(16, false);
    else { if (p.tab_bytes) W2B_LAUNCH_R2(0, false); else W2B_LAUNCH_R2(8, false); }
#undef W2B_LAUNCH_R2
#undef W2B_LAUNCH_R
    return hipGetLastError();
  });
}
