#pragma once
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
#define W2B_T 13   // most target rows kept in registers per chunk (LDS sizing); see TFor below
#endif
// target rows per chunk of the plain kernels: 13 (negative=24 -> 25 targets = 13 + 12), the widest chunk that stays
// within 128 VGPRs without spilling -- with and without the loss bookkeeping (round 4: the log-sigmoid terms are
// booked from the f values parked in LDS after phase C, when the chunk's registers are free, and the reg * sum q^2
// terms share one accumulator per thread; rounds 1-3 ran the loss-computing instantiation with 9-row chunks)
#ifndef W2B_T_LOSS
#define W2B_T_LOSS W2B_T
#endif
template <bool LOSS> struct TFor { static constexpr int value = LOSS ? W2B_T_LOSS : W2B_T; };
#ifndef W2B_CA
#define W2B_CA 8   // context rows loaded per sub-chunk
#endif
#ifndef W2B_STASH
#define W2B_STASH 8 // context rows whose raw fp32 columns stay in LDS between phase A and phase C
#endif
#ifndef W2B_MINWAVES
#define W2B_MINWAVES 4  // waves per SIMD the 256-thread kernels are register-allocated for (= workgroups per CU)
#endif

namespace {

// ------------------------------------------------------------------------------------ quantizer
// QM: 0 identity, 1 one bit, 2 two bits, 3 generic run-time bitlevel (>=3)
struct QParam { int bitlevel; int steps_i; float steps_f; };

template <int QM>
__device__ __forceinline__ float quant(float x, const QParam &q) {
  if (QM == 0) return x;
  if (QM == 2) {                                       // ref :80,:91-94 in four instructions (|x| is an operand modifier):
    const float lvl = (__builtin_fabsf(x) <= .5f) ? .25f : .75f;   // mag = x * sgn = |x|; NaN fails the compare -> .75
    return (x < 0.f) ? -lvl : lvl;                     // +0, -0, NaN -> + (ref :80)
  }
  const float sgn = (x < 0.f) ? -1.f : 1.f;          // +0, -0, NaN -> +1 (ref :80)
  if (QM == 1) return sgn / 3.f;                      // ref :85-87
  const float mag = x * sgn;
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
// MM (memory mode, run-time selectable per trainer): 0 = sc1, agent scope: coherent between the eight
// XCD L2s (Hogwild as on a cache-coherent CPU; the default); 1 = plain cached accesses (relaxed: a hot
// row is private to an XCD's L2 / a CU's L1 until it is evicted or the launch ends); 2 = nontemporal.
// experimental: 2 = nontemporal (L1-bypassing, L2-cached) loads + sc1 write-through stores;
//               3 = plain loads + sc1 write-through stores
// MM 4 = coherent rows (as 0) + EXACT serial reduction: the dot product of ref :461-467 is accumulated in the
// reference's own order (c = 0 .. D-1, one rounding per add) instead of the wavefront tree, which makes a
// single-worker run bit-identical to the CPU program (w2b_config.exact_reduction; parity mode, not a fast path).
#define W2B_MM_EXACT 4
#define W2B_EXACT_COLS 256    // columns whose products sit in LDS at a time in the exact mode
// MM 5 (W2B_MM_XCD) = nt loads + nt stores: past the CU's L1, served by and kept in the XCD's L2 -- the per-XCD copies
// of the hottest rows (XHot below)
// MM 6 = nt loads + PLAIN (write-back) stores.  Round 6: what the per-XCD copies of the hot rows are stored with.  On gfx950 an
// nt store is written through to memory (round-5 counters: the copies took 19 GB of reads per launch off the fabric and only
// 2 GB of writes); a plain store leaves the line dirty in the XCD's L2 -- the only cache that serves a copy during a launch
// (the loads stay nt: past the CU's L1) -- and it reaches memory when it is evicted or the kernel ends (k_xhot_fold runs
// behind a kernel boundary).  W2B_XCD_STORE_AUX=2 builds the round-5 policy for same-box A/Bs.
#ifndef W2B_XCD_STORE_AUX
#define W2B_XCD_STORE_AUX 2   // (until the same-box A/B of round 6 says otherwise)
#endif
template <int MM> struct Aux {
  static constexpr int load = (MM == 0 || MM == W2B_MM_EXACT) ? 16 : ((MM == 2 || MM == 5 || MM == 6) ? 2 : 0);
  static constexpr int store = (MM == 1) ? 0 : ((MM == 5) ? 2 : ((MM == 6) ? W2B_XCD_STORE_AUX : 16));
};
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// `tab_bytes` != 0: the whole table is addressable with 32-bit offsets (< 4 GiB): ONE buffer resource per table
// (loop invariant, built once) and the row start goes into the instruction's scalar offset -- one s_mul per row
// instead of a 4-SGPR descriptor per row (the kernels were close to issue-bound with as many SALU as VALU
// instructions).  Lanes beyond the row are masked by the callers (`active`).  tab_bytes == 0 (tables >= 4 GiB,
// e.g. V=3.7M x D=1000): per-row resource whose size is the row length.
// TB: -1 = decide at run time from tab_bytes; 0 / 1 = the caller's kernel was instantiated for tab_bytes != 0 / == 0
// (the sentence-resident kernel: its instruction stream is the bottleneck, and the run-time form costs ~20 scalar
// instructions and three branches per row access).
template <int VEC, int MM, int TB = -1>
__device__ __forceinline__ Col<VEC> load_col(const float *tab, long long row, int dim, int col0, unsigned tab_bytes) {
  Col<VEC> c;
  const int urow = __builtin_amdgcn_readfirstlane((int)row);
  __amdgpu_buffer_rsrc_t r;
  int soff;
  if (TB == 0 || (TB < 0 && tab_bytes)) {
    r = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, (int)tab_bytes, 0x27000);
    soff = urow * dim * 4;
  } else {
    r = __builtin_amdgcn_make_buffer_rsrc((void *)(tab + urow * (long long)dim), 0, dim * 4, 0x27000);
    soff = 0;
  }
  if (VEC == 4) {
    u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, col0 * 4, soff, Aux<MM>::load);
    c.e[0] = __uint_as_float(t.x); c.e[1 % VEC] = __uint_as_float(t.y);
    c.e[2 % VEC] = __uint_as_float(t.z); c.e[3 % VEC] = __uint_as_float(t.w);
  } else if (VEC == 2) {
    u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, col0 * 4, soff, Aux<MM>::load);
    c.e[0] = __uint_as_float(t.x); c.e[1 % VEC] = __uint_as_float(t.y);
  } else {
    c.e[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, col0 * 4, soff, Aux<MM>::load));
  }
  return c;
}
template <int VEC, int MM, int TB = -1>
__device__ __forceinline__ void store_col(float *tab, long long row, int dim, int col0, const Col<VEC> &c,
                                          unsigned tab_bytes) {
  const int urow = __builtin_amdgcn_readfirstlane((int)row);
  __amdgpu_buffer_rsrc_t r;
  int soff;
  if (TB == 0 || (TB < 0 && tab_bytes)) {
    r = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, (int)tab_bytes, 0x27000);
    soff = urow * dim * 4;
  } else {
    r = __builtin_amdgcn_make_buffer_rsrc((void *)(tab + urow * (long long)dim), 0, dim * 4, 0x27000);
    soff = 0;
  }
  if (VEC == 4) {
    u32x4 t;
    t.x = __float_as_uint(c.e[0]); t.y = __float_as_uint(c.e[1 % VEC]);
    t.z = __float_as_uint(c.e[2 % VEC]); t.w = __float_as_uint(c.e[3 % VEC]);
    __builtin_amdgcn_raw_buffer_store_b128(t, r, col0 * 4, soff, Aux<MM>::store);
  } else if (VEC == 2) {
    u32x2 t;
    t.x = __float_as_uint(c.e[0]); t.y = __float_as_uint(c.e[1 % VEC]);
    __builtin_amdgcn_raw_buffer_store_b64(t, r, col0 * 4, soff, Aux<MM>::store);
  } else {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(c.e[0]), r, col0 * 4, soff, Aux<MM>::store);
  }
}

// Wavefront (64-lane) reductions on the DPP cross-lane network: six dependent VALU operations, no LDS
// round trips (a __shfl_xor ladder is six ds_bpermute latencies).  Steps: swap inside quads, mirror
// inside half rows / rows of 16, then row_bcast:15 and row_bcast:31 carry the row sums up; lane 63 holds
// the total and v_readlane returns it to every lane.  Deterministic association order.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u(unsigned x) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false);
}
#define W2B_DPP_QUAD_1032 0xB1
#define W2B_DPP_QUAD_2301 0x4E
#define W2B_DPP_ROW_HALF_MIRROR 0x141
#define W2B_DPP_ROW_MIRROR 0x140
#define W2B_DPP_ROW_BCAST15 0x142
#define W2B_DPP_ROW_BCAST31 0x143
__device__ __forceinline__ float wave_sum(float x) {
  x += dpp_f<W2B_DPP_QUAD_1032, 0xf>(x);
  x += dpp_f<W2B_DPP_QUAD_2301, 0xf>(x);
  x += dpp_f<W2B_DPP_ROW_HALF_MIRROR, 0xf>(x);
  x += dpp_f<W2B_DPP_ROW_MIRROR, 0xf>(x);
  x += dpp_f<W2B_DPP_ROW_BCAST15, 0xa>(x);
  x += dpp_f<W2B_DPP_ROW_BCAST31, 0xc>(x);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ unsigned wave_xor(unsigned x) {
  x ^= dpp_u<W2B_DPP_QUAD_1032, 0xf>(x);
  x ^= dpp_u<W2B_DPP_QUAD_2301, 0xf>(x);
  x ^= dpp_u<W2B_DPP_ROW_HALF_MIRROR, 0xf>(x);
  x ^= dpp_u<W2B_DPP_ROW_MIRROR, 0xf>(x);
  x ^= dpp_u<W2B_DPP_ROW_BCAST15, 0xa>(x);
  x ^= dpp_u<W2B_DPP_ROW_BCAST31, 0xc>(x);
  return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
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
  int *prev;    // [maxt]  index of the previous occurrence of the same target row, or -1 (prep_lists only; process_word
                //         parks the dot products f of the targets here for the loss bookkeeping)
  int *cend;    // [maxt]  end index of every target chunk (a chunk is cut at W2B_T rows or at a repeated row)
  float *red;   // [2][W2B_T][W2B_MAXW] cross-wave partial dot products (double buffered)
  float *stash; // [W2B_STASH][blockDim][VEC] raw u columns of the first context rows, private to the
                // owning thread: phase C updates them without a second trip to memory
  float *xprod; // exact mode only: [W2B_T][W2B_EXACT_COLS + 1] products of the current column block
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
  L.cend = p; p += maxt;
  L.xprod = nullptr;   // set by the kernels that support the exact mode
  return L;
}


// ------------------------------------------------------------------------------------ list bookkeeping
// Executed by ONE wavefront (all 64 lanes, wave-uniform arguments) right after it has written the lists:
//   prev[i]  = index of the previous occurrence of target row i (-1: none)      -> duplicates serialise
//   cend[k]  = end of the k-th chunk of at most T targets, cut early at a repeated row
//   umult[j] = multiplicity of context row j at its first occurrence, 0 at later ones (plain kernel only)
// Lane-parallel compares on register copies (v_readlane broadcasts) instead of O(n^2) LDS loops.
template <int T, typename IP>
__device__ __forceinline__ int prep_lists(IP tgt, IP prev, IP cend, int nt, IP ctx, IP umult, int cw,
                                          int lane) {
  W2B_WAVE_SYNC();
  for (int i0 = 0; i0 < nt; i0 += 64) {
    const int i = i0 + lane;
    const int me = (i < nt) ? tgt[i] : -1;
    int pd = -1;
    for (int j = 0; j < min(nt, i0 + 64); j++) {
      const int tj = (j >= i0) ? __builtin_amdgcn_readlane(me, j - i0) : tgt[j];
      pd = (j < i && tj == me) ? j : pd;
    }
    if (i < nt) prev[i] = pd;
  }
  if (umult) {
    for (int i0 = 0; i0 < cw; i0 += 64) {
      const int i = i0 + lane;
      const int me = (i < cw) ? ctx[i] : -1;
      bool first = true;
      int mult = 0;
      for (int j = 0; j < cw; j++) {
        const int cj = (j >= i0 && j < i0 + 64) ? __builtin_amdgcn_readlane(me, j - i0) : ctx[j];
        first = first && !(j < i && cj == me);
        mult += (j >= i && cj == me) ? 1 : 0;
      }
      if (i < cw) umult[i] = first ? mult : 0;
    }
  }
  W2B_WAVE_SYNC();
  int start = 0, k = 0;
  while (start < nt) {
    const int lim = min(start + T, nt);
    int end = lim;
    for (int i0 = start + 1; i0 < lim; i0 += 64) {
      const int i = i0 + lane;
      const bool hit = (i < lim) && (prev[i] >= start);
      const unsigned long long m = __ballot(hit);
      if (m) { end = i0 + __ffsll((long long)m) - 1; break; }
    }
    if (lane == 0) cend[k] = end;
    k++;
    start = end;
  }
  return k;     // number of chunks
}

// ------------------------------------------------------------------------------------ atomic row updates
// tab[row] += d, element by element, as fp32 atomic adds performed by the memory system (no read-modify-write window in
// the kernel: nothing another worker adds meanwhile is lost).  When nobody else touches the row the result is
// fl(x + d) -- the very value a load / add / store sequence produces, so a single worker stays bit-identical.
template <int VEC, int TB = -1>
__device__ __forceinline__ void add_col(float *tab, long long row, int dim, int col0, const Col<VEC> &d, unsigned tab_bytes) {
  const int urow = __builtin_amdgcn_readfirstlane((int)row);
  __amdgpu_buffer_rsrc_t r;
  int soff;
  if (TB == 0 || (TB < 0 && tab_bytes)) {
    r = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, (int)tab_bytes, 0x27000);
    soff = urow * dim * 4;
  } else {
    r = __builtin_amdgcn_make_buffer_rsrc((void *)(tab + urow * (long long)dim), 0, dim * 4, 0x27000);
    soff = 0;
  }
#pragma unroll
  // aux 16 = sc1: agent scope, like every other coherent row access (the adds of workers on different XCDs meet at the
  // memory side, and they are ordered against the sc1 loads / stores of the non-atomic rows' accesses; round 3 issued
  // them without scope bits -- tools/atomic_probe.hip checks both forms for lost adds)
  for (int e = 0; e < VEC; e++) (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(d.e[e], r, (col0 + e) * 4, soff, 16);
}

// The same add for 16-byte columns with the wavefront's 256 deltas TRANSPOSED first, so that instruction e adds the
// dwords [64 e, 64 e + 64) of the wavefront's 1 KiB segment -- two whole cache lines per instruction instead of eight
// lines with 8 dwords each.  The memory side pays per line an instruction touches: tools/atomic_probe.hip measures 97 M
// row-updates/s for the per-lane layout above and 397 M/s for this one (16-byte stores of the same rows: 1350 M/s).  The
// transpose is 16 ds_bpermute (crossbar only, no LDS memory) + 12 selects.  ALL 64 lanes of the wavefront must call it
// (inactive lanes pass zeros); dest lane l of instruction e takes element l & 3 of lane 16 e + (l >> 2).
template <int TB = -1>
__device__ __forceinline__ void add_col_contig(float *tab, long long row, int dim, const Col<4> &d, unsigned tab_bytes) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int urow = __builtin_amdgcn_readfirstlane((int)row);
  __amdgpu_buffer_rsrc_t r;
  int soff;
  if (TB == 0 || (TB < 0 && tab_bytes)) {
    r = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, (int)tab_bytes, 0x27000);
    soff = urow * dim * 4;
  } else {
    r = __builtin_amdgcn_make_buffer_rsrc((void *)(tab + urow * (long long)dim), 0, dim * 4, 0x27000);
    soff = 0;
  }
  const int sel = lane & 3;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int src = (16 * e + (lane >> 2)) << 2;
    const float p0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(d.e[0])));
    const float p1 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(d.e[1])));
    const float p2 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(d.e[2])));
    const float p3 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(d.e[3])));
    const float v = sel == 0 ? p0 : (sel == 1 ? p1 : (sel == 2 ? p2 : p3));
    const int c = wave * 256 + e * 64 + lane;
    if (c < dim) (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, r, c * 4, soff, 16);
  }
}

// ------------------------------------------------------------------------------------ XCD-local copies of the hot rows
// With coherent (agent-scope) rows every access to a row goes to its memory line, and the few most frequent rows of
// u (context words) and v (targets) queue there: a 3200-byte row sustains ~7 M read-modify-writes per second, against
// e.g. 0.32 target uses of row 1 per centre word on Zipf(1) ids at 25 M words/s.  Every XCD therefore works on ITS OWN
// COPY of rows 1..nu of u and 1..nv of v (the vocabulary is sorted by count; the numbers come from the word counts and
// the number of workers), accessed with `nt` loads and stores: past the CU's L1, served by and kept in the XCD's L2
// (MI355X_MICROARCH.md, inter-workgroup visibility; tools/coherence_probe2.hip: all workgroups of an XCD see each
// other's read-modify-writes).  Inside an XCD a hot row is the reference's racy shared row (ref :490,501), at L2 speed.
// Across XCDs the copies are kept together by CONSENSUS merges.  One wavefront at a time per 1 KiB segment of a copy
// (try-lock; a wavefront that finds it taken skips its turn) loads the copy c, the value e it left there at its last
// merge and the master row m, and stores
//      n = c                  if m == e  (nobody else has published since: this XCD's copy IS the consensus -- a single
//                                         worker stays bit-identical to a run without copies)
//          m                  if c == e  (nothing of ours: adopt)
//          m + w * (c - m)    otherwise  (w = 1/8 by default: eight XCDs pulling with weight 1/8 each make the master
//                                         the running average of the copies)
// to master, copy and entry.  Deliberately NOT a sum of deltas: dozens of workers have a hot row in flight at any
// moment, all with gradients of the same stale value; adding all of them up over-shoots (measured: lossless atomic
// adds and delta sums moved the first-epoch loss of the text8-sized run by 7 % and made a 4-replica exchange diverge),
// whereas the reference applies its updates one after the other.  And deliberately free of delta bookkeeping: a worker's
// store that lands after an adoption undoes the adoption for that copy; with "copy - entry = our contribution" that
// became a negative contribution (first version of this scheme); with averaging it only delays the consensus.
// Workers take turns: every hot_period centre words a worker merges xhot_m rows, rotating through the set.
// k_xhot_fold (w2b_kernels_misc.hip) applies the same rule for all eight copies before and after every launch, so
// between launches the master rows are complete and copy == entry == master.  16-byte columns only (VEC == 4).
#define W2B_MM_XCD 5          // Aux<>: nt loads + nt stores (XCD scope)
#define W2B_MM_XHOT_ST 6      // Aux<>: nt loads + W2B_XCD_STORE_AUX stores: the hot-row copies' stores (round 6: plain write-back)
struct XHot {
  float *cu, *cv, *eu, *ev;    // this XCD's copies of the hot rows of u / v, and their entry values
  unsigned *lu, *lv;           // merge locks [row][W2B_MAXW]
  int nu, nv;
};
__device__ __forceinline__ XHot xhot_here(const W2bParams &P) {
  XHot X;
  X.nu = P.xhot ? P.xhot_u : 0;
  X.nv = P.xhot ? P.xhot_v : 0;
  // s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4): the XCD this workgroup runs on (a different placement would only be slower)
  const int xcd = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & (W2B_NXCD - 1);
  const long long rows = X.nu + X.nv;
  float *base = P.xhot + (long long)xcd * (2 * rows * P.dim + rows * W2B_MAXW);
  X.cu = base;
  X.cv = base + (long long)X.nu * P.dim;
  X.eu = X.cv + (long long)X.nv * P.dim;
  X.ev = X.eu + (long long)X.nu * P.dim;
  X.lu = reinterpret_cast<unsigned *>(X.ev + (long long)X.nv * P.dim);
  X.lv = X.lu + (long long)X.nu * W2B_MAXW;
  return X;
}
__device__ __forceinline__ Col<4> xhot_ld(const float *rows, int k, int n, int dim, int col0) {
  return load_col<4, W2B_MM_XCD, 0>(rows, k, dim, col0, (unsigned)(n * dim * 4));
}
__device__ __forceinline__ void xhot_st(float *rows, int k, int n, int dim, int col0, const Col<4> &c) {
  store_col<4, W2B_MM_XHOT_ST, 0>(rows, k, dim, col0, c, (unsigned)(n * dim * 4));
}
// hot row k (master row k + 1 of `tab`) of this XCD meets memory: this wavefront's segment of the row.
// MM / TB: how the master rows are accessed; w: weight of this XCD's copy in the consensus.
template <int MM, int TB>
__device__ __forceinline__ void xhot_merge_row(float *tab, float *copy, float *entry, unsigned *locks, int k, int n, int dim,
                                               int col0, bool active, int wave, int lane, unsigned tab_bytes, float w) {
  unsigned *lock = locks + k * W2B_MAXW + wave;
  unsigned got = 1u;
  if (lane == 0) got = __hip_atomic_exchange(lock, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (__builtin_amdgcn_readfirstlane((int)got) != 0) return;       // somebody is merging this segment right now
  Col<4> c, e, m, o;
#pragma unroll
  for (int i = 0; i < 4; i++) { c.e[i] = 0.f; e.e[i] = 0.f; m.e[i] = 0.f; }
  if (active) {
    c = xhot_ld(copy, k, n, dim, col0);
    e = xhot_ld(entry, k, n, dim, col0);
    m = load_col<4, MM, TB>(tab, k + 1, dim, col0, tab_bytes);
  }
  bool own = false, oth = false;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    own = own || __float_as_uint(c.e[i]) != __float_as_uint(e.e[i]);
    oth = oth || __float_as_uint(m.e[i]) != __float_as_uint(e.e[i]);
  }
  const bool any_own = __ballot(own) != 0ull, any_oth = __ballot(oth) != 0ull;      // per 1 KiB segment
#pragma unroll
  for (int i = 0; i < 4; i++) o.e[i] = !any_oth ? c.e[i] : (!any_own ? m.e[i] : m.e[i] + w * (c.e[i] - m.e[i]));
  if (active) {
    if (any_own) store_col<4, MM, TB>(tab, k + 1, dim, col0, o, tab_bytes);
    if (any_oth) xhot_st(copy, k, n, dim, col0, o);
    if (any_own || any_oth) xhot_st(entry, k, n, dim, col0, o);
  }
  __builtin_amdgcn_s_waitcnt(0);                                    // everything above has reached the memory system
  if (lane == 0) __hip_atomic_store(lock, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one merge event of a workgroup of the plain kernels: P.xhot_m rows of each table, rotating through the sets
template <int MM, int TB = -1>
__device__ __forceinline__ void xhot_merge_event(const W2bParams &P, const XHot &X, int &cursor, int col0, bool active) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = 0; j < P.xhot_m; j++) {
    const int i = cursor + j;
    if (X.nu > 0 && j < X.nu)
      xhot_merge_row<MM, TB>(P.u, X.cu, X.eu, X.lu, i % X.nu, X.nu, P.dim, col0, active, wave, lane, P.tab_bytes, P.xhot_w);
    if (X.nv > 0 && j < X.nv)
      xhot_merge_row<MM, TB>(P.v, X.cv, X.ev, X.lv, i % X.nv, X.nv, P.dim, col0, active, wave, lane, P.tab_bytes, P.xhot_w);
  }
  cursor += P.xhot_m;
}

// ------------------------------------------------------------------------------------ one centre word
// Preconditions: L.ctx[0..cw), L.tgt[0..nt) and prep_lists() results published by a __syncthreads();
// cw >= 1, nt >= 1.
// Ends with a __syncthreads() (lists may be overwritten afterwards).
// X: this XCD's copies of the hottest rows (nu = nv = 0: none; VEC == 4 only): a row k <= nu of u / k <= nv of v is read
// and written at its copy instead of its master address.  Passed by reference, so that its fields stay in registers.
// P.atomic_rank: the other rows among 1..atomic_rank are updated with atomic adds at their master address.
// ATOM (16-byte columns only): which tables have rows that are updated with atomic adds -- 0 none, 1 u (context rows, phase
// C), 2 u and v.  Instantiations of their own, so that the transposes of the atomic form cost the others no registers
// (compiled into one kernel they spilled 7 more VGPRs: 77.9 % -> 73.2 % of the roofline at the headline shape).
// TB: how rows are addressed (load_col): -1 = decided at run time from P.tab_bytes, 0 = one buffer resource per table (tables
// below 2 GiB: the worker kernel's 16-byte-column instantiations are compiled for it since round 5 -- the run-time form costs
// ~20 scalar instructions per row access and SGPRs that the kernel, at exactly its register budget, spills)
template <int QM, int VEC, bool LOSS, int MM, int ATOM = 0, int TB = -1>
__device__ __forceinline__ void process_word(const W2bParams &P, const WordLds &L, const QParam &qp,
                                             const int cw, const int nt, const float alpha,
                                             double &loss_acc, const XHot &X) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int dim = P.dim, col0 = tid * VEC;
  const bool active = col0 < dim;
  const float ar2 = (2.f * alpha) * P.reg;                      // 2*alpha*reg (ref :490,:501)
  constexpr int TC = TFor<LOSS>::value;                         // rows per chunk

  Col<VEC> x[TC];
  int rows[TC];           // row ids of the chunk, wave-uniform (SGPRs): one LDS read, then v_readlane
  int start = 0, chunk = 0, end = L.cend[0];
  auto chunk_rows = [&]() {
    const int mine = L.tgt[min(start + lane, nt - 1)];
#pragma unroll
    for (int i = 0; i < TC; i++) rows[i] = __builtin_amdgcn_readlane(mine, i);
  };
  chunk_rows();
  const int nhu = (VEC == 4) ? X.nu : 0, nhv = (VEC == 4) ? X.nv : 0;
  // row accesses: a hot row at this XCD's copy (VEC == 4 only), every other row at its master address
  auto ld_u = [&](int row) -> Col<VEC> {
    if constexpr (VEC == 4) { if ((unsigned)(row - 1) < (unsigned)nhu) return xhot_ld(X.cu, row - 1, nhu, dim, col0); }
    return load_col<VEC, MM, TB>(P.u, row, dim, col0, P.tab_bytes);
  };
  // row <- val (= old + d): a store (hot rows: to this XCD's copy), or an atomic add of d for rows 1..atomic_rank
  // (VEC == 4: only the ATOM instantiations look at the ranks; the 4-byte-column kernels decide at run time)
  const int atomic_rank = (VEC == 4 && ATOM < 2) ? 0 : P.atomic_rank, atomic_rank_u = (VEC == 4 && ATOM < 1) ? 0 : P.atomic_rank_u;
  auto up_u = [&](int row, const Col<VEC> &val, const Col<VEC> &d) {
    if constexpr (VEC == 4) { if ((unsigned)(row - 1) < (unsigned)nhu) { xhot_st(X.cu, row - 1, nhu, dim, col0, val); return; } }
    if (row <= atomic_rank_u) add_col<VEC, TB>(P.u, row, dim, col0, d, P.tab_bytes);
    else store_col<VEC, MM, TB>(P.u, row, dim, col0, val, P.tab_bytes);
  };
  auto ld_v = [&](int row) -> Col<VEC> {
    if constexpr (VEC == 4) { if ((unsigned)(row - 1) < (unsigned)nhv) return xhot_ld(X.cv, row - 1, nhv, dim, col0); }
    return load_col<VEC, MM, TB>(P.v, row, dim, col0, P.tab_bytes);
  };
  auto up_v = [&](int row, const Col<VEC> &val, const Col<VEC> &d) {
    if constexpr (VEC == 4) { if ((unsigned)(row - 1) < (unsigned)nhv) { xhot_st(X.cv, row - 1, nhv, dim, col0, val); return; } }
    if (row <= atomic_rank) add_col<VEC, TB>(P.v, row, dim, col0, d, P.tab_bytes);
    else store_col<VEC, MM, TB>(P.v, row, dim, col0, val, P.tab_bytes);
  };
  // one chunk of target rows
  auto load_targets = [&](bool zero) {
#pragma unroll
    for (int i = 0; i < TC; i++) {
      if (zero) {
#pragma unroll
        for (int e = 0; e < VEC; e++) x[i].e[e] = 0.f;
      }
      if (active && start + i < end) x[i] = ld_v(rows[i]);
    }
  };
  // issue the first chunk of target-row loads before the context phase so both gathers overlap
  load_targets(true);

  // ---- phase A: context_avg = (1/cw) * sum_j quantize(u[ctx_j])   (ref :431-449)
  Col<VEC> avg;
  float regsq = 0.f;                       // LOSS with reg != 0: this thread's sum of q^2 over the window rows AND the target rows
  const bool reg_on = P.reg != 0.f;
  float *fsave = reinterpret_cast<float *>(L.prev);
#pragma unroll
  for (int e = 0; e < VEC; e++) avg.e[e] = 0.f;
  for (int j0 = 0; j0 < cw; j0 += W2B_CA) {
    Col<VEC> r[W2B_CA];
#pragma unroll
    for (int jj = 0; jj < W2B_CA; jj++)
      if (active && j0 + jj < cw) r[jj] = ld_u(__builtin_amdgcn_readfirstlane(L.ctx[j0 + jj]));
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
          if (LOSS && reg_on) regsq += q * q;
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
    float p[TC];
#pragma unroll
    for (int i = 0; i < TC; i++) {
      float t[VEC];
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        const float q = quant<QM>(x[i].e[e], qp);
        t[e] = avg.e[e] * q;                                    // ref :466 (re-associated as a binary tree)
      }
      // pairwise tree over the elements of a column
      const float s = (VEC == 4) ? (t[0] + t[1 % VEC]) + (t[2 % VEC] + t[3 % VEC]) : ((VEC == 2) ? t[0] + t[1 % VEC] : t[0]);
      p[i] = active ? s : 0.f;
    }
    float *red = L.red + par * (TC * W2B_MAXW);
    if (MM == W2B_MM_EXACT) {
      // ref :461-467 in the reference's own order: f = 0; for c: f += context_avg[c] * quantize(v[c]) -- every
      // product rounded, then added to the running sum.  The products of a block of columns go to LDS, lane i of
      // wavefront 0 continues the chain of target i over them, block after block.
      float fchain = 0.f;
      for (int b0 = 0; b0 < dim; b0 += W2B_EXACT_COLS) {
        if (active && col0 >= b0 && col0 < b0 + W2B_EXACT_COLS) {
#pragma unroll
          for (int i = 0; i < TC; i++)
            if (i < n) {
#pragma unroll
              for (int e = 0; e < VEC; e++)
                L.xprod[i * (W2B_EXACT_COLS + 1) + (col0 - b0) + e] = avg.e[e] * quant<QM>(x[i].e[e], qp);
            }
        }
        __syncthreads();
        if (wave == 0 && lane < n) {
          const int cnt = min(W2B_EXACT_COLS, dim - b0);
          const float *src = L.xprod + lane * (W2B_EXACT_COLS + 1);
          for (int c = 0; c < cnt; c++) fchain += src[c];
        }
        __syncthreads();
      }
      if (wave == 0 && lane < n) {
        red[lane * W2B_MAXW] = fchain;
        for (int w = 1; w < nwaves; w++) red[lane * W2B_MAXW + w] = 0.f;   // the sum below adds exact zeros
      }
    } else {
#pragma unroll
      for (int i = 0; i < TC; i++) p[i] = wave_sum(p[i]);     // unconditional: W2B_T independent chains interleave
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < TC; i++)
          if (i < n) red[i * W2B_MAXW + wave] = p[i];
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
      if (LOSS && wave == 0) fsave[start + lane] = f;           // the log-sigmoid term of ref :480-483 is booked after phase C
    }
    // error accumulation + row update, in target order (ref :486-491)
#pragma unroll
    for (int i = 0; i < TC; i++) {
      if (i < n) {
        const float g = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gl), i));
        // (wave-uniform choice: rows[i] lives in SGPRs) a frequent row below the hot ones gets its delta as an atomic add;
        // every lane of the wavefront takes part in the transpose of that form, idle lanes with zeros
        const bool by_add = (VEC == 4 && ATOM >= 2) && rows[i] <= atomic_rank && !((unsigned)(rows[i] - 1) < (unsigned)nhv);
        Col<VEC> dl;
        if constexpr (VEC == 4 && ATOM >= 2) {
#pragma unroll
          for (int e = 0; e < VEC; e++) dl.e[e] = 0.f;
        }
        if (active) {
#pragma unroll
          for (int e = 0; e < VEC; e++) {
            float xv = x[i].e[e];
            // opaque copy: re-derive the quantized value here instead of keeping VEC extra registers per
            // row alive since the dot product (halves the register footprint of a chunk)
            if (QM != 0) asm volatile("" : "+v"(xv));
            const float q = quant<QM>(xv, qp);
            if (LOSS && reg_on) regsq += q * q;                 // reg * sum q^2 of every target row (ref :463,:468-471)
            err.e[e] += g * q;
            dl.e[e] = g * avg.e[e] - ar2 * xv;
            x[i].e[e] = xv + dl.e[e];
          }
          if (!by_add) up_v(rows[i], x[i], dl);
        }
        if constexpr (VEC == 4 && ATOM >= 2) {
          if (by_add) add_col_contig<TB>(P.v, rows[i], dim, dl, P.tab_bytes);
        }
      }
    }
    start = end;
    if (start >= nt) break;
    end = L.cend[++chunk];
    par ^= 1;
    chunk_rows();
    load_targets(false);
  }

  // ---- phase C: u[ctx_j] += context_avge - 2*alpha*reg*u[ctx_j]   (ref :494-503)
  for (int j0 = 0; j0 < cw; j0 += W2B_CA) {
    Col<VEC> r[W2B_CA];
#pragma unroll
    for (int jj = 0; jj < W2B_CA; jj++)
      if (active && j0 + jj < cw && L.umult[j0 + jj] > 0) {
        const int crow = __builtin_amdgcn_readfirstlane(L.ctx[j0 + jj]);
        // rows 1..fresh_rank_u are read AGAIN here instead of taken from the stash: the reference's update is
        // `u[c] += e[c]` on the current memory value (ref :500-502); the stashed value is a whole centre word (~30 us) old,
        // and storing stash + e would erase what every other worker added to a frequent row in between
        if (j0 + jj < W2B_STASH && crow > P.fresh_rank_u) {
#pragma unroll
          for (int e = 0; e < VEC; e++) r[jj].e[e] = L.stash[((j0 + jj) * blockDim.x + tid) * VEC + e];
        } else {
          r[jj] = ld_u(crow);
        }
      }
#pragma unroll
    for (int jj = 0; jj < W2B_CA; jj++) {
      if constexpr (VEC == 4 && ATOM >= 1) {
        if (j0 + jj < cw) {                    // (wave-uniform: the atomic form needs every lane of the wavefront)
          const int m = L.umult[j0 + jj];
          if (m > 0) {
            const int crow = __builtin_amdgcn_readfirstlane(L.ctx[j0 + jj]);
            const bool by_add = crow <= atomic_rank_u && !((unsigned)(crow - 1) < (unsigned)nhu);
            Col<VEC> dl;
#pragma unroll
            for (int e = 0; e < VEC; e++) dl.e[e] = 0.f;
            for (int k = 0; k < m; k++) {      // a row that occurs m times in the window is updated m times
              if (active) {
#pragma unroll
                for (int e = 0; e < VEC; e++) {
                  dl.e[e] = err.e[e] - ar2 * r[jj].e[e];
                  r[jj].e[e] = r[jj].e[e] + dl.e[e];
                }
              }
              if (by_add) add_col_contig<TB>(P.u, crow, dim, dl, P.tab_bytes);   // (every one of the m updates is an add of its own)
            }
            if (!by_add && active) up_u(crow, r[jj], dl);
          }
        }
      } else {
        if (active && j0 + jj < cw) {
          const int m = L.umult[j0 + jj];
          if (m > 0) {
            const int crow = __builtin_amdgcn_readfirstlane(L.ctx[j0 + jj]);
            const bool by_add = crow <= atomic_rank_u && !(VEC == 4 && (unsigned)(crow - 1) < (unsigned)nhu);   // (4-byte columns)
            Col<VEC> dl;
            for (int k = 0; k < m; k++) {      // a row that occurs m times in the window is updated m times
#pragma unroll
              for (int e = 0; e < VEC; e++) {
                dl.e[e] = err.e[e] - ar2 * r[jj].e[e];
                r[jj].e[e] = r[jj].e[e] + dl.e[e];
              }
              if (by_add && k + 1 < m) up_u(crow, r[jj], dl);    // (every one of the m updates is an add of its own)
            }
            up_u(crow, r[jj], dl);
          }
        }
      }
    }
  }
  if (LOSS) {
    // ref :480-483 for all targets of this centre word, now that the chunk registers are free: lane j of wavefront 0
    // takes target j (f parked in LDS by the same wavefront; in-order LDS + the barriers in between)
    if (wave == 0) {
      for (int j = lane; j < nt; j += 64) {
        const float f = fsave[j];
        const float dp = (j == 0) ? f : -f;                     // target 0 is the centre word (label 1)
        float sg;
        if (dp > 6.f) sg = 1.f;
        else if (dp < -6.f) sg = 1e-9f;
        else sg = 1.f / (1.f + expf(-dp));
        loss_acc += (double)logf(sg);
      }
    }
    if (reg_on) {
      const float s = wave_sum(active ? regsq : 0.f);
      if (lane == 0) loss_acc -= (double)(P.reg * s);           // ref :437-445 and :463-471: window rows + target rows
    }
  }
  __syncthreads();
}


// ------------------------------------------------------------------------------------ one centre word, wide rows
// Rows too long for one thread per column (more than 1024 columns of 16 / 4 bytes: -size > 4096, or > 1024 when not a
// multiple of 4; the reference has no limit, ref :598).  Same preconditions and semantics as process_word, organised for
// generality, not speed: 1024 threads, thread t owns the columns t, t + 1024, ...; the window average and the
// accumulated error live in a per-workgroup scratch row in global memory (P.wide_scratch), the targets are taken one
// at a time (so duplicates are ordered by construction), every element is accessed on its own (4 bytes per lane).
// The dot product is a tree over threads / wavefronts in the fast mode and the reference's serial chain (blocks of
// W2B_EXACT_COLS products through LDS, continued by thread 0) in the parity mode.
template <int QM, bool LOSS, int MM>
__device__ __forceinline__ void process_word_wide(const W2bParams &P, const WordLds &L, const QParam &qp, const int cw,
                                                  const int nt, const float alpha, double &loss_acc) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6, nthr = blockDim.x;
  const int dim = P.dim;
  const float ar2 = (2.f * alpha) * P.reg;
  float *avg = P.wide_scratch + (long long)blockIdx.x * 2 * dim, *errv = avg + dim;
  auto ld = [&](float *tab, int row, int c) -> float { return load_col<1, MM>(tab, row, dim, c, P.tab_bytes).e[0]; };
  auto st = [&](float *tab, int row, int c, float x) { Col<1> v; v.e[0] = x; store_col<1, MM>(tab, row, dim, c, v, P.tab_bytes); };
  // ---- phase A (ref :431-449)
  float regsq = 0.f;
  for (int c = tid; c < dim; c += nthr) {
    float s = 0.f;
    for (int j = 0; j < cw; j++) {
      const float q = quant<QM>(ld(P.u, L.ctx[j], c), qp);
      s += q;
      if (LOSS) regsq += q * q;
    }
    avg[c] = s / (float)cw;
    errv[c] = 0.f;
  }
  if (LOSS && P.reg != 0.f) {
    const float s = wave_sum(regsq);
    if (lane == 0) loss_acc -= (double)(P.reg * s);
  }
  __syncthreads();
  // ---- phase B (ref :450-492), one target at a time
  for (int t = 0; t < nt; t++) {
    const int row = L.tgt[t];
    float f = 0.f;
    if (MM == W2B_MM_EXACT) {
      for (int b0 = 0; b0 < dim; b0 += W2B_EXACT_COLS) {
        if (tid < W2B_EXACT_COLS && b0 + tid < dim) L.xprod[tid] = avg[b0 + tid] * quant<QM>(ld(P.v, row, b0 + tid), qp);
        __syncthreads();
        if (tid == 0) {
          const int cnt = min(W2B_EXACT_COLS, dim - b0);
          for (int c = 0; c < cnt; c++) f += L.xprod[c];
        }
        __syncthreads();
      }
      if (tid == 0) L.red[0] = f;
    } else {
      float p = 0.f;
      for (int c = tid; c < dim; c += nthr) p += avg[c] * quant<QM>(ld(P.v, row, c), qp);
      p = wave_sum(p);
      if (lane == 0) L.red[wave] = p;
      __syncthreads();
      if (tid == 0) {
        for (int w = 0; w < nwaves; w++) f += L.red[w];
        L.red[0] = f;
      }
    }
    __syncthreads();
    f = L.red[0];
    const float label = (t == 0) ? 1.f : 0.f;
    float g;
    if (f > 6.f) g = (label - 1.f) * alpha;
    else if (f < -6.f) g = label * alpha;
    else g = (label - P.exp_table[(int)((f + 6.f) * 83.f)]) * alpha;
    if (LOSS && tid == 0) {                                       // ref :480-483
      const float dp = (label != 0.f) ? f : -f;
      float sg;
      if (dp > 6.f) sg = 1.f;
      else if (dp < -6.f) sg = 1e-9f;
      else sg = 1.f / (1.f + expf(-dp));
      loss_acc += (double)logf(sg);
    }
    float s2 = 0.f;
    for (int c = tid; c < dim; c += nthr) {
      const float xv = ld(P.v, row, c), q = quant<QM>(xv, qp);
      if (LOSS) s2 += q * q;
      errv[c] += g * q;
      st(P.v, row, c, xv + (g * avg[c] - ar2 * xv));
    }
    if (LOSS && P.reg != 0.f) {
      s2 = wave_sum(s2);
      if (lane == 0) loss_acc -= (double)(P.reg * s2);
    }
    __syncthreads();            // the next target may be the same row: its stores are complete (one workgroup, in order)
  }
  // ---- phase C (ref :494-503), window order; a row that occurs twice is updated twice
  for (int c = tid; c < dim; c += nthr) {
    const float e = errv[c];
    for (int j = 0; j < cw; j++) {
      const int row = L.ctx[j];
      const float r = ld(P.u, row, c);
      st(P.u, row, c, r + (e - ar2 * r));
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------ worker helpers
// exact n % d for a run-time divisor: magic = floor(2^64 / d) precomputed on the host; the quotient
// estimate is at most 2 too small.  Replaces the ~150-instruction software 64-bit division.
__device__ __forceinline__ unsigned long long fast_mod(unsigned long long n, unsigned long long d,
                                                     unsigned long long magic) {
  if (d <= 1) return 0;
  const unsigned long long q = __umul64hi(n, magic);
  unsigned long long r = n - q * d;
  if (r >= d) r -= d;
  if (r >= d) r -= d;
  return r;
}
// word_count_actual of ALL replicas for the alpha schedule (ref :391); see W2bShared
__device__ __forceinline__ long long w2b_global_progress(const W2bParams &P, long long wca_local) {
  const long long others_per_self = P.total_threads / P.num_threads - 1;        // replicas - 1
  return wca_local + (long long)P.shared->wca_others + others_per_self * (wca_local - (long long)P.shared->wca_at_sync);
}
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
      if (!((min_ >> e) & 1ull)) {
        eof = 1;
        if (P.corpus_more && lane == 0) P.shared->corpus_overrun = 1;   // end of the resident slice, not of the file
      }
    }
  }
  len_out = len;
}


struct WorkerLds {            // scalars of one worker, owned by wavefront 0
  unsigned long long rng;
  long long cursor, wc, last_wc;
  int sen_len, sen_pos, override_, eof, done, cw, nt, pad;
  float alpha;
};

template <typename F>
hipError_t dispatch_q(int bitlevel, F &&f) {
#ifdef W2B_QUICK_BUILD      // developer builds: only the 1-bit instantiations (register / ISA studies)
  return f(std::integral_constant<int, 1>());
#else
  switch (bitlevel) {
    case 0: return f(std::integral_constant<int, 0>());
    case 1: return f(std::integral_constant<int, 1>());
    case 2: return f(std::integral_constant<int, 2>());
    default: return f(std::integral_constant<int, 3>());
  }
#endif
}


// mem-mode dispatch for the kernels that implement the exact serial reduction (plain worker / tuple kernels)
template <typename F>
hipError_t dispatch_mm_exact(int mem_mode, int exact, F &&f);

// mem-mode dispatch: f(std::integral_constant<int, MM>)
template <typename F>
hipError_t dispatch_mm(int mem_mode, F &&f) {
  switch (mem_mode) {
    case 1: return f(std::integral_constant<int, 1>());
#ifdef W2B_EXPERIMENTAL_MEMMODES
    case 2: return f(std::integral_constant<int, 2>());
    case 3: return f(std::integral_constant<int, 3>());
#endif
    default: return f(std::integral_constant<int, 0>());
  }
}

template <typename F>
hipError_t dispatch_mm_exact(int mem_mode, int exact, F &&f) {
  if (exact) return f(std::integral_constant<int, W2B_MM_EXACT>());
  return dispatch_mm(mem_mode, f);
}

}  // namespace
