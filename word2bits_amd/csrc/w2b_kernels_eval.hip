// w2b_kernels_eval.hip -- the analogy evaluator's scan (ref src/compute-accuracy.c:106-110,155-177) on gfx950.
//
// The reference scores one question at a time: dist[c] = sum_a vec[a] * M[c][a] for every row c, strictly
// sequential in a, then keeps the first row with the largest dist > 0.  All questions together are a
// [Q x D] . [D x V] product -- the one dense contraction in this code base -- but the transcript has to stay
// byte-identical, ties included, and 1-bit vectors tie massively (scores are sums of +-m^2, +-3m^2), so the
// arg-max is decided by the rounding of every partial sum: every accumulator must walk a = 0..D-1 in order
// (no split-K, no tree) with exactly the arithmetic of the reference build being replaced:
//   FUSED  (the Makefile:6 build: acc = fma(a, b, acc))  -> k_eval_scores_mfma: v_mfma_f32_32x32x2_f32, whose
//          accumulators ARE sequential fmaf chains in k order, bit for bit (measured here by the parity tests);
//   !FUSED (-ffp-contract=off: acc = acc + a*b, two roundings) -> k_eval_scores<false> on the vector ALU with
//          v_pk_mul_f32 + v_pk_add_f32 (no matrix instruction rounds twice); this TU is built -ffp-contract=off.
//   k_eval_scores<true> (v_pk_fma_f32) is the same fused chain on the vector ALU (W2B_EVAL_KERNEL=0; cross-check).
//
// Vector-ALU tile: 256 threads = 16 x 16, 128 questions x 128 rows per workgroup, 8 x 8 accumulators per thread,
// K step 16 through LDS (both operands stored k-major so a thread reads its 8+8 values as four ds_read_b128),
// next K slab prefetched into registers during the current one.  Operands are padded on the device
// ([Qp][Dp], [Vp][Dp], Dp % 16 == 0, zero filled): x + 0*0 == x and fma(0,0,x) == x, so padding never
// changes a comparison.  The arg-max is fused into the epilogue: per question a 64-bit key
// (score bits << 32 | ~row) -- positive floats order like their bit patterns, ~row makes the lowest row win
// ties -- reduced with ds_max_u64 in LDS and one global atomic max per question and workgroup; the Q x V score
// matrix never exists in memory.
//
// Bound: fp32 issue.  2 flop per multiply-add; 256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz = 157.3 TFLOP/s
// = 78.6 T multiply-adds/s for fused arithmetic on either pipe (v_pk_fma_f32 or the f32 MFMA), half of that
// (39.3 T/s) for the two-instruction unfused form.  XCD-aware launch order: each XCD owns a contiguous stripe
// of row tiles (see k_eval_scores_mfma for the finer order).
#include "w2b_device.hpp"

namespace {

constexpr int EBM = 128, EBN = 128, EBK = 16, ETHREADS = 256;
constexpr int ELD = EBM + 4;   // LDS row pitch (floats): keeps b128 reads aligned, staggers the k rows

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ref :106-110, one thread per row: quantize, float length in column order, sqrt, then the row is scaled
// by k_eval_scale.  (sqrt() in the reference is the double one on a float, stored to float: the correctly
// rounded float square root.)
template <int QM, bool FUSED>
__global__ void k_eval_row_len(const float *__restrict__ raw, long long words, long long size, long long ld,
                               QParam qp, float *__restrict__ len_out) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= words) return;
  const float *row = raw + b * ld;
  float len = 0.f;
  for (long long a = 0; a < size; a++) {
    const float x = quant<QM>(row[a], qp);
    len = FUSED ? __builtin_fmaf(x, x, len) : len + x * x;
  }
  len_out[b] = sqrtf(len);   // correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt)
}

template <int QM>
__global__ void k_eval_scale(float *__restrict__ M, long long words, long long size, long long ld, QParam qp,
                             const float *__restrict__ len) {
  const long long n = words * size;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const long long b = i / size, a = i - b * size;
    M[b * ld + a] = quant<QM>(M[b * ld + a], qp) / len[b];   // IEEE division (same default)
  }
}

// ref :155: vec = (M[b2] - M[b1]) + M[b3], in that order
__global__ void k_eval_queries(const float *__restrict__ M, long long ld, long long nq, const int *__restrict__ b1,
                               const int *__restrict__ b2, const int *__restrict__ b3, float *__restrict__ Q) {
  const long long n = nq * ld;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const long long q = i / ld, a = i - q * ld;
    Q[i] = (M[(long long)b2[q] * ld + a] - M[(long long)b1[q] * ld + a]) + M[(long long)b3[q] * ld + a];
  }
}

template <bool FUSED>
__device__ __forceinline__ f32x2 mac2(float a, f32x2 b, f32x2 acc) {
  const f32x2 av = {a, a};
  if (FUSED) return __builtin_elementwise_fma(av, b, acc);
  return acc + av * b;    // two roundings: -ffp-contract=off for this TU
}

template <bool FUSED>
__global__ void __launch_bounds__(ETHREADS, 2)
k_eval_scores(const float *__restrict__ Q, const float *__restrict__ M, int nq, int words, int ld, int q_tiles,
              int c_tiles, int c_per_xcd, const int *__restrict__ b1, const int *__restrict__ b2,
              const int *__restrict__ b3, unsigned long long *__restrict__ best) {
  __shared__ float As[2][EBK][ELD];
  __shared__ float Bs[2][EBK][ELD];
  __shared__ unsigned long long skey[EBM];

  // XCD-aware tile order (workgroup ids are dealt round-robin to the 8 XCDs)
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int qt = local / c_per_xcd, ct = xcd * c_per_xcd + local % c_per_xcd;
  if (qt >= q_tiles || ct >= c_tiles) return;
  const int m0 = qt * EBM, n0 = ct * EBN;

  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  if (tid < EBM) skey[tid] = 0ull;

  // global -> register staging: 128 rows x 16 k = 512 float4 per operand, two per thread
  const int lrow0 = tid >> 2, lk = (tid & 3) * 4;          // rows lrow0 and lrow0 + 64
  const float *ga = Q + (long long)(m0 + lrow0) * ld + lk;
  const float *gb = M + (long long)(n0 + lrow0) * ld + lk;
  const long long half = 64ll * ld;
  f32x4 ra0 = *(const f32x4 *)ga, ra1 = *(const f32x4 *)(ga + half);
  f32x4 rb0 = *(const f32x4 *)gb, rb1 = *(const f32x4 *)(gb + half);

  f32x2 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x2{0.f, 0.f};

  const int nk = ld / EBK;
  for (int kt = 0; kt < nk; kt++) {
    const int buf = kt & 1;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      As[buf][lk + e][lrow0] = ra0[e];
      As[buf][lk + e][lrow0 + 64] = ra1[e];
      Bs[buf][lk + e][lrow0] = rb0[e];
      Bs[buf][lk + e][lrow0 + 64] = rb1[e];
    }
    __syncthreads();
    if (kt + 1 < nk) {
      ga += EBK;
      gb += EBK;
      ra0 = *(const f32x4 *)ga;
      ra1 = *(const f32x4 *)(ga + half);
      rb0 = *(const f32x4 *)gb;
      rb1 = *(const f32x4 *)(gb + half);
    }
#pragma unroll
    for (int k = 0; k < EBK; k++) {
      const f32x4 a0 = *(const f32x4 *)&As[buf][k][ty * 4], a1 = *(const f32x4 *)&As[buf][k][64 + ty * 4];
      const f32x4 b0 = *(const f32x4 *)&Bs[buf][k][tx * 4], b1v = *(const f32x4 *)&Bs[buf][k][64 + tx * 4];
      const float a[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      const f32x2 b[4] = {{b0[0], b0[1]}, {b0[2], b0[3]}, {b1v[0], b1v[1]}, {b1v[2], b1v[3]}};
#pragma unroll
      for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = mac2<FUSED>(a[i], b[j], acc[i][j]);
    }
    // the other buffer is rewritten only after the next barrier of the following iteration's compute
  }

  // epilogue: strict-greater arg-max with ties to the lowest row (ref :166-175, N = 1, bestd starts at 0)
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int ml = (i < 4) ? ty * 4 + i : 64 + ty * 4 + (i - 4);
    const int q = m0 + ml;
    if (q >= nq) continue;
    const int e1 = b1[q], e2 = b2[q], e3 = b3[q];
    unsigned long long key = 0ull;
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int c = n0 + ((j < 2) ? tx * 4 + j * 2 + h : 64 + tx * 4 + (j - 2) * 2 + h);
        const float d = acc[i][j][h];
        if (c < words && c != e1 && c != e2 && c != e3 && d > 0.f) {   // NaN fails d > 0 like `dist > bestd`
          const unsigned long long k2 =
              ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)c);
          key = k2 > key ? k2 : key;
        }
      }
    if (key) atomicMax(&skey[ml], key);
  }
  __syncthreads();
  if (tid < EBM && skey[tid] && m0 + tid < nq) atomicMax(&best[m0 + tid], skey[tid]);
}

// ------------------------------------------------------------------------------------ fused mode on the matrix cores
// v_mfma_f32_32x32x2_f32 evaluates each of its 32x32 accumulators as acc = fma(a[k0+1], b[k0+1], fma(a[k0], b[k0],
// acc)) -- bitwise an fmaf chain in k order (MI355X_MICROARCH.md, "F32 (f32 in): exact f32") -- which is precisely
// the arithmetic of the reference's FMA-contracted build.  So the FUSED scan can run on the matrix pipe without
// giving up bit parity (the tests compare best rows AND best scores bit-for-bit on tie-dominated inputs); the
// two-rounding mode cannot (it stays on k_eval_scores above).  Same peak as the vector ALU (64 flop/clk/SIMD), but
// one instruction retires 4096 flop instead of 256, so issue slots, VGPR ports and LDS are no longer the limit.
//
// Tile: 128 vocabulary rows ("m") x 128 questions ("n") per workgroup, K slab 16 through LDS (k-major, as above);
// 4 wavefronts as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles = 64 accumulator VGPRs.  Rows are the MFMA's M side on
// purpose: a lane then holds ONE question (column) and 16 rows per tile, so the arg-max over rows is mostly
// in-lane; one exchange with lane^32 and one ds_max_u64 finish it.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) volatile float lds_vf;

// What the matrix pipe leaves to everything else (tools/mfma_probe.hip; matrix-pipe cycles taken from a saturated
// stream of these 64-cycle MFMAs, three wavefronts per SIMD): an ordinary vector instruction ~4, ds_write2_b32 ~3.5,
// ds_read2_b32 ~2, buffer_load_dwordx4 with a 32-bit offset ~12, global_load_dwordx4 with a 64-bit address pair ~50.
// And a wavefront OUTSIDE its MFMA loop gets a vector instruction issued only about once per MFMA of the others
// (55-80 cycles each, s_setprio changes nothing): a 250-instruction arg-max lasts 17 K cycles next to 25.6 K cycles
// of matrix work, and for that long its SIMD has one MFMA stream fewer.  Hence:
//   * software pipeline with HALF a slab (8 k = 4 MFMA k-pairs x 4 tiles = 16 MFMAs = 1024 matrix cycles) as the unit:
//     fragment reads of the next half, staging writes of the next slab and the loads of the slab after it are
//     issued under the MFMAs of the current half; one barrier per slab.  `nh` halves, nh = ceil(size / 8): the zero
//     columns that pad a row to a multiple of 16 are never multiplied;
//   * loads through buffer descriptors of the two 128-row operand tiles, one 32-bit lane offset for both and the
//     slab / row-block advance in scalar registers: no vector instruction computes an address in the loop;
//   * an arg-max that usually ends after 40 instructions: the lane maxima (v_max3) are compared with the best key
//     the question already has in memory; only a wavefront that can still improve one scans for row numbers;
//   * fragments through single ds_read_b32 (volatile keeps them from being merged): 16-bit offsets from ONE address
//     register per operand, where ds_read2_b32 (8-bit offsets) needs one per k row -- ~50 registers, the difference
//     between two and THREE workgroups per CU (168 registers), which is worth +6 % here: a workgroup's prologue
//     (~5 K cycles), its arg-max and the ~6 K cycles until its successor starts in the same slot are covered by
//     two other workgroups instead of one.  At most 8 reads per burst: the LDS wait counter has four bits.
// Measured and rejected: persistent workgroups that walk their tiles as one slab stream
// (with or without the arg-max of tile t riding between the MFMAs of tile t+1): 5-10 % slower than this.
__global__ void __launch_bounds__(ETHREADS, 3)
k_eval_scores_mfma(const float *__restrict__ Q, const float *__restrict__ M, int nq, int words, int ld, int nh,
                   int q_tiles, int c_tiles, int c_per_xcd, int q_group, const int *__restrict__ b1,
                   const int *__restrict__ b2, const int *__restrict__ b3, unsigned long long *__restrict__ best) {
  constexpr int MK = 16;
  // row pitch 130 floats: the staging writes of a 32-lane group (4 k-quads x 8 rows) fall on 32 different banks
  // (4 * 130 = 8 mod 32); the fragment reads are 32 consecutive floats of one k row whatever the pitch
  constexpr int MLD = EBM + 2;
  __shared__ float As[2][MK][MLD];       // rows of M
  __shared__ float Bs[2][MK][MLD];       // questions
  __shared__ unsigned long long skey[EBN];

  // XCD-aware order: an XCD owns a stripe of row tiles; inside it consecutive workgroups take `q_group` question
  // tiles of ONE row tile before moving to the next row tile, so a row tile fetched into that XCD's L2 is reused
  // q_group times while it is hot (the stripe itself, 6 MB at text8 size, does not fit the 4 MB L2).
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int per_blk = q_group * c_per_xcd, blk = local / per_blk, r = local - blk * per_blk;
  const int qt = blk * q_group + r % q_group, ct = xcd * c_per_xcd + r / q_group;
  if (qt >= q_tiles || ct >= c_tiles) return;
  const int m0 = ct * EBM, n0 = qt * EBN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
  const int lk2 = lane >> 5, l32 = lane & 31;
  if (tid < EBN) skey[tid] = 0ull;

  // what the epilogue needs from memory, requested now: the question words of this lane's two questions (excluded
  // from the arg-max, ref :169-171) and the best key each question has so far (possibly stale: then it is only lower)
  int qw[2][3];
  unsigned long long seen[2];
#pragma unroll
  for (int nt = 0; nt < 2; nt++) {
    const int q = n0 + wn + nt * 32 + l32;
    qw[nt][0] = q < nq ? b1[q] : -1;
    qw[nt][1] = q < nq ? b2[q] : -1;
    qw[nt][2] = q < nq ? b3[q] : -1;
    seen[nt] = q < nq ? __hip_atomic_load(&best[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
  }

  // global -> register staging: 128 rows x 16 k per operand; a thread covers rows lrow0 and lrow0 + 64
  constexpr int KQ = MK / 4, RS = ETHREADS / KQ, NL = EBM / RS;   // float4 per row, rows per pass, passes
  const int lrow0 = tid / KQ, lk = (tid % KQ) * 4;
  const int tile_bytes = EBM * ld * 4, rstep_bytes = RS * ld * 4;
  const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(M + (long long)m0 * ld), 0, tile_bytes, 0x27000);
  const __amdgpu_buffer_rsrc_t rb_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(Q + (long long)n0 * ld), 0, tile_bytes, 0x27000);
  const int voff = (lrow0 * ld + lk) * 4;
  // two staging sets: slab s+2 is requested while slab s+1 still waits in registers for its LDS buffer, i.e. a
  // load has two slabs of matrix work (>= 4096 cycles) to arrive -- one slab is not enough for a miss of the XCD's L2
  struct Stage { f32x4 a[NL], b[NL]; };
  Stage st0, st1;
  int next_slab = 0, slab_off = 0;      // the slab the next stage_load fetches and its byte offset in a row
  auto stage_load = [&](Stage &st, int nslab) {
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const u32x4 ta = __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, voff, slab_off + i * rstep_bytes, 0);
      const u32x4 tb = __builtin_amdgcn_raw_buffer_load_b128(rb_rsrc, voff, slab_off + i * rstep_bytes, 0);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        st.a[i][e] = __uint_as_float(ta[e]);
        st.b[i][e] = __uint_as_float(tb[e]);
      }
    }
    slab_off += next_slab + 1 < nslab ? MK * 4 : 0;     // never past the last slab (a repeated load is never stored)
    next_slab++;
  };
  auto stage_store = [&](const Stage &st, int buf) {
#pragma unroll
    for (int i = 0; i < NL; i++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        As[buf][lk + e][lrow0 + RS * i] = st.a[i][e];
        Bs[buf][lk + e][lrow0 + RS * i] = st.b[i][e];
      }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  // fragments of one half: k-pair p of half h is k = 8 h + 2 p + (lane / 32)
  auto read_pairs = [&](float (&fa)[4][2], float (&fq)[4][2], int buf, int h, int p0, int p1) {
#pragma unroll
    for (int p = p0; p < p1; p++) {
      const int k = h * 8 + 2 * p + lk2;
      fa[p][0] = *(lds_vf *)&As[buf][k][wm + l32];
      fa[p][1] = *(lds_vf *)&As[buf][k][wm + 32 + l32];
      fq[p][0] = *(lds_vf *)&Bs[buf][k][wn + l32];
      fq[p][1] = *(lds_vf *)&Bs[buf][k][wn + 32 + l32];
    }
  };
  auto read_half = [&](float (&fa)[4][2], float (&fq)[4][2], int buf, int h) { read_pairs(fa, fq, buf, h, 0, 4); };
  auto mma_pairs = [&](const float (&fa)[4][2], const float (&fq)[4][2], int p0, int p1) {
#pragma unroll
    for (int p = p0; p < p1; p++) {             // strictly increasing k: the chain order of the reference
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p][0], fq[p][0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p][0], fq[p][1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p][1], fq[p][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p][1], fq[p][1], acc[1][1], 0, 0, 0);
    }
  };
  auto mma_half = [&](const float (&fa)[4][2], const float (&fq)[4][2]) { mma_pairs(fa, fq, 0, 4); };
#define W2B_PIN() __builtin_amdgcn_sched_barrier(0)    /* the machine scheduler would re-serialise the pipeline */

  const int nslab = (nh + 1) >> 1;
  float f0a[4][2], f0q[4][2], f1a[4][2], f1q[4][2];
  stage_load(st0, nslab);
  stage_store(st0, 0);
  __syncthreads();
  stage_load(st0, nslab);               // slab 1
  stage_load(st1, nslab);               // slab 2
  read_half(f0a, f0q, 0, 0);
  // steady state, free of branches so that the wait counters stay exact: slab kt has both halves and a successor,
  // which waits in `st`
  auto slab = [&](int kt, Stage &st) {
    const int buf = kt & 1;
    read_pairs(f1a, f1q, buf, 1, 0, 2);
    W2B_PIN();
    mma_pairs(f0a, f0q, 0, 2);
    W2B_PIN();
    read_pairs(f1a, f1q, buf, 1, 2, 4);
    W2B_PIN();
    mma_pairs(f0a, f0q, 2, 3);
    W2B_PIN();
    stage_store(st, buf ^ 1);           // buf^1 was last read two halves ago, before the previous barrier
    W2B_PIN();
    mma_pairs(f0a, f0q, 3, 4);          // the staging writes complete under these four
    W2B_PIN();
    __syncthreads();
    stage_load(st, nslab);              // slab kt + 3
    read_pairs(f0a, f0q, buf ^ 1, 0, 0, 2);
    W2B_PIN();
    mma_pairs(f1a, f1q, 0, 2);
    W2B_PIN();
    read_pairs(f0a, f0q, buf ^ 1, 0, 2, 4);
    W2B_PIN();
    mma_pairs(f1a, f1q, 2, 4);
    W2B_PIN();
  };
  int kt = 0;
  for (; kt + 2 < nslab; kt += 2) {
    slab(kt, st0);
    slab(kt + 1, st1);
  }
  if (kt + 1 < nslab) slab(kt, st0);
  {                                     // last slab: one half when nh is odd
    const int buf = (nslab - 1) & 1;
    if (nh & 1) {
      mma_half(f0a, f0q);
    } else {
      read_half(f1a, f1q, buf, 1);
      mma_half(f0a, f0q);
      mma_half(f1a, f1q);
    }
  }
#undef W2B_PIN

  // epilogue (ref :166-175, N = 1).  Accumulator e of tile (mt, nt) in lane l is
  // row m = wm + mt*32 + 8*(e/4) + 4*(l/32) + e%4, question n = wn + nt*32 + l%32; a lane walks its rows in
  // increasing order, so "strictly greater" keeps the lowest row among equal scores (the reference's first-wins).
  const int r0 = m0 + wm;
#pragma unroll
  for (int nt = 0; nt < 2; nt++) {
    const int nl = wn + nt * 32 + l32, q = n0 + nl;
    const bool live = q < nq;
    const int e1 = qw[nt][0], e2 = qw[nt][1], e3 = qw[nt][2];
    // can any of this wavefront's 64 x 32 scores still improve its question's key?  A key orders by (score, lower
    // row): a score below the one already recorded never does; an equal one only with a lower row (another XCD's
    // stripe may have recorded a higher row first), so "greater or equal" goes on to the exact comparison.
    float m = acc[0][nt][0];
#pragma unroll
    for (int e = 1; e < 15; e += 2) m = __builtin_fmaxf(__builtin_fmaxf(m, acc[0][nt][e]), acc[0][nt][e + 1]);
    m = __builtin_fmaxf(m, acc[0][nt][15]);
#pragma unroll
    for (int e = 0; e < 16; e += 2) m = __builtin_fmaxf(__builtin_fmaxf(m, acc[1][nt][e]), acc[1][nt][e + 1]);
    const bool may = live && m > 0.f && __float_as_uint(m) >= (unsigned)(seen[nt] >> 32);     // (NaN: m > 0 fails)
    if (!__any(may)) continue;
    // no question word of these 32 questions among this wavefront's 64 rows and no row past the vocabulary (the
    // usual case): a float compare and two selects per accumulator
    const bool excl = (unsigned)(e1 - r0) < 64u || (unsigned)(e2 - r0) < 64u || (unsigned)(e3 - r0) < 64u;
    unsigned long long key = 0ull;
    if (r0 + 64 <= words && !__any(excl)) {
      float bd = 0.f;
      int bo = 0;
#pragma unroll
      for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const float d = acc[mt][nt][e];
          const bool g = d > bd;                       // NaN fails like `dist > bestd`
          bd = g ? d : bd;
          bo = g ? mt * 32 + 8 * (e >> 2) + (e & 3) : bo;
        }
      if (live && bd > 0.f)
        key = ((unsigned long long)__float_as_uint(bd) << 32) |
              (unsigned long long)(0xFFFFFFFFu - (unsigned)(r0 + 4 * lk2 + bo));
    } else {
#pragma unroll
      for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int c = r0 + mt * 32 + 8 * (e >> 2) + 4 * lk2 + (e & 3);
          const float d = acc[mt][nt][e];
          if (live && c < words && c != e1 && c != e2 && c != e3 && d > 0.f) {
            const unsigned long long k2 =
                ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)c);
            key = k2 > key ? k2 : key;
          }
        }
    }
    const unsigned long long other = __shfl_xor(key, 32, 64);
    key = other > key ? other : key;
    if (lk2 == 0 && key > seen[nt]) atomicMax(&skey[nl], key);
  }
  __syncthreads();
  if (tid < EBN && skey[tid] && n0 + tid < nq) atomicMax(&best[n0 + tid], skey[tid]);
}

}  // namespace

// ------------------------------------------------------------------------------------ launchers
hipError_t w2b_launch_eval_normalize(float *M, long long words, long long size, long long ld, int bitlevel,
                                     int fused, float *len, hipStream_t s) {
  QParam qp;
  qp.bitlevel = bitlevel;
  qp.steps_i = (bitlevel >= 4) ? (1 << (bitlevel - 1)) : 1;
  qp.steps_f = (float)qp.steps_i;
  if (words <= 0) return hipSuccess;
  const int blocks = (int)((words + 63) / 64);
  return dispatch_q(bitlevel, [&](auto qm) {
    constexpr int QM = decltype(qm)::value;
    if (fused) hipLaunchKernelGGL((k_eval_row_len<QM, true>), dim3(blocks), dim3(64), 0, s, M, words, size, ld, qp, len);
    else hipLaunchKernelGGL((k_eval_row_len<QM, false>), dim3(blocks), dim3(64), 0, s, M, words, size, ld, qp, len);
    hipLaunchKernelGGL((k_eval_scale<QM>), dim3(2048), dim3(256), 0, s, M, words, size, ld, qp, len);
    return hipGetLastError();
  });
}

hipError_t w2b_launch_eval_queries(const float *M, long long ld, long long nq, const int *b1, const int *b2,
                                   const int *b3, float *Q, int variant, hipStream_t s) {
  if (nq <= 0) return hipSuccess;
  (void)variant;
  hipLaunchKernelGGL(k_eval_queries, dim3(2048), dim3(256), 0, s, M, ld, nq, b1, b2, b3, Q);
  return hipGetLastError();
}

hipError_t w2b_launch_eval_scores(const float *Q, const float *M, int nq, int words, int size, int ld, int fused,
                                  const int *b1, const int *b2, const int *b3, unsigned long long *best,
                                  int variant, hipStream_t s) {
  if (nq <= 0 || words <= 0) return hipSuccess;
  const int q_tiles = (nq + EBM - 1) / EBM, c_tiles = (words + EBN - 1) / EBN;
  const int c_per_xcd = (c_tiles + 7) / 8;
  const long long grid = 8ll * c_per_xcd * q_tiles;
  if (fused && variant != 0) {   // matrix cores: bitwise the fused chain (the two-rounding mode has no MFMA form)
    int g = variant > 1 ? variant : 8;
    if (g > q_tiles) g = q_tiles;
    const long long grid2 = 8ll * c_per_xcd * ((q_tiles + g - 1) / g * g);
    hipLaunchKernelGGL(k_eval_scores_mfma, dim3((unsigned)grid2), dim3(ETHREADS), 0, s, Q, M, nq, words, ld,
                       (size + 7) / 8, q_tiles, c_tiles, c_per_xcd, g, b1, b2, b3, best);
  }
  else if (fused)
    hipLaunchKernelGGL((k_eval_scores<true>), dim3((unsigned)grid), dim3(ETHREADS), 0, s, Q, M, nq, words, ld,
                       q_tiles, c_tiles, c_per_xcd, b1, b2, b3, best);
  else
    hipLaunchKernelGGL((k_eval_scores<false>), dim3((unsigned)grid), dim3(ETHREADS), 0, s, Q, M, nq, words, ld,
                       q_tiles, c_tiles, c_per_xcd, b1, b2, b3, best);
  return hipGetLastError();
}
