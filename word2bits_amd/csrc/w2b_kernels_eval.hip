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

template <int MK>
__global__ void __launch_bounds__(ETHREADS, 2)
k_eval_scores_mfma(const float *__restrict__ Q, const float *__restrict__ M, int nq, int words, int ld, int q_tiles,
                   int c_tiles, int c_per_xcd, int q_group, const int *__restrict__ b1, const int *__restrict__ b2,
                   const int *__restrict__ b3, unsigned long long *__restrict__ best) {
  __shared__ float As[2][MK][ELD];       // rows of M
  __shared__ float Bs[2][MK][ELD];       // questions
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
  if (tid < EBN) skey[tid] = 0ull;

  // global -> register staging: 128 rows x MK k per operand; a thread covers rows lrow0 + (256 / KQ) * i
  constexpr int KQ = MK / 4, RS = ETHREADS / KQ, NL = EBM / RS;   // float4 per row, rows per pass, passes
  const int lrow0 = tid / KQ, lk = (tid % KQ) * 4;
  const float *ga = M + (long long)(m0 + lrow0) * ld + lk;
  const float *gb = Q + (long long)(n0 + lrow0) * ld + lk;
  const long long rstep = (long long)RS * ld;
  f32x4 ra[NL], rb[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) {
    ra[i] = *(const f32x4 *)(ga + i * rstep);
    rb[i] = *(const f32x4 *)(gb + i * rstep);
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  const int lk2 = lane >> 5, l32 = lane & 31;
  const int nk = ld / MK;
  for (int kt = 0; kt < nk; kt++) {
    const int buf = kt & 1;
#pragma unroll
    for (int i = 0; i < NL; i++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        As[buf][lk + e][lrow0 + RS * i] = ra[i][e];
        Bs[buf][lk + e][lrow0 + RS * i] = rb[i][e];
      }
    __syncthreads();
    if (kt + 1 < nk) {
      ga += MK;
      gb += MK;
#pragma unroll
      for (int i = 0; i < NL; i++) {
        ra[i] = *(const f32x4 *)(ga + i * rstep);
        rb[i] = *(const f32x4 *)(gb + i * rstep);
      }
    }
    // all fragments of the slab first (the LDS latency is paid once per slab, not once per four MFMAs)
#pragma unroll
    for (int g = 0; g < MK / 16; g++) {         // groups of 8 k-pairs: 32 fragment registers live at a time
      float fa[8][2], fq[8][2];
#pragma unroll
      for (int p = 0; p < 8; p++) {
        const int k = g * 16 + 2 * p + lk2;
        fa[p][0] = As[buf][k][wm + l32];
        fa[p][1] = As[buf][k][wm + 32 + l32];
        fq[p][0] = Bs[buf][k][wn + l32];
        fq[p][1] = Bs[buf][k][wn + 32 + l32];
      }
      __builtin_amdgcn_sched_barrier(0);        // keep the reads ahead of the MFMAs (the scheduler would sink them)
#pragma unroll
      for (int p = 0; p < 8; p++) {             // strictly increasing k: the chain order of the reference
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p][0], fq[p][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p][0], fq[p][1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p][1], fq[p][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p][1], fq[p][1], acc[1][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // epilogue (ref :166-175, N = 1).  Accumulator e of tile (mt, nt) in lane l is
  // row m = wm + mt*32 + 8*(e/4) + 4*(l/32) + e%4, question n = wn + nt*32 + l%32.
#pragma unroll
  for (int nt = 0; nt < 2; nt++) {
    const int nl = wn + nt * 32 + l32, q = n0 + nl;
    const bool live = q < nq;
    const int e1 = live ? b1[q] : -1, e2 = live ? b2[q] : -1, e3 = live ? b3[q] : -1;
    unsigned long long key = 0ull;
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int c = m0 + wm + mt * 32 + 8 * (e >> 2) + 4 * lk2 + (e & 3);
        const float d = acc[mt][nt][e];
        if (live && c < words && c != e1 && c != e2 && c != e3 && d > 0.f) {
          const unsigned long long k2 =
              ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)c);
          key = k2 > key ? k2 : key;
        }
      }
    const unsigned long long other = __shfl_xor(key, 32, 64);
    key = other > key ? other : key;
    if (lk2 == 0 && key) atomicMax(&skey[nl], key);
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

hipError_t w2b_launch_eval_scores(const float *Q, const float *M, int nq, int words, int ld, int fused,
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
    if (ld % 32 == 0 && getenv("W2B_EVAL_MK32"))   // experiment: fewer barriers, half the occupancy -- slower
      hipLaunchKernelGGL(k_eval_scores_mfma<32>, dim3((unsigned)grid2), dim3(ETHREADS), 0, s, Q, M, nq, words, ld,
                         q_tiles, c_tiles, c_per_xcd, g, b1, b2, b3, best);
    else
      hipLaunchKernelGGL(k_eval_scores_mfma<16>, dim3((unsigned)grid2), dim3(ETHREADS), 0, s, Q, M, nq, words, ld,
                         q_tiles, c_tiles, c_per_xcd, g, b1, b2, b3, best);
  }
  else if (fused)
    hipLaunchKernelGGL((k_eval_scores<true>), dim3((unsigned)grid), dim3(ETHREADS), 0, s, Q, M, nq, words, ld,
                       q_tiles, c_tiles, c_per_xcd, b1, b2, b3, best);
  else
    hipLaunchKernelGGL((k_eval_scores<false>), dim3((unsigned)grid), dim3(ETHREADS), 0, s, Q, M, nq, words, ld,
                       q_tiles, c_tiles, c_per_xcd, b1, b2, b3, best);
  return hipGetLastError();
}
