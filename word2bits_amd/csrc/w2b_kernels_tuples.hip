// w2b_kernels_tuples.hip -- form (ii): explicit (centre, context, negatives) tuples.  See w2b_device.hpp.
#include "w2b_device.hpp"

namespace {
// ------------------------------------------------------------------------------------ form (ii): tuples
template <int QM, int VEC, bool LOSS, int MAXTHREADS, int MM, int ATOM = 0>
__global__ void __launch_bounds__(MAXTHREADS, (MAXTHREADS <= 256 ? W2B_MINWAVES : 1)) k_train_tuples(const W2bParams P, const long long n,
                                                      const int32_t *__restrict__ center,
                                                      const int32_t *__restrict__ ctx_off,
                                                      const int32_t *__restrict__ ctx,
                                                      const int32_t *__restrict__ neg,
                                                      const float alpha) {
  extern __shared__ int smem[];
  WordLds L = carve_word_lds(smem, P.window, P.negative, VEC);
  int *s_cnt = L.cend + round4(P.negative + 1);   // [0] cw, [1] nt
  if (MM == W2B_MM_EXACT) L.xprod = reinterpret_cast<float *>(s_cnt + 4);
  const int tid = threadIdx.x, lane = tid & 63;
  // this XCD's copies of the hottest rows of u and v (16-byte columns, coherent rows, not in the parity mode)
  const bool hot = (VEC == 4 && MM == 0 && P.xhot != nullptr && P.xhot_u + P.xhot_v > 0);
  XHot XH = xhot_here(P);
  if (!hot) { XH.nu = 0; XH.nv = 0; }
  int since_merge = 0, merge_cursor = (int)(blockIdx.x >> 3) * P.xhot_m;   // (workgroup b runs on XCD b % 8: take turns)
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
      if (cw > 0) prep_lists<TFor<LOSS>::value, int *>(L.tgt, L.prev, L.cend, 1 + cnt, L.ctx, L.umult, cw, lane);
    }
    __syncthreads();
    const int cw = s_cnt[0], nt = s_cnt[1];
    bool wide = false;
    if constexpr (VEC == 1 && MAXTHREADS == 1024) wide = P.wide != 0;
    if (cw > 0) {
      if (wide) { if constexpr (VEC == 1 && MAXTHREADS == 1024) process_word_wide<QM, LOSS, MM>(P, L, qp, cw, nt, alpha, loss_acc); }
      else process_word<QM, VEC, LOSS, MM, ATOM>(P, L, qp, cw, nt, alpha, loss_acc, XH);
    } else __syncthreads();
    if (VEC == 4 && hot && ++since_merge >= P.hot_period) {
      since_merge = 0;
      xhot_merge_event<MM>(P, XH, merge_cursor, tid * VEC, tid * VEC < P.dim);
    }
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

}  // namespace

// instantiation), so the grid-stride loop has no tail of late-starting workgroups.
int w2b_tuple_max_grid(int num_cus) { return num_cus * 32; }   // 32 wavefronts per CU is all the hardware holds
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
  const size_t lds = w2b_lds_bytes(p.dim, p.window, p.negative, false, p.exact != 0);
  return dispatch_mm_exact(p.mem_mode, p.exact, [&](auto mm) -> hipError_t {
  constexpr int MM = decltype(mm)::value;
  return dispatch_q(p.bitlevel, [&](auto qm) -> hipError_t {
    constexpr int QM = decltype(qm)::value;
#define W2B_LAUNCH_T2(VEC, LOSS, MAXT)                                                                      \
    do {                                                                                                     \
      auto kern = k_train_tuples<QM, VEC, LOSS, MAXT, MM>;                                                       \
      int g = grid > 0 ? grid : auto_grid(kern, threads, lds, num_cus, per_cu_override, n);                  \
      hipLaunchKernelGGL(kern, dim3(g), dim3(threads), lds, s, p, n, center, ctx_off, ctx, neg, alpha); \
    } while (0)
#define W2B_LAUNCH_T(VEC, LOSS) \
    do { if (threads <= 256) W2B_LAUNCH_T2(VEC, LOSS, 256); else W2B_LAUNCH_T2(VEC, LOSS, 1024); } while (0)
    if constexpr (MM == 0) {          // rows updated with atomic adds: the ATOM instantiations (16-byte columns, <= 256 threads)
      const int atom = p.atomic_rank > 0 ? 2 : (p.atomic_rank_u > 0 ? 1 : 0);
      if (atom && vec == 4 && threads <= 256) {
#define W2B_LAUNCH_TA(LOSS, ATOM)                                                                            \
        do {                                                                                                 \
          auto kern = k_train_tuples<QM, 4, LOSS, 256, 0, ATOM>;                                             \
          int g = grid > 0 ? grid : auto_grid(kern, threads, lds, num_cus, per_cu_override, n);              \
          hipLaunchKernelGGL(kern, dim3(g), dim3(threads), lds, s, p, n, center, ctx_off, ctx, neg, alpha);  \
        } while (0)
        if (atom == 2) { if (loss) W2B_LAUNCH_TA(true, 2); else W2B_LAUNCH_TA(false, 2); }
        else { if (loss) W2B_LAUNCH_TA(true, 1); else W2B_LAUNCH_TA(false, 1); }
#undef W2B_LAUNCH_TA
        return hipGetLastError();
      }
    }
    if (vec == 4) { if (loss) W2B_LAUNCH_T(4, true); else W2B_LAUNCH_T(4, false); }
    else { if (loss) W2B_LAUNCH_T(1, true); else W2B_LAUNCH_T(1, false); }
#undef W2B_LAUNCH_T
#undef W2B_LAUNCH_T2
    return hipGetLastError();
  });
  });
}

