// w2b_kernels_workers.hip -- form (i): one workgroup per Hogwild worker, walking its corpus shard like
// TrainModelThread (ref src/word2bits.cpp:363-516).  See w2b_device.hpp.
#include "w2b_device.hpp"

namespace {
// ------------------------------------------------------------------------------------ form (i): workers

template <int QM, int VEC, bool LOSS, int MAXTHREADS, int MM, int ATOM = 0, int TB = -1>
__global__ void __launch_bounds__(MAXTHREADS, (MAXTHREADS <= 256 ? W2B_MINWAVES : 1)) k_train_workers(const W2bParams P, const long long max_positions) {
  extern __shared__ int smem[];
  WordLds L = carve_word_lds(smem, P.window, P.negative, VEC);
  int *s_sen = L.cend + round4(P.negative + 1);
  WorkerLds *S = reinterpret_cast<WorkerLds *>(s_sen + round4(W2B_MAX_SEN) + 4);
  if (MM == W2B_MM_EXACT) L.xprod = reinterpret_cast<float *>(reinterpret_cast<int *>(S) + (sizeof(WorkerLds) + 3) / 4 + 4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wid = blockIdx.x + P.worker_base;
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
  // this XCD's copies of the hottest rows of u and v (16-byte columns, coherent rows, not in the parity mode)
  const bool hot = (VEC == 4 && MM == 0 && P.xhot != nullptr && P.xhot_u + P.xhot_v > 0);
  XHot XH = xhot_here(P);
  if (!hot) { XH.nu = 0; XH.nv = 0; }
  int since_merge = 0, merge_cursor = (wid >> 3) * P.xhot_m;      // (workgroup b runs on XCD b % 8: take turns)
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
          const long long wca_all = w2b_global_progress(P, (long long)wca);
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
        const int b = (int)fast_mod(rng, (unsigned long long)W, P.window_magic);
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
              t = P.table[fast_mod(x >> 16, (unsigned long long)P.table_size, P.table_magic)];
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
          prep_lists<TFor<LOSS>::value, int *>(L.tgt, L.prev, L.cend, nt, L.ctx, L.umult, cw, lane);
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
    bool wide = false;
    if constexpr (VEC == 1 && MAXTHREADS == 1024) wide = P.wide != 0;
    if (cw > 0) {
      if (wide) { if constexpr (VEC == 1 && MAXTHREADS == 1024) process_word_wide<QM, LOSS, MM>(P, L, qp, cw, nt, alpha, loss_acc); }
      else process_word<QM, VEC, LOSS, MM, ATOM, TB>(P, L, qp, cw, nt, alpha, loss_acc, XH);
    } else __syncthreads();
    if (VEC == 4 && hot && ++since_merge >= P.hot_period) {
      since_merge = 0;
      xhot_merge_event<MM, TB>(P, XH, merge_cursor, tid * VEC, tid * VEC < P.dim);
    }
  }
  // save the worker
  __syncthreads();
  const int sl = S->sen_len;
  for (int i = tid; i < sl; i += blockDim.x) G->sen[i] = s_sen[i];
  double lsum = 0.0;
  if (LOSS) {
    // wave 0 holds the log-sigmoid terms on its lanes; lane 0 of every wave holds reg terms
    if (wave == 0) lsum = wave_sum_d(loss_acc);
    else if (lane == 0 && loss_acc != 0.0) { atomicAdd(&G->loss, loss_acc); atomicAdd(&P.shared->loss_epoch, loss_acc); }
  }
  if (tid == 0) {
    G->rng = S->rng; G->cursor = S->cursor; G->word_count = S->wc; G->last_word_count = S->last_wc;
    G->sen_len = S->sen_len; G->sen_pos = S->sen_pos; G->first_override = S->override_;
    if (LOSS) { atomicAdd(&G->loss, lsum); atomicAdd(&P.shared->loss_epoch, lsum); }
    if (S->done) { G->done = 1; atomicAdd(&P.shared->workers_done, 1); }
  }
}

}  // namespace

int w2b_workers_per_cu(const W2bParams &p, bool loss) {
  int vec;
  const int threads = w2b_block_threads(p.dim, &vec);
  const size_t lds = w2b_lds_bytes(p.dim, p.window, p.negative, true, p.exact != 0);
  int nb = 0;
  (void)dispatch_mm_exact(p.mem_mode, p.exact, [&](auto mm) -> hipError_t {
    constexpr int MM = decltype(mm)::value;
    return dispatch_q(p.bitlevel, [&](auto qm) -> hipError_t {
      constexpr int QM = decltype(qm)::value;
      // (the LOSS / VEC / launch-bound variants share the register budget of their block size)
      if (threads <= 256) return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_train_workers<QM, 4, false, 256, MM>, threads, lds);
      return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_train_workers<QM, 4, false, 1024, MM>, threads, lds);
    });
  });
  return nb > 0 ? nb : 1;
}

hipError_t w2b_launch_workers(const W2bParams &p, long long max_positions, bool loss, hipStream_t s, int grid) {
  if (grid <= 0) grid = p.num_threads - p.worker_base;
  int vec;
  const int threads = w2b_block_threads(p.dim, &vec);
  const size_t lds = w2b_lds_bytes(p.dim, p.window, p.negative, true, p.exact != 0);
  return dispatch_mm_exact(p.mem_mode, p.exact, [&](auto mm) -> hipError_t {
  constexpr int MM = decltype(mm)::value;
  return dispatch_q(p.bitlevel, [&](auto qm) -> hipError_t {
    constexpr int QM = decltype(qm)::value;
#define W2B_LAUNCH_W(VEC, LOSS) \
    do { if (threads <= 256) hipLaunchKernelGGL((k_train_workers<QM, VEC, LOSS, 256, MM>), dim3(grid), dim3(threads), lds, s, p, max_positions); \
         else hipLaunchKernelGGL((k_train_workers<QM, VEC, LOSS, 1024, MM>), dim3(grid), dim3(threads), lds, s, p, max_positions); } while (0)
    if constexpr (MM == 0) {          // coherent rows, 16-byte columns, at most 256 threads: the instantiations with the row addressing
      // fixed at compile time (TB) and, where rows are updated with atomic adds (w2b_tuning.atomic_rank*), the ATOM ones
      const int atom = p.atomic_rank > 0 ? 2 : (p.atomic_rank_u > 0 ? 1 : 0);
      if (vec == 4 && threads <= 256 && (atom || p.tab_bytes != 0)) {
#define W2B_LAUNCH_A(LOSS, ATOM, TB) hipLaunchKernelGGL((k_train_workers<QM, 4, LOSS, 256, 0, ATOM, TB>), dim3(grid), dim3(threads), lds, s, p, max_positions)
#define W2B_LAUNCH_AT(LOSS, ATOM) do { if (p.tab_bytes != 0) W2B_LAUNCH_A(LOSS, ATOM, 0); else W2B_LAUNCH_A(LOSS, ATOM, -1); } while (0)
        if (atom == 2) { if (loss) W2B_LAUNCH_AT(true, 2); else W2B_LAUNCH_AT(false, 2); }
        else if (atom == 1) { if (loss) W2B_LAUNCH_AT(true, 1); else W2B_LAUNCH_AT(false, 1); }
        else { if (loss) W2B_LAUNCH_A(true, 0, 0); else W2B_LAUNCH_A(false, 0, 0); }
#undef W2B_LAUNCH_AT
#undef W2B_LAUNCH_A
        return hipGetLastError();
      }
    }
    if (vec == 4) { if (loss) W2B_LAUNCH_W(4, true); else W2B_LAUNCH_W(4, false); }
    else { if (loss) W2B_LAUNCH_W(1, true); else W2B_LAUNCH_W(1, false); }
#undef W2B_LAUNCH_W
    return hipGetLastError();
  });
  });
}

