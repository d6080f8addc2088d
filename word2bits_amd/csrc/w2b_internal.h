// w2b_internal.h -- structures shared by the HIP kernels (w2b_kernels.hip) and the host side of the
// C ABI (w2b_trainer.cpp).  Not part of the public interface (include/word2bits_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define W2B_LCG_A 25214903917ULL   // ref src/word2bits.cpp:352 (same constants at :405,:428,:455)
#define W2B_LCG_C 11ULL
#define W2B_MAX_SEN 1000           // ref src/word2bits.cpp:32
#define W2B_MAXW 16                // max wavefronts per workgroup (1024 threads)
#define W2B_NXCD 8                 // accelerator complex dies of an MI355X, each with its own L2 (HW_REG_XCC_ID: 0..7)
#define W2B_XHOT_MAX 128           // most rows of one table with per-XCD copies

// Racy globals of the reference that all workers share (ref src/word2bits.cpp:51,53):
// alpha and word_count_actual.  One instance in device memory per trainer.
struct W2bShared {
  float alpha;
  int pad0;
  unsigned long long word_count_actual;
  double loss_tuples;   // loss sum of the tuple form (form ii)
  int workers_done;
  int corpus_overrun;   // a worker reached the end of a corpus SLICE (w2b_set_corpus_slice) before its quota: the slice was cut too short
  // replicas (one per GPU): word_count_actual above is LOCAL.  The alpha schedule (ref :391) runs on the global count:
  // what the other replicas had done at the last exchange + the assumption that each of them has advanced like this
  // one since (exact at every exchange; single replica: both fields stay 0 and the schedule is the reference's).
  unsigned long long wca_others;      // sum of the other replicas' word_count_actual at the last exchange
  unsigned long long wca_at_sync;     // this replica's word_count_actual at the last exchange
  double loss_epoch;                  // sum of the workers' total_loss of the running epoch (ref :537-538)
  int launch_done, pad1;              // worker workgroups of the running launch that have finished (the refreshers of the row-group kernel wait for it)
  unsigned long long dbg[16];   // phase timers of workgroup 0 (builds with -DW2B_PHASE_TIMERS only)
};

// Per-worker locals of TrainModelThread (ref src/word2bits.cpp:364-375) that must survive between
// launches because one epoch is cut into several kernel launches.
struct W2bWorker {
  unsigned long long rng;         // next_random
  long long cursor;               // index of the next unread corpus token
  long long word_count, last_word_count;
  double loss;                    // total_loss
  int sen_len, sen_pos;           // sentence_length, sentence_position
  int first_override;             // pending truncated first word of the shard (-2 none)
  int done;                       // epoch finished (local_iter reached 0)
  int sen[W2B_MAX_SEN];           // sen[]
};

struct W2bParams {
  float *u, *v;                   // [vocab_size][dim] fp32 masters
  const float *exp_table;         // [1000]
  const int32_t *table;           // unigram table
  long long table_size;
  const float *keep;              // sub-sampling threshold per word (nullptr when sample <= 0)
  const int32_t *corpus;          // token ids, 0 = </s>
  long long n_tokens;
  int corpus_more;                // the resident tokens are a slice and the file goes on behind them (end of slice != EOF)
  W2bWorker *workers;
  W2bShared *shared;
  const unsigned long long *jump_a, *jump_c;   // LCG jump-ahead: x_{n+k} = jump_a[k]*x_n + jump_c[k]
  long long vocab_size, train_words, iter;
  unsigned tab_bytes;             // bytes of one table when < 4 GiB (32-bit addressing of rows), else 0
  unsigned long long table_magic, window_magic;   // floor(2^64 / table_size), floor(2^64 / window)
  int dim, window, negative, bitlevel, num_threads;
  int total_threads;              // workers across all replicas (quota ref :414, alpha extrapolation)
  int mem_mode;                   // 0 coherent (sc1 row accesses), 1 relaxed (plain cached accesses)
  float *entry;                   // sentence-resident kernel: scratch rows [num_threads][2][slots][dim] (see w2b_kernels_resident.hip)
  // XCD-shared copies of the hottest rows (w2b_device.hpp "XHot"): [W2B_NXCD][copies of u rows 1..xhot_u | copies of v rows
  // 1..xhot_v | entries of the same][dim], then the merge locks [rows][W2B_MAXW].  0 / 0 = every row is accessed at its
  // master address.
  float *xhot;
  int xhot_u, xhot_v;
  int hot_period;                 // centre words between two merge events of a worker / workgroup (power of two)
  int xhot_m;                     // hot rows of each table that one merge event brings up to date
  int uavg_rank;                  // sentence-resident kernel: context rows 1..uavg_rank are merged by consensus and refreshed
  int win_refresh;                // ... at the latest after this many steps in a worker's window (0 = never)
  float xhot_w;                   // weight of one XCD's copy in the consensus of a merge (1 / W2B_NXCD by default)
  float *wide_scratch;            // rows too long for one thread per column: [workgroups][2][dim] (process_word_wide)
  int wide;                       // 1: such rows (1024 threads, every thread owns several columns)
  int atomic_rank;                // rows 1..atomic_rank (by count) of v are updated with fp32 atomic adds: no lost updates (see w2b_tuning)
  int atomic_rank_u;              // the same for u (plain worker / tuple kernels: the context rows' phase C)
  int fresh_rank_u;               // plain kernels, phase C: context rows 1..fresh_rank_u are re-read before their update instead of
                                  // taken from the LDS stash of phase A (the update lands on the current value, ref :500-502)
  int exact;                      // serial dot product in the reference's order (plain worker / tuple kernels)
  // Refreshed read copies of the hottest context rows (row-group kernel, w2b_kernels_groups.hip): rows 1..rc_rows of u are READ
  // at this XCD's copy (nt: served by the XCD's L2) while their updates stay lossless adds at the master address; one
  // refresher workgroup per XCD copies master -> copy in a loop for as long as the launch's workers run.
  int rc_rows;                    // 0 = none
  float *rc;                      // [W2B_NXCD][rc_rows][dim]
  int *rc_flags;                  // [0..8) claim (one refresher per XCD), [16..24) alive (the XCD's copies have been filled in this launch)
  int worker_base;                // plain worker kernel: workgroup b is worker worker_base + b (a launch may cover a slice of the workers:
                                  // w2b_tuning.concurrent_workers)
  float starting_alpha, sample, reg;
};

// launchers implemented in w2b_kernels.hip --------------------------------------------------------
// block size chosen from dim: one thread per 16-byte (or 4-byte) column of a row
int w2b_block_threads(int dim, int *vec_out, int *wide_out = nullptr);
size_t w2b_lds_bytes(int dim, int window, int negative, bool worker_form, bool exact = false);
int w2b_tuple_max_grid(int num_cus);                                   // most workgroups w2b_launch_tuples starts
hipError_t w2b_launch_tuples(const W2bParams &p, long long n, const int32_t *center,
                             const int32_t *ctx_off, const int32_t *ctx, const int32_t *neg,
                             float alpha, int grid, int num_cus, int per_cu_override, bool loss,
                             hipStream_t s);
hipError_t w2b_launch_workers(const W2bParams &p, long long max_positions, bool loss, hipStream_t s, int grid = 0);   // grid workgroups = workers worker_base .. worker_base + grid
// sentence-resident variant (w2b_kernels_resident.hip): radius >= 0 when it can run for this shape
int w2b_resident_plan(int dim, int window, int negative);
bool w2b_resident_atomic_ok(const W2bParams &p, int radius);           // atomic_rank > 0: can the sentence-resident kernel do it?
long long w2b_resident_scratch_rows(int radius);                       // scratch rows per worker
hipError_t w2b_launch_resident(const W2bParams &p, long long max_positions, int radius, bool loss, hipStream_t s,
                               bool debug = false);
// row-group variant of the worker kernel (w2b_kernels_groups.hip): short rows / few workers, every row shared
bool w2b_groups_ok(const W2bParams &p);                                 // can it run this shape / these row rules?
size_t w2b_groups_lds_bytes(int dim, int window, int negative);
int w2b_groups_per_cu(const W2bParams &p, bool loss);                   // resident workgroups per CU
hipError_t w2b_launch_groups(const W2bParams &p, long long max_positions, bool loss, hipStream_t s);
hipError_t w2b_launch_refresher(const W2bParams &p, hipStream_t s);        // k_refresh_rows, beside a launch of the row-group kernel
#define W2B_RC_MAX 64               // most rows of u with refreshed read copies
#define W2B_RC_BLOCKS 16            // refresher workgroups started per launch (one per XCD claims its copy, the others leave)
int w2b_workers_per_cu(const W2bParams &p, bool loss);                  // resident workgroups per CU, plain kernel
int w2b_resident_per_cu(const W2bParams &p, int radius, bool loss);     // ... sentence-resident kernel
hipError_t w2b_launch_init_net(float *u, float *v, long long n_per_table, const float *lut,
                               hipStream_t s);
hipError_t w2b_launch_export(const float *u, const float *v, float *out, long long n, int bitlevel,
                             hipStream_t s);
hipError_t w2b_launch_export_packed(const float *u, const float *v, unsigned long long *out, long long rows, int dim,
                                    int bitlevel, hipStream_t s);      // bitlevel 1 / 2: include/word2bits_corpus.h layout
// analogy evaluator (w2b_kernels_eval.hip; ref src/compute-accuracy.c:106-110,155-177)
hipError_t w2b_launch_eval_normalize(float *M, long long words, long long size, long long ld, int bitlevel,
                                     int fused, float *len_scratch, hipStream_t s);
hipError_t w2b_launch_eval_queries(const float *M, long long ld, long long nq, const int *b1, const int *b2,
                                   const int *b3, float *Q, int variant, hipStream_t s);
hipError_t w2b_launch_eval_scores(const float *Q, const float *M, int nq, int words, int size, int ld, int fused,
                                  const int *b1, const int *b2, const int *b3, unsigned long long *best,
                                  int variant /* 0: vector-ALU kernel always; else MFMA when fused */, hipStream_t s);
// bit-packed model files (w2b_corpus.cpp; format in include/word2bits_corpus.h)
#include <string>
#include <vector>
bool w2b_internal_is_packed(const unsigned char *d, size_t n);
int w2b_internal_parse_packed(const unsigned char *d, size_t n, std::vector<std::string> &words, std::vector<float> &values,
                              int64_t *dim_out);
int w2b_internal_fail(int code, const char *msg);   // sets w2b_last_error() (w2b_trainer.cpp)
struct w2b_trainer;
// what the evaluator needs from a live trainer (w2b_trainer.cpp): device tables, shape, bitlevel, device, stream
void w2b_internal_trainer_view(w2b_trainer *t, float **u, float **v, long long *V, long long *D, int *bitlevel,
                               int *device, hipStream_t *stream);
// per-XCD copies of the hottest rows meet their master rows (before / after every training launch; idempotent)
hipError_t w2b_launch_xhot_fold(const W2bParams &p, hipStream_t s);
hipError_t w2b_launch_wca_pack(const W2bShared *sh, unsigned long long *buf, hipStream_t s);   // buf[0] = local word count
hipError_t w2b_launch_wca_unpack(W2bShared *sh, const unsigned long long *buf, hipStream_t s); // from buf[1] = global sum
// replica exchange, one chunk (w2b_kernels_misc.hip): d = s = w - base;  w += a * s - d, base += a * s
hipError_t w2b_launch_xchg_delta(float *w, const float *base, float *d, float *s_, long long n, hipStream_t s);
hipError_t w2b_launch_xchg_apply(float *w, float *base, const float *d, const float *s_, float a, long long n,
                                 const float *fac /* per-row factors on the sum (k_xchg_factor) or nullptr */, long long first, int dim,
                                 int bitlevel, int cells /* 1: per element the whole sum where it stays in the safe step's quantization cell */,
                                 hipStream_t s);
// cnt[2 V]: contributors per row (in) -> factor on the summed delta (out); rule 0 exponential saturation (rate = expected updates
// per centre word), 1 hard threshold
hipError_t w2b_launch_xchg_factor(float *cnt, const float *rate, float words, float tau_u, float tau_v, long long V, int rule,
                                  int sat_u, int sat_v, hipStream_t s);
hipError_t w2b_launch_xchg_touched(const float *w, const float *base, float *cnt, long long rows, int dim, hipStream_t s);
