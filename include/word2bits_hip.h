/*
 * word2bits_hip.h -- C ABI of the MI355X (gfx950) training hot path of Word2Bits.
 *
 * The reference (agnusmaximus/Word2Bits) has no plugin/FFI layer: its hot path is the pthread
 * start routine  void *TrainModelThread(void *id)  (src/word2bits.cpp:363-516) that communicates
 * through process globals (src/word2bits.cpp:45-61).  This header is the seam a host program
 * binds instead of that routine: the globals become an opaque trainer object, the per-epoch
 * pthread_create/pthread_join pair (src/word2bits.cpp:535-536) becomes
 * w2b_epoch_begin() + w2b_train_step()... until finished, and the save loop's
 * quantize(u+v) (src/word2bits.cpp:549-550,568-569) becomes w2b_export_quantized().
 *
 * Conventions: plain C symbols, opaque handle, int return (0 = W2B_OK, <0 = error, text via
 * w2b_last_error()), caller-owned host buffers, library-owned device buffers, no exceptions,
 * no torch / C++ types.  All "ref" citations are paths inside the reference repository.
 *
 * There is NO CPU fallback: every compute entry point fails with W2B_ENOGPU when no
 * gfx950 device is usable.
 */
#ifndef WORD2BITS_HIP_H
#define WORD2BITS_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define W2B_OK            0
#define W2B_EINVAL       -1   /* bad argument */
#define W2B_ENOGPU       -2   /* no usable HIP device */
#define W2B_EHIP         -3   /* HIP runtime error (see w2b_last_error) */
#define W2B_ENOMEM       -4
#define W2B_EUNSUPPORTED -5
#define W2B_ERCCL        -6
#define W2B_ESTATE       -7   /* call order (e.g. train before the corpus was set) */
#define W2B_EIO          -8

#define W2B_EXP_TABLE_SIZE 1000        /* ref src/word2bits.cpp:30 */
#define W2B_MAX_EXP 6                  /* ref src/word2bits.cpp:31 */
#define W2B_MAX_SENTENCE_LENGTH 1000   /* ref src/word2bits.cpp:32 */
#define W2B_UNIGRAM_TABLE_SIZE 100000000LL /* ref src/word2bits.cpp:60 */

typedef struct w2b_trainer w2b_trainer;

/* The reference's training globals (src/word2bits.cpp:45-61) that the hot path reads. */
typedef struct w2b_config {
  int64_t vocab_size;    /* vocab_size                         ref :49 */
  int64_t train_words;   /* train_words                        ref :51 */
  int64_t iter;          /* -iter                              ref :51,608 */
  int32_t layer1_size;   /* -size                              ref :50,598 */
  int32_t window;        /* -window                            ref :48,605 */
  int32_t negative;      /* -negative                          ref :59,607 */
  int32_t bitlevel;      /* -bitlevel                          ref :48,597 */
  int32_t num_threads;   /* -threads: number of Hogwild workers; one workgroup each (ref :48,608) */
  float alpha;           /* starting alpha                     ref :53,603 */
  float sample;          /* -sample                            ref :53,606 */
  float reg;             /* -reg                               ref :54,599 */
  int32_t compute_loss;  /* keep the "Cost"/"Epoch Loss" bookkeeping (ref :437-445,480-483) */
  int32_t device;        /* HIP device ordinal */
  /* multi-GPU (one replica per GPU): this trainer runs the workers with global ids
   * [worker_offset, worker_offset + num_threads) out of total_threads (0 = num_threads).  Seeds
   * (ref :368) and the per-worker quota train_words/num_threads (ref :414) use the GLOBAL numbers;
   * the alpha schedule (ref :391) extrapolates local progress by total_threads/num_threads. */
  int32_t worker_offset;
  int32_t total_threads;
  /* Hogwild memory semantics of the fp32 master rows.  0 (default): every row access is agent-scope
   * (sc1), i.e. coherent between the eight XCD L2s of an MI355X -- workers see each other's updates as
   * threads do on a cache-coherent CPU (ref :490,501 write to shared memory).  1: plain cached
   * accesses -- faster when ids are heavily skewed, but a hot row is then private to an XCD's L2 (or
   * a CU's L1) until it is evicted or the launch ends; see DESIGN.md section 4. */
  int32_t relaxed_coherence;
  /* form (i) has three kernels.  Plain: one workgroup per worker, one thread per column, every context row of every position
   * read from / written to memory.  Row groups (round 5): a worker's rows spread over four groups of wavefronts, all targets of a
   * centre word in flight at once, a producer wavefront one word ahead, an adder wavefront for the lossless context-row adds
   * (rows of at most 1024 floats, window <= 16, negative <= 26; bit-identical to the plain kernel with one worker).
   * Sentence-resident: the fp32 rows of the sliding context window stay in LDS while a worker walks a sentence.
   * 0 = automatic: the row-group kernel for rows of at most 512 floats below a full device where the fidelity budget is not
   *     already thin (DESIGN.md sections 0a, 3.3c), else the plain kernel; never the sentence-resident one (round 4: it keeps
   *     context rows private for up to 2 x window + 1 positions, up to 13 % off the reference's epoch loss on a held-out regime);
   * 1 = plain, 2 = sentence-resident whenever it fits, 3 = row groups whenever they fit -- for coherent rows: with
   * relaxed_coherence or exact_reduction set the plain kernel runs whatever this field says (w2b_worker_kernel_info tells
   * which one a trainer uses: *resident = 0 plain, 1 sentence-resident, 2 row groups). */
  int32_t plain_worker_kernel;
  /* 1: parity mode.  The dot product of ref :461-467 is accumulated serially in the reference's own order
   * (c = 0 .. layer1_size-1, product rounded, then added) instead of the wavefront reduction tree -- the one
   * place where the fast path re-associates.  With it a single-worker run (num_threads == 1) reproduces the
   * CPU program built with -ffp-contract=off bit for bit: same u, v, same output file.  Implemented by the plain
   * worker kernel and the tuple kernel with coherent rows; several times slower, not meant for throughput. */
  int32_t exact_reduction;
  int32_t reserved[2];   /* must be zero */
} w2b_config;

/* ---- library ------------------------------------------------------------------------- */
const char *w2b_version(void);
const char *w2b_last_error(void);       /* thread-local text of the last failure */
int w2b_device_count(void);             /* number of visible HIP devices (0 when none) */
int w2b_device_compute_units(int32_t device);   /* compute units of that device (what w2b_plan_rows takes as num_cus); <= 0: no such device */

/* ---- host-side tables of the reference (pure host code, usable without a GPU) ---------- */
/* expTable, ref src/word2bits.cpp:614-618; out[1000] */
void w2b_build_exp_table(float *out);
/* InitUnigramTable, ref src/word2bits.cpp:112-128 */
int w2b_build_unigram_table(const int64_t *cn, int64_t vocab_size, int32_t *table, int64_t table_size);
/* sub-sampling keep threshold per word, ref src/word2bits.cpp:403-404; out[vocab_size] */
void w2b_build_keep_prob(const int64_t *cn, int64_t vocab_size, float sample, int64_t train_words,
                         float *out);
/* scalar quantizer (host twin of the device function), ref src/word2bits.cpp:73-108 */
float w2b_quantize(float x, int32_t bitlevel);

/* ---- trainer lifetime ------------------------------------------------------------------ */
int w2b_trainer_create(const w2b_config *cfg, w2b_trainer **out);
void w2b_trainer_destroy(w2b_trainer *t);

/* ---- tuning knobs --------------------------------------------------------------------------
 * Everything that changes which code path runs or trades fidelity for speed is in this struct (round 2 read these from
 * environment variables, which a host binding could neither set nor see).  A new trainer holds the defaults;
 * w2b_get_tuning returns the current values, w2b_set_tuning replaces them (before the first launch, or between
 * launches).  struct_size = sizeof(w2b_tuning) versions the struct; reserved fields must be zero.
 *
 * Hot rows: with coherent rows the few most frequent rows of u (context words) and v (targets) queue at their memory
 * lines.  Rows 1..hot_rows_u / 1..hot_rows_v (the vocabulary is sorted by count) therefore get one copy per XCD, shared
 * by all workers of the XCD through its L2, and every hot_period centre words a worker brings a few copies and their
 * master rows together (a running average over the eight XCDs).  -1 = automatic: NONE unless the launch fills
 * the device (>= 3 workgroups per CU: below that copies cost fidelity and buy nothing, round 4), then as many rows as reach a load
 * threshold computed from the word counts of w2b_set_vocab_counts and the number of workers (0 on flat distributions), at
 * most hot_cap.  0 = every access goes to the master rows.  A single worker is bit-identical with and without copies. */
typedef struct w2b_tuning {
  int32_t struct_size;     /* sizeof(w2b_tuning) */
  int32_t hot_rows_v;      /* -1 automatic (default), else 0..128 leading rows of v */
  int32_t hot_rows_u;      /* same for u (plain worker kernel and tuple kernel; the sentence-resident kernel keeps context rows in LDS) */
  int32_t hot_period;      /* centre words between two merge events of a worker; a power of two; 0 (default) = automatic: 16 (32 until round 4) */
  int32_t hot_cap;         /* most rows the automatic choice takes; default 128 */
  int32_t force_row_desc;  /* 1: address rows through per-row buffer descriptors (the form tables >= 2 GiB use) on any table */
  int32_t grid_per_cu;     /* tuple form: workgroups per CU (0 = occupancy query) */
  int32_t mem_mode;        /* -1 = from w2b_config.relaxed_coherence (default); 0 / 1 override it */
  /* Rows 1..atomic_rank (by count; beyond the hot rows) are updated with fp32 atomic adds at their own address instead of
   * load / modify / store: nothing another worker adds during the ~10 us a chunk of rows is in flight is lost, at the
   * price of more memory time per update.  -1 = automatic: every row when the vocabulary is so small and flat that even
   * its least frequent row is hit by several workers at once (computed from the word counts and the number of
   * workers) and a table is at most 8 MB, none otherwise; atomic_cap > 0 limits the number of rows. */
  int32_t atomic_rank;
  int32_t atomic_cap;
  int32_t hot_weight_permille;   /* weight of one XCD's copy of a hot row when it meets the master row; default 125 (1/8) */
  /* sentence-resident kernel: the most frequent context words (the rows that would be hot rows of u) are merged with
   * memory at the latest after this many steps in a worker's window, and stay resident; 0 = only when they leave */
  int32_t window_refresh;
  /* round 4 (carved out of the reserved fields: struct_size is unchanged, and 0 keeps meaning "the library decides") */
  int32_t atomic_rank_u;   /* rows of u (context words) updated with atomic adds: 0 = as atomic_rank (default), > 0 = rows 1..N,
                            * -1 = none.  The reference's own update of a context row is `u[c] += e[c]` on the CURRENT memory
                            * value (ref :500-502): an add of a gradient computed a whole centre word earlier, never a lost one. */
  int32_t fresh_rank_u;    /* plain worker / tuple kernels: context rows 1..N are read again right before their update (phase C)
                            * instead of taken from the LDS stash of phase A, so that the update lands on the row's CURRENT value
                            * as the reference's `u[c] += e[c]` does (ref :500-502); 0 = the library decides, -1 = none */
  int32_t exchange_sat_updates;  /* replica exchange, mode 2 with exchange_rule 1 (rounds 4-5): a row counts as SATURATED (moves by the mean of its
                                  * contributors' deltas instead of their sum) when every replica has updated it at least this
                                  * often since the last exchange; 0 = the library's default */
  /* round 5 (the last reserved field; struct_size unchanged): row-group worker kernel -- context rows 1..N are READ at a copy
   * per XCD that one refresher workgroup per XCD keeps re-filling from the master rows for as long as a launch runs, while
   * their updates stay lossless adds at the master address (a read of a line that is being added to at the memory side is
   * what bounds the shared-row mode: DESIGN.md section 3.3c).  0 = the library decides from the word counts and the number
   * of workers (none for a few workers), -1 = none, > 0 = rows 1..N (at most 64, and never beyond the rows with lossless adds). */
  int32_t refresh_rows_u;
  /* round 6 (the struct grew by 16 bytes: struct_size tells the versions apart).  Replica exchange, mode 2: how the deltas of the
   * c replicas that changed a row are combined.
   *   exchange_rule 2 = exponential saturation, a per-row factor on the SUM: k = (1 - exp(-c n / tau)) / (c (1 - exp(-n / tau))),
   *     n = expected updates of the row per replica since the last exchange (from the word counts): the sum for rarely updated
   *     rows, the mean for rows every replica has saturated, everything in between; exchange_tau_u / exchange_tau_v = tau for rows
   *     of u / v in updates (0 = the library's default, 64);
   *   exchange_rule 0 (default) = the same, and at -bitlevel 1 per ELEMENT the whole sum wherever it keeps the sign the safe step
   *     gives (the quantization cell of one bit): the forward values of the safe rule, the fp32 masters' inertia of one shared
   *     model.  Every other bitlevel: rule 2 (with more bits a master's magnitude is part of the forward value; measured);
   *   exchange_rule 1 = the hard threshold of rounds 4-5 (mean for rows with n >= exchange_sat_updates, sum otherwise).
   * The expected update counts come from w2b_set_vocab_counts; without word counts every row keeps the plain sum.
 * What was measured, including a rule that looked optimal and diverged: DESIGN.md section 3.5. */
  int32_t exchange_rule;
  int32_t exchange_tau_u;
  int32_t exchange_tau_v;
  /* plain worker kernel: how many workers run AT ONCE; a launch of more workers runs them in slices of this many, one after the
   * other (every worker still advances by max_positions per w2b_train_step).  0 = the library decides: all of them, except on
   * vocabularies so small and flat that every row collides, where 3/8 of them run at a time (at least 16): there concurrency x the
   * time a row is open is what moves the epoch loss, and a GPU workgroup has a row open several times longer than a CPU thread. */
  int32_t concurrent_workers;
} w2b_tuning;
/* What the library decides for a launch of the automatic worker kernel with `workers` concurrent workers on a GPU with `num_cus`
 * compute units, from the word counts alone (pure host arithmetic: usable -- and tested -- without a GPU): per-XCD copies
 * of the leading rows (none below a full device), rows updated by lossless adds, merge period.  tune = NULL: the defaults. */
typedef struct w2b_row_plan {
  int32_t copies_u, copies_v;          /* rows 1..N of u / v with per-XCD copies */
  int32_t atomic_rank_u, atomic_rank_v;/* rows 1..N of u / v updated by atomic adds (rows with copies excepted) */
  int32_t full_device;                 /* 1: >= 3 workgroups per compute unit */
  int32_t merge_period;                /* centre words between two merge events of a worker */
  int32_t row_group_kernel;            /* round 5: 1 = the row-group worker kernel runs (short rows, fidelity budget not thin), 0 = the plain one */
  int32_t refresh_rows_u;              /* round 5: context rows 1..N read at refreshed per-XCD copies (row-group kernel only) */
  int32_t concurrent_workers;          /* round 6: workers of the plain kernel that run at once (= workers, except on small flat vocabularies) */
} w2b_row_plan;
int w2b_plan_rows(const w2b_config *cfg, const w2b_tuning *tune, const int64_t *cn, int32_t num_cus, int32_t workers,
                  w2b_row_plan *out);
int w2b_get_tuning(w2b_trainer *t, w2b_tuning *out);
int w2b_set_tuning(w2b_trainer *t, const w2b_tuning *in);

/* ---- model: u ("syn0"), v ("syn1neg"), [vocab_size][layer1_size] fp32 row-major ---------- */
int w2b_init_net(w2b_trainer *t);                                   /* InitNet, ref :343-361 */
int w2b_set_model(w2b_trainer *t, const float *u, const float *v);  /* host -> device */
int w2b_get_model(w2b_trainer *t, float *u, float *v);              /* device -> host */
/* device addresses of the two tables (contiguous: v directly follows u) */
int w2b_model_device_ptrs(w2b_trainer *t, void **u_dev, void **v_dev);
/* quantize(u+v) of the save loop (ref :549-550,568-569) into a host buffer [V][D] */
int w2b_export_quantized(w2b_trainer *t, float *out);
/* the same values bit-packed on the device ("optional bit-packed output"; -bitlevel 1 and 2 only, else
 * W2B_EUNSUPPORTED): out[vocab_size * w2b_packed_words_per_row(layer1_size, bitlevel)] 64-bit words in the layout of
 * include/word2bits_corpus.h -- 1/32 (1/16) of the bytes of w2b_export_quantized leave the device; w2b_unpack_quantized
 * gives back exactly the floats w2b_export_quantized would have delivered. */
int w2b_export_packed(w2b_trainer *t, uint64_t *out);

/* ---- sampler state ------------------------------------------------------------------------ */
/* word counts vocab[].cn: builds the keep-probability table and, when table_size > 0, the
 * unigram table (InitUnigramTable) of that size (the reference uses 1e8). */
int w2b_set_vocab_counts(w2b_trainer *t, const int64_t *cn, int64_t table_size);
int w2b_set_unigram_table(w2b_trainer *t, const int32_t *table, int64_t table_size);
int w2b_set_exp_table(w2b_trainer *t, const float *exp_table);      /* default: w2b_build_exp_table */

/* ---- form (i): the worker, TrainModelThread(id), ref :363-516 --------------------------------
 * The corpus is the in-vocabulary token-id stream of the training file (0 = "</s>"; words not in
 * the vocabulary are already dropped, ref :398).  Worker w starts at ids[starts[w]]; when the
 * reference's fseek (ref :377) lands inside a word, first_override[w] is the id that the truncated
 * word maps to (-1: not in vocabulary, -2: seek landed on a token boundary). */
int w2b_set_corpus(w2b_trainer *t, const int32_t *ids, int64_t n_tokens);
int w2b_set_corpus_device(w2b_trainer *t, const void *ids_dev, int64_t n_tokens); /* not copied */
/* Replicas: only the part of the stream a replica's workers read has to be resident -- from its first worker's start to
 * where its last worker stops (quota train_words/total_threads, ref :414, plus the sentence it is in).  ids[0..n) is
 * that slice, shard starts are relative to it.  more_follows != 0: the file continues behind the slice; a worker that
 * reaches the end of the slice anyway (slice cut too short) makes w2b_epoch_poll / w2b_epoch_status fail with
 * W2B_ESTATE instead of silently ending its shard as if the file had ended. */
int w2b_set_corpus_slice(w2b_trainer *t, const int32_t *ids, int64_t n_tokens, int32_t more_follows);
int w2b_set_shards(w2b_trainer *t, const int64_t *starts, const int32_t *first_override /*or NULL*/);
/* pthread_create of one epoch (ref :535): re-seed every worker (next_random = id, ref :368),
 * rewind it to its shard start.  alpha / word_count_actual keep running across epochs. */
int w2b_epoch_begin(w2b_trainer *t);
/* Advance every unfinished worker by at most max_positions passes of the reference's main loop
 * (one pass = one sentence position, ref :424-509).  Asynchronous on the trainer's stream. */
int w2b_train_step(w2b_trainer *t, int64_t max_positions);
/* blocks; *finished = 1 when all workers have ended their epoch (pthread_join, ref :536) */
int w2b_epoch_status(w2b_trainer *t, int32_t *finished, int64_t *word_count_actual, float *alpha,
                     double *loss_sum);

/* Non-blocking form of w2b_epoch_status: the state after the launch `lag` launches before the latest one (lag 0..3;
 * waits only for THAT launch, so with lag = 1 the next launch is already running while the host looks at the previous
 * one -- an epoch is then noticed to be finished one launch late, and that extra launch returns at once).  loss_sum is
 * the epoch's loss accumulated on the device (atomic adds: same terms as w2b_epoch_status, summed in a different order). */
int w2b_epoch_poll(w2b_trainer *t, int32_t lag, int32_t *finished, int64_t *word_count_actual, float *alpha,
                   double *loss_sum);

/* Number of Hogwild workers (-threads) that exactly fills this GPU for the configured shape: the workgroups
 * of the worker kernel that are resident at once (more workers run in rounds; fewer leave CUs idle) -- but, when
 * cfg.train_words is known, never more than train_words / 50000 (20000 until round 4): a worker re-computes alpha only after >10000 of
 * its own words (ref :379-393), so shorter shards would freeze the learning rate for the whole epoch; and, when that is
 * fewer than 3 workgroups per CU (not a full device) with the plain kernel, never more than 256 -- the scale of the
 * reference's own runs: beyond it the shared-row mode gains little speed and drifts from the reference. */
int w2b_suggested_threads(w2b_trainer *t, int32_t *out);

/* Which form-(i) kernel w2b_train_step() runs for this trainer: *resident = 1 for the sentence-resident kernel
 * (*radius = sentence positions on either side of the centre word whose rows stay in LDS, *column_bytes = bytes
 * of a row owned by one lane), 0 for the plain kernel; *hot_rows = leading rows of v with per-XCD copies (w2b_tuning;
 * chosen from the word counts of w2b_set_vocab_counts and num_threads: 0 on flat distributions).  Any pointer may be NULL. */
int w2b_worker_kernel_info(w2b_trainer *t, int32_t *resident, int32_t *radius, int32_t *column_bytes,
                           int32_t *workgroups_per_cu, int32_t *hot_rows);

/* ---- form (ii): explicit tuples (benchmark / single-step parity form) ------------------------
 * n centre words; ctx_off[n+1] CSR into ctx[] (context rows of u, ref :431-447);
 * neg[n*negative] rows of v drawn for d = 1..negative, -1 = skipped draw (ref :458).
 * The centre word itself is target d = 0 with label 1 (ref :451-453).
 * serial != 0: one workgroup applies the tuples strictly in order (the reference's -threads 1
 * semantics); serial == 0: Hogwild over `grid` workgroups (0 = fill the device). */
int w2b_train_tuples(w2b_trainer *t, int64_t n, const int32_t *center, const int32_t *ctx_off,
                     const int32_t *ctx, const int32_t *neg, float alpha, int32_t serial,
                     double *loss_out /* or NULL */);
/* same, operands already resident in device memory; asynchronous on the trainer's stream */
int w2b_train_tuples_device(w2b_trainer *t, int64_t n, const void *center_dev, const void *ctx_off_dev,
                            const void *ctx_dev, const void *neg_dev, float alpha, int32_t grid);

/* ---- streams / timing ------------------------------------------------------------------------- */
int w2b_synchronize(w2b_trainer *t);
/* hipEvent timing of the training kernels launched since the last reset (sum over launches). */
int w2b_timing_enable(w2b_trainer *t, int32_t on);
int w2b_timing_read(w2b_trainer *t, double *kernel_ms, int64_t *launches); /* syncs, then resets */
/* the same launches one by one: ms_out[0..min(capacity, *launches)) (syncs; does not reset) -- min / median / max per launch */
int w2b_timing_launches(w2b_trainer *t, double *ms_out, int64_t capacity, int64_t *launches);

/* ---- multi-GPU: one process per GPU, replicas + periodic all-reduce over RCCL ----------------
 * Replaces the shared-memory Hogwild of ref :535-536 across devices.  Every replica keeps `base`, the
 * state all replicas agreed on at the last exchange.  An exchange sums what every replica has added since:
 *      d_r = W_r - base;   S = sum_r d_r;   W_r += a * S - d_r;   base += a * S     (a = 1: delta-sum, a = 1/R: average)
 * chunk by chunk (256 MB) on two exchange streams of its own, so that the elementwise kernels of one chunk overlap with
 * the collective of the other AND with the training launches issued meanwhile: w2b_sync_replicas returns at once, the
 * other replicas' contribution lands on top of whatever this replica has trained in between, and only a reader of the
 * model (w2b_get_model, w2b_export_quantized, w2b_epoch_status, w2b_synchronize, ...) waits for the exchange. */
#define W2B_UNIQUE_ID_BYTES 128
int w2b_comm_unique_id(void *out128);                      /* rank 0 creates, others receive */
/* Call while all replicas hold the same model (after w2b_init_net / w2b_set_model).  nranks == 1 with id128 == NULL
 * creates no communicator (w2b_sync_replicas is then a no-op); with an id a communicator of size 1 is created and the
 * whole exchange path runs (a way to exercise it on a one-GPU machine; the model stays bit-identical). */
int w2b_comm_init(w2b_trainer *t, int32_t nranks, int32_t rank, const void *id128);
/* ranks of the RCCL communicator this trainer exchanges over, asked of RCCL itself (ncclCommCount): 0 = no communicator.
 * What bench.py prints as `rccl_ranks`, so that a line claiming N GPUs can be checked against the collective that ran. */
int w2b_comm_count(w2b_trainer *t, int32_t *nranks_out);
/* mode 0: delta-sum (a = 1);  mode 1: average of the deltas (a = 1/R);  mode 2 (what ./word2bits -gpus N uses; round 6): a
 * per-row factor on the summed delta from the row's expected number of updates per replica since the last exchange -- the sum
 * for rarely updated rows, towards the mean of the c replicas that changed it for rows every replica has saturated -- decides
 * every element's QUANTIZED value, and at -bitlevel 1 the whole sum is taken wherever it keeps that sign (w2b_tuning.
 * exchange_rule; a row that only one replica saw keeps that replica's whole update, which mode 1 divides by R; mode 0 diverges
 * from 4 replicas on).  Measured in tests/test_gpu_exchange.py: 2 / 4 / 8 replicas on one GPU through the phase API below.
 * Asynchronous (see above). */
int w2b_sync_replicas(w2b_trainer *t, int32_t mode);
/* Centre words per replica between two exchanges that ./word2bits -gpus N uses when nothing else is said: 1 / 32 of a replica's
 * words per epoch, between 32 768 and 1 048 576 (where the staleness of a replica and the rule's own bias were measured to cancel on
 * four regimes, and where one exchange of the whole model per launch fits the xGMI links).  Pure arithmetic. */
int64_t w2b_suggested_exchange_words(int64_t train_words_per_epoch, int32_t replicas);
/* (Round 4 also had a HOT TIER -- w2b_sync_hot_rows / w2b_exchange_begin_hot / w2b_exchange_hot_rows: the leading rows of both
 * tables exchanged after every launch.  It measured no gain over the full exchanges alone -- the rows that are rare individually
 * are a quarter of all draws and want the short interval as much as the frequent ones -- and was removed in round 5.) */
/* exchanges since the last call and their summed device time (begin of the first chunk -> end of the last; waits for
 * the exchanges in flight); resets both */
int w2b_sync_stats(w2b_trainer *t, int64_t *exchanges, double *device_ms);

/* The same exchange for hosts that bring their own collective (MPI, torch.distributed over gloo or RCCL, ...):
 *   w2b_exchange_init once, while all replicas hold the same model; then per exchange
 *   w2b_exchange_begin(&n_chunks, &my_words)
 *   (mode 2 only) w2b_exchange_counts(&cnt, &m): cnt[0..m) = 1 for every row of [u||v] this replica changed;
 *                            the host sums cnt over the replicas in place; the first w2b_exchange_apply turns the sums into the
 *                            per-row factors of w2b_tuning.exchange_rule and applies them (scale = 1)
 *   for c in [0, n_chunks): w2b_exchange_delta(c, &buf, &n)   -- buf[0..n) = this replica's delta (device memory, complete
 *                            on return);  the host sums buf over all replicas IN PLACE with its collective;
 *                           w2b_exchange_apply(c, a)          -- expects the sum to be complete
 *   w2b_exchange_end(sum of all replicas' my_words, or -1)     -- the alpha schedule (ref :391) runs on the global count
 * Chunks c and c + 1 use different staging buffers and streams, so a host may pipeline them. */
int w2b_exchange_init(w2b_trainer *t);
int w2b_exchange_begin(w2b_trainer *t, int64_t *n_chunks, int64_t *local_word_count /* or NULL */);
int w2b_exchange_counts(w2b_trainer *t, void **buf_dev, int64_t *elems);
int w2b_exchange_delta(w2b_trainer *t, int64_t chunk, void **buf_dev, int64_t *elems);
int w2b_exchange_apply(w2b_trainer *t, int64_t chunk, float scale);
int w2b_exchange_end(w2b_trainer *t, int64_t word_count_all_replicas);

#ifdef __cplusplus
}
#endif
#endif
