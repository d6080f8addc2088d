/*
 * oracle/w2b_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-source restatement of the reference CPU algorithm of
 * agnusmaximus/Word2Bits for the training hot path, written from the prose
 * specification in SURVEY.md Appendix A and checked bit-for-bit against the
 * unmodified reference program (oracle/_ref/word2bits_nofma, see
 * tests/test_oracle_vs_ref.py).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product
 * (word2bits_amd/) never does.
 *
 * Every function cites the reference lines it follows
 * (paths relative to /root/reference).
 */
#ifndef W2B_ORACLE_H
#define W2B_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define W2BO_EXP_TABLE_SIZE 1000      /* src/word2bits.cpp:30 */
#define W2BO_MAX_EXP 6                /* src/word2bits.cpp:31 */
#define W2BO_MAX_SENTENCE 1000        /* src/word2bits.cpp:32 */
#define W2BO_TABLE_SIZE 100000000LL   /* src/word2bits.cpp:60 */

/* ---- scalar pieces ------------------------------------------------------------------- */
float w2bo_quantize(float x, int bitlevel);                 /* src/word2bits.cpp:73-108 */
void  w2bo_quantize_array(const float *x, float *out, long long n, int bitlevel);
float w2bo_sigmoid(float x);                                 /* src/word2bits.cpp:67-71 */
void  w2bo_build_exp_table(float *tab /*[1000]*/);           /* src/word2bits.cpp:614-618 */
uint64_t w2bo_lcg_next(uint64_t s);                          /* e.g. src/word2bits.cpp:405 */
/* table of the sigmoid-bin index used at src/word2bits.cpp:475 */
int   w2bo_exp_index(float f);

/* ---- model / sampler state ----------------------------------------------------------- */
void w2bo_init_net(long long vocab_size, long long dim, float *u, float *v);      /* :343-361 */
void w2bo_build_unigram_table(const long long *cn, long long vocab_size,
                              int *table, long long table_size);                   /* :112-128 */
/* keep-probability of the frequent-word subsampler, src/word2bits.cpp:403-404 */
float w2bo_keep_prob(long long cn, float sample, long long train_words);

typedef struct w2bo_model {
  long long vocab_size, dim, train_words, iter;
  int window, negative, bitlevel, num_threads;
  float starting_alpha, sample, reg;
  const long long *cn;      /* [vocab_size] word counts (vocab[].cn) */
  float *u, *v;             /* [vocab_size*dim] fp32 masters, row-major */
  const float *exp_table;   /* [1000] */
  const int *table;         /* unigram table */
  long long table_size;
  /* shared, racy (Hogwild) state -- src/word2bits.cpp:51,53 */
  volatile float alpha;
  volatile long long word_count_actual;
  int compute_loss;         /* 0: skip the logf/expf bookkeeping (timing runs) */
} w2bo_model;

/* One centre word: phases A (gather+average), B (targets), C (scatter).
 * src/word2bits.cpp:426-503.  ctx/targets are row ids; labels[i] in {0,1}.
 * scratch must hold 2*dim floats.  Returns the loss contribution (as the
 * reference accumulates it, in double). */
double w2bo_center_update(w2bo_model *m, const int *ctx, int cw,
                          const int *targets, const int *labels, int nt,
                          float alpha, float *scratch);

/* A batch of explicit tuples, applied strictly in order (serial semantics).
 * ctx_off is CSR [n+1]; neg is [n*negative] with -1 meaning "skipped draw"
 * (target == word, src/word2bits.cpp:458).  Returns summed loss. */
double w2bo_train_tuples(w2bo_model *m, long long n, const int *center,
                         const int *ctx_off, const int *ctx, const int *neg, float alpha);

/* ---- the worker: TrainModelThread(id), src/word2bits.cpp:363-516 --------------------- */
/* Token stream form: ids[] are vocabulary indices (0 == "</s>", -1 == word not in
 * vocabulary), the worker starts at ids[start] and sees end-of-file at n.
 * first_override >= -1 replaces the token at ids[start-1]... see w2bo_file_* below;
 * pass -2 for "none".  One call == one epoch of that worker (local_iter == 1).
 * Returns total_loss of the worker (thread_losses[id]). */
double w2bo_train_worker_tokens(w2bo_model *m, long long id, const int *ids, long long n,
                                long long start, int first_override);

/* run `nthreads` workers concurrently with pthreads (Hogwild), worker w starting at
 * starts[w] with overrides[w]; returns the epoch loss (sum of worker losses). */
double w2bo_train_epoch_tokens(w2bo_model *m, const int *ids, long long n,
                               const long long *starts, const int *overrides, int nthreads);

/* ---- file level: vocabulary + tokenisation (src/word2bits.cpp:131-301) --------------- */
typedef struct w2bo_vocab w2bo_vocab;
w2bo_vocab *w2bo_vocab_learn(const char *train_file, int min_count);   /* :265-301 */
/* the same with the reference's vocab_hash_size (:35) as a parameter: ReduceVocab (:245-263) runs whenever the
 * vocabulary outgrows 70 % of it (:293) */
w2bo_vocab *w2bo_vocab_learn_ex(const char *train_file, int min_count, int vocab_hash_size);
void w2bo_vocab_free(w2bo_vocab *);
long long w2bo_vocab_size(const w2bo_vocab *);
long long w2bo_vocab_train_words(const w2bo_vocab *);
long long w2bo_vocab_file_size(const w2bo_vocab *);
const char *w2bo_vocab_word(const w2bo_vocab *, long long i);
long long w2bo_vocab_count(const w2bo_vocab *, long long i);
int w2bo_vocab_search(const w2bo_vocab *, const char *word);           /* :166-174 */

/* Tokenise the whole file as ReadWordIndex would from offset 0 (:131-155,177-185):
 * returns number of tokens; ids_out/begin_out (malloc'ed, caller frees with free()).
 * begin_out[i] is the byte offset of the first character of token i
 * (for "</s>" the offset of the '\n'). */
long long w2bo_tokenize_file(const w2bo_vocab *, const char *train_file,
                             int **ids_out, long long **begin_out);
/* For a worker that fseek()s to byte `offset` (:377): index of the first token it will
 * read and, when the seek lands inside a word, the id that the truncated word maps to
 * (override, >= -1); *override = -2 when the seek lands on a token boundary. */
long long w2bo_shard_start(const w2bo_vocab *, const char *train_file, long long offset,
                           const long long *begin, long long n_tokens, int *override);

/* Whole-program restatement: learn vocab, init, train `iter` epochs with `num_threads`
 * workers, write the output file exactly like src/word2bits.cpp:560-576.  Returns 0. */
int w2bo_run(const char *train_file, const char *output_file, int bitlevel, int dim, int window,
             int negative, int num_threads, int iter, int min_count, float alpha, float sample,
             float reg, int binary, double *epoch_losses /*[iter] or NULL*/);
/* the same program with the reference's vocab_hash_size (:35) as a parameter (ReduceVocab, :245-263,293) */
int w2bo_run_ex(const char *train_file, const char *output_file, int bitlevel, int dim, int window,
                int negative, int num_threads, int iter, int min_count, float alpha, float sample,
                float reg, int binary, double *epoch_losses /*[iter] or NULL*/, int vocab_hash_size);

/* ------------------------------------------------------------------ evaluator (src/compute-accuracy.c)
 * Numerics of the analogy evaluator, restated; file parsing and the stdout transcript are
 * restated in oracle/eval_oracle.py on top of these two.  `fma` != 0 evaluates every
 * `acc += a * b` as one fused multiply-add (what the reference's Makefile:6 build does on an
 * FMA host), fma == 0 as two roundings (the -ffp-contract=off build).  Both are pinned to the
 * corresponding build of the unmodified evaluator (tests/golden/eval_*). */
/* :106-110 : quantize again with the evaluator's bitlevel, then divide each row by its length */
void w2bo_eval_normalize(float *M, long long words, long long size, int bitlevel, int fma);
/* :146-177 with N = 1: best[q] = first c (not b1,b2,b3) with the largest dist > 0, or -1 */
void w2bo_eval_top1(const float *M, long long words, long long size, long long nq, const int *b1,
                    const int *b2, const int *b3, int fma, int *best, float *bestd);

#ifdef __cplusplus
}
#endif
#endif
