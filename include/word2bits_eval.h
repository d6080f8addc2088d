/*
 * word2bits_eval.h -- C ABI of the MI355X analogy evaluator (the consumer of the trained vectors file).
 *
 * Replaces the reference's evaluator program, src/compute-accuracy.c (`compute_accuracy <FILE> <bitlevel>
 * <threshold> < questions-words.txt`), whose whole cost is the exhaustive scan of ref :158-177: for every
 * question a [1 x D] . [D x V] product over the normalised matrix followed by a strict-greater arg-max.
 * On the GPU that scan is one batched fp32 product (all questions x all rows) with the arg-max fused into its
 * epilogue (word2bits_amd/csrc/w2b_kernels_eval.hip): on the matrix cores (f32 MFMA = sequential fmaf chains,
 * bit for bit) in fused mode, on the vector ALU (packed mul + add) in the two-rounding mode.
 *
 * Parity contract: answers -- and therefore the stdout transcript -- are IDENTICAL to the reference's, ties
 * included.  Every score is accumulated in the reference's order (a = 0 .. size-1, one accumulator per
 * (question,row) pair), candidates tie-break to the lowest row, and `fused` selects the arithmetic of the build
 * being replaced: 1 = `acc += a*b` is one fused multiply-add (what the reference's Makefile:6 flags,
 * -O3 -march=native, produce on any FMA-capable x86), 0 = two roundings (-ffp-contract=off).  The two builds of
 * the unmodified reference disagree with each other on tie-heavy 1-bit vectors; each mode matches its build.
 *
 * No CPU fallback: W2B_ENOGPU when no device is visible.  Error codes and w2b_last_error() are those of
 * word2bits_hip.h.
 */
#ifndef WORD2BITS_EVAL_H
#define WORD2BITS_EVAL_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct w2b_eval w2b_eval;

/* ref src/compute-accuracy.c:80-112: read "<words> <size>", cap words at `threshold` (0 = off, ref :86), read
 * each row (word up to the first ' ', '\n' bytes dropped, at most max_w = 50 characters kept, upper-cased; then
 * `size` raw float32), apply quantize(x, bitlevel) (ref :26-61,106) and divide the row by its length
 * (ref :107-110; float accumulation in column order, a zero row becomes NaN as in the reference).
 * W2B_EIO when the file cannot be opened ("Input file not found", ref :81-84). */
int w2b_eval_load(const char *file, int32_t bitlevel, int64_t threshold, int32_t fused, int32_t device,
                  w2b_eval **out);
void w2b_eval_free(w2b_eval *e);

/* The evaluator on a live trainer (include/word2bits_hip.h), without the file round trip: exactly what
 * w2b_eval_load(file, bitlevel, threshold, ...) would hold after the trainer's vectors had been written to `file` in
 * the binary format (quantize(u+v) with the trainer's bitlevel, ref src/word2bits.cpp:565-574) -- the values stay
 * on the device, `words[i]` is the vocabulary word of row i (w2b_corpus_word), n_words = vocab_size. */
struct w2b_trainer;
int w2b_eval_from_trainer(struct w2b_trainer *t, int64_t n_words, const char *const *words, int32_t bitlevel,
                          int64_t threshold, int32_t fused, w2b_eval **out);

int64_t w2b_eval_words(const w2b_eval *e);                  /* `words` after the threshold, ref :85-86 */
int64_t w2b_eval_size(const w2b_eval *e);                   /* `size`, ref :87 */
const char *w2b_eval_word(const w2b_eval *e, int64_t row);  /* &vocab[row * max_w], ref :99-104 */
/* ref :140-145,152: first row whose upper-cased word equals `upper_word`, or w2b_eval_words() if none */
int64_t w2b_eval_lookup(const w2b_eval *e, const char *upper_word);
/* the normalised matrix M of ref :106-110, [words][size] (parity tests) */
int w2b_eval_get_matrix(w2b_eval *e, float *out);

/* ref :155-177 with N = 1, for `nq` questions at once: vec = (M[b2] - M[b1]) + M[b3]; best[q] = the first row c
 * (c != b1,b2,b3) whose score  sum_a vec[a] * M[c][a]  is the largest one > 0, bestd[q] = that score;
 * best[q] = -1, bestd[q] = 0 when no row scores above 0.  bestd may be NULL. */
int w2b_eval_top1(w2b_eval *e, int64_t nq, const int32_t *b1, const int32_t *b2, const int32_t *b3,
                  int32_t *best, float *bestd);

/* ref :94,113-188: the program's stdout for the question stream `questions[0..len)` (what the reference reads
 * from stdin with scanf("%s")), including "Starting eval...".  *out is malloc'ed; release it with
 * w2b_eval_free_text. */
int w2b_eval_transcript(w2b_eval *e, const char *questions, int64_t len, char **out, int64_t *out_len);
void w2b_eval_free_text(char *text);

/* Which kernel scores the fused (FMA) mode: 1 (default) = the f32 MFMA kernel; 0 = the same fused chain on the vector ALU
 * (v_pk_fma_f32; a cross-check of the MFMA path); n > 1 = MFMA with n question tiles per row tile in the launch order.
 * The two-rounding mode always runs on the vector ALU.  (Round 2 read this from the environment.) */
int w2b_eval_set_kernel(w2b_eval *e, int32_t variant);

/* Device time (HIP events on the evaluator's stream) and launch count of the score kernel since load or since
 * the last call; `macs` = multiply-adds those launches performed (questions x padded rows x padded size). */
int w2b_eval_timing_read(w2b_eval *e, double *kernel_ms, int64_t *launches, double *macs);

#ifdef __cplusplus
}
#endif
#endif
