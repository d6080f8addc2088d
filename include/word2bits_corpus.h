/*
 * word2bits_corpus.h -- C ABI of the host-side corpus ingest that feeds the GPU hot path.
 *
 * Restates, for an in-memory token stream, what the reference does with stdio on every epoch:
 * LearnVocabFromTrainFile / ReadWord / SearchVocab / SortVocab (ref src/word2bits.cpp:131-301),
 * the per-thread fseek shard arithmetic (ref :377) and the output writer (ref :560-576).
 * Pure host code: usable (and tested) without a GPU.
 */
#ifndef WORD2BITS_CORPUS_H
#define WORD2BITS_CORPUS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct w2b_corpus w2b_corpus;

/* Learn the vocabulary of `train_file` with -min-count `min_count` (ref :265-301, 215-242) and
 * tokenise the file once into vocabulary ids (0 = "</s>", out-of-vocabulary words dropped as the
 * reference's reader does at ref :398).  Returns 0, or W2B_EIO if the file cannot be read. */
int w2b_corpus_load(const char *train_file, int32_t min_count, w2b_corpus **out);
/* The same with the reference's `vocab_hash_size` (ref :35) as an argument; 0 = its value, 30 000 000.  The constant
 * matters in one place: while the reference learns the vocabulary, ReduceVocab (ref :245-263) drops every word whose
 * count so far is not above `min_reduce` (1, 2, 3, ... from call to call) whenever the table holds more than
 * 0.7 x vocab_hash_size words (ref :293) -- from 21 M distinct words on.  Reproduced exactly, including what depends
 * on the order of the stream (a dropped word that returns starts again at 1 and sorts behind its peers) and the
 * reference's quirk that "</s>" itself can be dropped, after which another word owns row 0 and ends sentences. */
int w2b_corpus_load_ex(const char *train_file, int32_t min_count, int32_t vocab_hash_size, w2b_corpus **out);
void w2b_corpus_free(w2b_corpus *c);

int64_t w2b_corpus_vocab_size(const w2b_corpus *c);    /* vocab_size,  ref :49 */
int64_t w2b_corpus_train_words(const w2b_corpus *c);   /* train_words, ref :233 */
int64_t w2b_corpus_file_size(const w2b_corpus *c);     /* file_size,   ref :299 */
const char *w2b_corpus_word(const w2b_corpus *c, int64_t i);
const int64_t *w2b_corpus_counts(const w2b_corpus *c); /* vocab[].cn, [vocab_size] */
int32_t w2b_corpus_search(const w2b_corpus *c, const char *word); /* SearchVocab, -1 if absent */
int64_t w2b_corpus_num_tokens(const w2b_corpus *c);
const int32_t *w2b_corpus_tokens(const w2b_corpus *c); /* [num_tokens] */

/* Shard starts of `num_threads` workers: worker w would fseek to file_size/num_threads*w (ref :377).
 * starts[w] = index (into the token stream) of the first whole token at or after that byte;
 * first_override[w] = id of the truncated word when the seek lands inside a word (-1 = not in the
 * vocabulary), -2 when it lands on a token boundary. */
int w2b_corpus_shards(const w2b_corpus *c, int32_t num_threads, int64_t *starts, int32_t *first_override);

/* Output writer of ref :560-576: "V D\n", then per row "word " + D values + "\n";
 * binary != 0 -> raw little-endian float32, else "%lf ".  `values` is [vocab_size][dim],
 * already quantize(u+v). */
int w2b_save_vectors(const char *path, const w2b_corpus *c, const float *values, int64_t dim,
                     int32_t binary);

/* ---- bit-packed vectors ("optional bit-packed output"; not a format of the reference) ----------------
 * A 1-bit (2-bit) model is 1 (2) bits of information per value; the reference's file spends 32.  Packed layout of a row
 * of `dim` values quantized at -bitlevel 1 or 2: ceil(dim / 64) blocks of 64 consecutive columns; per block one 64-bit
 * word of SIGN bits (bit c % 64 set: the value of column c is negative) and, at bitlevel 2, a second word of MAGNITUDE
 * bits (set: 0.75, clear: 0.25; bitlevel 1 has the single magnitude 1/3).  Bits of columns >= dim are zero.
 * w2b_packed_words_per_row = ceil(dim / 64) * bitlevel, or -1 when bitlevel is neither 1 nor 2.  Lossless: unpacking
 * returns the exact float bit patterns quantize() (ref :73-108) produces.  Pure host code; the device-side producer is
 * w2b_export_packed (word2bits_hip.h). */
int64_t w2b_packed_words_per_row(int64_t dim, int32_t bitlevel);
/* values[rows][dim] must already be quantize()d at `bitlevel` (anything else is W2B_EINVAL) */
int w2b_pack_quantized(const float *values, int64_t rows, int64_t dim, int32_t bitlevel, uint64_t *out);
int w2b_unpack_quantized(const uint64_t *packed, int64_t rows, int64_t dim, int32_t bitlevel, float *out);
/* Packed model file: "W2BP1 <vocab_size> <dim> <bitlevel>\n", the vocabulary (one word per line, row order), then
 * vocab_size x words_per_row little-endian 64-bit words. */
int w2b_save_vectors_packed(const char *path, const w2b_corpus *c, const uint64_t *packed, int64_t dim, int32_t bitlevel);
/* Packed file -> the reference's output file (ref :560-576; binary != 0: raw float32, else "%lf "), byte for byte what
 * w2b_save_vectors / the reference would have written for the same model, so that the unmodified compute_accuracy (or any
 * other consumer of the reference's format) can read it.  ./compute_accuracy and w2b_eval_load also take a packed file
 * directly. */
int w2b_unpack_vectors_file(const char *packed_path, const char *out_path, int32_t binary);

#ifdef __cplusplus
}
#endif
#endif
