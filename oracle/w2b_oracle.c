/*
 * oracle/w2b_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference algorithm (agnusmaximus/Word2Bits,
 * src/word2bits.cpp) for the CBOW/negative-sampling update with bit-level
 * quantisation.  Written from SURVEY.md Appendix A; organised differently from
 * the reference (explicit model struct, list-building separated from the
 * arithmetic, memory-buffer reader instead of stdio) but arithmetically
 * identical operation-by-operation so that, compiled WITHOUT fused
 * multiply-add contraction, it reproduces the reference binary's output
 * bit-for-bit at -threads 1 (pinned by tests/test_oracle_golden.py against
 * oracle/_ref/word2bits_nofma, and by the fixtures under tests/golden/).
 *
 * Parity status: PINNED (see above).  Build: oracle/Makefile (-ffp-contract=off).
 *
 * All "ref:" comments cite /root/reference/src/word2bits.cpp.
 */
#define _GNU_SOURCE
#include "w2b_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LCG_A 25214903917ULL
#define LCG_C 11ULL
#define MAX_WORD 4096 /* ref :29 MAX_STRING */

/* ------------------------------------------------------------------ scalar pieces */

uint64_t w2bo_lcg_next(uint64_t s) { return s * LCG_A + LCG_C; } /* ref :352,405,428,455 */

/* ref :73-108.  bitlevel 0 identity; 1 -> +-1/3; 2 -> +-{.25,.75}; >=4 -> grid of
 * 2^(b-1) steps clamped to 1; 3 falls through to +-0. */
float w2bo_quantize(float x, int bitlevel) {
  if (bitlevel == 0) return x;
  float sgn = (x < 0) ? -1.0f : 1.0f; /* +0,-0,NaN -> +1 */
  float mag = x * sgn;
  if (bitlevel == 1) return sgn / 3;
  float level = 0;
  if (bitlevel == 2) {
    /* double literal .5 in the reference: float promoted to double, same ordering */
    level = (mag >= 0 && (double)mag <= .5) ? .25f : .75f;
  }
  if (bitlevel >= 4) {
    int steps = (int)pow(2, bitlevel - 1);
    int k = (int)((mag * steps) + (float).5); /* float mul, float add, truncate */
    if (k > steps) k = steps;
    level = k / (float)steps;
  }
  return sgn * level;
}

void w2bo_quantize_array(const float *x, float *out, long long n, int bitlevel) {
  for (long long i = 0; i < n; i++) out[i] = w2bo_quantize(x[i], bitlevel);
}

/* ref :67-71 (float exp: the argument is a float, C++ picks the float overload) */
float w2bo_sigmoid(float x) {
  if (x > W2BO_MAX_EXP) return 1;
  if (x < -W2BO_MAX_EXP) return 1e-9f;
  return 1 / (1 + expf(-x));
}

/* ref :614-618 */
void w2bo_build_exp_table(float *tab) {
  for (int i = 0; i < W2BO_EXP_TABLE_SIZE; i++) {
    float e = expf((i / (float)W2BO_EXP_TABLE_SIZE * 2 - 1) * W2BO_MAX_EXP);
    tab[i] = e / (e + 1);
  }
}

/* ref :475 -- EXP_TABLE_SIZE / MAX_EXP / 2 is integer arithmetic = 83 */
int w2bo_exp_index(float f) {
  return (int)((f + W2BO_MAX_EXP) * (W2BO_EXP_TABLE_SIZE / W2BO_MAX_EXP / 2));
}

/* ------------------------------------------------------------------ model state */

/* ref :343-361: one LCG (seed 1) fills v first, then u, with (low16/65536 - 0.5). */
void w2bo_init_net(long long V, long long D, float *u, float *v) {
  uint64_t s = 1;
  float *dst[2] = {v, u};
  for (int t = 0; t < 2; t++)
    for (long long i = 0; i < V * D; i++) {
      s = w2bo_lcg_next(s);
      dst[t][i] = (float)(((s & 0xFFFF) / (float)65536) - 0.5);
    }
}

/* ref :112-128 */
void w2bo_build_unigram_table(const long long *cn, long long V, int *table, long long tsz) {
  double total = 0, power = 0.75;
  for (long long a = 0; a < V; a++) total += pow((double)cn[a], power);
  long long i = 0;
  double edge = pow((double)cn[0], power) / total;
  for (long long a = 0; a < tsz; a++) {
    table[a] = (int)i;
    if (a / (double)tsz > edge) {
      i++;
      /* the reference reads vocab[i].cn one past the end when i == V; that value is
       * then discarded by the clamp below and never used again (edge only grows). */
      edge += (i < V) ? pow((double)cn[i], power) / total : 0.0;
    }
    if (i >= V) i = V - 1;
  }
}

/* ref :403-404 */
float w2bo_keep_prob(long long cn, float sample, long long train_words) {
  float st = sample * train_words;
  return (sqrtf(cn / st) + 1) * st / cn;
}

/* ------------------------------------------------------------------ one centre word */

/* ref :426-503, with the context rows and the target rows already chosen. */
double w2bo_center_update(w2bo_model *m, const int *ctx, int cw, const int *targets,
                          const int *labels, int nt, float alpha, float *scratch) {
  const long long D = m->dim;
  const int bl = m->bitlevel;
  const float reg = m->reg;
  float *havg = scratch, *herr = scratch + D;
  double loss = 0;
  for (long long c = 0; c < D; c++) havg[c] = 0;
  for (long long c = 0; c < D; c++) herr[c] = 0;
  if (cw == 0) return 0;
  /* phase A, ref :431-449 */
  for (int j = 0; j < cw; j++) {
    const float *row = m->u + (long long)ctx[j] * D;
    float rl = 0;
    for (long long c = 0; c < D; c++) {
      float q = w2bo_quantize(row[c], bl);
      havg[c] += q;
      rl += q * q;
    }
    rl = reg * rl;
    loss += -rl;
  }
  for (long long c = 0; c < D; c++) havg[c] /= cw;
  /* phase B, ref :450-492 */
  for (int d = 0; d < nt; d++) {
    float *row = m->v + (long long)targets[d] * D;
    long long label = labels[d];
    float f = 0, rl = 0, g;
    for (long long c = 0; c < D; c++) {
      float q = w2bo_quantize(row[c], bl);
      f += havg[c] * q;
      rl += q * q;
    }
    rl = reg * rl;
    if (f > W2BO_MAX_EXP) g = (label - 1) * alpha;
    else if (f < -W2BO_MAX_EXP) g = (label - 0) * alpha;
    else g = (label - m->exp_table[w2bo_exp_index(f)]) * alpha;
    if (m->compute_loss) {
      float dp = (float)(f * pow(-1, 1 - label));
      float ll = logf(w2bo_sigmoid(dp));
      loss += ll - rl;
    }
    for (long long c = 0; c < D; c++) herr[c] += g * w2bo_quantize(row[c], bl);
    for (long long c = 0; c < D; c++) row[c] += g * havg[c] - 2 * alpha * reg * row[c];
  }
  /* phase C, ref :494-503 */
  for (int j = 0; j < cw; j++) {
    float *row = m->u + (long long)ctx[j] * D;
    for (long long c = 0; c < D; c++) row[c] += herr[c] - 2 * alpha * reg * row[c];
  }
  return loss;
}

double w2bo_train_tuples(w2bo_model *m, long long n, const int *center, const int *ctx_off,
                         const int *ctx, const int *neg, float alpha) {
  const int K = m->negative;
  int *tg = (int *)malloc(sizeof(int) * (K + 1) * 2), *lb = tg + K + 1;
  float *scratch = (float *)malloc(sizeof(float) * 2 * m->dim);
  double loss = 0;
  for (long long i = 0; i < n; i++) {
    int nt = 0;
    tg[nt] = center[i]; lb[nt] = 1; nt++;
    for (int d = 0; d < K; d++) {
      int t = neg[i * K + d];
      if (t < 0 || t == center[i]) continue; /* skipped draw, ref :458 */
      tg[nt] = t; lb[nt] = 0; nt++;
    }
    loss += w2bo_center_update(m, ctx + ctx_off[i], ctx_off[i + 1] - ctx_off[i], tg, lb, nt,
                               alpha, scratch);
  }
  free(tg); free(scratch);
  return loss;
}

/* ------------------------------------------------------------------ the worker */

typedef struct {
  const int *ids; long long n, pos; int override; /* -2 none */
} tok_src;

/* ReadWordIndex over the token stream: returns id (>= -1), sets *eof at end. */
static int src_next(tok_src *s, int *eof) {
  if (s->override != -2) { int w = s->override; s->override = -2; return w; }
  if (s->pos >= s->n) { *eof = 1; return -1; }
  return s->ids[s->pos++];
}

double w2bo_train_worker_tokens(w2bo_model *m, long long id, const int *ids, long long n,
                                long long start, int first_override) {
  const int W = m->window, K = m->negative;
  tok_src src = {ids, n, start, first_override};
  uint64_t rng = (uint64_t)id;                          /* ref :368 */
  long long sen[W2BO_MAX_SENTENCE + 1];
  long long slen = 0, spos = 0, wc = 0, last_wc = 0;
  int eof = 0;
  double total_loss = 0;
  int *ctx = (int *)malloc(sizeof(int) * (2 * W + 2 + 2 * (K + 1)));
  int *tg = ctx + 2 * W + 2, *lb = tg + K + 1;
  float *scratch = (float *)malloc(sizeof(float) * 2 * m->dim);
  sen[0] = 0;
  for (;;) {
    if (wc - last_wc > 10000) {                          /* ref :379-393 */
      m->word_count_actual += wc - last_wc;
      last_wc = wc;
      float a = m->starting_alpha *
                (1 - m->word_count_actual / (float)(m->iter * m->train_words + 1));
      if (a < m->starting_alpha * 0.0001) a = m->starting_alpha * 0.0001;
      m->alpha = a;
    }
    if (slen == 0) {                                     /* ref :394-413 */
      for (;;) {
        int w = src_next(&src, &eof);
        if (eof) break;
        if (w == -1) continue;
        wc++;
        if (w == 0) break;
        if (m->sample > 0) {
          float keep = w2bo_keep_prob(m->cn[w], m->sample, m->train_words);
          rng = w2bo_lcg_next(rng);
          if (keep < (rng & 0xFFFF) / (float)65536) continue;
        }
        sen[slen++] = w;
        if (slen >= W2BO_MAX_SENTENCE) break;
      }
      spos = 0;
    }
    if (eof || wc > m->train_words / m->num_threads) {  /* ref :414-423, local_iter == 1 */
      m->word_count_actual += wc - last_wc;
      break;
    }
    long long word = sen[spos];
    rng = w2bo_lcg_next(rng);                            /* ref :428-429 */
    long long b = rng % (uint64_t)W;
    int cw = 0;
    for (long long a = b; a < W * 2 + 1 - b; a++) {      /* ref :431-436 */
      if (a == W) continue;
      long long c = spos - W + a;
      if (c < 0 || c >= slen) continue;
      ctx[cw++] = (int)sen[c];
    }
    if (cw) {
      int nt = 0;
      tg[nt] = (int)word; lb[nt] = 1; nt++;              /* ref :451-453 */
      for (int d = 1; d < K + 1; d++) {                  /* ref :455-459 */
        rng = w2bo_lcg_next(rng);
        long long t = m->table[(rng >> 16) % (uint64_t)m->table_size];
        if (t == 0) t = rng % (uint64_t)(m->vocab_size - 1) + 1;
        if (t == word) continue;
        tg[nt] = (int)t; lb[nt] = 0; nt++;
      }
      total_loss += w2bo_center_update(m, ctx, cw, tg, lb, nt, m->alpha, scratch);
    }
    spos++;                                              /* ref :505-509 */
    if (spos >= slen) slen = 0;
  }
  free(ctx); free(scratch);
  return total_loss;
}

typedef struct {
  w2bo_model *m; long long id; const int *ids; long long n, start; int ov; double loss;
} worker_arg;

static void *worker_main(void *p) {
  worker_arg *a = (worker_arg *)p;
  a->loss = w2bo_train_worker_tokens(a->m, a->id, a->ids, a->n, a->start, a->ov);
  return NULL;
}

/* ref :532-539 (one iteration of the epoch loop) */
double w2bo_train_epoch_tokens(w2bo_model *m, const int *ids, long long n,
                               const long long *starts, const int *overrides, int nthreads) {
  worker_arg *args = (worker_arg *)calloc(nthreads, sizeof(worker_arg));
  pthread_t *pt = (pthread_t *)calloc(nthreads, sizeof(pthread_t));
  for (int w = 0; w < nthreads; w++) {
    args[w] = (worker_arg){m, w, ids, n, starts[w], overrides ? overrides[w] : -2, 0};
    if (nthreads == 1) worker_main(&args[w]);
    else pthread_create(&pt[w], NULL, worker_main, &args[w]);
  }
  double loss = 0;
  for (int w = 0; w < nthreads; w++) {
    if (nthreads > 1) pthread_join(pt[w], NULL);
    loss += args[w].loss;
  }
  free(args); free(pt);
  return loss;
}

/* ------------------------------------------------------------------ file level */

struct w2bo_vocab {
  char **word; long long *cn; long long size, cap, train_words, file_size;
  int *slot; long long nslot; /* open addressing, power of two */
};
typedef struct { char *w; long long cn; } vw_pair;

static uint64_t str_hash(const char *s) {
  uint64_t h = 1469598103934665603ULL;
  for (; *s; s++) { h ^= (unsigned char)*s; h *= 1099511628211ULL; }
  return h;
}
static void map_rebuild(w2bo_vocab *v) {
  long long want = 1024;
  while (want < v->size * 3) want <<= 1;
  free(v->slot);
  v->nslot = want;
  v->slot = (int *)malloc(sizeof(int) * want);
  for (long long i = 0; i < want; i++) v->slot[i] = -1;
  for (long long i = 0; i < v->size; i++) {
    uint64_t h = str_hash(v->word[i]) & (want - 1);
    while (v->slot[h] != -1) h = (h + 1) & (want - 1);
    v->slot[h] = (int)i;
  }
}
int w2bo_vocab_search(const w2bo_vocab *v, const char *word) { /* ref :166-174 */
  uint64_t h = str_hash(word) & (v->nslot - 1);
  for (;;) {
    int i = v->slot[h];
    if (i == -1) return -1;
    if (!strcmp(word, v->word[i])) return i;
    h = (h + 1) & (v->nslot - 1);
  }
}
static long long vocab_add(w2bo_vocab *v, const char *word) { /* ref :188-204 */
  if (v->size + 1 >= v->cap) {
    v->cap = v->cap ? v->cap * 2 : 1024;
    v->word = (char **)realloc(v->word, sizeof(char *) * v->cap);
    v->cn = (long long *)realloc(v->cn, sizeof(long long) * v->cap);
  }
  v->word[v->size] = strdup(word);
  v->cn[v->size] = 0;
  v->size++;
  if (v->size * 3 > v->nslot) map_rebuild(v);
  else {
    uint64_t h = str_hash(word) & (v->nslot - 1);
    while (v->slot[h] != -1) h = (h + 1) & (v->nslot - 1);
    v->slot[h] = (int)(v->size - 1);
  }
  return v->size - 1;
}

/* ReadWord over a byte buffer (ref :131-155).  Returns 1 at end-of-file (word dropped). */
static int read_word(const unsigned char *buf, long long size, long long *pos, char *word,
                     long long *begin) {
  int a = 0;
  for (;;) {
    if (*pos >= size) return 1;
    int ch = buf[(*pos)++];
    if (ch == 13) continue;
    if (ch == ' ' || ch == '\t' || ch == '\n') {
      if (a > 0) { if (ch == '\n') (*pos)--; break; }
      if (ch == '\n') { strcpy(word, "</s>"); *begin = *pos - 1; return 0; }
      continue;
    }
    if (a == 0) *begin = *pos - 1;
    word[a++] = (char)ch;
    if (a >= MAX_WORD - 1) a--;
  }
  word[a] = 0;
  return 0;
}

static unsigned char *slurp(const char *path, long long *size) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  *size = ftell(f);
  fseek(f, 0, SEEK_SET);
  unsigned char *b = (unsigned char *)malloc(*size + 1);
  if (fread(b, 1, *size, f) != (size_t)*size) { fclose(f); free(b); return NULL; }
  fclose(f);
  return b;
}

static int by_count_desc(const void *a, const void *b) { /* ref :207-212 */
  long long d = ((const vw_pair *)b)->cn - ((const vw_pair *)a)->cn;
  return d > 0 ? 1 : (d < 0 ? -1 : 0);
}

/* ReduceVocab (ref :245-263): called from the learning loop whenever the vocabulary outgrows 70 % of the hash table
 * (ref :293).  Every entry -- "</s>" at index 0 included, the reference does not protect it -- whose count is not
 * above min_reduce is removed, the survivors keep their relative order, and min_reduce goes up by one. */
static void reduce_vocab(w2bo_vocab *v, long long *min_reduce) {
  long long b = 0;
  for (long long a = 0; a < v->size; a++) {
    if (v->cn[a] > *min_reduce) { v->cn[b] = v->cn[a]; v->word[b] = v->word[a]; b++; }
    else free(v->word[a]);
  }
  v->size = b;
  map_rebuild(v);
  (*min_reduce)++;
}

/* ref :265-301, 215-242.  vocab_hash_size is the reference's constant of that name (ref :35: 30 000 000); it only
 * matters through the ReduceVocab trigger `vocab_size > vocab_hash_size * 0.7` (int times double, as written there). */
w2bo_vocab *w2bo_vocab_learn_ex(const char *train_file, int min_count, int vocab_hash_size) {
  long long size, pos = 0, begin, min_reduce = 1;          /* min_reduce: ref :48 */
  unsigned char *buf = slurp(train_file, &size);
  if (!buf) return NULL;
  w2bo_vocab *v = (w2bo_vocab *)calloc(1, sizeof(*v));
  map_rebuild(v);
  vocab_add(v, "</s>");
  char word[MAX_WORD];
  while (!read_word(buf, size, &pos, word, &begin)) {
    int i = w2bo_vocab_search(v, word);
    if (i == -1) { long long a = vocab_add(v, word); v->cn[a] = 1; }
    else v->cn[i]++;
    if (v->size > vocab_hash_size * 0.7) reduce_vocab(v, &min_reduce);   /* ref :293 */
  }
  vw_pair *p = (vw_pair *)malloc(sizeof(vw_pair) * v->size);
  for (long long i = 0; i < v->size; i++) { p[i].w = v->word[i]; p[i].cn = v->cn[i]; }
  qsort(p + 1, v->size - 1, sizeof(vw_pair), by_count_desc); /* same libc qsort as the reference */
  long long kept = 0;
  v->train_words = 0;
  for (long long i = 0; i < v->size; i++) {
    if (p[i].cn < min_count && i != 0) { free(p[i].w); continue; }
    v->word[kept] = p[i].w; v->cn[kept] = p[i].cn; kept++;
    v->train_words += p[i].cn;
  }
  v->size = kept;
  free(p);
  map_rebuild(v);
  v->file_size = size;
  free(buf);
  return v;
}
w2bo_vocab *w2bo_vocab_learn(const char *train_file, int min_count) {
  return w2bo_vocab_learn_ex(train_file, min_count, 30000000);          /* ref :35 */
}
void w2bo_vocab_free(w2bo_vocab *v) {
  if (!v) return;
  for (long long i = 0; i < v->size; i++) free(v->word[i]);
  free(v->word); free(v->cn); free(v->slot); free(v);
}
long long w2bo_vocab_size(const w2bo_vocab *v) { return v->size; }
long long w2bo_vocab_train_words(const w2bo_vocab *v) { return v->train_words; }
long long w2bo_vocab_file_size(const w2bo_vocab *v) { return v->file_size; }
const char *w2bo_vocab_word(const w2bo_vocab *v, long long i) { return v->word[i]; }
long long w2bo_vocab_count(const w2bo_vocab *v, long long i) { return v->cn[i]; }

long long w2bo_tokenize_file(const w2bo_vocab *v, const char *train_file, int **ids_out,
                             long long **begin_out) {
  long long size, pos = 0, begin = 0, n = 0, cap = 1 << 16;
  unsigned char *buf = slurp(train_file, &size);
  if (!buf) return -1;
  int *ids = (int *)malloc(sizeof(int) * cap);
  long long *bg = (long long *)malloc(sizeof(long long) * cap);
  char word[MAX_WORD];
  while (!read_word(buf, size, &pos, word, &begin)) {
    if (n == cap) {
      cap *= 2;
      ids = (int *)realloc(ids, sizeof(int) * cap);
      bg = (long long *)realloc(bg, sizeof(long long) * cap);
    }
    ids[n] = w2bo_vocab_search(v, word);
    bg[n] = begin;
    n++;
  }
  free(buf);
  *ids_out = ids; *begin_out = bg;
  return n;
}

long long w2bo_shard_start(const w2bo_vocab *v, const char *train_file, long long offset,
                           const long long *begin, long long n, int *override) {
  long long size, pos = offset, wb = -1;
  unsigned char *buf = slurp(train_file, &size);
  char word[MAX_WORD];
  *override = -2;
  /* first token whose first byte is at or after the seek offset */
  long long lo = 0, hi = n;
  while (lo < hi) { long long mid = (lo + hi) / 2; if (begin[mid] >= offset) hi = mid; else lo = mid + 1; }
  if (buf && !read_word(buf, size, &pos, word, &wb)) {
    if (lo < n && wb == begin[lo]) { /* landed on a token boundary */ }
    else *override = w2bo_vocab_search(v, word); /* truncated word (mid-word seek) */
  }
  free(buf);
  return lo;
}

/* ------------------------------------------------------------------ whole program */

static void save_vectors(const char *path, const w2bo_vocab *vb, const w2bo_model *m, int binary) {
  FILE *fo = fopen(path, "wb"); /* ref :560-576 */
  fprintf(fo, "%lld %lld\n", m->vocab_size, m->dim);
  for (long long a = 0; a < m->vocab_size; a++) {
    fprintf(fo, "%s ", vb->word[a]);
    for (long long b = 0; b < m->dim; b++) {
      float s = m->u[a * m->dim + b] + m->v[a * m->dim + b];
      s = w2bo_quantize(s, m->bitlevel);
      if (binary) fwrite(&s, sizeof(float), 1, fo);
      else fprintf(fo, "%lf ", s);
    }
    fprintf(fo, "\n");
  }
  fclose(fo);
}

int w2bo_run(const char *train_file, const char *output_file, int bitlevel, int dim, int window,
             int negative, int num_threads, int iter, int min_count, float alpha, float sample,
             float reg, int binary, double *epoch_losses) {
  return w2bo_run_ex(train_file, output_file, bitlevel, dim, window, negative, num_threads, iter, min_count, alpha,
                     sample, reg, binary, epoch_losses, 30000000);
}

int w2bo_run_ex(const char *train_file, const char *output_file, int bitlevel, int dim, int window,
                int negative, int num_threads, int iter, int min_count, float alpha, float sample,
                float reg, int binary, double *epoch_losses, int vocab_hash_size) {
  w2bo_vocab *vb = w2bo_vocab_learn_ex(train_file, min_count, vocab_hash_size);
  if (!vb) return 1;
  w2bo_model m;
  memset(&m, 0, sizeof m);
  m.vocab_size = vb->size; m.dim = dim; m.train_words = vb->train_words; m.iter = iter;
  m.window = window; m.negative = negative; m.bitlevel = bitlevel; m.num_threads = num_threads;
  m.starting_alpha = alpha; m.alpha = alpha; m.sample = sample; m.reg = reg;
  m.cn = vb->cn; m.compute_loss = 1;
  m.u = (float *)malloc(sizeof(float) * m.vocab_size * dim);
  m.v = (float *)malloc(sizeof(float) * m.vocab_size * dim);
  float *et = (float *)malloc(sizeof(float) * (W2BO_EXP_TABLE_SIZE + 1));
  w2bo_build_exp_table(et);
  m.exp_table = et;
  w2bo_init_net(m.vocab_size, dim, m.u, m.v);
  int *table = NULL;
  if (negative > 0) {
    table = (int *)malloc(sizeof(int) * W2BO_TABLE_SIZE);
    w2bo_build_unigram_table(vb->cn, vb->size, table, W2BO_TABLE_SIZE);
  }
  m.table = table; m.table_size = W2BO_TABLE_SIZE;
  int *ids; long long *begin;
  long long n = w2bo_tokenize_file(vb, train_file, &ids, &begin);
  long long *starts = (long long *)malloc(sizeof(long long) * num_threads);
  int *ov = (int *)malloc(sizeof(int) * num_threads);
  for (int w = 0; w < num_threads; w++)
    starts[w] = w2bo_shard_start(vb, train_file, vb->file_size / (long long)num_threads * w,
                                 begin, n, &ov[w]);   /* ref :377 */
  for (int it = 0; it < iter; it++) {
    double l = w2bo_train_epoch_tokens(&m, ids, n, starts, ov, num_threads);
    if (epoch_losses) epoch_losses[it] = l;
  }
  save_vectors(output_file, vb, &m, binary);
  free(ids); free(begin); free(starts); free(ov); free(table); free(et);
  free(m.u); free(m.v);
  w2bo_vocab_free(vb);
  return 0;
}

/* ------------------------------------------------------------------ evaluator numerics */

/* src/compute-accuracy.c:106-110.  The evaluator carries its own copy of quantize (:26-61),
 * identical to the trainer's; len accumulates in float, sqrt() is the double one applied to
 * a float and stored back to float. */
void w2bo_eval_normalize(float *M, long long words, long long size, int bitlevel, int fma) {
  for (long long b = 0; b < words; b++) {
    float *row = M + b * size;
    for (long long a = 0; a < size; a++) row[a] = w2bo_quantize(row[a], bitlevel);
    float len = 0;
    for (long long a = 0; a < size; a++) len = fma ? fmaf(row[a], row[a], len) : len + row[a] * row[a];
    len = (float)sqrt((double)len);
    for (long long a = 0; a < size; a++) row[a] /= len;
  }
}

/* src/compute-accuracy.c:146-177 (N = 1): bestd starts at 0, a candidate replaces it only when
 * strictly greater, candidates are visited in row order, the three question rows are skipped. */
void w2bo_eval_top1(const float *M, long long words, long long size, long long nq, const int *b1,
                    const int *b2, const int *b3, int fma, int *best, float *bestd) {
  float *vec = (float *)malloc(sizeof(float) * (size > 0 ? size : 1));
  for (long long q = 0; q < nq; q++) {
    for (long long a = 0; a < size; a++)
      vec[a] = (M[a + b2[q] * size] - M[a + b1[q] * size]) + M[a + b3[q] * size];
    float bd = 0;
    int bi = -1;
    for (long long c = 0; c < words; c++) {
      if (c == b1[q] || c == b2[q] || c == b3[q]) continue;
      const float *row = M + c * size;
      float dist = 0;
      if (fma) for (long long a = 0; a < size; a++) dist = fmaf(vec[a], row[a], dist);
      else for (long long a = 0; a < size; a++) dist += vec[a] * row[a];
      if (dist > bd) { bd = dist; bi = (int)c; }
    }
    best[q] = bi;
    bestd[q] = bd;
  }
  free(vec);
}
