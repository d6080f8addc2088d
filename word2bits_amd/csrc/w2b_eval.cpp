// w2b_eval.cpp -- host side of include/word2bits_eval.h: the vector-file reader, the question-stream state
// machine and the stdout transcript of the reference evaluator (ref src/compute-accuracy.c:80-188), around the
// GPU scan in w2b_kernels_eval.hip.  No arithmetic on scores happens here and there is no CPU fallback.
#include "../../include/word2bits_eval.h"
#include "../../include/word2bits_hip.h"
#include "w2b_internal.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace {
constexpr int64_t kMaxW = 50;            // ref :24 max_w
constexpr int64_t kTile = 256;           // padding unit of rows / questions (covers both kernels' tiles)
constexpr int64_t kChunkQ = 1 << 16;     // questions per launch

inline bool is_space(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }   // isspace, C locale
inline char c_upper(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }     // toupper, C locale

int efail(int code, const std::string &msg) { return w2b_internal_fail(code, msg.c_str()); }
#define EHIP(x)                                                                                   \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess) return efail(W2B_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
  } while (0)
}  // namespace

struct w2b_eval {
  int device = 0;
  hipStream_t stream = nullptr;
  int64_t words = 0, size = 0, ld = 0, rows_padded = 0;
  int fused = 1;
  int variant = 1;                                      // W2B_EVAL_KERNEL=0: vector-ALU kernel also in fused mode (default: MFMA)
  std::vector<char> vocab;                              // flat [words * max_w] (+ slack), as ref :88
  std::unordered_map<std::string, int64_t> first;       // upper-cased word -> first row (ref :140)
  float *M = nullptr;                                   // [rows_padded][ld], zero padded, normalised
  // per-call scratch (grown on demand)
  float *Q = nullptr;
  int32_t *b123 = nullptr;
  unsigned long long *best = nullptr;
  int64_t cap_q = 0;
  double kernel_ms = 0;                                 // score-kernel time since the last timing_read
  int64_t launches = 0;
  double macs = 0;
};

static void eval_release(w2b_eval *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->M) (void)hipFree(e->M);
  if (e->Q) (void)hipFree(e->Q);
  if (e->b123) (void)hipFree(e->b123);
  if (e->best) (void)hipFree(e->best);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

// fscanf(f, "%lld", &x): skip white space, optional sign, digits
static bool scan_ll(const std::vector<unsigned char> &d, size_t &pos, long long *out) {
  while (pos < d.size() && is_space(d[pos])) pos++;
  size_t st = pos;
  if (pos < d.size() && (d[pos] == '+' || d[pos] == '-')) pos++;
  size_t dig = pos;
  while (pos < d.size() && d[pos] >= '0' && d[pos] <= '9') pos++;
  if (pos == dig) return false;
  *out = strtoll(std::string(d.begin() + st, d.begin() + pos).c_str(), nullptr, 10);
  return true;
}

// One row's name as the reference's reader leaves it (ref :97-105): bytes up to the first ' ', '\n' bytes dropped, at
// most max_w characters kept (the rest lands on index max_w = the next row's first byte, overwritten by that row
// later), a terminating 0, then upper-cased.  Consumes from d[pos...].
static void read_name(const unsigned char *d, size_t n, size_t &pos, char *row) {
  long long a = 0;
  for (;;) {
    const bool at_end = pos >= n;
    const unsigned char ch = at_end ? 0xFF : d[pos];              // (char)EOF
    if (!at_end) pos++;
    row[a] = (char)ch;
    if (at_end || ch == ' ') break;
    if (a < kMaxW && ch != '\n') a++;
  }
  row[a] = 0;
  for (long long i = 0; i < kMaxW; i++) row[i] = c_upper(row[i]);
}

// common tail of the two constructors: device buffers, quantize(x, bitlevel) + normalisation of the rows (ref :106-110).
// `host_rows` ([words][size], may be null) or `dev_rows` ([words][size] on the device, may be null) hold the raw values.
static int eval_finish(w2b_eval *e, int32_t bitlevel, const float *host_rows, const float *dev_rows, w2b_eval **out) {
  const long long words = e->words, size = e->size;
  char *vocab = e->vocab.data();
  for (long long b = 0; b < words; b++) e->first.emplace(std::string(vocab + b * kMaxW), b);   // first wins
  auto bail = [&](int rc) { eval_release(e); return rc; };
  if (hipSetDevice(e->device) != hipSuccess) return bail(efail(W2B_EHIP, "hipSetDevice failed"));
  if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess)
    return bail(efail(W2B_EHIP, "hipStreamCreate failed"));
  const size_t mbytes = (size_t)e->rows_padded * e->ld * 4;
  float *len = nullptr;
  if (hipMalloc(&e->M, mbytes) != hipSuccess || hipMalloc(&len, (size_t)(words + 1) * 4) != hipSuccess)
    return bail(efail(W2B_ENOMEM, "w2b_eval: device allocation failed"));
  hipError_t he = hipMemsetAsync(e->M, 0, mbytes, e->stream);
  if (he == hipSuccess && words > 0 && host_rows)
    he = hipMemcpy2DAsync(e->M, (size_t)e->ld * 4, host_rows, (size_t)size * 4, (size_t)size * 4, (size_t)words,
                          hipMemcpyHostToDevice, e->stream);
  if (he == hipSuccess && words > 0 && dev_rows)
    he = hipMemcpy2DAsync(e->M, (size_t)e->ld * 4, dev_rows, (size_t)size * 4, (size_t)size * 4, (size_t)words,
                          hipMemcpyDeviceToDevice, e->stream);
  if (he == hipSuccess) he = w2b_launch_eval_normalize(e->M, words, size, e->ld, bitlevel, e->fused, len, e->stream);
  if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
  (void)hipFree(len);
  if (he != hipSuccess) return bail(efail(W2B_EHIP, std::string("w2b_eval: ") + hipGetErrorString(he)));
  *out = e;
  return W2B_OK;
}

static w2b_eval *eval_new(long long words, long long size, int32_t fused, int32_t device) {
  w2b_eval *e = new w2b_eval;
  e->device = device;
  e->words = words;
  e->size = size;
  e->fused = fused ? 1 : 0;
  e->ld = (size + 15) / 16 * 16;
  e->rows_padded = (words + kTile - 1) / kTile * kTile;
  if (e->rows_padded == 0) e->rows_padded = kTile;
  e->vocab.assign((size_t)(words * kMaxW + kMaxW + 2), 0);
  return e;
}

extern "C" int w2b_eval_load(const char *file, int32_t bitlevel, int64_t threshold, int32_t fused, int32_t device,
                             w2b_eval **out) {
  if (!file || !out) return efail(W2B_EINVAL, "w2b_eval_load: null argument");
  *out = nullptr;
  FILE *f = fopen(file, "rb");
  if (!f) return efail(W2B_EIO, "Input file not found");           // ref :81-84
  std::vector<unsigned char> d;
  {
    fseek(f, 0, SEEK_END);
    const long long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    d.resize(n > 0 ? (size_t)n : 0);
    if (n > 0 && fread(d.data(), 1, (size_t)n, f) != (size_t)n) {
      fclose(f);
      return efail(W2B_EIO, "w2b_eval_load: short read");
    }
    fclose(f);
  }
  if (w2b_internal_is_packed(d.data(), d.size())) {
    // a bit-packed model file (include/word2bits_corpus.h): rebuilt in memory as the bytes of the reference's binary file
    // (ref src/word2bits.cpp:560-576), which then go through the reader below like any other file
    std::vector<std::string> names;
    std::vector<float> values;
    int64_t dim = 0;
    if (w2b_internal_parse_packed(d.data(), d.size(), names, values, &dim) != W2B_OK)
      return efail(W2B_EIO, "w2b_eval_load: damaged bit-packed file");
    std::vector<unsigned char> b;
    char head[64];
    const int hl = snprintf(head, sizeof head, "%lld %lld\n", (long long)names.size(), (long long)dim);
    b.insert(b.end(), head, head + hl);
    for (size_t a = 0; a < names.size(); a++) {
      b.insert(b.end(), names[a].begin(), names[a].end());
      b.push_back(' ');
      const unsigned char *row = (const unsigned char *)(values.data() + a * (size_t)dim);
      b.insert(b.end(), row, row + (size_t)dim * 4);
      b.push_back('\n');
    }
    d.swap(b);
  }
  size_t pos = 0;
  long long words = 0, size = 0;
  if (!scan_ll(d, pos, &words)) return efail(W2B_EIO, "w2b_eval_load: no <words> header");
  if (threshold && words > threshold) words = threshold;            // ref :86
  if (!scan_ll(d, pos, &size)) return efail(W2B_EIO, "w2b_eval_load: no <size> header");
  if (words < 0 || size <= 0 || words > 0x7FFFFF00ll || size > (1 << 24))
    return efail(W2B_EINVAL, "w2b_eval_load: unsupported <words> <size>");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return efail(W2B_ENOGPU, "w2b_eval_load: no HIP device visible (the evaluator has no CPU fallback)");
  if (device < 0 || device >= ndev) return efail(W2B_EINVAL, "w2b_eval_load: bad device index");

  w2b_eval *e = eval_new(words, size, fused, device);
  std::vector<float> raw((size_t)(words * size), 0.f);
  char *vocab = e->vocab.data();
  for (long long b = 0; b < words; b++) {                           // ref :96-105
    read_name(d.data(), d.size(), pos, vocab + b * kMaxW);
    const size_t want = (size_t)size * 4, have = d.size() - pos;
    const size_t take = (want < have ? want : have) / 4 * 4;
    memcpy(raw.data() + b * size, d.data() + pos, take);
    pos = want <= have ? pos + want : d.size();   // a short fread also swallows the 1-3 bytes of a cut float
  }
  d.clear();
  d.shrink_to_fit();
  return eval_finish(e, bitlevel, raw.data(), nullptr, out);
}

// The evaluator on a LIVE trainer: what `compute_accuracy <file> <bitlevel> <threshold>` would load after
// `./word2bits -binary 1` had written <file> from this trainer -- without the file: quantize(u+v) (ref src/word2bits.cpp
// :568-569) is exported on the device straight into the evaluator's matrix, the names go through the same reader
// logic as a file's would (each row of a binary file is "word" + ' ' + floats + '\n', ref :565-574).
extern "C" int w2b_eval_from_trainer(w2b_trainer *t, int64_t n_words, const char *const *words_in, int32_t bitlevel,
                                     int64_t threshold, int32_t fused, w2b_eval **out) {
  if (!t || !out || (n_words > 0 && !words_in)) return efail(W2B_EINVAL, "w2b_eval_from_trainer: null argument");
  *out = nullptr;
  float *u = nullptr, *v = nullptr;
  long long V = 0, D = 0;
  int tb = 0, dev = 0;
  hipStream_t ts = nullptr;
  w2b_internal_trainer_view(t, &u, &v, &V, &D, &tb, &dev, &ts);
  if (n_words != V) return efail(W2B_EINVAL, "w2b_eval_from_trainer: one word per vocabulary row is needed");
  long long words = V;
  if (threshold && words > threshold) words = threshold;            // ref :86
  w2b_eval *e = eval_new(words, D, fused, dev);
  char *vocab = e->vocab.data();
  std::string rowbytes;
  for (long long b = 0; b < words; b++) {
    rowbytes.assign("\n");                                          // what the previous row (or the header) left behind
    rowbytes += words_in[b];
    rowbytes += ' ';
    size_t pos = 0;
    read_name((const unsigned char *)rowbytes.data(), rowbytes.size(), pos, vocab + b * kMaxW);
  }
  if (hipSetDevice(dev) != hipSuccess) { eval_release(e); return efail(W2B_EHIP, "hipSetDevice failed"); }
  float *q = nullptr;
  if (hipMalloc(&q, sizeof(float) * (size_t)(words > 0 ? words : 1) * D) != hipSuccess) {
    eval_release(e);
    return efail(W2B_ENOMEM, "w2b_eval_from_trainer: device allocation failed");
  }
  hipError_t he = w2b_launch_export(u, v, q, words * D, tb, ts);     // quantize(u+v) with the TRAINER's bitlevel
  if (he == hipSuccess) he = hipStreamSynchronize(ts);
  if (he != hipSuccess) {
    (void)hipFree(q);
    eval_release(e);
    return efail(W2B_EHIP, std::string("w2b_eval_from_trainer: ") + hipGetErrorString(he));
  }
  const int rc = eval_finish(e, bitlevel, nullptr, q, out);
  (void)hipFree(q);
  return rc;
}

extern "C" void w2b_eval_free(w2b_eval *e) { eval_release(e); }
extern "C" int64_t w2b_eval_words(const w2b_eval *e) { return e ? e->words : 0; }
extern "C" int64_t w2b_eval_size(const w2b_eval *e) { return e ? e->size : 0; }
extern "C" const char *w2b_eval_word(const w2b_eval *e, int64_t row) {
  if (!e || row < 0 || row >= e->words) return nullptr;
  return e->vocab.data() + row * kMaxW;
}
extern "C" int64_t w2b_eval_lookup(const w2b_eval *e, const char *upper_word) {
  if (!e || !upper_word) return 0;
  auto it = e->first.find(upper_word);
  return it == e->first.end() ? e->words : it->second;
}

extern "C" int w2b_eval_get_matrix(w2b_eval *e, float *out) {
  if (!e || !out) return efail(W2B_EINVAL, "w2b_eval_get_matrix: null argument");
  EHIP(hipSetDevice(e->device));
  if (e->words > 0)
    EHIP(hipMemcpy2D(out, (size_t)e->size * 4, e->M, (size_t)e->ld * 4, (size_t)e->size * 4, (size_t)e->words,
                     hipMemcpyDeviceToHost));
  return W2B_OK;
}

extern "C" int w2b_eval_top1(w2b_eval *e, int64_t nq, const int32_t *b1, const int32_t *b2, const int32_t *b3,
                             int32_t *best, float *bestd) {
  if (!e || nq < 0 || (nq > 0 && (!b1 || !b2 || !b3 || !best)))
    return efail(W2B_EINVAL, "w2b_eval_top1: bad argument");
  for (int64_t q = 0; q < nq; q++)
    if (b1[q] < 0 || b1[q] >= e->words || b2[q] < 0 || b2[q] >= e->words || b3[q] < 0 || b3[q] >= e->words)
      return efail(W2B_EINVAL, "w2b_eval_top1: question row out of range");
  EHIP(hipSetDevice(e->device));
  std::vector<unsigned long long> keys;
  for (int64_t q0 = 0; q0 < nq; q0 += kChunkQ) {
    const int64_t n = (nq - q0 < kChunkQ) ? nq - q0 : kChunkQ;
    const int64_t np = (n + kTile - 1) / kTile * kTile;
    if (np > e->cap_q) {
      if (e->Q) (void)hipFree(e->Q);
      if (e->b123) (void)hipFree(e->b123);
      if (e->best) (void)hipFree(e->best);
      e->Q = nullptr; e->b123 = nullptr; e->best = nullptr; e->cap_q = 0;
      if (hipMalloc(&e->Q, (size_t)np * e->ld * 4) != hipSuccess || hipMalloc(&e->b123, (size_t)np * 12) != hipSuccess ||
          hipMalloc(&e->best, (size_t)np * 8) != hipSuccess)
        return efail(W2B_ENOMEM, "w2b_eval_top1: device allocation failed");
      e->cap_q = np;
    }
    int32_t *d1 = e->b123, *d2 = e->b123 + np, *d3 = e->b123 + 2 * np;
    EHIP(hipMemcpyAsync(d1, b1 + q0, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
    EHIP(hipMemcpyAsync(d2, b2 + q0, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
    EHIP(hipMemcpyAsync(d3, b3 + q0, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
    EHIP(hipMemsetAsync(e->Q, 0, (size_t)np * e->ld * 4, e->stream));
    EHIP(hipMemsetAsync(e->best, 0, (size_t)np * 8, e->stream));
    EHIP(w2b_launch_eval_queries(e->M, e->ld, n, d1, d2, d3, e->Q, e->variant, e->stream));
    hipEvent_t t0, t1;
    EHIP(hipEventCreate(&t0));
    EHIP(hipEventCreate(&t1));
    EHIP(hipEventRecord(t0, e->stream));
    hipError_t le = w2b_launch_eval_scores(e->Q, e->M, (int)n, (int)e->words, (int)e->size, (int)e->ld, e->fused, d1, d2, d3,
                                           e->best, e->variant, e->stream);
    if (le == hipSuccess) le = hipEventRecord(t1, e->stream);
    keys.resize((size_t)n);
    if (le == hipSuccess) le = hipMemcpyAsync(keys.data(), e->best, (size_t)n * 8, hipMemcpyDeviceToHost, e->stream);
    if (le == hipSuccess) le = hipStreamSynchronize(e->stream);
    float ms = 0;
    if (le == hipSuccess) le = hipEventElapsedTime(&ms, t0, t1);
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    if (le != hipSuccess) return efail(W2B_EHIP, std::string("w2b_eval_top1: ") + hipGetErrorString(le));
    e->kernel_ms += ms;
    e->launches++;
    e->macs += (double)n * (double)e->words * (double)e->size;   // algorithmic: padding is not work
    for (int64_t q = 0; q < n; q++) {
      const unsigned long long k = keys[(size_t)q];
      best[q0 + q] = k ? (int32_t)(0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull)) : -1;
      if (bestd) {
        const uint32_t bits = (uint32_t)(k >> 32);
        memcpy(&bestd[q0 + q], &bits, 4);
      }
    }
  }
  return W2B_OK;
}

extern "C" int w2b_eval_set_kernel(w2b_eval *e, int32_t variant) {
  if (!e) return efail(W2B_EINVAL, "w2b_eval_set_kernel: null evaluator");
  if (variant < 0 || variant > 64) return efail(W2B_EINVAL, "w2b_eval_set_kernel: variant must be 0..64");
  e->variant = variant;
  return W2B_OK;
}

extern "C" int w2b_eval_timing_read(w2b_eval *e, double *kernel_ms, int64_t *launches, double *macs) {
  if (!e) return efail(W2B_EINVAL, "w2b_eval_timing_read: null handle");
  EHIP(hipSetDevice(e->device));
  const double ms = e->kernel_ms;
  e->kernel_ms = 0;
  if (kernel_ms) *kernel_ms = ms;
  if (launches) *launches = e->launches;
  if (macs) *macs = e->macs;
  e->launches = 0;
  e->macs = 0;
  return W2B_OK;
}

// ------------------------------------------------------------------------------------ transcript
namespace {
// scanf("%s", st) over a buffer.  `hit_end` mirrors feof(stdin): it latches as soon as a read runs into the end
// of the input, which also happens while reading a last token that has no trailing white space.
struct TokenIn {
  const char *p;
  int64_t n, pos = 0;
  bool hit_end = false;
  bool next(std::string &st) {             // false: nothing read, st keeps its old contents
    while (pos < n && is_space((unsigned char)p[pos])) pos++;
    if (pos >= n) { hit_end = true; return false; }
    const int64_t s = pos;
    while (pos < n && !is_space((unsigned char)p[pos])) pos++;
    if (pos >= n) hit_end = true;
    st.assign(p + s, (size_t)(pos - s));
    return true;
  }
};
void upper_inplace(std::string &s) { for (char &c : s) c = c_upper(c); }

struct Step {                // what the loop of ref :114-186 does, in stream order
  enum Kind { SectionEnd, SectionName, Question } kind;
  int qid;                   // QID at that moment
  std::string text;          // section name / expected word (st4)
  int64_t q;                 // index into the batched questions
};

void appendf(std::string &out, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void appendf(std::string &out, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  const int n = vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  out.append(buf, (size_t)(n < (int)sizeof buf ? n : (int)sizeof buf - 1));
}
}  // namespace

extern "C" int w2b_eval_transcript(w2b_eval *e, const char *questions, int64_t len, char **out, int64_t *out_len) {
  if (!e || !out || len < 0 || (len > 0 && !questions)) return efail(W2B_EINVAL, "w2b_eval_transcript: bad argument");
  *out = nullptr;
  TokenIn in{questions, len};
  std::string st1, st2, st3, st4;
  std::vector<Step> steps;
  std::vector<int32_t> b1s, b2s, b3s;
  int QID = 0, TQ = 0, TQS = 0;
  // pass 1: parse the stream exactly like the scanf loop; collect the answerable questions
  for (;;) {
    in.next(st1);
    upper_inplace(st1);
    if (st1 == ":" || st1 == "EXIT" || in.hit_end) {                 // ref :119
      steps.push_back({Step::SectionEnd, QID, std::string(), 0});
      QID++;
      in.next(st1);                                                   // section name, printed as read (ref :126-128)
      if (in.hit_end) break;
      steps.push_back({Step::SectionName, QID, st1, 0});
      continue;
    }
    in.next(st2); upper_inplace(st2);
    in.next(st3); upper_inplace(st3);
    in.next(st4); upper_inplace(st4);
    const int64_t r1 = w2b_eval_lookup(e, st1.c_str()), r2 = w2b_eval_lookup(e, st2.c_str()),
                  r3 = w2b_eval_lookup(e, st3.c_str());
    TQ++;
    if (r1 == e->words || r2 == e->words || r3 == e->words) continue;  // ref :149-151
    if (w2b_eval_lookup(e, st4.c_str()) == e->words) continue;          // ref :152-153
    TQS++;
    steps.push_back({Step::Question, QID, st4, (int64_t)b1s.size()});
    b1s.push_back((int32_t)r1);
    b2s.push_back((int32_t)r2);
    b3s.push_back((int32_t)r3);
  }
  // the scan of ref :155-177 for all of them at once, on the GPU
  std::vector<int32_t> best(b1s.size());
  if (!b1s.empty()) {
    const int rc = w2b_eval_top1(e, (int64_t)b1s.size(), b1s.data(), b2s.data(), b3s.data(), best.data(), nullptr);
    if (rc != W2B_OK) return rc;
  }
  // pass 2: replay the counters and print (ref :120-131,178-187)
  std::string txt = "Starting eval...\n";
  int TCN = 0, CCN = 0, TACN = 0, CACN = 0, SECN = 0, SYCN = 0, SEAC = 0, SYAC = 0;
  for (const Step &s : steps) {
    if (s.kind == Step::SectionEnd) {
      if (TCN == 0) TCN = 1;
      if (s.qid != 0) {
        appendf(txt, "ACCURACY TOP1: %.2f %%  (%d / %d)\n", CCN / (float)TCN * 100, CCN, TCN);
        appendf(txt, "Total accuracy: %.2f %%   Semantic accuracy: %.2f %%   Syntactic accuracy: %.2f %% \n",
                CACN / (float)TACN * 100, SEAC / (float)SECN * 100, SYAC / (float)SYCN * 100);
      }
    } else if (s.kind == Step::SectionName) {
      txt += s.text;
      txt += ":\n";
      TCN = 0;
      CCN = 0;
    } else {
      const int32_t c = best[(size_t)s.q];
      const char *bestw = c >= 0 ? e->vocab.data() + (int64_t)c * kMaxW : "";
      if (s.text == bestw) {                                           // strcmp on the words (ref :178)
        CCN++;
        CACN++;
        if (s.qid <= 5) SEAC++; else SYAC++;
      }
      if (s.qid <= 5) SECN++; else SYCN++;
      TCN++;
      TACN++;
    }
  }
  appendf(txt, "Questions seen / total: %d %d   %.2f %% \n", TQS, TQ, TQS / (float)TQ * 100);
  char *buf = (char *)malloc(txt.size() + 1);
  if (!buf) return efail(W2B_ENOMEM, "w2b_eval_transcript: out of memory");
  memcpy(buf, txt.data(), txt.size());
  buf[txt.size()] = 0;
  *out = buf;
  if (out_len) *out_len = (int64_t)txt.size();
  return W2B_OK;
}

extern "C" void w2b_eval_free_text(char *text) { free(text); }
