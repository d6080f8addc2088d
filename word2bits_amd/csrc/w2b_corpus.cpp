// w2b_corpus.cpp -- host-side corpus ingest behind include/word2bits_corpus.h.
// One pass over a memory-mapped training file builds the vocabulary, a second pass emits the
// int32 token stream the GPU workers walk.  Semantics follow the reference's stdio reader
// (ref src/word2bits.cpp:131-301, :377) -- see the notes at each function.
#include "../../include/word2bits_corpus.h"
#include "../../include/word2bits_hip.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxWord = 4096;          // MAX_STRING, ref :29
constexpr int64_t kCheckpointEvery = 1 << 12;

struct Reader {                          // byte cursor over the mapped file
  const unsigned char *buf;
  int64_t size, pos;
};

// One token as ReadWord delivers it (ref :131-155): bytes 13 are skipped, ' ' '\t' '\n' separate,
// a newline is its own token "</s>", words are cut at kMaxWord-1 characters (the last slot keeps
// being overwritten), a final word that runs into end-of-file is dropped.
// Returns false at end of file.  `begin` = offset of the token's first byte.
inline bool next_token(Reader &r, char *word, int &len, int64_t &begin) {
  int a = 0;
  while (true) {
    if (r.pos >= r.size) return false;
    const int ch = r.buf[r.pos++];
    if (ch == 13) continue;
    if (ch == ' ' || ch == '\t' || ch == '\n') {
      if (a > 0) {
        if (ch == '\n') r.pos--;       // the newline is delivered as the next token
        break;
      }
      if (ch == '\n') {
        memcpy(word, "</s>", 5);
        len = 4;
        begin = r.pos - 1;
        return true;
      }
      continue;
    }
    if (a == 0) begin = r.pos - 1;
    word[a++] = (char)ch;
    if (a >= kMaxWord - 1) a--;
  }
  word[a] = 0;
  len = a;
  return true;
}

inline uint64_t fnv1a(const char *s, int len) {
  uint64_t h = 1469598103934665603ULL;
  for (int i = 0; i < len; i++) {
    h ^= (unsigned char)s[i];
    h *= 1099511628211ULL;
  }
  return h;
}

struct StringMap {                        // open addressing: word -> dense id
  std::vector<int32_t> slot;
  std::vector<uint64_t> hash;
  std::vector<uint32_t> off;              // offset of each word in the arena
  std::vector<char> arena;
  uint64_t mask = 0;

  void rehash(size_t cap) {
    slot.assign(cap, -1);
    mask = cap - 1;
    for (size_t i = 0; i < hash.size(); i++) {
      uint64_t h = hash[i] & mask;
      while (slot[h] != -1) h = (h + 1) & mask;
      slot[h] = (int32_t)i;
    }
  }
  const char *word(int32_t id) const { return arena.data() + off[id]; }
  int32_t find(const char *w, int len, uint64_t h) const {
    uint64_t p = h & mask;
    while (true) {
      const int32_t id = slot[p];
      if (id == -1) return -1;
      if (hash[id] == h && !strcmp(word(id), w)) return id;
      p = (p + 1) & mask;
    }
  }
  int32_t add(const char *w, int len, uint64_t h) {
    if ((hash.size() + 1) * 2 > slot.size()) rehash(slot.empty() ? 1 << 16 : slot.size() * 2);
    const int32_t id = (int32_t)hash.size();
    hash.push_back(h);
    off.push_back((uint32_t)arena.size());
    arena.insert(arena.end(), w, w + len + 1);
    uint64_t p = h & mask;
    while (slot[p] != -1) p = (p + 1) & mask;
    slot[p] = id;
    return id;
  }
};

}  // namespace

struct w2b_corpus {
  std::vector<std::string> words;         // final vocabulary order (row order of u / v)
  std::vector<int64_t> counts;
  StringMap final_map;                    // word -> final id
  int64_t train_words = 0, file_size = 0;
  std::vector<int32_t> tokens;            // in-vocabulary token stream
  // sparse index for the shard arithmetic: raw-token checkpoints
  std::vector<int64_t> cp_byte, cp_index; // byte offset of a raw token / number of in-vocab tokens before it
  int fd = -1;
  const unsigned char *map = nullptr;
};

extern "C" void w2b_corpus_free(w2b_corpus *c) {
  if (!c) return;
  if (c->map && c->file_size > 0) munmap((void *)c->map, (size_t)c->file_size);
  if (c->fd >= 0) close(c->fd);
  delete c;
}

extern "C" int w2b_corpus_load(const char *train_file, int32_t min_count, w2b_corpus **out) {
  return w2b_corpus_load_ex(train_file, min_count, 0, out);
}

extern "C" int w2b_corpus_load_ex(const char *train_file, int32_t min_count, int32_t vocab_hash_size, w2b_corpus **out) {
  if (!train_file || !out || vocab_hash_size < 0) return W2B_EINVAL;
  if (vocab_hash_size == 0) vocab_hash_size = 30000000;    // ref :35
  *out = nullptr;
  const int fd = open(train_file, O_RDONLY);
  if (fd < 0) return W2B_EIO;
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); return W2B_EIO; }
  w2b_corpus *c = new w2b_corpus();
  c->fd = fd;
  c->file_size = st.st_size;
  if (st.st_size > 0) {
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { w2b_corpus_free(c); return W2B_EIO; }
    c->map = (const unsigned char *)m;
  }
  // ---- pass 1, parallel: the file is cut at token boundaries (right after a ' ', '\t' or '\n': the reader's
  // state there is "no word open", so every piece tokenises exactly like the sequential scan) and each host
  // thread tokenises and counts its piece into a private table (ref :277-293 per piece).
  int nthreads = (int)std::thread::hardware_concurrency();
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 32) nthreads = 32;     // the merge below is serial in (pieces x distinct words per piece)
  int64_t min_piece = 4 << 20;
  if (const char *e = getenv("W2B_INGEST_THREADS")) nthreads = std::max(1, atoi(e));
  if (const char *e = getenv("W2B_INGEST_MIN_PIECE")) min_piece = std::max<int64_t>(1, atoll(e));
  const int npieces = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, c->file_size / min_piece));
  std::vector<int64_t> cut(npieces + 1, c->file_size);
  cut[0] = 0;
  for (int t = 1; t < npieces; t++) {
    int64_t p = std::max(cut[t - 1], c->file_size / npieces * t);
    while (p < c->file_size && p > 0 && !(c->map[p - 1] == ' ' || c->map[p - 1] == '\t' || c->map[p - 1] == '\n')) p++;
    cut[t] = p;
  }
  struct Piece {
    StringMap seen;                       // private first-appearance ids
    std::vector<int64_t> cn;
    std::vector<int32_t> raw;             // private id of every raw token
    std::vector<int64_t> cp_raw, cp_byte; // every kCheckpointEvery-th raw token: its index and first byte
    std::vector<int32_t> to_final;        // private id -> final vocabulary id (-1: dropped)
    std::vector<int32_t> out;             // in-vocabulary tokens of the piece
    std::vector<int64_t> cp_index;        // in-vocabulary tokens of the piece before each checkpoint
  };
  std::vector<Piece> pieces(npieces);
  const bool timing = getenv("W2B_INGEST_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[ingest] %-18s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  auto run_parallel = [&](auto &&fn) {
    std::vector<std::thread> th;
    for (int t = 1; t < npieces; t++) th.emplace_back(fn, t);
    fn(0);
    for (auto &x : th) x.join();
  };
  run_parallel([&](int t) {
    Piece &P = pieces[t];
    char word[kMaxWord];
    int len;
    int64_t begin = 0;
    // a piece that is not the last one ends right after a separator, so treating its end as end-of-file never
    // drops a word; the last piece has the real end-of-file semantics (ref :135-138,278-279)
    Reader r{c->map, cut[t + 1], cut[t]};
    while (next_token(r, word, len, begin)) {
      const uint64_t h = fnv1a(word, len);
      int32_t id = P.seen.hash.empty() ? -1 : P.seen.find(word, len, h);
      if (id < 0) { id = P.seen.add(word, len, h); P.cn.push_back(0); }
      P.cn[id]++;
      if ((P.raw.size() % kCheckpointEvery) == 0) {
        P.cp_raw.push_back((int64_t)P.raw.size());
        P.cp_byte.push_back(begin);
      }
      P.raw.push_back(id);
    }
  });
  lap("tokenise+count");
  // ---- merge in file order: walking the pieces in order and each piece's words in its own first-appearance
  // order reproduces the sequential first-appearance order, which is what breaks ties in SortVocab.
  StringMap seen;
  std::vector<int64_t> cn;
  seen.add("</s>", 4, fnv1a("</s>", 4));      // "</s>" is vocabulary entry 0 from the start (ref :276)
  cn.push_back(0);
  std::vector<std::vector<int32_t>> to_global(npieces);
  for (int t = 0; t < npieces; t++) {
    Piece &P = pieces[t];
    to_global[t].resize(P.cn.size());
    for (size_t i = 0; i < P.cn.size(); i++) {
      const char *w = P.seen.word((int32_t)i);
      const int len = (int)strlen(w);
      const uint64_t h = P.seen.hash[i];
      int32_t id = seen.find(w, len, h);
      if (id < 0) { id = seen.add(w, len, h); cn.push_back(0); }
      cn[id] += P.cn[i];
      to_global[t][i] = id;
    }
  }
  lap("merge");
  // ---- ReduceVocab (ref :245-263, called at :293 whenever `vocab_size > vocab_hash_size * 0.7`).  The number of
  // entries the sequential scan holds never exceeds the number of distinct words, so below that limit it never runs
  // (every corpus with fewer than 21 M distinct words).  Above it, what survives depends on the order of the stream:
  // replay the raw tokens once, sequentially, keeping per word its running count, whether it is in the table, and when
  // it (last) entered -- ReduceVocab compacts the array in place, so the survivors keep their order and a word that
  // comes back after having been removed is appended at the end with a count of 1.
  const int32_t nseen = (int32_t)cn.size();
  std::vector<int32_t> order;                 // the vocabulary array as SortVocab finds it (ids of `seen`)
  const double reduce_above = vocab_hash_size * 0.7;       // int times double, as the reference writes it
  if (!((double)nseen > reduce_above)) {
    order.resize(nseen);
    for (int32_t i = 0; i < nseen; i++) order[i] = i;       // first-appearance order, "</s>" first
  } else {
    std::vector<int64_t> run(nseen, 0), entered(nseen, -1);
    int64_t clock = 0, live = 1, min_reduce = 1;            // min_reduce: ref :48
    entered[0] = clock++;                                   // AddWordToVocab("</s>"), count 0 (ref :276)
    for (int t = 0; t < npieces; t++) {
      const std::vector<int32_t> &g = to_global[t];
      for (const int32_t r : pieces[t].raw) {
        const int32_t id = g[r];
        if (entered[id] < 0) { entered[id] = clock++; run[id] = 1; live++; }
        else run[id]++;
        if ((double)live > reduce_above) {                  // ReduceVocab: nothing is protected, not even "</s>"
          for (int32_t i = 0; i < nseen; i++)
            if (entered[i] >= 0 && !(run[i] > min_reduce)) { entered[i] = -1; run[i] = 0; live--; }
          min_reduce++;
        }
      }
    }
    for (int32_t i = 0; i < nseen; i++) if (entered[i] >= 0) order.push_back(i);
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return entered[a] < entered[b]; });
    cn.swap(run);                                           // the counts the reference ends up with
  }
  lap("reduce");
  // ---- SortVocab (ref :215-242): entry 0 ("</s>", unless ReduceVocab has removed it: then whatever word came
  // next) stays first, the rest by count descending.  glibc's qsort is a merge sort for arrays of this size, i.e.
  // ties keep their order in the array: stable_sort.
  if (order.size() > 1)
    std::stable_sort(order.begin() + 1, order.end(), [&](int32_t a, int32_t b) { return cn[a] > cn[b]; });
  std::vector<int32_t> remap(nseen, -1);
  for (size_t k = 0; k < order.size(); k++) {
    const int32_t id = order[k];
    if (cn[id] < min_count && k != 0) continue;          // ref :225
    remap[id] = (int32_t)c->words.size();
    c->words.emplace_back(seen.word(id));
    c->counts.push_back(cn[id]);
    c->train_words += cn[id];                             // ref :233
  }
  for (size_t i = 0; i < c->words.size(); i++)
    c->final_map.add(c->words[i].c_str(), (int)c->words[i].size(),
                     fnv1a(c->words[i].c_str(), (int)c->words[i].size()));
  lap("sort+vocab");
  // ---- pass 2, parallel, no second look at the text: private ids -> final ids, out-of-vocabulary words dropped
  // (ref :398), plus the sparse (byte, token index) index used by the shard arithmetic
  run_parallel([&](int t) {
    Piece &P = pieces[t];
    P.to_final.resize(P.cn.size());
    for (size_t i = 0; i < P.cn.size(); i++) P.to_final[i] = remap[to_global[t][i]];
    P.out.reserve(P.raw.size());
    size_t cp = 0;
    for (size_t k = 0; k < P.raw.size(); k++) {
      if (cp < P.cp_raw.size() && (int64_t)k == P.cp_raw[cp]) { P.cp_index.push_back((int64_t)P.out.size()); cp++; }
      const int32_t id = P.to_final[P.raw[k]];
      if (id >= 0) P.out.push_back(id);
    }
    std::vector<int32_t>().swap(P.raw);
  });
  lap("remap");
  std::vector<int64_t> base(npieces + 1, 0);
  for (int t = 0; t < npieces; t++) base[t + 1] = base[t] + (int64_t)pieces[t].out.size();
  c->tokens.resize((size_t)base[npieces]);
  run_parallel([&](int t) {
    if (!pieces[t].out.empty())
      memcpy(c->tokens.data() + base[t], pieces[t].out.data(), pieces[t].out.size() * sizeof(int32_t));
  });
  for (int t = 0; t < npieces; t++)
    for (size_t i = 0; i < pieces[t].cp_byte.size(); i++) {
      c->cp_byte.push_back(pieces[t].cp_byte[i]);
      c->cp_index.push_back(base[t] + pieces[t].cp_index[i]);
    }
  lap("concat");
  *out = c;
  return W2B_OK;
}

extern "C" int64_t w2b_corpus_vocab_size(const w2b_corpus *c) { return (int64_t)c->words.size(); }
extern "C" int64_t w2b_corpus_train_words(const w2b_corpus *c) { return c->train_words; }
extern "C" int64_t w2b_corpus_file_size(const w2b_corpus *c) { return c->file_size; }
extern "C" const char *w2b_corpus_word(const w2b_corpus *c, int64_t i) { return c->words[i].c_str(); }
extern "C" const int64_t *w2b_corpus_counts(const w2b_corpus *c) { return c->counts.data(); }
extern "C" int64_t w2b_corpus_num_tokens(const w2b_corpus *c) { return (int64_t)c->tokens.size(); }
extern "C" const int32_t *w2b_corpus_tokens(const w2b_corpus *c) { return c->tokens.data(); }

extern "C" int32_t w2b_corpus_search(const w2b_corpus *c, const char *w) {
  const int len = (int)strlen(w);
  return c->final_map.find(w, len, fnv1a(w, len));
}

extern "C" int w2b_corpus_shards(const w2b_corpus *c, int32_t num_threads, int64_t *starts,
                                 int32_t *first_override) {
  if (!c || num_threads < 1 || !starts) return W2B_EINVAL;
  char word[kMaxWord];
  int len;
  for (int32_t w = 0; w < num_threads; w++) {
    const int64_t off = c->file_size / (int64_t)num_threads * (int64_t)w;    // ref :377
    // walk whole tokens from the last checkpoint at or before `off`
    size_t cp = std::upper_bound(c->cp_byte.begin(), c->cp_byte.end(), off) - c->cp_byte.begin();
    int64_t index = 0, begin = 0, boundary = -1;
    Reader r{c->map, c->file_size, 0};
    if (cp > 0) { r.pos = c->cp_byte[cp - 1]; index = c->cp_index[cp - 1]; }
    while (true) {
      const int64_t save = r.pos;
      if (!next_token(r, word, len, begin)) { boundary = -1; break; }      // ran into end of file
      if (begin >= off) { boundary = begin; r.pos = save; break; }
      if (w2b_corpus_search(c, word) >= 0) index++;
    }
    starts[w] = index;
    int32_t ov = -2;
    // what the reference's reader sees first after the seek: either the whole token at `boundary`,
    // or the tail of a word that started before `off`.
    Reader q{c->map, c->file_size, off};
    int64_t qb = 0;
    if (next_token(q, word, len, qb)) {
      if (qb != boundary) ov = w2b_corpus_search(c, word);
    }
    if (first_override) first_override[w] = ov;
  }
  return W2B_OK;
}

// "%lf " of one float, exactly as glibc prints it (the value is exact in binary, printf rounds the exact decimal
// expansion half-to-even at 6 decimals), without going through printf: the text form of a cfg2 model is 320 M
// values (ref :571 calls fprintf once per value).  |x| < 2^31 is done in integer arithmetic: x = m * 2^e, so
// x * 10^6 = (m * 10^6) * 2^e with m * 10^6 < 2^44; everything else (huge, inf, nan) falls back to snprintf.
static inline char *format_lf(char *p, float x) {
  uint32_t bits;
  memcpy(&bits, &x, 4);
  const uint32_t ex = (bits >> 23) & 0xFF, frac = bits & 0x7FFFFF;
  if (ex >= 127 + 31) return p + snprintf(p, 64, "%lf ", (double)x);    // >= 2^31, inf, nan
  const uint64_t m = ex ? (uint64_t)(frac | 0x800000) : frac;               // denormals: no hidden bit
  const int e = (ex ? (int)ex : 1) - 127 - 23;                               // x = m * 2^e
  const uint64_t scaled = m * 1000000ull;
  uint64_t q;
  if (e >= 0) {
    q = scaled << e;
  } else if (-e >= 64) {
    q = 0;                                                                   // < 2^44 / 2^64: far below one half
  } else {
    const int sh = -e;
    q = scaled >> sh;
    const uint64_t rem = scaled & ((1ull << sh) - 1), halfway = 1ull << (sh - 1);
    if (rem > halfway || (rem == halfway && (q & 1))) q++;                   // round half to even
  }
  if (bits >> 31) *p++ = '-';                                                // also "-0.000000"
  const uint64_t ip = q / 1000000ull;
  uint32_t fp = (uint32_t)(q % 1000000ull);
  char tmp[24];
  int n = 0;
  uint64_t t = ip;
  do { tmp[n++] = (char)('0' + t % 10); t /= 10; } while (t);
  while (n) *p++ = tmp[--n];
  *p++ = '.';
  for (int i = 5; i >= 0; i--) { p[i] = (char)('0' + fp % 10); fp /= 10; }
  p += 6;
  *p++ = ' ';
  return p;
}

// the writer of ref :560-576 over any source of row names
template <typename NameOf>
static int write_vectors(const char *path, int64_t V, NameOf &&name_of, const float *values, int64_t dim, int32_t binary) {
  FILE *fo = fopen(path, "wb");
  if (!fo) return W2B_EIO;
  std::vector<char> big(1 << 20);          // stdio buffer of this call (released after fclose)
  setvbuf(fo, big.data(), _IOFBF, big.size());
  fprintf(fo, "%lld %lld\n", (long long)V, (long long)dim);
  std::vector<char> line;
  if (!binary) line.resize((size_t)dim * 64 + 64);
  for (int64_t a = 0; a < V; a++) {
    fputs(name_of(a), fo);
    fputc(' ', fo);
    const float *row = values + a * dim;
    if (binary) {
      fwrite(row, sizeof(float), (size_t)dim, fo);
    } else {                                                                 // ref :571  fprintf(fo, "%lf ", ...)
      char *p = line.data();
      for (int64_t b = 0; b < dim; b++) p = format_lf(p, row[b]);
      fwrite(line.data(), 1, (size_t)(p - line.data()), fo);
    }
    fputc('\n', fo);
  }
  return fclose(fo) == 0 ? W2B_OK : W2B_EIO;
}

extern "C" int w2b_save_vectors(const char *path, const w2b_corpus *c, const float *values, int64_t dim,
                                int32_t binary) {
  if (!path || !c || !values) return W2B_EINVAL;
  return write_vectors(path, (int64_t)c->words.size(), [&](int64_t a) { return c->words[(size_t)a].c_str(); }, values, dim, binary);
}

// ------------------------------------------------------------------------------------ bit-packed vectors
// (include/word2bits_corpus.h: layout and file format)
extern "C" int64_t w2b_packed_words_per_row(int64_t dim, int32_t bitlevel) {
  if (dim < 1 || (bitlevel != 1 && bitlevel != 2)) return -1;
  return (dim + 63) / 64 * bitlevel;
}

static inline uint32_t f32_bits(float x) { uint32_t b; memcpy(&b, &x, 4); return b; }
static inline float bits_f32(uint32_t b) { float x; memcpy(&x, &b, 4); return x; }
static const uint32_t kThird = 0x3EAAAAABu, kQuarter = 0x3E800000u, kThreeQuarters = 0x3F400000u;   // 1/3, .25, .75

extern "C" int w2b_pack_quantized(const float *values, int64_t rows, int64_t dim, int32_t bitlevel, uint64_t *out) {
  const int64_t wpr = w2b_packed_words_per_row(dim, bitlevel);
  if (!values || !out || rows < 0) return W2B_EINVAL;
  if (wpr < 0) return W2B_EUNSUPPORTED;
  for (int64_t r = 0; r < rows; r++) {
    const float *row = values + r * dim;
    uint64_t *o = out + r * wpr;
    for (int64_t w = 0; w < wpr; w++) o[w] = 0;
    for (int64_t c = 0; c < dim; c++) {
      const uint32_t b = f32_bits(row[c]), mag = b & 0x7FFFFFFFu;
      const uint64_t bit = 1ull << (c & 63);
      uint64_t *blk = o + (c >> 6) * bitlevel;
      if (bitlevel == 1) { if (mag != kThird) return W2B_EINVAL; }
      else if (mag == kThreeQuarters) blk[1] |= bit;
      else if (mag != kQuarter) return W2B_EINVAL;
      if (b >> 31) blk[0] |= bit;
    }
  }
  return W2B_OK;
}

extern "C" int w2b_unpack_quantized(const uint64_t *packed, int64_t rows, int64_t dim, int32_t bitlevel, float *out) {
  const int64_t wpr = w2b_packed_words_per_row(dim, bitlevel);
  if (!packed || !out || rows < 0) return W2B_EINVAL;
  if (wpr < 0) return W2B_EUNSUPPORTED;
  for (int64_t r = 0; r < rows; r++) {
    const uint64_t *in = packed + r * wpr;
    float *row = out + r * dim;
    for (int64_t c = 0; c < dim; c++) {
      const uint64_t *blk = in + (c >> 6) * bitlevel;
      const int sh = (int)(c & 63);
      const uint32_t mag = bitlevel == 1 ? kThird : (((blk[1] >> sh) & 1) ? kThreeQuarters : kQuarter);
      row[c] = bits_f32(mag | ((uint32_t)((blk[0] >> sh) & 1) << 31));
    }
  }
  return W2B_OK;
}

extern "C" int w2b_save_vectors_packed(const char *path, const w2b_corpus *c, const uint64_t *packed, int64_t dim,
                                       int32_t bitlevel) {
  const int64_t wpr = w2b_packed_words_per_row(dim, bitlevel);
  if (!path || !c || !packed) return W2B_EINVAL;
  if (wpr < 0) return W2B_EUNSUPPORTED;
  FILE *fo = fopen(path, "wb");
  if (!fo) return W2B_EIO;
  const int64_t V = (int64_t)c->words.size();
  fprintf(fo, "W2BP1 %lld %lld %d\n", (long long)V, (long long)dim, (int)bitlevel);
  for (int64_t a = 0; a < V; a++) { fputs(c->words[(size_t)a].c_str(), fo); fputc('\n', fo); }
  fwrite(packed, sizeof(uint64_t), (size_t)(V * wpr), fo);
  return fclose(fo) == 0 ? W2B_OK : W2B_EIO;
}

// A packed file in memory -> its vocabulary and unpacked rows.  Shared with the evaluator's loader (w2b_eval.cpp).
bool w2b_internal_is_packed(const unsigned char *d, size_t n) { return n >= 6 && !memcmp(d, "W2BP1 ", 6); }
int w2b_internal_parse_packed(const unsigned char *d, size_t n, std::vector<std::string> &words, std::vector<float> &values,
                              int64_t *dim_out) {
  if (!w2b_internal_is_packed(d, n)) return W2B_EINVAL;
  const unsigned char *nl = (const unsigned char *)memchr(d, '\n', n);
  if (!nl || nl - d > 100) return W2B_EIO;
  long long V = 0, D = 0;
  int bitlevel = 0;
  if (sscanf(std::string((const char *)d + 6, (size_t)(nl - d) - 6).c_str(), "%lld %lld %d", &V, &D, &bitlevel) != 3) return W2B_EIO;
  const int64_t wpr = w2b_packed_words_per_row(D, bitlevel);
  if (V < 0 || wpr < 0 || V > 0x7FFFFF00ll || D > (1 << 24)) return W2B_EIO;
  size_t pos = (size_t)(nl - d) + 1;
  // a damaged header must not size an allocation: every word takes at least its '\n', every row its words
  if ((unsigned long long)V > n - pos || (unsigned long long)V * (unsigned long long)wpr > (n - pos) / sizeof(uint64_t)) return W2B_EIO;
  words.clear();
  words.reserve((size_t)V);
  for (long long a = 0; a < V; a++) {
    const unsigned char *e = pos < n ? (const unsigned char *)memchr(d + pos, '\n', n - pos) : nullptr;
    if (!e) return W2B_EIO;
    words.emplace_back((const char *)d + pos, (size_t)(e - (d + pos)));
    pos = (size_t)(e - d) + 1;
  }
  if (n - pos < (size_t)(V * wpr) * sizeof(uint64_t)) return W2B_EIO;
  std::vector<uint64_t> packed((size_t)(V * wpr));
  if (!packed.empty()) memcpy(packed.data(), d + pos, packed.size() * sizeof(uint64_t));     // (unaligned in the file)
  values.assign((size_t)(V * D), 0.f);
  if (dim_out) *dim_out = D;
  return V > 0 ? w2b_unpack_quantized(packed.data(), V, D, bitlevel, values.data()) : W2B_OK;
}

extern "C" int w2b_unpack_vectors_file(const char *packed_path, const char *out_path, int32_t binary) {
  if (!packed_path || !out_path) return W2B_EINVAL;
  FILE *f = fopen(packed_path, "rb");
  if (!f) return W2B_EIO;
  std::vector<unsigned char> d;
  fseek(f, 0, SEEK_END);
  const long long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  d.resize(n > 0 ? (size_t)n : 0);
  const bool ok = n <= 0 || fread(d.data(), 1, (size_t)n, f) == (size_t)n;
  fclose(f);
  if (!ok) return W2B_EIO;
  std::vector<std::string> words;
  std::vector<float> values;
  int64_t dim = 0;
  if (int rc = w2b_internal_parse_packed(d.data(), d.size(), words, values, &dim)) return rc;
  return write_vectors(out_path, (int64_t)words.size(), [&](int64_t a) { return words[(size_t)a].c_str(); }, values.data(), dim, binary);
}
