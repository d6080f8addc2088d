// word2bits_main.cpp -- the ./word2bits command line on top of the HIP C ABI.
//
// Same flags, defaults, stdout lines and output-file format as the reference program
// (ref src/word2bits.cpp:579-621 main/ArgPos, :518-577 TrainModel) so that the reference's
// compute_accuracy evaluator runs unchanged on the result.  What differs is where the work
// happens: the corpus is tokenised once on the host (word2bits_corpus.h) and every epoch is a
// sequence of GPU launches in which each of the -threads Hogwild workers is one workgroup.
// GPU-only additions use new flag names: -gpus, -sync-every, -positions, -device, -table-size, -relaxed,
// -window-cache, -exact, -eval, -hot-rows, -hot-period, -row-desc, -atomic-rank, -atomic-cap, -row-groups, -refresh-rows, ...
#include <pthread.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/word2bits_corpus.h"
#include "../../include/word2bits_eval.h"
#include "../../include/word2bits_hip.h"

namespace {

struct Options {                       // defaults: ref src/word2bits.cpp:48-54,59
  std::string train_file, output_file;
  int binary = 0, debug_mode = 2, window = 5, min_count = 5, num_threads = 12, bitlevel = 1;
  long long layer1_size = 100, iter = 5, classes = 0;
  int save_every_epoch = 0, negative = 5;
  float alpha = 0.05f, sample = 1e-3f, reg = 0.f;
  // GPU-only
  int gpus = 1, device = 0;
  long long sync_every = 1;            // -gpus > 1: launches between two replica exchanges (8 until round 4: the interval is what
                                       // costs epoch loss -- tests/test_gpu_exchange.py; keep -positions short with replicas)
  long long positions = 4096;          // sentence positions per worker per launch
  long long table_size = W2B_UNIGRAM_TABLE_SIZE;
  int relaxed = 0;                     // 1: plain cached row accesses instead of agent-scope ones
  int window_cache = -1;               // -1 automatic, 0 plain worker kernel, 1 sentence-resident kernel
  int row_groups = -1;                 // -row-groups N: -1 automatic, 0 never, 1 the row-group worker kernel wherever it fits
  int exact = 0;                       // 1: serial dot product in the reference's order (bit parity at -threads 1)
  std::string eval_file;               // -eval FILE: questions to score on the GPU after the final save (-binary 1)
  int hot_rows = -1;                   // -hot-rows N: leading rows of v (and u) with per-XCD copies; -1 = from the counts
  int hot_rows_u = -1, hot_rows_v = -1; // -hot-rows-u / -hot-rows-v N: the same for one table only
  int hot_cap = -1;                    // -hot-cap N: most rows the automatic choice takes (-1 = default)
  int hot_period = 0;                  // -hot-period N: centre words between two merge events of a worker (0 = default)
  int atomic_rank = -2;                // -atomic-rank N: rows 1..N are updated with atomic adds (-1 automatic; default: library's)
  int window_refresh = -1;             // -window-refresh N: w2b_tuning.window_refresh (-1 = default)
  int hot_weight = 0;                  // -hot-weight N: w2b_tuning.hot_weight_permille (0 = default)
  int atomic_cap = -1;                 // -atomic-cap N: most rows the automatic choice takes
  int atomic_rank_u = 0;               // -atomic-rank-u N: w2b_tuning.atomic_rank_u (0 = as -atomic-rank, -1 = none)
  int fresh_rank_u = 0;                // -fresh-rank-u N: w2b_tuning.fresh_rank_u (0 = the library decides, -1 = none)
  int refresh_rows_u = 0;              // -refresh-rows N: w2b_tuning.refresh_rows_u (0 = the library decides, -1 = none)
  std::string packed_file;             // -packed FILE: also write the final vectors bit-packed (-bitlevel 1 / 2; word2bits_corpus.h)
  int row_desc = 0;                    // -row-desc 1: the row addressing of tables >= 2 GiB on any table (w2b_tuning.force_row_desc)
  long long sync_words = 0;            // -sync-words N: centre words per replica between two exchanges (0 = automatic, below); sets -positions / -sync-every
  int threads_literal = 0;             // -threads-literal 1: keep an explicit -threads N even where the library would rather fill the device
  int concurrent = 0;                  // -concurrent N: w2b_tuning.concurrent_workers (0 = the library decides)
  int xchg_rule = 0, xchg_tau_u = 0, xchg_tau_v = 0;   // -exchange-rule / -exchange-tau-u / -exchange-tau-v: w2b_tuning.exchange_* (replicas)
};

// w2b_config.plain_worker_kernel from -window-cache / -row-groups: 0 automatic, 1 plain, 2 sentence-resident, 3 row groups
int worker_kernel_choice(const Options &o) {
  if (o.window_cache > 0) return 2;
  if (o.row_groups > 0) return 3;
  if (o.window_cache == 0 || o.row_groups == 0) return 1;
  return 0;
}

// ArgPos, ref :579-589: exact-match search; a flag in last position has no value -> exit(1)
int arg_pos(const char *flag, int argc, char **argv) {
  for (int a = 1; a < argc; a++)
    if (!strcmp(flag, argv[a])) {
      if (a == argc - 1) {
        printf("Argument missing for %s\n", flag);
        exit(1);
      }
      return a;
    }
  return -1;
}

void die(const char *what, int rc) {
  fprintf(stderr, "word2bits: %s failed (%d): %s\n", what, rc, w2b_last_error());
  exit(2);
}
#define CK(call)                         \
  do {                                   \
    int rc_ = (call);                    \
    if (rc_ != W2B_OK) die(#call, rc_);  \
  } while (0)

struct Replica {                        // one GPU
  int index = 0;
  w2b_trainer *t = nullptr;
};

// -gpus N in one process: the collectives of the N communicators must be issued concurrently, one host thread per
// replica.  The threads live as long as the run (round 3 created and joined N threads per exchange) and take one
// command at a time: 1 = full exchange (w2b_sync_replicas mode 2), 0 = exit.
struct ExchangeCrew {
  struct Slot { pthread_t th; w2b_trainer *t; ExchangeCrew *crew; };
  std::vector<Slot> slots;
  pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
  pthread_cond_t cv = PTHREAD_COND_INITIALIZER, done_cv = PTHREAD_COND_INITIALIZER;
  long long generation = 0;
  int command = 0, remaining = 0;
  static void *run(void *p) {
    Slot *s = (Slot *)p;
    ExchangeCrew *c = s->crew;
    long long seen = 0;
    for (;;) {
      pthread_mutex_lock(&c->mu);
      while (c->generation == seen) pthread_cond_wait(&c->cv, &c->mu);
      seen = c->generation;
      const int cmd = c->command;
      pthread_mutex_unlock(&c->mu);
      if (cmd == 0) return nullptr;
      CK(w2b_sync_replicas(s->t, 2));
      pthread_mutex_lock(&c->mu);
      if (--c->remaining == 0) pthread_cond_signal(&c->done_cv);
      pthread_mutex_unlock(&c->mu);
    }
  }
  void start(std::vector<Replica> &reps) {
    slots.resize(reps.size());
    for (size_t g = 0; g < reps.size(); g++) {
      slots[g].t = reps[g].t;
      slots[g].crew = this;
      pthread_create(&slots[g].th, nullptr, run, &slots[g]);
    }
  }
  void issue(int cmd) {   // returns when every replica has issued its exchange
    pthread_mutex_lock(&mu);
    command = cmd;
    remaining = cmd == 0 ? 0 : (int)slots.size();
    generation++;
    pthread_cond_broadcast(&cv);
    while (remaining > 0) pthread_cond_wait(&done_cv, &mu);
    pthread_mutex_unlock(&mu);
  }
  void stop() {
    if (slots.empty()) return;
    issue(0);
    for (auto &s : slots) pthread_join(s.th, nullptr);
    slots.clear();
  }
};

void save(const Options &o, const w2b_corpus *c, w2b_trainer *t, const std::string &path) {
  const long long V = w2b_corpus_vocab_size(c), D = o.layer1_size;
  if (o.classes != 0) {                 // ref :542,562: nothing but the fopen/fclose happens
    FILE *f = fopen(path.c_str(), "wb");
    if (f) fclose(f);
    return;
  }
  std::vector<float> q((size_t)V * D);
  CK(w2b_export_quantized(t, q.data()));                 // quantize(u+v), ref :549-550,568-569
  int rc = w2b_save_vectors(path.c_str(), c, q.data(), D, o.binary);
  if (rc != W2B_OK) {
    fprintf(stderr, "word2bits: cannot write %s\n", path.c_str());
    exit(2);
  }
}

}  // namespace

int main(int argc, char **argv) {
  Options o;
  int i;
  // flag list and order: ref :596-611
  if ((i = arg_pos("-save-every-epoch", argc, argv)) > 0) o.save_every_epoch = atoi(argv[i + 1]);
  if ((i = arg_pos("-bitlevel", argc, argv)) > 0) o.bitlevel = atoi(argv[i + 1]);
  if ((i = arg_pos("-size", argc, argv)) > 0) o.layer1_size = atoi(argv[i + 1]);
  if ((i = arg_pos("-reg", argc, argv)) > 0) o.reg = (float)atof(argv[i + 1]);
  if ((i = arg_pos("-train", argc, argv)) > 0) o.train_file = argv[i + 1];
  if ((i = arg_pos("-debug", argc, argv)) > 0) o.debug_mode = atoi(argv[i + 1]);
  if ((i = arg_pos("-binary", argc, argv)) > 0) o.binary = atoi(argv[i + 1]);
  if ((i = arg_pos("-alpha", argc, argv)) > 0) o.alpha = (float)atof(argv[i + 1]);
  if ((i = arg_pos("-output", argc, argv)) > 0) o.output_file = argv[i + 1];
  if ((i = arg_pos("-window", argc, argv)) > 0) o.window = atoi(argv[i + 1]);
  if ((i = arg_pos("-sample", argc, argv)) > 0) o.sample = (float)atof(argv[i + 1]);
  if ((i = arg_pos("-negative", argc, argv)) > 0) o.negative = atoi(argv[i + 1]);
  if ((i = arg_pos("-threads", argc, argv)) > 0) o.num_threads = atoi(argv[i + 1]);
  if ((i = arg_pos("-iter", argc, argv)) > 0) o.iter = atoi(argv[i + 1]);
  if ((i = arg_pos("-min-count", argc, argv)) > 0) o.min_count = atoi(argv[i + 1]);
  if ((i = arg_pos("-classes", argc, argv)) > 0) o.classes = atoi(argv[i + 1]);
  // GPU-only flags (new names; the reference ignores unknown flags, so scripts stay portable)
  if ((i = arg_pos("-gpus", argc, argv)) > 0) o.gpus = atoi(argv[i + 1]);
  if ((i = arg_pos("-device", argc, argv)) > 0) o.device = atoi(argv[i + 1]);
  if ((i = arg_pos("-sync-every", argc, argv)) > 0) o.sync_every = atoll(argv[i + 1]);
  if ((i = arg_pos("-positions", argc, argv)) > 0) o.positions = atoll(argv[i + 1]);
  if ((i = arg_pos("-table-size", argc, argv)) > 0) o.table_size = atoll(argv[i + 1]);
  if ((i = arg_pos("-relaxed", argc, argv)) > 0) o.relaxed = atoi(argv[i + 1]);
  if ((i = arg_pos("-window-cache", argc, argv)) > 0) o.window_cache = atoi(argv[i + 1]);
  if ((i = arg_pos("-row-groups", argc, argv)) > 0) o.row_groups = atoi(argv[i + 1]);
  if ((i = arg_pos("-exact", argc, argv)) > 0) o.exact = atoi(argv[i + 1]);
  if ((i = arg_pos("-eval", argc, argv)) > 0) o.eval_file = argv[i + 1];
  if ((i = arg_pos("-packed", argc, argv)) > 0) o.packed_file = argv[i + 1];
  if ((i = arg_pos("-hot-rows", argc, argv)) > 0) o.hot_rows = atoi(argv[i + 1]);
  if ((i = arg_pos("-hot-rows-u", argc, argv)) > 0) o.hot_rows_u = atoi(argv[i + 1]);
  if ((i = arg_pos("-hot-rows-v", argc, argv)) > 0) o.hot_rows_v = atoi(argv[i + 1]);
  if ((i = arg_pos("-hot-period", argc, argv)) > 0) o.hot_period = atoi(argv[i + 1]);
  if ((i = arg_pos("-hot-cap", argc, argv)) > 0) o.hot_cap = atoi(argv[i + 1]);
  if ((i = arg_pos("-row-desc", argc, argv)) > 0) o.row_desc = atoi(argv[i + 1]);
  if ((i = arg_pos("-atomic-rank", argc, argv)) > 0) o.atomic_rank = atoi(argv[i + 1]);
  if ((i = arg_pos("-atomic-cap", argc, argv)) > 0) o.atomic_cap = atoi(argv[i + 1]);
  if ((i = arg_pos("-atomic-rank-u", argc, argv)) > 0) o.atomic_rank_u = atoi(argv[i + 1]);
  if ((i = arg_pos("-fresh-rank-u", argc, argv)) > 0) o.fresh_rank_u = atoi(argv[i + 1]);
  if ((i = arg_pos("-refresh-rows", argc, argv)) > 0) o.refresh_rows_u = atoi(argv[i + 1]);
  if ((i = arg_pos("-hot-weight", argc, argv)) > 0) o.hot_weight = atoi(argv[i + 1]);
  if ((i = arg_pos("-window-refresh", argc, argv)) > 0) o.window_refresh = atoi(argv[i + 1]);
  if ((i = arg_pos("-threads-literal", argc, argv)) > 0) o.threads_literal = atoi(argv[i + 1]);
  if ((i = arg_pos("-sync-words", argc, argv)) > 0) o.sync_words = atoll(argv[i + 1]);
  if ((i = arg_pos("-concurrent", argc, argv)) > 0) o.concurrent = atoi(argv[i + 1]);
  if ((i = arg_pos("-exchange-rule", argc, argv)) > 0) o.xchg_rule = atoi(argv[i + 1]);
  if ((i = arg_pos("-exchange-tau-u", argc, argv)) > 0) o.xchg_tau_u = atoi(argv[i + 1]);
  if ((i = arg_pos("-exchange-tau-v", argc, argv)) > 0) o.xchg_tau_v = atoi(argv[i + 1]);

  // ---- TrainModel, ref :518-577
  printf("Starting training using file %s\n", o.train_file.c_str());
  w2b_corpus *corpus = nullptr;
  if (w2b_corpus_load(o.train_file.c_str(), o.min_count, &corpus) != W2B_OK) {
    printf("ERROR: training data file not found!\n");      // ref :272-273
    exit(1);
  }
  const long long V = w2b_corpus_vocab_size(corpus);
  const long long train_words = w2b_corpus_train_words(corpus);
  if (o.debug_mode > 0) {                                    // ref :295-298
    printf("Vocab size: %lld\n", V);
    printf("Words in train file: %lld\n", train_words);
  }
  if (o.output_file.empty()) return 0;                       // ref :527
  if (!o.packed_file.empty() && w2b_packed_words_per_row(o.layer1_size, o.bitlevel) < 0) {
    fprintf(stderr, "word2bits: -packed needs -bitlevel 1 or 2\n");       // said before the training, not after it
    return 2;
  }

  // A worker re-computes alpha only after more than 10000 of its own words (ref :379-393).  The reference has the same
  // property, but nobody starts it with hundreds of threads on a small file; a GPU invites exactly that.
  if (o.num_threads > 32 && train_words / o.num_threads < 20000)      // (worker counts no CPU run would use)
    fprintf(stderr, "word2bits: warning: -threads %d leaves %lld words per worker and epoch; with fewer than 20000 the learning "
                    "rate schedule (re-computed per worker every 10000 words) hardly runs, and -threads 0 keeps 50000: it picks at most "
                    "%lld workers for this file\n", o.num_threads, train_words / o.num_threads,
            train_words / 50000 > 1 ? train_words / 50000 : 1);
  int ndev = w2b_device_count();
  if (ndev <= 0) {
    fprintf(stderr, "word2bits: no HIP device visible; this build has no CPU path\n");
    return 2;
  }
  if (o.gpus < 1) o.gpus = 1;
  if (o.gpus > ndev) {
    fprintf(stderr, "word2bits: -gpus %d requested but %d visible\n", o.gpus, ndev);
    return 2;
  }
  {
    // What the library itself would run for this file (-threads 0): a probe trainer with the real vocabulary and its counts --
    // which worker kernel runs, and with it how many workers fill the device, depends on how much of the text the most
    // frequent words are.
    w2b_config probe_cfg;
    memset(&probe_cfg, 0, sizeof probe_cfg);
    probe_cfg.train_words = w2b_corpus_train_words(corpus);   // (with total_threads = replicas below: caps workers on small corpora)
    probe_cfg.total_threads = o.gpus;
    probe_cfg.vocab_size = V; probe_cfg.layer1_size = (int32_t)o.layer1_size; probe_cfg.window = o.window;
    probe_cfg.sample = o.sample;
    probe_cfg.negative = o.negative; probe_cfg.bitlevel = o.bitlevel; probe_cfg.num_threads = 1;
    probe_cfg.alpha = o.alpha; probe_cfg.compute_loss = 1; probe_cfg.device = o.device;
    probe_cfg.relaxed_coherence = o.relaxed;
    probe_cfg.plain_worker_kernel = worker_kernel_choice(o);
    probe_cfg.exact_reduction = o.exact;
    w2b_trainer *probe = nullptr;
    int32_t per_gpu = 1024;
    CK(w2b_trainer_create(&probe_cfg, &probe));
    CK(w2b_set_vocab_counts(probe, w2b_corpus_counts(corpus), 0));
    CK(w2b_suggested_threads(probe, &per_gpu));               // workgroups resident at once on one GPU
    w2b_trainer_destroy(probe);
    const int32_t cus = w2b_device_compute_units(o.device);
    bool picked = false;                                      // the library chose the number of workers: say which
    if (o.num_threads < 1) {                                  // GPU extension: -threads 0 = what the library picks
      o.num_threads = per_gpu * o.gpus;
      picked = true;
    } else if (cus > 0 && !o.exact && !o.relaxed) {
      // An explicit count BETWEEN the reference's own scale and a full device, on a corpus that could fill the device.  In
      // that range every row is shared by all workers (lossless context rows), and on a long stream that mode drifts with the
      // worker count: BASELINE configs[1] literally (100 M tokens) ends -0.5 / -0.8 / -3.6 / -11 % off the reference's epoch
      // loss at 64 / 128 / 256 / 512 workers, where the full-device mode (per-XCD copies, consensus merges) is -0.6 %
      // (profiles/r05_sessions/r05o_long_stream.txt).  The reference's -threads is a speed knob -- its own results do not
      // depend on it (0.3 % between 1 and 256 threads) -- so the faithful reading of `-threads 256` on such a corpus is "train
      // this file", not "use the mode that drifts": the library's own choice runs, with a notice (round 5 only warned).
      // -threads-literal 1 keeps the count.  Thresholds from the device (advisor, round 5): a full device is 3 workgroups per
      // compute unit; the drift passes the 1.5 % gate between 128 and 256 workers (a quarter of that and more is taken).
      w2b_row_plan plan_n, plan_auto;
      const int32_t n_per = o.num_threads / o.gpus > 0 ? o.num_threads / o.gpus : 1;
      CK(w2b_plan_rows(&probe_cfg, nullptr, w2b_corpus_counts(corpus), cus, n_per, &plan_n));
      CK(w2b_plan_rows(&probe_cfg, nullptr, w2b_corpus_counts(corpus), cus, per_gpu, &plan_auto));
      // (not where the row-group kernel runs the explicit count: rows of at most 512 floats -- round 6 checked that kernel on 100 M-token
      // streams at 256 workers: +0.1 % at -size 200, -0.6 % at -size 400 / 2 bits, tests/test_gpu_fidelity.py)
      if (!plan_n.full_device && plan_auto.full_device && 4 * n_per >= 3 * cus && !plan_n.row_group_kernel) {
        if (o.threads_literal)
          fprintf(stderr, "word2bits: warning: -threads %d on %lld words: below a full device all workers share every row, which "
                          "drifts on long streams (measured -3.6 %% of the reference's epoch loss at 256 workers on a 100 M-token "
                          "stream); -threads 0 fills the device for this file and stays within 1 %%\n", o.num_threads, train_words);
        else {
          fprintf(stderr, "word2bits: notice: -threads %d on %lld words is between the reference's scale and a full device, where "
                          "shared rows drift on long streams; running %d workers (what -threads 0 picks for this file; "
                          "-threads-literal 1 keeps %d)\n", o.num_threads, train_words, per_gpu * o.gpus, o.num_threads);
          o.num_threads = per_gpu * o.gpus;
          picked = true;
        }
      }
    }
    if (o.debug_mode > 0 && picked)
      printf("Hogwild workers (workgroups): %d\n", o.num_threads);
  }
  if (o.gpus > 1 && o.num_threads % o.gpus != 0) {
    fprintf(stderr, "word2bits: -threads must be a multiple of -gpus\n");
    return 2;
  }

  // shards of all workers (ref :377) -- worker ids are global across GPUs
  std::vector<int64_t> starts(o.num_threads);
  std::vector<int32_t> overrides(o.num_threads);
  CK(w2b_corpus_shards(corpus, o.num_threads, starts.data(), overrides.data()));

  const int per_gpu = o.num_threads / o.gpus;
  if (o.gpus > 1 && (o.sync_words > 0 || arg_pos("-sync-every", argc, argv) <= 0)) {
    // The interval between two exchanges, in centre words per replica.  What it costs is measured: while
    // a replica trains alone it misses what the others learn -- with a PERFECT combination rule 8 replicas end 1.5 / 4.9 / 16.8 %
    // off the single replica on the 22 M-token file at 16 K / 131 K / 1 M words per replica -- while the shipped rule errs the other
    // way at short intervals (+5 % at 131 K words on the literal 100 M-token stream).  Where the two meet, 8 replicas against one,
    // as a fraction of a replica's words per epoch: 1/34 (22 M tokens at the configs[1] shape: +0.8 / -2.7 / -8 % at 65 K / 131 K /
    // 262 K words), 1/23 (100 M tokens: +5.0 / +2.0 / +0.2 / -2.1 % at 131 K / 262 K / 524 K / 1 M), 1/26 (heldout_v1m, 85 M words:
    // +2.5 / -0.3 / -7.2 % at 221 K / 442 K / 1 M) and below 1/66 (text8-sized corpus, 32 workers per replica: -1.8 / -4.2 / -8.4 %
    // at 32 K / 65 K / 131 K).  Automatic: 1 / 32 of the replica's words per epoch, between 32 768 words and the 1 048 576 at which
    // one full exchange of a 2.56 GB model per launch fits the xGMI links; -sync-words N sets it, -sync-every / -positions keep
    // their old meaning when given.
    long long words = o.sync_words > 0 ? o.sync_words : w2b_suggested_exchange_words(train_words, o.gpus);
    long long pos = words / (per_gpu > 0 ? per_gpu : 1);
    if (pos < 16) pos = 16;
    if (arg_pos("-positions", argc, argv) > 0 && o.positions < pos) pos = o.positions;     // (an explicit, shorter launch is kept)
    o.positions = pos;
    o.sync_every = words / (pos * (per_gpu > 0 ? per_gpu : 1));
    if (o.sync_every < 1) o.sync_every = 1;
    if (o.debug_mode > 0)
      printf("Replica exchange: every %lld launches of %lld positions (%lld centre words per replica)\n", o.sync_every, o.positions,
             o.sync_every * o.positions * per_gpu);
  }
  std::vector<Replica> reps(o.gpus);
  char uid[W2B_UNIQUE_ID_BYTES];
  if (o.gpus > 1) CK(w2b_comm_unique_id(uid));
  struct InitArg { const Options *o; Replica *r; const w2b_corpus *c; const int64_t *st; const int32_t *ov;
                   int per_gpu; long long V, tw; const char *uid; };
  auto init_replica = [](void *p) -> void * {
    InitArg *a = (InitArg *)p;
    const Options &o = *a->o;
    w2b_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.vocab_size = a->V;
    cfg.train_words = a->tw;
    cfg.iter = o.iter;
    cfg.layer1_size = (int32_t)o.layer1_size;
    cfg.window = o.window;
    cfg.negative = o.negative;
    cfg.bitlevel = o.bitlevel;
    cfg.num_threads = a->per_gpu;
    cfg.alpha = o.alpha;
    cfg.sample = o.sample;
    cfg.reg = o.reg;
    cfg.compute_loss = 1;
    cfg.device = o.device + a->r->index;
    cfg.worker_offset = a->r->index * a->per_gpu;             // worker_offset: global id of local worker 0
    cfg.relaxed_coherence = o.relaxed;
    cfg.plain_worker_kernel = worker_kernel_choice(o);
    cfg.exact_reduction = o.exact;
    cfg.total_threads = o.num_threads;                  // total_threads across all GPUs
    CK(w2b_trainer_create(&cfg, &a->r->t));
    if (o.hot_rows >= 0 || o.hot_rows_u >= 0 || o.hot_rows_v >= 0 || o.hot_cap >= 0 || o.hot_period > 0 || o.row_desc || o.atomic_rank >= -1 || o.atomic_cap >= 0 || o.hot_weight > 0 || o.window_refresh >= 0 || o.atomic_rank_u != 0 || o.fresh_rank_u != 0 || o.refresh_rows_u != 0 ||
        o.concurrent > 0 || o.xchg_rule > 0 || o.xchg_tau_u > 0 || o.xchg_tau_v > 0) {
      w2b_tuning tn;
      CK(w2b_get_tuning(a->r->t, &tn));
      if (o.hot_rows >= 0) tn.hot_rows_v = tn.hot_rows_u = o.hot_rows;
      if (o.hot_rows_u >= 0) tn.hot_rows_u = o.hot_rows_u;
      if (o.hot_rows_v >= 0) tn.hot_rows_v = o.hot_rows_v;
      if (o.hot_period > 0) tn.hot_period = o.hot_period;
      if (o.hot_cap >= 0) tn.hot_cap = o.hot_cap;
      tn.force_row_desc = o.row_desc ? 1 : 0;
      if (o.atomic_rank >= -1) tn.atomic_rank = o.atomic_rank;
      if (o.atomic_cap >= 0) tn.atomic_cap = o.atomic_cap;
      if (o.hot_weight > 0) tn.hot_weight_permille = o.hot_weight;
      if (o.window_refresh >= 0) tn.window_refresh = o.window_refresh;
      if (o.atomic_rank_u != 0) tn.atomic_rank_u = o.atomic_rank_u;
      if (o.fresh_rank_u != 0) tn.fresh_rank_u = o.fresh_rank_u;
      if (o.refresh_rows_u != 0) tn.refresh_rows_u = o.refresh_rows_u;
      if (o.concurrent > 0) tn.concurrent_workers = o.concurrent;
      if (o.xchg_rule > 0) tn.exchange_rule = o.xchg_rule;
      if (o.xchg_tau_u > 0) tn.exchange_tau_u = o.xchg_tau_u;
      if (o.xchg_tau_v > 0) tn.exchange_tau_v = o.xchg_tau_v;
      CK(w2b_set_tuning(a->r->t, &tn));
    }
    CK(w2b_init_net(a->r->t));                              // ref :528
    CK(w2b_set_vocab_counts(a->r->t, w2b_corpus_counts(a->c), o.negative > 0 ? o.table_size : 0));  // ref :529
    if (o.gpus == 1) {
      CK(w2b_set_corpus(a->r->t, w2b_corpus_tokens(a->c), w2b_corpus_num_tokens(a->c)));
      CK(w2b_set_shards(a->r->t, a->st, a->ov));
    } else {
      // Replicas: only the tokens this replica's workers read go to its GPU (the reference's threads share one file and
      // read file_size / num_threads bytes each, ref :377,414).  A worker starts at its shard start and stops after the
      // sentence in which its word count passes train_words / num_threads -- which may lie beyond the next shard's
      // start -- so the slice ends where the last of them can stop: quota + 1 tokens, then on to the next "</s>", at
      // most 64000 tokens further (a sentence is 1000 KEPT tokens; sub-sampling may drop many in between).  Should
      // that ever be too short the library reports it (w2b_set_corpus_slice) instead of ending the shard early.
      const int32_t *tok = w2b_corpus_tokens(a->c);
      const int64_t n = w2b_corpus_num_tokens(a->c);
      const int64_t quota = a->tw / o.num_threads;
      const int64_t *st = a->st + a->r->index * a->per_gpu;
      int64_t lo = n, hi = 0;
      for (int w = 0; w < a->per_gpu; w++) {
        if (st[w] < lo) lo = st[w];
        int64_t e = st[w] + quota + 2;
        const int64_t cap = e + 64000;
        while (e < n && e < cap && tok[e - 1] != 0) e++;
        if (e > n) e = n;
        if (e > hi) hi = e;
      }
      if (lo > hi) lo = hi;
      std::vector<int64_t> rel(a->per_gpu);
      for (int w = 0; w < a->per_gpu; w++) rel[w] = st[w] - lo;
      CK(w2b_set_corpus_slice(a->r->t, tok + lo, hi - lo, hi < n));
      CK(w2b_set_shards(a->r->t, rel.data(), a->ov + a->r->index * a->per_gpu));
      if (o.debug_mode > 1) fprintf(stderr, "word2bits: replica %d holds tokens [%lld, %lld) of %lld\n", a->r->index,
                                    (long long)lo, (long long)hi, (long long)n);
    }
    if (o.gpus > 1) CK(w2b_comm_init(a->r->t, o.gpus, a->r->index, a->uid));
    return nullptr;
  };
  {
    std::vector<InitArg> args(o.gpus);
    std::vector<pthread_t> th(o.gpus);
    for (int g = 0; g < o.gpus; g++) {
      reps[g].index = g;
      args[g] = InitArg{&o, &reps[g], corpus, starts.data(), overrides.data(), per_gpu, V, train_words, uid};
      if (o.gpus == 1) init_replica(&args[g]);
      else pthread_create(&th[g], nullptr, init_replica, &args[g]);   // RCCL init must run concurrently
    }
    if (o.gpus > 1) for (int g = 0; g < o.gpus; g++) pthread_join(th[g], nullptr);
  }

  ExchangeCrew crew;
  if (o.gpus > 1) crew.start(reps);
  const auto t_start = std::chrono::steady_clock::now();
  for (int iteration = 0; iteration < o.iter; iteration++) {
    printf("Starting epoch: %d\n", iteration);                // ref :533
    for (auto &r : reps) CK(w2b_epoch_begin(r.t));             // pthread_create, ref :535
    double last_loss = 0, epoch_loss = 0;
    bool finished = false;
    long long launches = 0;
    // The host stays one launch behind the device (w2b_epoch_poll with lag 1 waits for the previous launch only, the
    // next one is already queued; the launch after the last one finds every worker done and returns at once).  With
    // replicas the exchange runs on streams of its own next to the following launches (w2b_sync_replicas returns at
    // once); the end of an epoch (w2b_epoch_status below) waits for it.
    const int lag = 1;
    while (!finished) {
      for (auto &r : reps) CK(w2b_train_step(r.t, o.positions));
      finished = true;
      long long wca = 0;
      float alpha = 0;
      epoch_loss = 0;
      for (auto &r : reps) {
        int32_t fin = 0;
        int64_t w = 0;
        float a = 0;
        double l = 0;
        CK(w2b_epoch_poll(r.t, lag, &fin, &w, &a, &l));
        finished = finished && fin;
        wca += w;
        alpha = a;
        epoch_loss += l;
      }
      launches++;
      if (o.gpus > 1) {
        // replicas: every sync_every launches (and always at the end of an epoch) an all-reduce of the deltas of the whole
        // [u||v] over RCCL, every row's sum shared among the replicas that trained it (w2b_sync_replicas mode 2, asynchronous)
        const long long every = o.sync_every > 0 ? o.sync_every : 1;
        if (finished || launches % every == 0) crew.issue(1);
      }
      if (o.debug_mode > 1) {                                 // progress line, ref :384-387
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        printf("%cAlpha: %f  Progress: %.2f%%  Cost: %f Words/thread/sec: %.2fk  ", 13, alpha,
               wca / (float)(o.iter * train_words + 1) * 100, epoch_loss - last_loss,
               wca / (secs + 1e-9) / 1000.0 / o.num_threads);
        fflush(stdout);
        last_loss = epoch_loss;
      }
    }
    epoch_loss = 0;                                           // the epoch is over: the workers' totals, in worker order (ref :537-538)
    for (auto &r : reps) {
      int32_t fin = 0;
      double l = 0;
      CK(w2b_epoch_status(r.t, &fin, nullptr, nullptr, &l));
      epoch_loss += l;
    }
    printf("Epoch Loss: %lf\n", epoch_loss);                  // ref :539
    if (o.save_every_epoch && o.classes == 0) {               // ref :540-542 (the per-epoch file only when classes == 0)
      char name[8192];
      snprintf(name, sizeof name, "%s_epoch%d", o.output_file.c_str(), iteration);
      save(o, corpus, reps[0].t, name);
    }
  }
  crew.stop();
  save(o, corpus, reps[0].t, o.output_file);                  // ref :560-576
  if (!o.packed_file.empty() && o.classes == 0) {
    // the same vectors at 1 (2) bits per value instead of 32: packed on the device, 1/32 (1/16) of the bytes cross the bus
    const int64_t wpr = w2b_packed_words_per_row(o.layer1_size, o.bitlevel);
    if (wpr < 0) {
      fprintf(stderr, "word2bits: -packed needs -bitlevel 1 or 2\n");
      return 2;
    }
    std::vector<uint64_t> pk((size_t)(V * wpr));
    CK(w2b_export_packed(reps[0].t, pk.data()));
    if (w2b_save_vectors_packed(o.packed_file.c_str(), corpus, pk.data(), o.layer1_size, o.bitlevel) != W2B_OK) {
      fprintf(stderr, "word2bits: cannot write %s\n", o.packed_file.c_str());
      return 2;
    }
  }
  int rc_eval = 0;
  if (!o.eval_file.empty()) {
    // what `compute_accuracy <output> 0 0 < FILE` prints for the vectors just saved (binary format), scored on the GPU
    // straight from the live model: no file round trip
    FILE *qf = fopen(o.eval_file.c_str(), "rb");
    if (!qf) {
      fprintf(stderr, "word2bits: cannot open %s\n", o.eval_file.c_str());
      rc_eval = 1;
    } else {
      std::string qs;
      char buf[1 << 16];
      size_t n;
      while ((n = fread(buf, 1, sizeof buf, qf)) > 0) qs.append(buf, n);
      fclose(qf);
      std::vector<const char *> names((size_t)V);
      for (long long a = 0; a < V; a++) names[(size_t)a] = w2b_corpus_word(corpus, a);
      w2b_eval *ev = nullptr;
      char *txt = nullptr;
      int64_t len = 0;
      if (w2b_eval_from_trainer(reps[0].t, V, names.data(), 0, 0, 1, &ev) != W2B_OK ||
          w2b_eval_transcript(ev, qs.data(), (int64_t)qs.size(), &txt, &len) != W2B_OK) {
        fprintf(stderr, "word2bits: -eval failed: %s\n", w2b_last_error());
        rc_eval = 1;
      } else {
        fwrite(txt, 1, (size_t)len, stdout);
        w2b_eval_free_text(txt);
      }
      if (ev) w2b_eval_free(ev);
    }
  }
  for (auto &r : reps) w2b_trainer_destroy(r.t);
  w2b_corpus_free(corpus);
  return rc_eval;
}
