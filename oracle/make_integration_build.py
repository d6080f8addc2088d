#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: builds oracle/_ref/word2bits_hipseam -- the reference program with ONLY its thread fan-out
replaced by calls into libword2bits_hip.so, exactly as INTEGRATION.md tells a maintainer of the reference to do it.

The patch is applied to a scratch copy of /root/reference/src/word2bits.cpp written to oracle/_ref/ (git-ignored, never
committed: this repository contains no reference source); everything else of the reference -- main(), flag parsing,
LearnVocabFromTrainFile, SortVocab, the save loops, the stdout lines -- is compiled as it stands.  What it proves
(tests/test_gpu_integration.py): the C ABI of include/word2bits_hip.h is a sufficient seam -- with `W2B_SEAM_EXACT=1
-threads 1` the patched reference writes byte for byte the files of the unmodified reference (tests/golden/*.vec).

Edits (anchors are lines of the reference; the script fails loudly if one is not found exactly once):
  1. after the includes          : #include of the two C-ABI headers
  2. TrainModel(), ref :528-536  : InitNet / InitUnigramTable / pthread_create / pthread_join  ->  w2b_* calls; the token
                                   stream and the per-thread shard starts (ref :377) come from word2bits_corpus.h, after
                                   checking that its vocabulary is the reference's own (same size, same words, same counts)
  3. ref :537-538                : thread_losses sum -> the loss returned by w2b_epoch_status
  4. both save loops, ref :549-550 and :568-569 : u+v / quantize  ->  the exported quantize(u+v) table
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("REF", "/root/reference")


def replace_once(src, old, new, count=1):
    n = src.count(old)
    if n != count:
        raise SystemExit("make_integration_build: anchor %r found %d times (expected %d)" % (old[:60], n, count))
    return src.replace(old, new)


HEADER = '''#include <pthread.h>
#include <vector>
#include "word2bits_hip.h"
#include "word2bits_corpus.h"
static w2b_trainer *w2b_seam_t = NULL;
static std::vector<float> w2b_seam_q;          // quantize(u+v), filled before each save loop
#define W2B_SEAM_CK(call) do { int rc_ = (call); if (rc_ != 0) { printf("%s failed (%d): %s\\n", #call, rc_, w2b_last_error()); exit(1); } } while (0)
'''

SETUP = '''  // ---- HIP seam: what InitNet / InitUnigramTable / the thread fan-out did now happens behind the C ABI
  {
    w2b_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.vocab_size = vocab_size;   cfg.train_words = train_words;   cfg.iter = iter;
    cfg.layer1_size = layer1_size; cfg.window = window;             cfg.negative = negative;
    cfg.bitlevel = bitlevel;       cfg.num_threads = num_threads;   cfg.alpha = starting_alpha;
    cfg.sample = sample;           cfg.reg = reg;                   cfg.compute_loss = 1;   cfg.device = 0;
    cfg.exact_reduction = getenv("W2B_SEAM_EXACT") ? atoi(getenv("W2B_SEAM_EXACT")) : 0;
    W2B_SEAM_CK(w2b_trainer_create(&cfg, &w2b_seam_t));
    W2B_SEAM_CK(w2b_init_net(w2b_seam_t));                                   // InitNet(): same LCG, same bits
    std::vector<int64_t> cn(vocab_size);
    for (long long i = 0; i < vocab_size; i++) cn[i] = vocab[i].cn;
    W2B_SEAM_CK(w2b_set_vocab_counts(w2b_seam_t, cn.data(), negative > 0 ? table_size : 0));   // InitUnigramTable()
    // token stream + shard starts (ref :377): the library's ingest must have learnt the reference's own vocabulary
    w2b_corpus *corpus = NULL;
    W2B_SEAM_CK(w2b_corpus_load(train_file, min_count, &corpus));
    if (w2b_corpus_vocab_size(corpus) != vocab_size || w2b_corpus_train_words(corpus) != train_words) {
      printf("HIP seam: vocabulary mismatch (%lld vs %lld words)\\n", (long long)w2b_corpus_vocab_size(corpus), vocab_size);
      exit(1);
    }
    for (long long i = 0; i < vocab_size; i++)
      if (strcmp(w2b_corpus_word(corpus, i), vocab[i].word) || w2b_corpus_counts(corpus)[i] != vocab[i].cn) {
        printf("HIP seam: vocabulary row %lld differs\\n", i);
        exit(1);
      }
    std::vector<int64_t> starts(num_threads);
    std::vector<int32_t> first(num_threads);
    W2B_SEAM_CK(w2b_corpus_shards(corpus, num_threads, starts.data(), first.data()));
    W2B_SEAM_CK(w2b_set_corpus(w2b_seam_t, w2b_corpus_tokens(corpus), w2b_corpus_num_tokens(corpus)));
    W2B_SEAM_CK(w2b_set_shards(w2b_seam_t, starts.data(), first.data()));
    w2b_corpus_free(corpus);
  }
'''

EPOCH = '''    double total_loss_epoch = 0;
    {
      W2B_SEAM_CK(w2b_epoch_begin(w2b_seam_t));                              // pthread_create x num_threads
      int32_t done = 0; int64_t wca = 0; float a_now = 0;
      while (!done) {                                                        // pthread_join
        W2B_SEAM_CK(w2b_train_step(w2b_seam_t, 4096));
        W2B_SEAM_CK(w2b_epoch_status(w2b_seam_t, &done, &wca, &a_now, &total_loss_epoch));
      }
    }
'''

EXPORT = '''  w2b_seam_q.resize((size_t)vocab_size * layer1_size);
  if (classes == 0) W2B_SEAM_CK(w2b_export_quantized(w2b_seam_t, w2b_seam_q.data()));
'''


def main():
    src_path = os.path.join(REF, "src", "word2bits.cpp")
    if not os.path.exists(src_path):
        print("make_integration_build: %s not present, nothing built" % src_path)
        return 0
    out_dir = os.path.join(HERE, "_ref")
    os.makedirs(out_dir, exist_ok=True)
    s = open(src_path).read()
    s = replace_once(s, "#include <pthread.h>\n", HEADER)
    s = replace_once(s, "  InitNet();\n  if (negative > 0) InitUnigramTable();\n", SETUP)
    s = replace_once(s, "    memset(thread_losses, 0, sizeof(double) * num_threads);\n"
                        "    for (a = 0; a < num_threads; a++) pthread_create(&pt[a], NULL, TrainModelThread, (void *)a);\n"
                        "    for (a = 0; a < num_threads; a++) pthread_join(pt[a], NULL);\n"
                        "    double total_loss_epoch = 0;\n"
                        "    for (a = 0; a < num_threads; a++) total_loss_epoch += thread_losses[a];\n", EPOCH)
    # per-epoch save (ref :542-556) and final save (ref :560-576): the value comes from the exported table
    s = replace_once(s, "    if (classes == 0 && save_every_epoch) {\n",
                     "    if (classes == 0 && save_every_epoch) {\n" + EXPORT.replace("  w2b", "      w2b").replace("  if (", "      if ("))
    s = replace_once(s, "  // Write an extra file\n", EXPORT + "  // Write an extra file\n")
    s = replace_once(s, "float avg = u[a*layer1_size+b] + v[a*layer1_size+b];\n", "float avg = w2b_seam_q[a*layer1_size+b];\n", count=2)
    s = replace_once(s, "avg = quantize(avg, bitlevel);\n", "/* quantize(u+v): done on the device */\n", count=2)
    patched = os.path.join(out_dir, "word2bits_hipseam.cpp")
    with open(patched, "w") as f:
        f.write(s)
    exe = os.path.join(out_dir, "word2bits_hipseam")
    lib_dir = os.path.join(ROOT, "word2bits_amd")
    cmd = ["g++", "-O3", "-ffp-contract=off", "-w", "-I" + os.path.join(ROOT, "include"), patched, "-o", exe,
           "-L" + lib_dir, "-lword2bits_hip", "-Wl,-rpath,$ORIGIN/../../word2bits_amd", "-L/opt/rocm/lib",
           "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-lm", "-pthread"]
    subprocess.check_call(cmd)
    os.remove(patched)                      # the scratch copy of the reference source does not stay around
    print("built %s" % exe)
    return 0


if __name__ == "__main__":
    sys.exit(main())
