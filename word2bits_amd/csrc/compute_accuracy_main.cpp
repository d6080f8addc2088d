// compute_accuracy_main.cpp -- drop-in for the reference's evaluator program (ref src/compute-accuracy.c:63-189):
//   ./compute_accuracy <FILE> <bitlevel> <threshold> [fma|nofma] < questions-words.txt
// Same positional arguments, same stdout.  The scan runs on the MI355X through include/word2bits_eval.h.
// The optional 4th argument (or W2B_EVAL_FUSED=0|1) selects which build of the reference the scores are
// bit-identical to: "fma" (default; the reference's own Makefile flags on an FMA-capable host) or "nofma"
// (-ffp-contract=off).  The reference ignores a 4th argument, so scripts can pass it to both.
#include "../../include/word2bits_eval.h"
#include "../../include/word2bits_hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

int main(int argc, char **argv) {
  if (argc < 2) {   // ref :73-76
    printf("Usage: ./compute-accuracy <FILE> <bitlevel> <threshold>\nwhere FILE contains word projections, and "
           "threshold is used to reduce vocabulary of the model for fast approximate evaluation (0 = off, "
           "otherwise typical value is 30000)\n");
    return 0;
  }
  const int bitlevel = argc > 2 ? atoi(argv[2]) : 0;          // ref :78
  const long long threshold = argc > 3 ? atoi(argv[3]) : 0;   // ref :79
  int fused = 1;
  if (const char *env = getenv("W2B_EVAL_FUSED")) fused = atoi(env) != 0;
  if (argc > 4) fused = strcmp(argv[4], "nofma") != 0;
  int device = 0;
  if (const char *env = getenv("W2B_DEVICE")) device = atoi(env);

  w2b_eval *e = nullptr;
  const int rc = w2b_eval_load(argv[1], bitlevel, threshold, fused, device, &e);
  if (rc == W2B_EIO && !strcmp(w2b_last_error(), "Input file not found")) {
    printf("Input file not found\n");                          // ref :81-84
    return -1;
  }
  if (rc != W2B_OK) {
    fprintf(stderr, "compute_accuracy: %s\n", w2b_last_error());
    return 1;
  }
  std::string in;
  char buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, stdin)) > 0) in.append(buf, n);
  char *txt = nullptr;
  int64_t len = 0;
  if (w2b_eval_transcript(e, in.data(), (int64_t)in.size(), &txt, &len) != W2B_OK) {
    fprintf(stderr, "compute_accuracy: %s\n", w2b_last_error());
    return 1;
  }
  fwrite(txt, 1, (size_t)len, stdout);
  w2b_eval_free_text(txt);
  w2b_eval_free(e);
  return 0;
}
