"""Records what the UNMODIFIED reference program does on the text8-sized synthetic corpus of the fidelity test
(tests/w2b_testlib.py write_zipf_text_corpus: 17 M Zipf(1) tokens over 70 000 words) at -bitlevel 1 -size 200 -window 8
-negative 24 -iter 3: the three "Epoch Loss" values, for -threads 8 (this container has 8 cores).
Writes tests/golden/fidelity_text8size.json.   ~20 minutes.   python tests/golden/make_fidelity_golden_text8size.py"""
import json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from w2b_testlib import write_zipf_text_corpus
corpus = write_zipf_text_corpus("/tmp/w2b_text8size.txt")
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
t0 = time.time()
p = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "word2bits_stock"), "-train", corpus, "-output", "/tmp/w2b_text8size_ref.bin",
                    "-bitlevel", "1", "-size", "200", "-window", "8", "-negative", "24", "-iter", "3", "-threads", str(threads),
                    "-min-count", "5", "-binary", "1"], capture_output=True, text=True, check=True)
out = {"corpus": "write_zipf_text_corpus(vocab=70000, n_tokens=17_000_000, seed=0)", "threads": threads,
       "flags": "-bitlevel 1 -size 200 -window 8 -negative 24 -iter 3 -min-count 5",
       "vocab_size": int(re.search(r"Vocab size: (\d+)", p.stdout).group(1)),
       "train_words": int(re.search(r"Words in train file: (\d+)", p.stdout).group(1)),
       "epoch_losses": [float(x) for x in re.findall(r"Epoch Loss: (-?[\d.]+)", p.stdout)], "secs": round(time.time() - t0, 1)}
print(json.dumps(out))
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "fidelity_text8size.json"), "w"), indent=1)
