#!/usr/bin/env python3
"""End-to-end wall time of ./word2bits beside the unmodified reference program on ONE file (round-3 review, missing #4:
ingest + table + train + save, ref progress line :383-387, README "several minutes").

Both programs print the same marker lines (the CLI is a drop-in), so the phases are read off their unbuffered stdout:
    launch -> 'Starting epoch: 0'   start-up: vocabulary, InitNet, unigram table (ref :520-531) [+ device upload]
    -> last 'Epoch Loss:'           training (all epochs)
    -> exit                         save (ref :540-576), written to a real file
File: the 22 M-token headline corpus of the fidelity tests (w2b_testlib.write_headline_corpus: V = 400 K, every word 5x +
20 M Zipf(1) tokens), flags of BASELINE configs[1]: -bitlevel 1 -size 800 -window 8 -negative 24 -iter 1 -sample 0 -binary 1.
usage: tools/e2e_compare.py out.json [--ref-threads N]   (run on the GPU box; test infrastructure, not product)"""
import json, os, select, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from w2b_testlib import write_headline_corpus


def phases(cmd, timeout=1500):
    t0 = time.time()
    p = subprocess.Popen(["stdbuf", "-o0"] + cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, bufsize=0)
    fd = p.stdout.fileno()
    buf, t_epoch, t_loss = b"", None, None
    while time.time() - t0 < timeout:
        r, _, _ = select.select([fd], [], [], 0.5)
        now = time.time()
        if r:
            chunk = os.read(fd, 65536)
            if not chunk:
                break
            buf = (buf + chunk)[-8192:]
            if t_epoch is None and b"Starting epoch: 0" in buf:
                t_epoch = now
            if b"Epoch Loss:" in chunk or (t_loss is None and b"Epoch Loss:" in buf):
                t_loss = now
        elif p.poll() is not None:
            break
    p.wait()
    t1 = time.time()
    if t_epoch is None or t_loss is None:
        return None
    return {"startup_s": round(t_epoch - t0, 2), "train_s": round(t_loss - t_epoch, 2), "save_s": round(t1 - t_loss, 2), "total_s": round(t1 - t0, 2)}


def main():
    out = sys.argv[1]
    ref_threads = int(sys.argv[sys.argv.index("--ref-threads") + 1]) if "--ref-threads" in sys.argv else (os.cpu_count() or 1)
    tmp = "/tmp/w2b_e2e"
    os.makedirs(tmp, exist_ok=True)
    corpus = write_headline_corpus(os.path.join(tmp, "headline_%d.txt" % os.getpid()))
    flags = ["-bitlevel", "1", "-size", "800", "-window", "8", "-negative", "24", "-iter", "1", "-sample", "0", "-binary", "1", "-min-count", "5"]
    res = {"file": "write_headline_corpus(vocab=400000, n_zipf=20000000): %d bytes, 22 M tokens" % os.path.getsize(corpus), "flags": " ".join(flags)}
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else "both"     # hip | ref | both (two calls merge into out.json)
    if os.path.exists(out):
        res.update({k: v for k, v in json.load(open(out)).items() if k in ("word2bits_hip", "reference", "reference_threads")})
    if only in ("hip", "both"):
        res["word2bits_hip"] = phases([os.path.join(ROOT, "word2bits"), "-train", corpus, "-output", os.path.join(tmp, "hip.bin"), "-threads", "0"] + flags)
    if only in ("ref", "both"):
        res["reference"] = phases([os.path.join(ROOT, "oracle", "_ref", "word2bits_stock"), "-train", corpus, "-output", os.path.join(tmp, "ref.bin"),
                                   "-threads", str(ref_threads)] + flags)
        res["reference_threads"] = ref_threads
    if res.get("word2bits_hip") and res.get("reference"):
        res["speedup_total"] = round(res["reference"]["total_s"] / res["word2bits_hip"]["total_s"], 1)
        res["speedup_train"] = round(res["reference"]["train_s"] / max(res["word2bits_hip"]["train_s"], 1e-3), 1)
    for f in ("hip.bin", "ref.bin", os.path.basename(corpus)):
        try:
            os.remove(os.path.join(tmp, f))
        except OSError:
            pass
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
