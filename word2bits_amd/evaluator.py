"""ctypes mirror of the reference's evaluator program (ref src/compute-accuracy.c) over include/word2bits_eval.h.

    ev = Evaluator("vectors.bin", bitlevel=0, threshold=0)          # ref :77-112
    print(ev.transcript(open("questions-words.txt", "rb").read()).decode())   # ref :113-188, same bytes

The exhaustive scan runs on the MI355X (w2b_kernels_eval.hip); there is no CPU path in this module.
"""
import ctypes as C

import numpy as np

from . import _lib


class Evaluator:
    """`fused=True` reproduces the reference built with its own Makefile flags (FMA-contracted dot products),
    `fused=False` the -ffp-contract=off build; answers are identical to that build's, ties included."""

    def __init__(self, path, bitlevel=0, threshold=0, fused=True, device=0, _handle=None):
        self._h = C.c_void_p()
        self._L = _lib.lib()
        if _handle is not None:
            self._h = _handle
        else:
            _lib.check(self._L.w2b_eval_load(str(path).encode(), int(bitlevel), int(threshold), int(bool(fused)),
                                             int(device), C.byref(self._h)))
        self.words = int(self._L.w2b_eval_words(self._h))
        self.size = int(self._L.w2b_eval_size(self._h))

    @classmethod
    def from_trainer(cls, trainer, words, bitlevel=0, threshold=0, fused=True):
        """The evaluator on a live Trainer (no file round trip): what Evaluator(path) would hold after the trainer's
        vectors had been saved to `path` with binary=1.  `words` = the vocabulary (Corpus.words())."""
        L = _lib.lib()
        arr = (C.c_char_p * len(words))(*[w if isinstance(w, bytes) else w.encode("latin1") for w in words])
        h = C.c_void_p()
        _lib.check(L.w2b_eval_from_trainer(trainer._h, len(words), arr, int(bitlevel), int(threshold), int(bool(fused)),
                                           C.byref(h)))
        return cls(None, _handle=h)

    def close(self):
        if self._h:
            self._L.w2b_eval_free(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def word(self, row):
        return self._L.w2b_eval_word(self._h, int(row))

    def lookup(self, upper_word):
        """First row whose upper-cased word equals `upper_word` (bytes), or `words` (ref :140-145)."""
        return int(self._L.w2b_eval_lookup(self._h, bytes(upper_word)))

    def matrix(self):
        out = np.empty((self.words, self.size), np.float32)
        _lib.check(self._L.w2b_eval_get_matrix(self._h, out.ctypes.data_as(_lib.f32p)))
        return out

    def top1(self, b1, b2, b3):
        """ref :155-177 for a batch: (best row or -1, its score) per question."""
        b1, b2, b3 = (np.ascontiguousarray(x, np.int32) for x in (b1, b2, b3))
        n = len(b1)
        best, bestd = np.empty(n, np.int32), np.empty(n, np.float32)
        p = lambda a: a.ctypes.data_as(_lib.i32p)
        _lib.check(self._L.w2b_eval_top1(self._h, n, p(b1), p(b2), p(b3), p(best), bestd.ctypes.data_as(_lib.f32p)))
        return best, bestd

    def transcript(self, questions):
        """stdout of `compute_accuracy FILE bitlevel threshold < questions` as bytes."""
        questions = bytes(questions)
        out, n = C.c_void_p(), C.c_int64()
        _lib.check(self._L.w2b_eval_transcript(self._h, questions, len(questions), C.byref(out), C.byref(n)))
        try:
            return C.string_at(out, n.value)
        finally:
            self._L.w2b_eval_free_text(out)

    def set_kernel(self, variant):
        """1 = f32 MFMA kernel (default), 0 = the same fused chain on the vector ALU (cross-check)"""
        _lib.check(self._L.w2b_eval_set_kernel(self._h, int(variant)))

    def timing(self):
        """(kernel ms, launches, multiply-adds) of the score kernel since the last call."""
        ms, n, macs = C.c_double(), C.c_int64(), C.c_double()
        _lib.check(self._L.w2b_eval_timing_read(self._h, C.byref(ms), C.byref(n), C.byref(macs)))
        return ms.value, n.value, macs.value


def compute_accuracy(path, questions, bitlevel=0, threshold=0, fused=True, device=0):
    """The reference's `main` (ref :63-189) as a function: returns the stdout bytes."""
    ev = Evaluator(path, bitlevel, threshold, fused, device)
    try:
        return ev.transcript(questions)
    finally:
        ev.close()
