"""diagnostic: per-phase s_memtime ticks of k_eval_scores_mfma (library built with -DW2B_EVAL_EXP=128)"""
import ctypes as C, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import word2bits_amd as w2b
from word2bits_amd import _lib
from w2b_testlib import write_vectors_file
V, D, Q = 60238, int(sys.argv[1]) if len(sys.argv) > 1 else 200, 19544
rng = np.random.default_rng(3)
M = ((rng.integers(0, 2, (V, D)) * 2 - 1).astype(np.float32) / np.float32(3))
path = write_vectors_file(os.path.join(tempfile.mkdtemp(), "v.bin"), [("w%d" % i).encode() for i in range(V)], M)
b = rng.integers(0, V, (3, Q)).astype(np.int32)
ev = w2b.Evaluator(path, 0, 0, fused=True)
ev.top1(*b)
L = _lib.lib()
out = (C.c_ulonglong * 16)()
L.w2b_debug_eval_ticks(out, 1)
ev.timing()
ev.top1(*b)
ms, launches, macs = ev.timing()
L.w2b_debug_eval_ticks(out, 0)
n = out[4]
print("epilogue pieces (nt=0: ids, scan, reduce; nt=1: ...; tail):", [round(out[i] / n) for i in range(8, 14)], round(out[2] / n))
print("dim %d: kernel %.3f ms; workgroups %d; per workgroup (wave 0): prologue %.0f, main loop %.0f, epilogue %.0f ticks; first start -> last end %.0f ticks (%.3f ms at 2.385 GHz)" %
      (D, ms / launches, n, out[0] / n, out[1] / n, out[2] / n, out[6] - out[5], (out[6] - out[5]) / 2.385e6))
