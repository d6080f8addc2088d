#!/usr/bin/env python3
"""The evaluator's measurement lives in bench.py (`--form eval`); this wrapper keeps the old entry point.

  python tools/eval_bench.py [--vocab 60238 --dim 200 --eval-questions 19544 --eval-kind 1bit --steps 5]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.execv(sys.executable, [sys.executable, os.path.join(ROOT, "bench.py"), "--form", "eval", "--steps", "5",
                          "--warmup", "1"] + sys.argv[1:])
