"""word2bits_amd -- MI355X (gfx950) native hot path of Word2Bits behind the reference's interface.

Only what the path needs lives here: `csrc/` (HIP kernels + C ABI + host ingest) and a thin ctypes
mirror of the reference's training interface.  Importing the package never pulls in the oracle.
"""
from ._lib import W2bError, LIB_PATH, lib        # noqa: F401
from .trainer import Corpus, Trainer, comm_unique_id, train_model   # noqa: F401
from .trainer import packed_words_per_row, pack_quantized, unpack_quantized, unpack_vectors_file   # noqa: F401
from .evaluator import Evaluator, compute_accuracy   # noqa: F401
