"""Host-side mirror of the reference's training interface on top of the HIP C ABI.

Names follow the reference (ref src/word2bits.cpp): `Trainer` holds what the reference keeps in
process globals (:45-61); `Trainer.train_epoch()` is one iteration of TrainModel's epoch loop
(:532-539, pthread_create/join of TrainModelThread); `train_model()` is TrainModel (:518-577).
All arithmetic happens in libword2bits_hip.so on the GPU.
"""
import ctypes as C
import numpy as np

from . import _lib
from ._lib import Config, check, lib


def _f32(a):
    return a.ctypes.data_as(_lib.f32p)


def _i32(a):
    return a.ctypes.data_as(_lib.i32p)


def _i64(a):
    return a.ctypes.data_as(_lib.i64p)


def packed_words_per_row(dim, bitlevel):
    n = lib().w2b_packed_words_per_row(int(dim), int(bitlevel))
    if n < 0:
        raise _lib.W2bError(_lib.W2B_EUNSUPPORTED, "bit-packed vectors exist for bitlevel 1 and 2")
    return n


def pack_quantized(values, bitlevel):
    """host twin of Trainer.export_packed: values [rows][dim] already quantized at `bitlevel`"""
    values = np.ascontiguousarray(values, np.float32)
    out = np.empty((values.shape[0], packed_words_per_row(values.shape[1], bitlevel)), np.uint64)
    check(lib().w2b_pack_quantized(_f32(values), values.shape[0], values.shape[1], int(bitlevel), out.ctypes.data_as(_lib.u64p)))
    return out


def unpack_quantized(packed, dim, bitlevel):
    packed = np.ascontiguousarray(packed, np.uint64)
    out = np.empty((packed.shape[0], dim), np.float32)
    check(lib().w2b_unpack_quantized(packed.ctypes.data_as(_lib.u64p), packed.shape[0], int(dim), int(bitlevel), _f32(out)))
    return out


def unpack_vectors_file(packed_path, out_path, binary=1):
    """packed model file -> the reference's output file format (ref :560-576)"""
    check(lib().w2b_unpack_vectors_file(packed_path.encode(), out_path.encode(), int(binary)))


class Corpus:
    """Vocabulary + token stream of a training file (LearnVocabFromTrainFile, ref :265-301)."""

    def __init__(self, train_file, min_count=5, vocab_hash_size=0):
        """vocab_hash_size: the reference's constant of that name (ref :35; 0 = 30 000 000) -- it decides when
        ReduceVocab (ref :245-263) runs while the vocabulary is learned (w2b_corpus_load_ex)."""
        self._h = _lib.vp()
        rc = lib().w2b_corpus_load_ex(train_file.encode(), int(min_count), int(vocab_hash_size), C.byref(self._h))
        if rc != 0:
            raise _lib.W2bError(rc, "ERROR: training data file not found!")   # ref :272
        L = lib()
        self.vocab_size = L.w2b_corpus_vocab_size(self._h)
        self.train_words = L.w2b_corpus_train_words(self._h)
        self.file_size = L.w2b_corpus_file_size(self._h)
        self.num_tokens = L.w2b_corpus_num_tokens(self._h)

    @property
    def handle(self):
        return self._h

    def words(self):
        L = lib()
        return [L.w2b_corpus_word(self._h, i).decode("latin1") for i in range(self.vocab_size)]

    def counts(self):
        p = lib().w2b_corpus_counts(self._h)
        return np.ctypeslib.as_array(p, shape=(self.vocab_size,)).copy()

    def tokens(self):
        if self.num_tokens == 0:
            return np.zeros(0, np.int32)
        p = lib().w2b_corpus_tokens(self._h)
        return np.ctypeslib.as_array(p, shape=(self.num_tokens,)).copy()

    def search(self, word):
        return lib().w2b_corpus_search(self._h, word.encode("latin1"))

    def shards(self, num_threads):
        starts = np.zeros(num_threads, np.int64)
        ov = np.zeros(num_threads, np.int32)
        check(lib().w2b_corpus_shards(self._h, num_threads, _i64(starts), _i32(ov)))
        return starts, ov

    def save_vectors(self, path, values, binary):
        values = np.ascontiguousarray(values, np.float32)
        check(lib().w2b_save_vectors(path.encode(), self._h, _f32(values), values.shape[1], int(binary)))

    def save_vectors_packed(self, path, packed, dim, bitlevel):
        """bit-packed model file (include/word2bits_corpus.h); `packed` as Trainer.export_packed / pack_quantized give it"""
        packed = np.ascontiguousarray(packed, np.uint64)
        check(lib().w2b_save_vectors_packed(path.encode(), self._h, packed.ctypes.data_as(_lib.u64p), int(dim), int(bitlevel)))

    def close(self):
        if self._h:
            lib().w2b_corpus_free(self._h)
            self._h = _lib.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Trainer:
    """One model replica on one GPU (the reference's globals u, v, alpha, word_count_actual, ...)."""

    # knobs of struct w2b_tuning applied to every new Trainer before the keyword arguments (tests re-run whole test
    # functions under e.g. force_row_desc=1 by patching this dict)
    default_tuning = {}

    def __init__(self, vocab_size, layer1_size=100, window=5, negative=5, bitlevel=1, num_threads=12,
                 iter=5, alpha=0.05, sample=1e-3, reg=0.0, train_words=0, compute_loss=True, device=0,
                 worker_offset=0, total_threads=0, relaxed_coherence=False, window_cache=None, exact=False, row_groups=None,
                 **tuning):
        cfg = Config()
        cfg.vocab_size, cfg.train_words, cfg.iter = int(vocab_size), int(train_words), int(iter)
        cfg.layer1_size, cfg.window, cfg.negative = int(layer1_size), int(window), int(negative)
        cfg.bitlevel, cfg.num_threads = int(bitlevel), int(num_threads)
        cfg.alpha, cfg.sample, cfg.reg = float(alpha), float(sample), float(reg)
        cfg.compute_loss, cfg.device = int(bool(compute_loss)), int(device)
        cfg.worker_offset, cfg.total_threads = int(worker_offset), int(total_threads)
        cfg.relaxed_coherence = int(bool(relaxed_coherence))
        # window_cache: None = automatic, True = sentence-resident kernel whenever it fits, False = plain
        # row_groups: None = automatic, True = the row-group kernel wherever it fits, False = never
        cfg.plain_worker_kernel = 2 if window_cache else (3 if row_groups else (1 if (window_cache is False or row_groups is False) else 0))
        # exact: serial dot product in the reference's order -> a 1-worker run is bit-identical to the CPU program
        cfg.exact_reduction = int(bool(exact))
        self.cfg = cfg
        self._h = _lib.vp()
        check(lib().w2b_trainer_create(C.byref(cfg), C.byref(self._h)))
        self.vocab_size, self.layer1_size = int(vocab_size), int(layer1_size)
        self.num_threads = int(num_threads)
        tuning = dict(self.default_tuning, **tuning)
        if tuning:                 # hot_rows_v, hot_rows_u, hot_period, hot_cap, force_row_desc, grid_per_cu, mem_mode
            self.set_tuning(**tuning)

    # ---- tuning knobs (struct w2b_tuning)
    def get_tuning(self):
        tn = _lib.Tuning()
        check(lib().w2b_get_tuning(self._h, C.byref(tn)))
        return {k: getattr(tn, k) for k, _ in _lib.Tuning._fields_ if k != "struct_size"}

    def set_tuning(self, **kw):
        tn = _lib.Tuning()
        check(lib().w2b_get_tuning(self._h, C.byref(tn)))
        for k, v in kw.items():
            if k == "struct_size" or not hasattr(tn, k):
                raise TypeError("unknown tuning knob %r" % k)
            setattr(tn, k, int(v))
        check(lib().w2b_set_tuning(self._h, C.byref(tn)))

    # ---- model
    def init_net(self):
        check(lib().w2b_init_net(self._h))

    def set_model(self, u, v):
        u = np.ascontiguousarray(u, np.float32)
        v = np.ascontiguousarray(v, np.float32)
        assert u.shape == v.shape == (self.vocab_size, self.layer1_size)
        check(lib().w2b_set_model(self._h, _f32(u), _f32(v)))

    def get_model(self):
        u = np.empty((self.vocab_size, self.layer1_size), np.float32)
        v = np.empty_like(u)
        check(lib().w2b_get_model(self._h, _f32(u), _f32(v)))
        return u, v

    def export_quantized(self):
        out = np.empty((self.vocab_size, self.layer1_size), np.float32)
        check(lib().w2b_export_quantized(self._h, _f32(out)))
        return out

    def export_packed(self):
        """quantize(u+v) bit-packed on the device (bitlevel 1 / 2): uint64 [vocab_size][words_per_row]"""
        wpr = packed_words_per_row(self.layer1_size, self.cfg.bitlevel)
        out = np.empty((self.vocab_size, wpr), np.uint64)
        check(lib().w2b_export_packed(self._h, out.ctypes.data_as(_lib.u64p)))
        return out

    def model_device_ptrs(self):
        u, v = _lib.vp(), _lib.vp()
        check(lib().w2b_model_device_ptrs(self._h, C.byref(u), C.byref(v)))
        return u.value, v.value

    def model_tensor(self):
        """torch view (no copy) of the flat [u || v] device buffer, for torch.distributed collectives."""
        import torch
        u_ptr, _ = self.model_device_ptrs()
        n = 2 * self.vocab_size * self.layer1_size

        class _View:       # __cuda_array_interface__ v2: torch wraps the library-owned allocation
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (u_ptr, False), "version": 2}
        return torch.as_tensor(_View(), device=torch.device("cuda", self.cfg.device))

    # ---- sampler state
    def set_vocab_counts(self, cn, table_size=100000000):
        cn = np.ascontiguousarray(cn, np.int64)
        assert len(cn) == self.vocab_size
        check(lib().w2b_set_vocab_counts(self._h, _i64(cn), int(table_size)))

    def set_unigram_table(self, table):
        table = np.ascontiguousarray(table, np.int32)
        check(lib().w2b_set_unigram_table(self._h, _i32(table), len(table)))

    def set_exp_table(self, tab):
        tab = np.ascontiguousarray(tab, np.float32)
        check(lib().w2b_set_exp_table(self._h, _f32(tab)))

    # ---- form (i): workers
    def set_corpus(self, ids):
        ids = np.ascontiguousarray(ids, np.int32)
        check(lib().w2b_set_corpus(self._h, _i32(ids), len(ids)))

    def set_corpus_slice(self, ids, more_follows):
        """the part of the token stream this replica's workers read (shard starts are relative to it)"""
        ids = np.ascontiguousarray(ids, np.int32)
        check(lib().w2b_set_corpus_slice(self._h, _i32(ids), len(ids), int(bool(more_follows))))

    def set_corpus_device(self, dev_ptr, n_tokens):
        check(lib().w2b_set_corpus_device(self._h, _lib.vp(dev_ptr), int(n_tokens)))

    def set_shards(self, starts, first_override=None):
        starts = np.ascontiguousarray(starts, np.int64)
        assert len(starts) == self.num_threads
        ov = None
        if first_override is not None:
            ov = np.ascontiguousarray(first_override, np.int32)
        check(lib().w2b_set_shards(self._h, _i64(starts), None if ov is None else _i32(ov)))

    def epoch_begin(self):
        check(lib().w2b_epoch_begin(self._h))

    def train_step(self, max_positions):
        check(lib().w2b_train_step(self._h, int(max_positions)))

    def epoch_status(self, want_loss=True):
        fin, wca, alpha, loss = C.c_int32(0), C.c_int64(0), C.c_float(0), C.c_double(0)
        check(lib().w2b_epoch_status(self._h, C.byref(fin), C.byref(wca), C.byref(alpha),
                                     C.byref(loss) if want_loss else None))
        return bool(fin.value), wca.value, alpha.value, loss.value

    def epoch_poll(self, lag=1):
        """state after the launch `lag` launches before the latest one, without waiting for the newer ones"""
        fin, wca, alpha, loss = C.c_int32(0), C.c_int64(0), C.c_float(0), C.c_double(0)
        check(lib().w2b_epoch_poll(self._h, int(lag), C.byref(fin), C.byref(wca), C.byref(alpha), C.byref(loss)))
        return bool(fin.value), wca.value, alpha.value, loss.value

    def train_epoch(self, positions_per_launch=4096):
        """pthread_create + pthread_join of one epoch (ref :535-536). Returns the epoch loss."""
        self.epoch_begin()
        while True:
            self.train_step(positions_per_launch)
            fin, _, _, loss = self.epoch_status()
            if fin:
                return loss

    def worker_kernel_info(self):
        """(resident, radius, column_bytes, workgroups_per_cu, hot_rows) of the form-(i) kernel train_step() runs"""
        a, b, c, d, e = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
        check(lib().w2b_worker_kernel_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e)))
        return a.value == 1, b.value, c.value, d.value, e.value

    def worker_kernel_name(self):
        """which form-(i) kernel train_step() runs: "plain", "resident" (sentence-resident) or "groups" (row groups)"""
        a = C.c_int32(0)
        check(lib().w2b_worker_kernel_info(self._h, C.byref(a), None, None, None, None))
        return ("plain", "resident", "groups")[a.value]

    def suggested_threads(self):
        n = C.c_int32(0)
        check(lib().w2b_suggested_threads(self._h, C.byref(n)))
        return n.value

    # ---- form (ii): tuples
    def train_tuples(self, center, ctx_off, ctx, neg, alpha, serial=False):
        center = np.ascontiguousarray(center, np.int32)
        ctx_off = np.ascontiguousarray(ctx_off, np.int32)
        ctx = np.ascontiguousarray(ctx, np.int32)
        neg = np.ascontiguousarray(neg, np.int32)
        loss = C.c_double(0)
        check(lib().w2b_train_tuples(self._h, len(center), _i32(center), _i32(ctx_off), _i32(ctx), _i32(neg),
                                     float(alpha), int(bool(serial)), C.byref(loss)))
        return loss.value

    def train_tuples_device(self, n, center_ptr, ctx_off_ptr, ctx_ptr, neg_ptr, alpha, grid=0):
        check(lib().w2b_train_tuples_device(self._h, int(n), _lib.vp(center_ptr), _lib.vp(ctx_off_ptr),
                                            _lib.vp(ctx_ptr), _lib.vp(neg_ptr), float(alpha), int(grid)))

    # ---- stream / timing / replicas
    def synchronize(self):
        check(lib().w2b_synchronize(self._h))

    def timing_enable(self, on=True):
        check(lib().w2b_timing_enable(self._h, int(bool(on))))

    def timing_read(self):
        ms, n = C.c_double(0), C.c_int64(0)
        check(lib().w2b_timing_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def timing_launches(self):
        """durations (ms) of the training launches since the last timing_read(), one by one"""
        n = C.c_int64(0)
        check(lib().w2b_timing_launches(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.float64)
        if n.value:
            check(lib().w2b_timing_launches(self._h, out.ctypes.data_as(_lib.f64p), n.value, C.byref(n)))
        return out

    def comm_init(self, nranks, rank, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), 128) if unique_id is not None else None
        check(lib().w2b_comm_init(self._h, int(nranks), int(rank), buf))

    def comm_count(self):
        """ranks of the library's RCCL communicator as RCCL reports them (0: none)"""
        n = C.c_int32(0)
        check(lib().w2b_comm_count(self._h, C.byref(n)))
        return n.value

    def sync_replicas(self, mode=0):
        check(lib().w2b_sync_replicas(self._h, int(mode)))

    def sync_stats(self):
        """(exchanges, summed device ms) since the last call"""
        n, ms = C.c_int64(0), C.c_double(0)
        check(lib().w2b_sync_stats(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    # ---- the exchange in phases, for a host-supplied collective (include/word2bits_hip.h)
    def exchange_init(self):
        """call while all replicas hold the same model"""
        check(lib().w2b_exchange_init(self._h))

    def exchange_begin(self):
        """-> (number of chunks, this replica's word_count_actual)"""
        n, w = C.c_int64(0), C.c_int64(0)
        check(lib().w2b_exchange_begin(self._h, C.byref(n), C.byref(w)))
        return n.value, w.value

    def exchange_counts(self):
        """-> (device pointer, floats): 1 for every row of [u||v] this replica changed since the last exchange; sum it over
        the replicas in place and exchange_apply divides every row's summed delta by it (contributor average)"""
        p, n = _lib.vp(), C.c_int64(0)
        check(lib().w2b_exchange_counts(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def exchange_delta(self, chunk):
        """-> (device pointer, floats) of this replica's delta of the chunk; sum it over the replicas in place"""
        p, n = _lib.vp(), C.c_int64(0)
        check(lib().w2b_exchange_delta(self._h, int(chunk), C.byref(p), C.byref(n)))
        return p.value, n.value

    def exchange_apply(self, chunk, scale=1.0):
        check(lib().w2b_exchange_apply(self._h, int(chunk), float(scale)))

    def exchange_end(self, word_count_all_replicas=-1):
        check(lib().w2b_exchange_end(self._h, int(word_count_all_replicas)))

    def device_tensor(self, ptr, n):
        """torch view (no copy) of n floats of library-owned device memory"""
        import torch

        class _View:
            __cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(_View(), device=torch.device("cuda", self.cfg.device))

    def close(self):
        if self._h:
            lib().w2b_trainer_destroy(self._h)
            self._h = _lib.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_unique_id():
    buf = C.create_string_buffer(128)
    check(lib().w2b_comm_unique_id(buf))
    return buf.raw


def train_model(train_file, output_file, bitlevel=1, size=100, window=5, negative=5, threads=12, iter=5,
                min_count=5, alpha=0.05, sample=1e-3, reg=0.0, binary=0, table_size=100000000,
                positions_per_launch=4096, device=0, verbose=False, relaxed_coherence=False, window_cache=None,
                exact=False):
    """TrainModel (ref :518-577) on one GPU: vocab, InitNet, unigram table, `iter` epochs, save.
    Returns the list of epoch losses."""
    corpus = Corpus(train_file, min_count)
    if threads < 1:      # GPU extension (as ./word2bits -threads 0): as many workers as fill the device for this shape
        probe = Trainer(corpus.vocab_size, size, window, negative, bitlevel, 1, iter, alpha, sample, reg, corpus.train_words,
                        True, device, relaxed_coherence=relaxed_coherence, window_cache=window_cache, exact=exact)
        probe.set_vocab_counts(corpus.counts(), 0)     # the kernel choice (and the fill) depends on the word counts
        threads = probe.suggested_threads()
        probe.close()
    t = Trainer(corpus.vocab_size, size, window, negative, bitlevel, threads, iter, alpha, sample, reg,
                corpus.train_words, True, device, relaxed_coherence=relaxed_coherence,
                window_cache=window_cache, exact=exact)
    t.init_net()
    t.set_vocab_counts(corpus.counts(), table_size if negative > 0 else 0)
    t.set_corpus(corpus.tokens())
    starts, ov = corpus.shards(threads)
    t.set_shards(starts, ov)
    losses = []
    for it in range(iter):
        if verbose:
            print("Starting epoch: %d" % it)
        losses.append(t.train_epoch(positions_per_launch))
        if verbose:
            print("Epoch Loss: %f" % losses[-1])
    corpus.save_vectors(output_file, t.export_quantized(), binary)
    t.close()
    corpus.close()
    return losses
