"""Multi-GPU plan of the hot path: one process per GPU, corpus shards <-> ranks, replicas of
[u || v] combined by a periodic all-reduce.  The reference's only parallelism is
Hogwild over pthreads that share one model (ref src/word2bits.cpp:535-536); across GPUs there is no
shared memory, so each rank trains a full replica on its own corpus shard and the replicas are
summed as DELTAS:  W <- base + sum_r (W_r - base)  -- the closest analogue of all threads adding
their updates into one shared table.

Three drivers of the same exchange:
  * the library's own RCCL communicator (w2b_comm_init / w2b_sync_replicas) -- used by the CLI
    and by bench.py on GPUs: asynchronous, chunked, overlapped with the training launches;
  * `PhasedReplicaSync` below: the library's exchange kernels (w2b_exchange_begin / delta / apply / end)
    with a torch.distributed collective for the sum -- RCCL when every rank has a GPU, gloo when several
    ranks share one GPU (how the training effect of the exchange is tested on a one-GPU box);
  * `TorchReplicaSync`: the protocol restated on plain torch tensors -- it runs on CPU tensors over gloo,
    which is how the N>1 arithmetic is tested without any GPU.
"""
import numpy as np


def worker_plan(total_workers, world, rank):
    """Global Hogwild worker ids owned by `rank`: a contiguous block, as the CLI assigns them
    (word2bits_main.cpp: worker_offset = rank * per_gpu).  total_workers must divide evenly."""
    if total_workers % world:
        raise ValueError("-threads (%d) must be a multiple of the number of GPUs (%d)" % (total_workers, world))
    per = total_workers // world
    return rank * per, per


def token_shard_starts(n_tokens, total_workers, worker_offset, num_workers):
    """Start index of each local worker inside a token stream of n_tokens when the stream is cut into
    total_workers equal shards (the token-stream analogue of file_size/num_threads*id, ref :377)."""
    ids = np.arange(worker_offset, worker_offset + num_workers, dtype=np.int64)
    return ids * (n_tokens // total_workers)


def replica_token_slice(tokens, starts, quota, slack=64000):
    """The part [lo, hi) of the token stream that the workers starting at `starts` (one replica's) read, and whether the
    stream goes on behind it.  A worker stops after the sentence in which its word count passes `quota`
    (train_words / total_threads, ref :414-423), which may lie beyond the next worker's start: quota + 2 tokens, then on
    to the next "</s>" (id 0), at most `slack` tokens further.  Same rule as ./word2bits -gpus N (word2bits_main.cpp)."""
    n = len(tokens)
    lo, hi = n, 0
    for s in starts:
        s = int(s)
        lo = min(lo, s)
        e = s + int(quota) + 2
        cap = min(n, e + slack)
        if e < cap:                                   # first e' >= e with tokens[e' - 1] == 0, else cap
            z = np.flatnonzero(np.asarray(tokens[e - 1:cap - 1]) == 0)
            e = e + int(z[0]) if len(z) else cap
        hi = max(hi, min(e, n))
    lo = min(lo, hi)
    return lo, hi, hi < n


def exchange_unique_id(dist, rank, make_id):
    """rank 0 creates the RCCL unique id, everyone receives it over the existing process group."""
    box = [make_id() if rank == 0 else None]
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast_object_list(box, src=0)
    return box[0]


def saturation_factors(rate, words, contributors, tau_u=64.0, tau_v=64.0, vocab_size=None):
    """The per-row factor on the summed delta of mode 2 (w2b_kernels_misc.hip k_xchg_factor, rules 0 and 2) on torch tensors:
    k = (1 - exp(-c n / tau)) / (c (1 - exp(-n / tau))) with n = rate * words expected updates of the row per replica since the
    last exchange and c replicas that changed it; 1 for c <= 1.  rate: [2 V] (rows of u, then rows of v)."""
    import torch
    V = vocab_size if vocab_size is not None else rate.numel() // 2
    tau = torch.cat([torch.full((V,), float(tau_u)), torch.full((rate.numel() - V,), float(tau_v))]).to(rate)
    x = (rate.double() * float(words) / tau.double()).clamp_min(1e-12)
    c = contributors.double().clamp_min(1.0)
    k = torch.expm1(-c * x) / (c * torch.expm1(-x))
    k = torch.minimum(torch.ones_like(k), torch.maximum(k, 1.0 / c))
    return torch.where((contributors > 1) & (x > 1e-6), k, torch.ones_like(k)).to(rate.dtype)


def quantization_cell(x, bitlevel):
    """an integer label of the quantization cell of every element (quantize(), ref src/word2bits.cpp:73-108): two values with the
    same label have the same forward value"""
    import torch
    neg = (x < 0).to(torch.int64)
    if bitlevel == 1 or bitlevel == 3:
        return neg
    mag = x.abs()
    if bitlevel == 2:
        return neg * 2 + (~(mag <= 0.5)).to(torch.int64)
    steps = 1 << (bitlevel - 1)
    k = torch.clamp((mag * steps + 0.5).to(torch.int64), max=steps)
    return torch.where(k == 0, torch.zeros_like(k), neg * (steps + 1) + k)   # (+0 and -0 are one forward value)


class TorchReplicaSync:
    """The replica exchange on torch tensors (CPU tensors over gloo: how the N > 1 arithmetic is tested without a GPU).

    `model` is the flat [u || v] tensor of this rank, `base` the snapshot taken at the previous exchange.
      mode 0: delta-sum;  mode 1: average;
      mode 2: the library's default rule -- per row the saturation factor on the summed delta decides every
              element's quantized value, and at bitlevel 1 the whole sum is taken wherever it keeps that sign.  Needs
              `rate` ([2 V] expected updates per centre word), `dim`, `bitlevel`, and per call `words` (centre words per replica
              since the last exchange).
    With world size 1 every mode leaves `model` bit-identical (no arithmetic is done)."""

    def __init__(self, dist, mode=0, rate=None, dim=None, bitlevel=1, tau_u=64.0, tau_v=64.0, cells=True):
        self.dist = dist
        self.mode = mode
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rate, self.dim, self.bitlevel, self.tau, self.cells = rate, dim, bitlevel, (tau_u, tau_v), cells

    def sync(self, model, base, words=0):
        import torch
        if self.world == 1:
            return model
        if self.mode == 0:
            model.sub_(base)                       # W <- W - base
            self.dist.all_reduce(model)            # sum of deltas
            model.add_(base)                       # W <- base + sum
            base.copy_(model)
        elif self.mode == 1:
            self.dist.all_reduce(model)
            model.mul_(1.0 / self.world)
            base.copy_(model)
        elif self.mode == 2:
            d = model - base
            touched = (d.view(-1, self.dim) != 0).any(1).to(d.dtype)      # rows this replica changed (k_xchg_touched)
            self.dist.all_reduce(touched)
            self.dist.all_reduce(d)                                        # S
            k = saturation_factors(self.rate, words, touched, *self.tau)
            safe = d * k.repeat_interleave(self.dim)
            if self.cells and self.bitlevel == 1:          # (one bit only: with more bits the cells do harm)
                same = quantization_cell(base + safe, self.bitlevel) == quantization_cell(base + d, self.bitlevel)
                comb = torch.where(same, d, safe)
            else:
                comb = safe
            base.add_(comb)
            model.copy_(base)                      # (synchronous: nothing was trained since the delta)
        else:
            raise ValueError("unknown sync mode")
        return model


class PhasedReplicaSync:
    """The library's exchange (delta / apply kernels, base snapshot, progress counters) around a torch.distributed
    all-reduce.  `trainer.exchange_init()` must have been called while the replicas were identical."""

    def __init__(self, dist, trainer, mode=0):
        self.dist, self.t, self.mode = dist, trainer, mode
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.on_device = dist.is_initialized() and dist.get_backend() == "nccl"

    def sync(self):
        import torch
        n_chunks, words = self.t.exchange_begin()
        scale = 1.0 / self.world if self.mode == 1 else 1.0
        if self.mode == 2:                                    # contributor average: who trained which row
            self._reduce(self.t.device_tensor(*self.t.exchange_counts()))
        for c in range(n_chunks):
            ptr, n = self.t.exchange_delta(c)                 # complete on return
            buf = self.t.device_tensor(ptr, n)
            self._reduce(buf)
            self.t.exchange_apply(c, scale)
        total = torch.tensor([words], dtype=torch.int64)
        if self.world > 1:
            if self.on_device:
                total = total.to(buf.device)
            self.dist.all_reduce(total)
        self.t.exchange_end(int(total.item()))

    def _reduce(self, buf):
        import torch
        if self.world > 1:
            if self.on_device:
                self.dist.all_reduce(buf)
            else:                                             # gloo: through host memory
                host = buf.cpu()
                self.dist.all_reduce(host)
                buf.copy_(host)
            if buf.is_cuda:
                torch.cuda.synchronize(buf.device)


def global_progress_alpha(starting_alpha, words_done_all_ranks, iters, train_words):
    """alpha schedule of ref :391 on the GLOBAL word count (float32 arithmetic as in the reference)."""
    a = np.float32(starting_alpha) * (np.float32(1) - np.float32(words_done_all_ranks) /
                                      np.float32(iters * train_words + 1))
    floor = np.float32(starting_alpha * 0.0001)
    return float(max(a, floor))
