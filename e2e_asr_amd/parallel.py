"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference has no distributed code at all; this is the MI355X-native
addition.  Utterances are independent, so the global batch is sharded by utterance with an
equal shard per rank and a full replica of the 10.6 M parameters per GPU.  The only exchange
is the gradient all-reduce(sum), placed between tf.gradients and clip_by_global_norm in the
seq2seq_model.py:148-155 sequence so that the clip sees the global-batch gradient; 1/N is
folded into the fused clip+Adam kernel.  Because loss = mean over utterances of a
length-normalised cost (losses.py:32-35), equal shards make mean-of-means exact.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce is bound by ONE link,
so fewer, larger messages win.  The flat fp32 gradient buffer (42.5 MB) is reduced in at most
five contiguous buckets -- decoders, then encoder layers top-down -- each launched
asynchronously as soon as its gradients are final, so the exchange hides under the encoder's
backward-through-time.
"""
import torch
import torch.distributed as dist


class TorchDistComm(object):
    """The exchange as DataParallel sees it: torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests)."""

    def __init__(self, process_group=None):
        if not dist.is_initialized():
            raise RuntimeError("DataParallel needs torch.distributed.init_process_group() first")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)

    def broadcast(self, t, src=0):
        dist.broadcast(t, src=src, group=self.group)

    def all_reduce(self, t, async_op=False):
        """Sum over ranks, in place.  Returns a work handle (`.wait()`) when async_op."""
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)


class DataParallel(object):
    def __init__(self, model, process_group=None, overlap=None, comm=None, grad_dtype=None):
        """overlap=None reads ASR_DP_OVERLAP (default off): the default path is ONE blocking all-reduce of
        the whole flat gradient after backward -- 42.5 MB, well under a millisecond of a >20 ms step,
        and the simplest thing that is correct by construction.  overlap=True launches the per-bucket
        all-reduces asynchronously under the encoder BPTT."""
        import os
        if overlap is None:
            overlap = os.environ.get("ASR_DP_OVERLAP", "0") == "1"
        # grad_dtype "bf16" (BASELINE config 3; ASR_DP_GRAD_DTYPE=bf16): the exchange carries bfloat16 -- 21.2 MB instead of
        # 42.5 MB per step over the xGMI ring -- and the sum comes back into the fp32 master gradient, so
        # clip, Adam moments and weights stay fp32.  Each shard gradient is rounded once (2^-9 relative) before the sum.
        if grad_dtype is None:
            grad_dtype = os.environ.get("ASR_DP_GRAD_DTYPE", "f32")
        if grad_dtype not in ("f32", "fp32", "float32", "bf16", "bfloat16"):
            raise ValueError("grad_dtype must be f32 or bf16, not %r" % (grad_dtype,))
        self.grad_bf16 = grad_dtype in ("bf16", "bfloat16")
        self.comm = TorchDistComm(process_group) if comm is None else comm
        self.world, self.rank = self.comm.world, self.comm.rank
        self.overlap = overlap
        self._pending = []
        self._done = set()
        self._n = model.variables.flat.numel()
        # identical initial weights everywhere (rank 0's), like a restored checkpoint
        self.comm.broadcast(model.variables.flat, src=0)
        self.buckets = self.bucket_ranges(model.variables)
        if overlap and not self.buckets_partition(self.buckets, self._n):
            # e.g. a VariableStore built in another key order: the decoder hull would overlap encoder ranges and those
            # regions would be reduced twice.  One blocking all-reduce of the whole buffer is always correct.
            self.overlap = False
        model.dist = self
        # dropout masks and the sampler's noise are per replica (a TF replica draws its own); the scheduled-sampling COIN
        # stays common to all ranks (attn_decoder.py:132 draws one scalar for the whole batch)
        model.rank_seed = (self.rank * 0x9E3779B1) & 0x7FFFFFFF

    @staticmethod
    def buckets_partition(buckets, n):
        """True when the bucket ranges are pairwise disjoint and their union is exactly [0, n)."""
        pos = 0
        for lo, hi in sorted(r for _, r in buckets):
            if lo != pos or hi <= lo:
                return False
            pos = hi
        return pos == n

    @staticmethod
    def bucket_ranges(variables):
        """Contiguous [lo, hi) ranges of the flat buffer in the order their gradients become
        final during backward: all decoders first, then encoder layers from the top."""
        specs = variables._specs
        groups = {}
        for name, _, off, n in specs:
            if "/encoder/RNNLayer" in name:
                key = int(name.split("RNNLayer")[1].split("/")[0])
            else:
                key = 0                       # decoders (and anything else): first bucket
            lo, hi = groups.get(key, (off, off))
            groups[key] = (min(lo, off), max(hi, off + (n + 3) // 4 * 4))
        order = [0] + sorted((k for k in groups if k > 0), reverse=True)
        return [(k, groups[k]) for k in order if k in groups]

    def grad_ready(self, key, flat_grad):
        """Called by the model's backward when bucket `key` is final: launch its all-reduce."""
        if not self.overlap or self.world == 1:
            return
        if flat_grad.is_cuda:
            from . import ops
            ops.side_join()                  # weight gradients are produced on the side stream
        for k, (lo, hi) in self.buckets:
            if k == key:
                if k in self._done:
                    raise RuntimeError("gradient bucket %r reduced twice in one step" % (k,))
                if self.grad_bf16:
                    half = flat_grad[lo:hi].to(torch.bfloat16)
                    self._pending.append((self.comm.all_reduce(half, async_op=True), half, flat_grad[lo:hi]))
                else:
                    self._pending.append((self.comm.all_reduce(flat_grad[lo:hi], async_op=True), None, None))
                self._done.add(k)

    def all_reduce_grads(self, flat_grad):
        """Finish the exchange; returns N so the caller scales by 1/N."""
        if self.world > 1:
            if self._pending:
                for w, half, dst in self._pending:
                    if w is not None:
                        w.wait()
                    if half is not None:
                        dst.copy_(half)            # bf16 sum -> fp32 master gradient
                missing = [k for k, _ in self.buckets if k not in self._done]
                if missing:
                    raise RuntimeError("gradient buckets incomplete: %r never became ready" % (missing,))
            elif self.grad_bf16:
                half = flat_grad.to(torch.bfloat16)
                self.comm.all_reduce(half)
                flat_grad.copy_(half)
            else:
                self.comm.all_reduce(flat_grad)
        self._pending, self._done = [], set()
        return self.world

    def all_reduce_scalar_mean(self, t):
        if self.world > 1:
            self.comm.all_reduce(t)
            t /= self.world
        return t


def shard_batch(batch, rank, world):
    """Equal utterance shards (train.py bucket batches are 128/64/32: all divisible by 8)."""
    B = len(batch["logmel_len"])
    if B % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (B, world))
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    shard = {k: (v[sl] if hasattr(v, "__len__") and len(v) == B else v) for k, v in batch.items()}
    # a replica's batch is padded to ITS longest utterance, as its own padded_batch would have done
    # (speech_dataset.py:53 padded_batch): the pyramid's pad-and-reshape (encoder.py:100-115) needs T == max(len)
    t_max = int(max(shard["logmel_len"]))
    if shard["logmel"].shape[1] > t_max:
        shard["logmel"] = shard["logmel"][:, :t_max]
    return shard
