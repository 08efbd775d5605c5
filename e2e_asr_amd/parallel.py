"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference has no distributed code at all; this is the MI355X-native
addition.  Utterances are independent, so the global batch is sharded by utterance with an
equal shard per rank and a full replica of the 10.6 M parameters per GPU.  The only exchange
is the gradient all-reduce(sum), placed between tf.gradients and clip_by_global_norm in the
seq2seq_model.py:148-155 sequence so that the clip sees the global-batch gradient; 1/N is
folded into the fused clip+Adam kernel.  Because loss = mean over utterances of a
length-normalised cost (losses.py:32-35), equal shards make mean-of-means exact.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce is bound by ONE link,
so fewer, larger messages win.  Default: ONE blocking all-reduce of the flat fp32 gradient buffer
(42.5 MB) after backward.

Overlap mode (`overlap=True` / ASR_DP_OVERLAP=1) obeys the rule every persistent kernel of this
library needs (never a second resident, spinning kernel next to a persistent
launch -- RCCL's channel kernels spin on their peers exactly as the recurrent kernels spin on their
group, and two such kernels can starve each other's workgroup dispatch): no collective is ever in
flight while a persistent kernel runs or can still be launched.  The only window of the backward
pass without one is its TAIL: after the last (lowest-layer) BPTT, while that layer's weight-gradient
GEMMs finish on the side stream.  So the buckets that are final by then -- decoders and every
encoder layer above the lowest, ~93 % of the bytes at config 2 -- are all-reduced in that window,
gated on (a) the main stream up to and including the last BPTT and (b) the side stream up to the
second-lowest layer's weight gradients; the lowest layer's bucket follows after the side join.  The
next step's first persistent kernel is ordered behind `all_reduce_grads` (the optimizer needs the
sums), so nothing persistent can start while a collective is still running.
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
        the whole flat gradient after backward -- 42.5 MB, well under a millisecond of a ~9 ms step,
        and the simplest thing that is correct by construction.  overlap=True all-reduces the buckets that
        are final in the tail window of the backward pass (module docstring); never next to a persistent kernel."""
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
        # world 1 normally skips the exchange; force_exchange (ASR_DP_FORCE_EXCHANGE=1) runs the collectives anyway, so that a
        # one-GPU box can put real RCCL calls next to the persistent kernels (tests/test_gpu_dp_train.py)
        self.force_exchange = os.environ.get("ASR_DP_FORCE_EXCHANGE", "0") == "1"
        self._pending = []
        self._done = set()
        self._ready = []
        self._reported = set()
        self._xs = None            # stream the tail-window collectives are launched from (overlap mode, CUDA)
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
        # stays common to all ranks (attn_decoder.py:132 draws one scalar for the whole batch): it is a pure
        # function of (coin_seed, task, global step, output step) -- attn_decoder.sampling_coins -- so shards that end at
        # different longest targets cannot drift apart (tests/test_parallel_gloo.py::test_sampling_coin_is_common_across_ranks)
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

    def _exchange(self, flat_grad, lo, hi, async_op):
        """All-reduce flat_grad[lo:hi] (bf16 on the wire when grad_bf16); returns the pending record."""
        if self.grad_bf16:
            half = flat_grad[lo:hi].to(torch.bfloat16)
            return (self.comm.all_reduce(half, async_op=async_op), half, flat_grad[lo:hi])
        return (self.comm.all_reduce(flat_grad[lo:hi], async_op=async_op), None, None)

    def begin_step(self):
        """Called at the top of every backward pass: a backward that raised mid-step, or two backward passes before one
        exchange, must not leave last step's bucket reports behind.  Collectives still pending are finished first (their
        sums land in a gradient buffer the new backward pass zeroes anyway), never abandoned in flight."""
        for w, _, _ in self._pending:
            if w is not None:
                w.wait()
        self._pending, self._done, self._ready, self._reported = [], set(), [], set()

    def grad_ready(self, key, flat_grad):
        """Called by the model's backward when bucket `key` is final on the caller's + side streams: encoder layers
        top-down (key = depth) as each layer's BPTT and weight-gradient GEMMs have been ENQUEUED, then key 0 after the side
        join.  Overlap mode launches nothing until the lowest encoder layer reports (see the module docstring)."""
        if not self.overlap or self.world == 1 and not self.force_exchange:
            return
        if key in self._reported:
            raise RuntimeError("gradient bucket %r reported ready twice in one step" % (key,))
        self._reported.add(key)
        lowest = min(k for k, _ in self.buckets if k > 0) if any(k > 0 for k, _ in self.buckets) else 0
        cuda = flat_grad.is_cuda
        if key > lowest:
            # side-stream work queued so far = this layer's (and every earlier bucket's) weight gradients
            self._ready.append(key)
            if cuda:
                from . import ops
                if self._xs is None:
                    self._xs = torch.cuda.Stream(device=flat_grad.device)
                ops.side_wait(self._xs)
            return
        if key == lowest and key > 0:
            # the last persistent kernel of the backward pass is enqueued on the caller's stream: open the tail window
            if not self._ready:
                return
            ready = [0] + self._ready if any(k == 0 for k, _ in self.buckets) else list(self._ready)
            if cuda:
                main = torch.cuda.current_stream(flat_grad.device)
                self._xs.wait_stream(main)                 # ... up to and including the lowest layer's BPTT
                with torch.cuda.stream(self._xs):          # torch's process group orders the collective behind this stream
                    for k, (lo, hi) in self.buckets:
                        if k in ready:
                            rec = self._exchange(flat_grad, lo, hi, True)
                            if rec[1] is not None:
                                rec[1].record_stream(main)
                            self._pending.append(rec)
            else:
                for k, (lo, hi) in self.buckets:
                    if k in ready:
                        self._pending.append(self._exchange(flat_grad, lo, hi, True))
            self._done.update(ready)
            self._ready = []
            return
        # key 0 (or a model without encoder buckets): everything is final and joined on the caller's stream
        for k, (lo, hi) in self.buckets:
            if k not in self._done:
                self._pending.append(self._exchange(flat_grad, lo, hi, True))
                self._done.add(k)
        self._ready = []

    def all_reduce_grads(self, flat_grad):
        """Finish the exchange; returns N so the caller scales by 1/N."""
        if self.world > 1 or self.force_exchange:
            if self._pending:
                if self._xs is not None and flat_grad.is_cuda:
                    # what was launched from the exchange stream (bf16 staging, a comm object that works eagerly) before the sums are read
                    torch.cuda.current_stream(flat_grad.device).wait_stream(self._xs)
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
        self._pending, self._done, self._ready, self._reported = [], set(), [], set()
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
