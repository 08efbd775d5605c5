"""CPU, world_size 2, gloo: the data-parallel host logic (bucketed flat-gradient all-reduce,
1/N folded into the update, equal utterance shards) reproduces the single-process
global-batch gradient.  Gradients come from the torch oracle twin (no GPU here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from e2e_asr_amd.weights import init_weights, synthetic_batch


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _FakeModel(object):
    def __init__(self, variables):
        self.variables = variables
        self.dist = None


def _worker(rank, world, port, q, rank_mode='overlap'):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from e2e_asr_amd.parallel import DataParallel, shard_batch
        from e2e_asr_amd.variables import VariableStore
        from oracle import torch_ref as R
        w = init_weights(feat=10, hidden=8, depth=2, vocab={"char": 13}, emb=8, hidden_dec=8, lm_hidden=8,
                         attn_vec=4, seed=1 + rank)              # ranks start DIFFERENT: broadcast must fix it
        st = VariableStore.from_arrays(w, "cpu")
        model = _FakeModel(st)
        dp = DataParallel(model, overlap=(rank_mode.startswith('overlap')), grad_dtype=('bf16' if rank_mode.endswith('bf16') else 'f32'))
        batch = synthetic_batch(B=4, T=9, F=10, t_dec=6, vocab=13, variable_len=True, seed=5)
        mine = shard_batch(batch, rank, world)
        mine["logmel"] = mine["logmel"].astype(np.float64)
        W = R.weights_to_torch(st.to_arrays())
        total, _, _ = R.seq2seq_loss(mine, W, num_layers={"char": 2})
        total.backward()
        st.ensure_grad()
        for name in st.names():
            st.grad_of(name).copy_(W[name].grad.float())
        # the model's order (Seq2SeqModel.backward): encoder layers top-down as their BPTT is enqueued, key 0 after the join
        dp.grad_ready(2, st.grad); dp.grad_ready(1, st.grad); dp.grad_ready(0, st.grad)
        n = dp.all_reduce_grads(st.grad)
        avg = {k: (st.grad_of(k) / n).numpy().copy() for k in st.names()}
        q.put((rank, st.to_arrays(), avg, [b for b in dp.buckets]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["overlap", "blocking", "overlap_bf16", "blocking_bf16"])
def test_dp_two_ranks_equals_global_batch(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, w0, g0, buckets), (_, w1, g1, _) = res
    for k in w0:                                     # broadcast made the replicas identical
        np.testing.assert_array_equal(w0[k], w1[k])
        np.testing.assert_array_equal(g0[k], g1[k])  # and both ranks hold the same reduced gradient
    # single-process reference on the whole batch
    from oracle import torch_ref as R
    batch = synthetic_batch(B=4, T=9, F=10, t_dec=6, vocab=13, variable_len=True, seed=5)
    batch["logmel"] = batch["logmel"].astype(np.float64)
    W = R.weights_to_torch({k: v.astype(np.float64) for k, v in w0.items()})
    # per-shard decoders stop at their own max target length; the global batch pads with zero-weight steps
    total, _, _ = R.seq2seq_loss(batch, W, num_layers={"char": 2})
    total.backward()
    for k in w0:
        ref = W[k].grad.numpy()
        if mode.endswith("bf16"):     # the exchange carried bfloat16: each shard gradient rounded once (2^-9), the sum once more
            np.testing.assert_allclose(g0[k], ref, rtol=0, atol=8e-3 * max(1e-6, np.abs(ref).max()))
            assert g0[k].dtype == np.float32
        else:
            np.testing.assert_allclose(g0[k], ref, rtol=0, atol=2e-6)
    # buckets tile the flat buffer without overlap, decoders first then layers top-down
    keys = [b[0] for b in buckets]
    assert keys == [0, 2, 1]
    spans = sorted(b[1] for b in buckets)
    assert spans[0][0] == 0 and all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))


def test_shard_batch_rejects_uneven():
    from e2e_asr_amd.parallel import shard_batch
    b = synthetic_batch(B=6, T=4, F=2, t_dec=4, vocab=9)
    with pytest.raises(ValueError):
        shard_batch(b, 0, 4)
    s = shard_batch(b, 1, 3)
    assert len(s["logmel_len"]) == 2 and s["logmel"].shape[0] == 2
