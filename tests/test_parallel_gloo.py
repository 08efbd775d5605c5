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


def _coin_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from e2e_asr_amd.attn_decoder import AttnDecoder
        from e2e_asr_amd.parallel import shard_batch
        dp = AttnDecoder.class_params()
        dp.samp_prob = 0.3
        dec = AttnDecoder(True, dp, scope="char")
        assert dec.feedback_mode() == 2
        out = []
        for step in range(4):
            # ragged targets whose per-shard maxima differ (rank 0's shard holds the longest target in even steps,
            # rank 1's in odd ones), so the ranks ask for different numbers of coins every step
            gb = synthetic_batch(B=8, T=12, F=4, t_dec=30, vocab=40, variable_len=True, seed=100 + step)
            ln = np.asarray(gb["char_len"]).copy()
            ln[:] = np.minimum(ln, 11 + step)
            ln[(step % 2) * 4] = 25 + step
            gb["char_len"] = ln
            mine = shard_batch(gb, rank, world)
            t_out = int(np.max(mine["char_len"]))
            dec.coin_step = step                  # what Seq2SeqModel.forward does with its global step
            coin = dec.draw_coins(t_out)
            fed = ~(coin < 1.0 - dp.samp_prob)    # the feedback mask the kernels derive (multi_decoder.py / decoder.hip)
            mine_len = torch.tensor([t_out])
            lens = [torch.zeros(1, dtype=torch.long) for _ in range(world)]
            dist.all_gather(lens, mine_len)
            pad = torch.zeros(64, dtype=torch.float64); pad[:t_out] = torch.from_numpy(coin)
            coins = [torch.zeros(64, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(coins, pad)
            out.append((t_out, [int(l) for l in lens], [c.numpy() for c in coins], fed))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_sampling_coin_is_common_across_ranks():
    """attn_decoder.py:131-133 draws ONE uniform per step for the whole batch; SURVEY 8e: the same coin on every rank.
    Shards padded to their own longest target ask for different numbers of coins -- the masks must still agree at
    every step both ranks run, at every optimizer step, and change from step to step."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_coin_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    seen = []
    differing_lengths = 0
    for step in range(4):
        t0, lens, coins, fed0 = res[0][step]
        t1, lens1, coins1, fed1 = res[1][step]
        assert lens == lens1 == [t0, t1]
        differing_lengths += t0 != t1
        n = min(t0, t1)
        np.testing.assert_array_equal(coins[0][:n], coins[1][:n])          # what each rank gathered from both
        np.testing.assert_array_equal(coins1[0][:n], coins1[1][:n])
        np.testing.assert_array_equal(fed0[:n], fed1[:n])                  # the feedback masks themselves
        assert 0.0 <= coins[0][:n].min() and coins[0][:n].max() < 1.0
        seen.append(coins[0][:10].copy())
    assert differing_lengths == 4
    for a in range(4):
        for b in range(a + 1, 4):
            assert not np.array_equal(seen[a], seen[b])                    # a fresh coin vector every optimizer step


def test_sampling_coins_prefix_stable_and_per_task():
    from e2e_asr_amd.attn_decoder import AttnDecoder, sampling_coins
    a, b = sampling_coins(0, 7, 3, 5), sampling_coins(0, 7, 3, 120)
    np.testing.assert_array_equal(a, b[:5])
    assert not np.array_equal(sampling_coins(0, 7, 4, 5), a) and not np.array_equal(sampling_coins(1, 7, 3, 5), a)
    c, p = AttnDecoder(True, scope="char"), AttnDecoder(True, scope="phone")
    assert c.coin_stream != p.coin_stream                                  # each task's decoder draws its own (one op per decoder graph)
    x = c.draw_coins(9); y = c.draw_coins(9)                               # stand-alone use: a new vector per call
    assert not np.array_equal(x, y) and c.coin_step == 2
