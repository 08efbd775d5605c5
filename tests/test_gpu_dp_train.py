"""GPU: the data-parallel hooks of the HIP train step with N = 2 on ONE card, and the training-loop branches that
interleave the char LM (train.py:269-291) and resume it from a checkpoint.

Two GPUs are not available to the tests, so rank 1 and rank 0 run one after the other in this process through a
loop-back `comm` object: rank 1's gradient buckets are RECORDED as `DataParallel` hands them to the exchange, then rank
0's exchange ADDS them -- exactly the sum an RCCL all-reduce would leave on rank 0.  Everything else is the product
path: `Encoder.backward(on_layer_done)` -> `grad_ready` -> `ops.side_join()`, the bucket ranges, and the 1/N fold inside
`asr_clip_adam_f32` (seq2seq_model.py:148-155 ordering: gradients -> exchange -> clip -> Adam)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class LoopbackComm(object):
    """world = 2.  mode "record": all_reduce clones the tensor it is given (keyed by call order); mode "replay": adds the
    clone recorded at the same call index.  Calls are made in the same order on both ranks (bucket order is fixed)."""

    def __init__(self, rank, store):
        self.world, self.rank, self.store, self.calls = 2, rank, store, 0
        self.mode = "record"

    def broadcast(self, t, src=0):
        pass                                   # both models are built from the same seed

    def all_reduce(self, t, async_op=False):
        if self.mode == "record":
            self.store.append(t.detach().clone())
        else:
            peer = self.store[self.calls]
            assert peer.shape == t.shape
            t.add_(peer)
        self.calls += 1
        return None


def _params(samp=0.0):
    from e2e_asr_amd.attn_decoder import AttnDecoder
    from e2e_asr_amd.seq2seq_model import Seq2SeqModel
    p = Seq2SeqModel.class_params()
    p.num_layers = {"char": 3}; p.max_output = {"char": 12}
    p.encoder_params.use_lstm = True; p.encoder_params.hidden_size = 64; p.encoder_params.out_prob = 1.0
    dp = AttnDecoder.class_params()
    dp.hidden_size_dec = 64; dp.lm_hidden_size = 64; dp.emb_size = 24; dp.attention_vec_size = 16; dp.vocab_size = 40
    dp.out_prob_dec = 1.0; dp.samp_prob = samp
    p.decoder_params = {"char": dp}
    return p


@pytest.mark.parametrize("overlap", [False, True])
def test_two_rank_step_equals_one_global_batch_step(overlap):
    """Two half-batches through DataParallel (N = 2) must leave rank 0 with the weights of ONE step on the global batch:
    loss = mean over utterances of a length-normalised cost (losses.py:32-35), equal shards, sum of shard gradients
    scaled by 1/N inside the fused clip+Adam kernel, clip on the global-batch gradient.  overlap=True sends the
    buckets that are final after the last BPTT in the tail window (from the exchange stream), the lowest layer's last."""
    from e2e_asr_amd import ops
    from e2e_asr_amd.parallel import DataParallel, shard_batch
    from e2e_asr_amd.seq2seq_model import Seq2SeqModel
    from e2e_asr_amd.weights import synthetic_batch
    p = _params()
    gb = synthetic_batch(B=8, T=40, F=20, t_dec=9, vocab=40, variable_len=True, seed=5)
    single = Seq2SeqModel(None, True, p, device=DEV, feat_length=20, seed=4)
    for step in range(2):                                        # two steps: the Adam moments matter in the second
        gb["logmel"] = gb["logmel"] * (1.0 + 0.1 * step)
        loss_g = single.step(gb)["char"].item()
        if step == 0:
            store = []
            ranks = [Seq2SeqModel(None, True, p, device=DEV, feat_length=20, seed=4) for _ in range(2)]
            dps = [DataParallel(m, overlap=overlap, comm=LoopbackComm(r, store)) for r, m in enumerate(ranks)]
            assert dps[0].buckets_partition(dps[0].buckets, ranks[0].variables.flat.numel())
        else:
            # rank 1 took a wrong (un-reduced) update in step 0: give it rank 0's state, as a real all-reduce would have
            ranks[1].variables.flat.copy_(ranks[0].variables.flat)
            for slot, (m_, v_) in ranks[0].variables.adam_slots.items():
                m1, v1 = ranks[1].variables.ensure_adam(slot)
                m1.copy_(m_); v1.copy_(v_)
        del store[:]
        dps[1].comm.mode, dps[1].comm.calls = "record", 0
        dps[0].comm.mode, dps[0].comm.calls = "replay", 0
        l1 = ranks[1].step(shard_batch(gb, 1, 2))["char"].item()
        n_calls = dps[1].comm.calls
        l0 = ranks[0].step(shard_batch(gb, 0, 2))["char"].item()
        ops.check_device_flag(torch.device(DEV))
        assert dps[0].comm.calls == n_calls == (len(dps[0].buckets) if overlap else 1)
        np.testing.assert_allclose(0.5 * (l0 + l1), loss_g, rtol=1e-5)      # mean of shard means = global mean
        a, b = ranks[0].variables.flat.cpu().numpy(), single.variables.flat.cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-5)
        np.testing.assert_allclose(np.sqrt(ranks[0]._gnorm_sq.item()) / 2.0, np.sqrt(single._gnorm_sq.item()), rtol=1e-4)
    assert ranks[0].global_step == single.global_step == 2
    assert ranks[0].rank_seed != ranks[1].rank_seed              # replicas draw their own dropout masks / sampler noise


def test_two_rank_sampling_coin_is_common_with_ragged_shards():
    """attn_decoder.py:131-133: ONE uniform per output step for the whole batch; SURVEY 8e: the same coin on all ranks.
    Two ranks (loop-back comm, HIP path) with samp_prob = 0.3 on shards whose longest targets differ, three optimizer
    steps: the coin vectors agree on every output step both ranks run, change from step to step, and tokens really are fed
    back.  (A per-process generator consumed t_out draws per step and drifted apart from step 2 on.)"""
    from e2e_asr_amd import ops
    from e2e_asr_amd.parallel import DataParallel, shard_batch
    from e2e_asr_amd.seq2seq_model import Seq2SeqModel
    from e2e_asr_amd.weights import synthetic_batch
    p = _params(samp=0.3)
    store = []
    ranks = [Seq2SeqModel(None, True, p, device=DEV, feat_length=20, seed=4) for _ in range(2)]
    dps = [DataParallel(m, overlap=False, comm=LoopbackComm(r, store)) for r, m in enumerate(ranks)]
    seen, fed_any = [], 0
    for step in range(3):
        gb = synthetic_batch(B=8, T=40, F=20, t_dec=12, vocab=40, variable_len=True, seed=50 + step)
        ln = np.asarray(gb["char_len"])
        ln[:4] = np.minimum(ln[:4], 5 + step)          # rank 0's shard ends earlier than rank 1's (whose last row has 11)
        for b in range(4):
            gb["char"][b, ln[b]:] = 0; gb["char"][b, ln[b]] = 2
        del store[:]
        dps[1].comm.mode, dps[1].comm.calls = "record", 0
        dps[0].comm.mode, dps[0].comm.calls = "replay", 0
        coins, t_outs = [None, None], [0, 0]
        for r in (1, 0):
            sb = shard_batch(gb, r, 2)
            ranks[r].forward(sb)
            dec = ranks[r].decoder["char"]
            coins[r], t_outs[r] = dec.last_coin.copy(), dec.saved["t_out"]
            tok = dec.saved["ws"]["tok"].cpu().numpy()
            teacher = np.asarray(sb["char"]).T[:tok.shape[0]]
            fed_any += int((tok[1:t_outs[r]] != teacher[1:t_outs[r]]).sum())
            ranks[r].backward()
            ranks[r].apply_gradients()
        ops.check_device_flag(torch.device(DEV))
        assert t_outs[0] != t_outs[1] and len(coins[0]) == t_outs[0] and len(coins[1]) == t_outs[1]
        n = min(t_outs)
        np.testing.assert_array_equal(coins[0][:n], coins[1][:n])
        np.testing.assert_array_equal(coins[0][:n] < 0.7, coins[1][:n] < 0.7)
        seen.append(coins[1][:5].copy())
    assert not np.array_equal(seen[0], seen[1]) and not np.array_equal(seen[1], seen[2])
    assert fed_any > 0                                  # scheduled sampling really fed drawn tokens
    assert ranks[0].global_step == ranks[1].global_step == 3


def test_rccl_world1_steps_with_and_without_overlap():
    """RCCL itself next to the persistent kernels, as far as one GPU allows: a child process initialises the "nccl" backend at
    world size 1 and runs config-2-width train steps plain, with the blocking exchange, with the tail overlap and with the
    bf16 exchange (tests/_rccl_world1.py; ASR_DP_FORCE_EXCHANGE=1 issues the collectives although world == 1)."""
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "_rccl_world1.py"), str(port)], env=env, capture_output=True,
                       text=True, timeout=300)
    if r.returncode != 0 or "rccl world-1 ok" not in r.stdout:
        out_dir = os.path.join(os.path.dirname(here), "gpurun_out")
        if os.path.isdir(out_dir):
            with open(os.path.join(out_dir, "rccl_world1_child.log"), "w") as f:
                f.write(r.stdout + "\n---- stderr ----\n" + r.stderr)
        lines = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "assert" in l.lower() or "Traceback" in l]
        raise AssertionError("child rc %d\n%s\n...\n%s" % (r.returncode, "\n".join(lines[-20:]), r.stderr[-1500:]))


def test_bucket_overlap_falls_back_when_ranges_do_not_partition():
    """A variable store whose decoder variables are NOT contiguous (another key order) would make the decoder bucket's
    hull overlap encoder ranges; DataParallel must notice and use the single blocking all-reduce."""
    from e2e_asr_amd.parallel import DataParallel
    from e2e_asr_amd.seq2seq_model import Seq2SeqModel
    from e2e_asr_amd.variables import VariableStore
    p = _params()
    m = Seq2SeqModel(None, True, p, device=DEV, feat_length=20, seed=4)
    arrays = m.variables.to_arrays()
    names = list(arrays)
    dec = [n for n in names if "rnn_decoder" in n]
    enc = [n for n in names if "rnn_decoder" not in n]
    shuffled = dec[:3] + enc + dec[3:]
    m2 = Seq2SeqModel(None, True, p, variables=VariableStore.from_arrays({n: arrays[n] for n in shuffled}, DEV), device=DEV,
                      feat_length=20)
    dp = DataParallel(m2, overlap=True, comm=LoopbackComm(0, []))
    assert dp.overlap is False
    assert DataParallel(m, overlap=True, comm=LoopbackComm(0, [])).overlap is True


class _Stop(Exception):
    pass


def _lm_batches(rng, n, B=6, T=9, V=12):
    out = []
    for _ in range(n):
        lens = rng.integers(2, T + 1, B); lens[0] = T
        ids = np.zeros((B, T + 1), np.int64)
        for b in range(B):
            ids[b, :lens[b] + 1] = rng.integers(1, V, lens[b] + 1)
        out.append({"char": ids, "char_len": lens})
    return out


def test_train_loop_lm_interleave_and_resume_of_lm_state(tmp_path, monkeypatch):
    """train.py:269-291: with lm_prob > 0 a coin per iteration chooses an LM step (own optimizer AdamLM, own step counter
    and learning rate) or an ASR step.  (1) lm_prob = 1: only LM steps run, so only the variables the LM shares with the
    decoder move and global_step stays 0.  (2) A run interrupted at a checkpoint and resumed must continue the LM
    exactly: lm_global_step, the LM learning rate and both epoch counters come back, so the next LM update equals the
    uninterrupted run's (AdamLM bias correction continues at t+1 instead of restarting at 1)."""
    from e2e_asr_amd import checkpoint
    from e2e_asr_amd.lm_encoder import LMEncoder
    from e2e_asr_amd.lm_model import LMModel
    from e2e_asr_amd.train import Train
    from e2e_asr_amd.weights import synthetic_batch
    p = _params()
    p.num_layers = {"char": 2}; p.max_output = {"char": 8}; p.decoder_params["char"].vocab_size = 12

    def tparams(d, lm_prob):
        tp = Train.class_params()
        tp.train_dir = str(tmp_path / d); tp.best_model_dir = str(tmp_path / d / "best")
        tp.steps_per_checkpoint = 3; tp.feat_length = 20; tp.max_epochs = 100; tp.min_steps = 0; tp.lm_prob = lm_prob
        ep = LMEncoder.class_params()
        ep.out_prob = 1.0; ep.lm_hidden_size = 64; ep.proj_size = 64; ep.emb_size = 24; ep.vocab_size = 12
        tp.lm_enc_params = ep
        tp.lm_params = LMModel.class_params()
        return tp
    b0 = synthetic_batch(B=4, T=16, F=20, t_dec=9, vocab=12, seed=1)
    rng = np.random.default_rng(3)
    lm_set = _lm_batches(rng, 4)

    # (1) lm_prob = 1: the ASR branch is never taken; stop the loop from the LM side after a few steps
    tp = tparams("lm_only", 1.0)
    tr = Train(p, tp, device=DEV)
    taken = []
    orig = LMModel.step

    def counting_step(self, batch=None):
        taken.append(self.lm_global_step)
        if len(taken) > 5:
            raise _Stop()
        return orig(self, batch)
    monkeypatch.setattr(LMModel, "step", counting_step)
    with pytest.raises(_Stop):
        tr.train([[b0] * 4], [b0], lm_set=lm_set)
    monkeypatch.setattr(LMModel, "step", orig)
    from e2e_asr_amd.seq2seq_model import Seq2SeqModel
    fresh = Seq2SeqModel(None, True, p, device=DEV, feat_length=20).variables.to_arrays()
    after = tr.model.variables.to_arrays()
    moved = sorted(k for k in after if not np.array_equal(after[k], fresh[k]))
    pre = "model/rnn_decoder_char/"
    assert moved == sorted(pre + l for l in ("decoder/embedding", "rnn/basic_lstm_cell/kernel", "rnn/basic_lstm_cell/bias",
                                             "rnn/OutputProjection/kernel", "rnn/OutputProjection/bias"))
    assert tr.model.global_step == 0 and taken == [0, 1, 2, 3, 4, 5]        # 4 LM batches per LM epoch: the iterator restarted

    # (2) interleaved run up to a checkpoint; then the SAME LM step once in the live process and once after a resume
    tp = tparams("split", 0.5)
    tr_a = Train(p, tp, device=DEV)
    tr_a.train([[b0] * 50], [b0], lm_set=lm_set, max_steps=3)       # returns right after the step-3 checkpoint
    ck = checkpoint.load(open(os.path.join(tp.train_dir, "checkpoint.txt")).read().strip())
    lm_t = int(ck["lm_global_step"])
    assert int(ck["global_step"]) == 3 and lm_t == tr_a.lm_model.lm_global_step and lm_t > 0
    assert float(ck["lm_learning_rate"]) == tr_a.lm_model.learning_rate and "epoch" in ck and "lm_epoch" in ck
    assert any(k.endswith("/AdamLM") for k in ck)
    tr_a.lm_model.learning_rate_decay_op()                          # a decayed LM rate must survive too
    checkpoint.save(os.path.join(tp.train_dir, "asr.ckpt-3"), tr_a.model.variables, 3, tr_a.model.learning_rate,
                    extra=dict(epoch=tr_a.model.epoch, lm_global_step=lm_t, lm_learning_rate=tr_a.lm_model.learning_rate,
                               lm_epoch=tr_a.lm_model.epoch))
    tr_c = Train(p, tparams("split", 0.5), device=DEV)
    tr_c.train([[b0] * 50], [b0], lm_set=lm_set, max_steps=0)       # restore only
    assert tr_c.model.global_step == 3 and tr_c.lm_model.lm_global_step == lm_t
    assert tr_c.lm_model.learning_rate == tr_a.lm_model.learning_rate == 0.5e-4
    assert tr_c.lm_model.epoch == tr_a.lm_model.epoch
    x = _lm_batches(np.random.default_rng(9), 1)[0]
    la = tr_a.lm_model.step(x).item()
    lc = tr_c.lm_model.step(x).item()
    assert la == pytest.approx(lc, rel=1e-6)
    wa, wc = tr_a.model.variables.flat.cpu().numpy(), tr_c.model.variables.flat.cpu().numpy()
    np.testing.assert_allclose(wc, wa, rtol=0, atol=1e-6)           # AdamLM continued at t+1 with the restored moments
    # without the restored step counter the update would differ by the bias-correction ratio (~1.5x at this t)
    assert tr_c.lm_model.lm_global_step == lm_t + 1


def test_train_from_tfrecord_buckets_end_to_end(tmp_path):
    """The whole caller side of the path on real files: TFRecord buckets `train_1k.<k>.*` / `dev*` / `lm*` written in the
    reference's SequenceExample layout (speech_dataset.py:15-45, lm_dataset.py:12-31) -> Train.get_data_sets / get_lm_set ->
    smallest-bucket-first schedule with the LM interleaved (train.py:94-131, 261-295) -> HIP train steps -> greedy dev decode,
    asr_err.txt / best.txt / checkpoints under the TF variable names.  No dataset is injected."""
    from e2e_asr_amd import checkpoint
    from e2e_asr_amd.lm_dataset import write_lm_tfrecord
    from e2e_asr_amd.lm_encoder import LMEncoder
    from e2e_asr_amd.lm_model import LMModel
    from e2e_asr_amd.speech_dataset import write_speech_tfrecord
    from e2e_asr_amd.train import Train
    rng = np.random.default_rng(11)
    F, V = 8, 12

    def corpus(n, tmin, tmax):
        utts = []
        for i in range(n):
            T, L = int(rng.integers(tmin, tmax)), int(rng.integers(2, 7))
            ch = np.concatenate([[1], rng.integers(3, V, L), [2]])
            utts.append({"utt_id": "u%d_%d" % (tmin, i), "logmel": rng.standard_normal((T, F)).astype(np.float32), "char": ch,
                         "char_len": len(ch) - 1, "phone": rng.integers(3, 9, L + 1), "phone_len": L})
        return utts
    data = tmp_path / "data"; data.mkdir()
    write_speech_tfrecord(str(data / "train_1k.0.a"), corpus(8, 9, 16))        # bucket 0: short utterances
    write_speech_tfrecord(str(data / "train_1k.1.a"), corpus(6, 20, 30))       # bucket 1: longer ones
    write_speech_tfrecord(str(data / "dev.0"), corpus(5, 9, 30))
    write_lm_tfrecord(str(data / "lm.0"), [np.concatenate([[1], rng.integers(3, V, int(rng.integers(2, 8))), [2]]).tolist()
                                           for _ in range(10)])
    p = _params()
    p.num_layers = {"char": 2}; p.max_output = {"char": 8}; p.decoder_params["char"].vocab_size = V
    tp = Train.class_params()
    tp.data_dir = tp.lm_data_dir = str(data)
    tp.train_dir = str(tmp_path / "run"); tp.best_model_dir = str(tmp_path / "run" / "best")
    tp.feat_length = F; tp.batch_size = 3; tp.buck_batch_size = [4, 2]; tp.steps_per_checkpoint = 3
    tp.max_epochs = 100; tp.min_steps = 0; tp.lm_prob = 0.3
    ep = LMEncoder.class_params()
    ep.out_prob = 1.0; ep.lm_hidden_size = 64; ep.proj_size = 64; ep.emb_size = 24; ep.vocab_size = V
    tp.lm_enc_params = ep
    tp.lm_params = LMModel.class_params(); tp.lm_params.lm_batch_size = 4
    tr = Train(p, tp, device=DEV)
    model = tr.train(max_steps=7)
    assert model.global_step == 7 and tr.lm_model.lm_global_step >= 1
    errs = [float(l) for l in open(os.path.join(tp.train_dir, "asr_err.txt"))]
    assert len(errs) == 2 and all(e >= 0.0 for e in errs)
    ck = checkpoint.load(open(os.path.join(tp.train_dir, "checkpoint.txt")).read().strip())
    assert int(ck["global_step"]) == 6 and "model/encoder/RNNLayer2/bidirectional_rnn/bw/basic_lstm_cell/kernel" in ck
    # bucket schedule: the first epoch consumes bucket 0 (8 utterances / 4 = 2 batches) before bucket 1 (3 batches of 2)
    assert model.epoch >= 1
