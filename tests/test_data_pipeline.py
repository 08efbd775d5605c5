"""Host-side data format, batching and scoring: TFRecord framing + SequenceExample
codec, SpeechDataset / LMDataset batching (speech_dataset.py, lm_dataset.py), bucket discovery (train.py:94-131),
transcript filtering and WER (data_utils.py:17-33, swbd_utils.py, eval_model.py:218-258)."""
import os
import struct

import numpy as np
import pytest

from e2e_asr_amd import data_utils, swbd_utils, tfrecord
from e2e_asr_amd.base_params import Bunch
from e2e_asr_amd.eval_model import Eval, edit_distance, edit_ops
from e2e_asr_amd.lm_dataset import LMDataset, write_lm_tfrecord
from e2e_asr_amd.speech_dataset import SpeechDataset, shuffle_buffer, write_speech_tfrecord


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors for CRC-32C
    assert tfrecord.crc32c(b"123456789") == 0xE3069283
    assert tfrecord.crc32c(bytes(32)) == 0x8A9136AA
    assert tfrecord.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert tfrecord.crc32c(bytes(range(32))) == 0x46DD794E


def test_record_framing_roundtrip_and_corruption(tmp_path):
    p = str(tmp_path / "x.tfrecord")
    recs = [b"", b"abc", os.urandom(1000)]
    tfrecord.write_records(p, recs)
    assert list(tfrecord.read_records(p, verify_payload=True)) == recs
    raw = bytearray(open(p, "rb").read())
    # layout of the first (empty) record: 8-byte length 0, crc, no data, crc
    assert struct.unpack("<Q", raw[:8])[0] == 0 and len(raw) == sum(16 + len(r) for r in recs)
    bad = bytearray(raw); bad[16] ^= 1                      # length field of record 2
    open(p, "wb").write(bad)
    with pytest.raises(ValueError):
        list(tfrecord.read_records(p))
    bad = bytearray(raw); bad[16 + 12 + 1] ^= 1             # payload byte of record 2
    open(p, "wb").write(bad)
    assert len(list(tfrecord.read_records(p))) == 3          # not checked by default
    with pytest.raises(ValueError):
        list(tfrecord.read_records(p, verify_payload=True))
    open(p, "wb").write(raw[:-3])
    with pytest.raises(ValueError):
        list(tfrecord.read_records(p))


def test_sequence_example_wire_format():
    """Hand-assembled bytes (field numbers of example.proto / feature.proto) parse to the expected values, incl.
    the unpacked encodings older writers emit, and negative int64."""
    def ld(fn, b): return bytes([(fn << 3) | 2, len(b)]) + b
    int_feat_packed = ld(3, ld(1, bytes([5, 0x96, 0x01])))                       # Int64List [5, 150] packed
    int_feat_unpacked = ld(3, bytes([0x08, 7, 0x08]) + b"\xff" * 9 + b"\x01")    # value=7, value=-1 as varints
    flt = np.asarray([1.5, -2.0], "<f4")
    flt_packed = ld(2, ld(1, flt.tobytes()))
    flt_unpacked = ld(2, bytes([0x0D]) + flt[:1].tobytes() + bytes([0x0D]) + flt[1:].tobytes())
    byt = ld(1, ld(1, b"sw02001-A_000098-001156"))
    ctx = b"".join(ld(1, ld(1, k) + ld(2, v)) for k, v in ((b"a", int_feat_packed), (b"b", int_feat_unpacked), (b"seg", byt)))
    fl = ld(1, ld(1, b"x") + ld(2, ld(1, flt_packed) + ld(1, flt_unpacked)))
    c, s = tfrecord.parse_sequence_example(ld(1, ctx) + ld(2, fl))
    assert c["a"].tolist() == [5, 150] and c["b"].tolist() == [7, -1] and c["seg"] == [b"sw02001-A_000098-001156"]
    assert len(s["x"]) == 2 and all(np.array_equal(v, flt) for v in s["x"])
    # the module's own writer round-trips
    rec = tfrecord.make_sequence_example({"n": np.int64(-3), "id": b"u1"}, {"f": [flt, flt * 2], "i": [np.int64(9), np.int64(1 << 40)]})
    c, s = tfrecord.parse_sequence_example(rec)
    assert c["n"].tolist() == [-3] and c["id"] == [b"u1"]
    assert np.array_equal(s["f"][1], flt * 2) and [int(v[0]) for v in s["i"]] == [9, 1 << 40]


def _corpus(rng, n, F=8):
    utts = []
    for i in range(n):
        T, L = int(rng.integers(3, 20)), int(rng.integers(2, 9))
        utts.append({"utt_id": "utt%03d" % i, "logmel": rng.standard_normal((T, F)).astype(np.float32),
                     "char": np.concatenate([[1], rng.integers(3, 30, L), [2]]), "phone": rng.integers(3, 9, L + 1)})
    return utts


def test_speech_dataset_padded_batches(tmp_path):
    rng = np.random.default_rng(0)
    utts = _corpus(rng, 11)
    f1, f2 = str(tmp_path / "dev.0"), str(tmp_path / "dev.1")
    write_speech_tfrecord(f1, utts[:6]); write_speech_tfrecord(f2, utts[6:])
    ds = SpeechDataset(Bunch(batch_size=4, feat_length=8), [f1, f2], isTraining=False, verify_crc=True)
    batches = list(ds)
    assert [len(b["utt_id"]) for b in batches] == [4, 4, 3]            # padded_batch keeps the remainder
    assert list(ds)[0]["utt_id"] == batches[0]["utt_id"]                # re-iterable (= iterator re-initialised)
    k = 0
    for b in batches:
        Tm, Lm = int(b["logmel_len"].max()), int(b["char_len"].max())
        assert b["logmel"].shape == (len(b["utt_id"]), Tm, 8) and b["logmel"].dtype == np.float32
        assert b["char"].shape == (len(b["utt_id"]), Lm) and b["char"].dtype == np.int64
        for i, uid in enumerate(b["utt_id"]):
            u = utts[k]; k += 1
            assert uid == u["utt_id"] and b["logmel_len"][i] == len(u["logmel"])
            np.testing.assert_array_equal(b["logmel"][i, :len(u["logmel"])], u["logmel"])
            assert not b["logmel"][i, len(u["logmel"]):].any()          # zero padding
            np.testing.assert_array_equal(b["char"][i, :len(u["char"])], u["char"])
            assert not b["char"][i, len(u["char"]):].any() and b["phone_len"][i] == len(u["phone"])
    with pytest.raises(ValueError):
        list(SpeechDataset(Bunch(batch_size=4, feat_length=9), [f1], isTraining=False))


def test_training_shuffle_is_a_buffered_permutation(tmp_path):
    rng = np.random.default_rng(1)
    utts = _corpus(rng, 40)
    f = str(tmp_path / "train_1k.0.0")
    write_speech_tfrecord(f, utts)
    ds = SpeechDataset(Bunch(batch_size=7, feat_length=8), [f], isTraining=True, seed=3)
    ids = [u for b in ds for u in b["utt_id"]]
    assert sorted(ids) == sorted(u["utt_id"] for u in utts) and ids != [u["utt_id"] for u in utts]
    # every pass has its own generator (the n-th child of the seed): a second pass is another permutation, a second dataset
    # with the same seed repeats both passes exactly -- also when the passes overlap in time, as in the training loop, which
    # starts the next epoch's reader once `files_exhausted` says this pass has parsed its last record
    ids2 = [u for b in ds for u in b["utt_id"]]
    assert sorted(ids2) == sorted(ids) and ids2 != ids
    ds_b = SpeechDataset(Bunch(batch_size=7, feat_length=8), [f], isTraining=True, seed=3)
    it1 = iter(ds_b)
    first = next(it1)                                  # 40 utterances < the shuffle buffer: everything is parsed before the first batch
    assert ds_b.files_exhausted
    it2 = iter(ds_b)                                   # the next pass starts while the first one still drains
    second_pass_first = next(it2)
    assert not ds_b.files_exhausted or len(utts) <= ds_b.SHUFFLE_BUFFER
    rest1 = [u for b in it1 for u in b["utt_id"]]
    rest2 = [u for b in it2 for u in b["utt_id"]]
    assert list(first["utt_id"]) + rest1 == ids and list(second_pass_first["utt_id"]) + rest2 == ids2
    # tf.data semantics: with buffer k, the element emitted at position i comes from the first i+k inputs
    out = list(shuffle_buffer(range(100), 10, np.random.default_rng(0)))
    assert sorted(out) == list(range(100)) and all(v < i + 10 for i, v in enumerate(out))


def test_lm_dataset(tmp_path):
    seqs = [[1, 5, 6, 2], [1, 9, 2], [1, 4, 4, 4, 4, 2]]
    f = str(tmp_path / "lm.0")
    write_lm_tfrecord(f, seqs)
    got = [b for b in LMDataset([f], 2, seed=0)]
    assert [b["char"].shape[0] for b in got] == [2, 1]
    rows = sorted(tuple(r[:n]) for b in got for r, n in zip(b["char"].tolist(), b["char_len"].tolist()))
    assert rows == sorted(tuple(s) for s in seqs)


def test_bucket_discovery(tmp_path):
    from e2e_asr_amd.train import Train
    rng = np.random.default_rng(2)
    for name in ("train_1k.0.a", "train_1k.0.b", "train_1k.1.a", "dev.0", "lm.0"):
        if name.startswith("lm"):
            write_lm_tfrecord(str(tmp_path / name), [[1, 3, 2]])
        else:
            write_speech_tfrecord(str(tmp_path / name), _corpus(rng, 5))
    p = Train.class_params()
    p.data_dir = p.lm_data_dir = str(tmp_path); p.feat_length = 8; p.batch_size = 3; p.buck_batch_size = [4, 2]
    tr = Train(None, p, device="cpu")
    bucks, dev = tr.get_data_sets(logging=False)
    assert [len(b.data_files) for b in bucks] == [2, 1] and [b.params.batch_size for b in bucks] == [4, 2]
    assert sum(len(b["utt_id"]) for b in bucks[0]) == 10 and bucks[0].is_training and not dev.is_training
    assert [len(b["utt_id"]) for b in dev] == [3, 2]
    (tmp_path / "subset.txt").write_text("train_1k.0.b\n")
    p.subset_file = str(tmp_path / "subset.txt")
    bucks, _ = tr.get_data_sets(logging=False)
    assert [len(b.data_files) for b in bucks] == [1, 0]
    assert [os.path.basename(f) for f in tr.get_lm_files()] == ["lm.0"]


def test_relevant_words_and_normaliser():
    norm = swbd_utils.reverse_swbd_normalizer()
    assert norm("yeah ! i @ know #") == "yeah [laughter] i [noise] know [vocalized-noise]"
    words, rel = data_utils.get_relevant_words("uh i<sp>thi- think [noise] so um")
    assert words == ["uh", "i", "thi-", "think", "[noise]", "so", "um"] and rel == ["i", "think", "so"]
    assert data_utils.get_relevant_words("") == ([], [])
    assert (data_utils.PAD_ID, data_utils.GO_ID, data_utils.EOS_ID) == (0, 1, 2)


def test_vocabulary_and_sentences(tmp_path):
    vp = tmp_path / "char.vocab"
    pieces = ["<pad>", "<go>", "<eos>", u"▁i", u"▁th", "ink", u"▁!", u"▁so"]
    vp.write_bytes(("\n".join(pieces) + "\n").encode("utf-8"))
    vocab, rev = data_utils.initialize_vocabulary(str(vp))
    assert rev[3] == u"▁i".encode("utf-8") and vocab[b"ink"] == 5
    with pytest.raises(ValueError):
        data_utils.initialize_vocabulary(str(tmp_path / "none"))
    sent = Eval.wp_array_to_sent([3, 4, 5, 6, 7, 2, 4, 4], rev, swbd_utils.reverse_swbd_normalizer())
    assert sent == "i think [laughter] so"                                  # cut at EOS, pieces joined, tag restored
    ev = Eval(None, params=Bunch(best_model_dir=str(tmp_path), vocab_dir=str(tmp_path)))
    assert ev.rev_char_vocab == rev
    assert Eval(None, params=Bunch(best_model_dir=str(tmp_path), vocab_dir="")).rev_char_vocab is None


def test_edit_distance_and_operation_counts():
    assert edit_distance("kitten", "sitting") == 3 and edit_distance([], [1, 2]) == 2 and edit_distance([1], [1]) == 0
    for a, b in (("a b c".split(), "a x c d".split()), ([], ["w"]), (["w", "v"], []), ("the cat sat".split(), "the cat sat".split())):
        d, i, dl, s = edit_ops(a, b)
        assert d == edit_distance(a, b) == i + dl + s
    assert edit_ops("a b c".split(), "a x c d".split()) == (2, 1, 0, 1)
    assert edit_ops(["w", "v"], []) == (2, 0, 2, 0)
    rng = np.random.default_rng(0)
    for _ in range(30):
        a, b = rng.integers(0, 4, rng.integers(0, 9)).tolist(), rng.integers(0, 4, rng.integers(0, 9)).tolist()
        d, i, dl, s = edit_ops(a, b)
        assert d == edit_distance(a, b) == i + dl + s and len(a) - dl + i == len(b)


class _FakeModel(object):
    """Stands in for the eval Seq2SeqModel: `forward(batch)` then `greedy_ids` returns the scripted hypotheses."""
    def __init__(self, hyps):
        self.hyps, self.k = hyps, 0
    def forward(self, batch):
        self.cur = self.hyps[self.k]; self.k += 1
    def greedy_ids(self, task):
        import torch
        return torch.tensor(self.cur)


def test_greedy_decode_wer_and_files(tmp_path):
    pieces = ["<pad>", "<go>", "<eos>", u"▁i", u"▁think", u"▁so", u"▁uh", u"▁no"]
    (tmp_path / "char.vocab").write_bytes(("\n".join(pieces) + "\n").encode("utf-8"))
    gold = np.asarray([[1, 3, 4, 5, 2, 0], [1, 7, 2, 0, 0, 0]])          # "i think so", "no"
    hyp = [np.asarray([[3, 6, 4, 7, 2, 2], [7, 7, 2, 0, 0, 0]])]         # "i uh think no" -> 1 sub; "no no" -> 1 ins
    ev = Eval(_FakeModel(hyp), params=Bunch(best_model_dir=str(tmp_path), vocab_dir=str(tmp_path)))
    err = ev.greedy_decode([{"char": gold, "utt_id": ["u1", "u2"]}])
    assert err == pytest.approx(2 / 4.0)
    assert (tmp_path / "gold_asr.txt").read_text() == "u1\ti think so\nu2\tno\n"
    assert (tmp_path / "decoded_asr.txt").read_text() == "u1\ti think no\nu2\tno no\n"      # filler dropped
    assert (tmp_path / "raw_asr.txt").read_text() == "u1\ti uh think no\nu2\tno no\n"
    # without a vocabulary: token-level rate over EOS-trimmed ids
    ev2 = Eval(_FakeModel(hyp), params=Bunch(best_model_dir="", vocab_dir=""))
    assert ev2.greedy_decode([{"char": gold}]) == pytest.approx((2 + 1) / 4.0)


def test_uniform_feature_lists_parse_in_one_shot_and_irregular_ones_fall_back():
    """tfrecord.parse_sequence_example: filterbank frames (packed FloatLists of one length) come back as ONE [steps, n] float32
    array and token id sequences (one non-negative value per step) as a [steps, 1] int64 array -- same values as the per-step
    parser; anything irregular (ragged frames, negative / multi-value / huge ints, bytes) keeps the per-step form."""
    from e2e_asr_amd import tfrecord
    rng = np.random.default_rng(3)
    frames = rng.standard_normal((37, 80)).astype(np.float32)
    ids = rng.integers(0, 30000, size=25).astype(np.int64)
    small = rng.integers(0, 100, size=9).astype(np.int64)
    ex = tfrecord.make_sequence_example(
        {"segment": [b"utt"], "logmel_len": np.array([37], np.int64)},
        {"logmel": [f for f in frames], "cint": [np.array([v], np.int64) for v in ids], "pint": [np.array([v], np.int64) for v in small],
         "ragged": [np.zeros(3, np.float32), np.zeros(5, np.float32)], "neg": [np.array([-4], np.int64), np.array([6], np.int64)],
         "multi": [np.array([1, 2], np.int64)], "big": [np.array([1 << 40], np.int64)], "names": [[b"a"], [b"bc"]],
         "one_frame": [np.arange(4, dtype=np.float32)]})
    ctx, seq = tfrecord.parse_sequence_example(ex)
    assert isinstance(seq["logmel"], np.ndarray) and seq["logmel"].dtype == np.float32 and seq["logmel"].shape == (37, 80)
    np.testing.assert_array_equal(seq["logmel"], frames)
    for name, want in (("cint", ids), ("pint", small)):
        assert isinstance(seq[name], np.ndarray) and seq[name].shape == (len(want), 1) and seq[name].dtype == np.int64
        np.testing.assert_array_equal(seq[name][:, 0], want)
    np.testing.assert_array_equal(seq["one_frame"], np.arange(4, dtype=np.float32)[None])
    assert isinstance(seq["ragged"], list) and [len(x) for x in seq["ragged"]] == [3, 5]
    assert [int(x[0]) for x in seq["neg"]] == [-4, 6] and list(seq["multi"][0]) == [1, 2] and int(seq["big"][0][0]) == 1 << 40
    assert seq["names"] == [[b"a"], [b"bc"]] and ctx["segment"] == [b"utt"] and int(ctx["logmel_len"][0]) == 37
    # truncated / corrupted frame lists must not be mis-read by the stride shortcut
    fl = bytes(tfrecord._ld(1, tfrecord._enc_feature(frames[0])) + tfrecord._ld(1, tfrecord._enc_feature(frames[1][:79])))
    assert tfrecord._uniform_float_frames(memoryview(fl)) is None


def test_int_step_shortcut_rejects_short_and_empty_trailing_features():
    """tfrecord._single_int_steps gathers header bytes of every step at once: a feature list whose LAST step is an empty or
    short Feature (`0A 00`, `0A 02 1A 00`) or a bytes list passes the entry walk and must fall back to the per-step parser
    (None), never raise IndexError out of parse_sequence_example."""
    from e2e_asr_amd import tfrecord
    one = bytes(tfrecord._ld(1, tfrecord._enc_feature(np.array([7], np.int64))))
    for tail in (b"\x0a\x00", b"\x0a\x02\x1a\x00", b"\x0a\x03\x1a\x01\x0a", bytes(tfrecord._ld(1, tfrecord._enc_feature([b"x"])))):
        assert tfrecord._single_int_steps(memoryview(one + tail)) is None
        assert tfrecord._single_int_steps(memoryview(tail)) is None
    assert tfrecord._single_int_steps(memoryview(one + one))[:, 0].tolist() == [7, 7]
    # through the public parser: an int list followed by an empty Feature step, and a bytes feature list
    ex = tfrecord.make_sequence_example({"segment": [b"u"]}, {"cint": [np.array([3], np.int64), np.array([], np.int64)],
                                                               "names": [[b"a"], [b""], [b"bcd"]]})
    _, seq = tfrecord.parse_sequence_example(ex)
    assert [list(map(int, x)) for x in seq["cint"]] == [[3], []] and seq["names"] == [[b"a"], [b""], [b"bcd"]]
