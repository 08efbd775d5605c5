"""Speech data sets from TFRecord files -- the role of speech_dataset.py:5-63 without TensorFlow.

Each record is a SequenceExample with context `segment` (bytes), `logmel_len`, `cint_len`, `pint_len` (int64) and
feature lists `logmel` ([feat_length] floats per frame), `cint`, `pint` (one int64 per step) (speech_dataset.py:15-24).
`SpeechDataset(params, data_files, isTraining)` is a RE-ITERABLE batch source (each `iter()` is one pass = the
reference's `sess.run(data_iter.initializer)`; exhaustion = its `OutOfRangeError`): training shuffles through a
4000-record buffer (speech_dataset.py:53) and every batch is zero-padded to its longest member
(`padded_batch`, :54-58; the last batch may be smaller).  Batches are the dicts `Seq2SeqModel.get_batch` consumes:
logmel [B,T,F] f32, char/phone [B,L] int64 (batch-major, like the iterator of the reference), *_len [B], utt_id [B].
"""
import threading
import numpy as np

from . import tfrecord


def _pad_stack(arrs, tail_shape=(), dtype=None):
    n = max((len(a) for a in arrs), default=0)
    out = np.empty((len(arrs), n) + tuple(tail_shape), dtype=dtype or arrs[0].dtype)
    for i, a in enumerate(arrs):                  # (zeroing only the padding: np.zeros + copy touched the 8 MB batch twice, 4.3 vs 0.6 ms)
        out[i, :len(a)] = a
        out[i, len(a):] = 0
    return out


def shuffle_buffer(items, buffer_size, rng):
    """tf.data `shuffle(buffer_size)`: keep a buffer, emit a uniformly chosen element, refill from the stream."""
    buf = []
    for it in items:
        if len(buf) < buffer_size:
            buf.append(it)
            continue
        j = int(rng.integers(len(buf)))
        out, buf[j] = buf[j], it
        yield out
    while buf:
        j = int(rng.integers(len(buf)))
        buf[j], buf[-1] = buf[-1], buf[j]
        yield buf.pop()


class SpeechDataset(object):
    SHUFFLE_BUFFER = 4000

    def __init__(self, params, data_files, isTraining, seed=None, verify_crc=False):
        self.params = params                      # batch_size, feat_length
        self.is_training = isTraining
        self.data_files = list(data_files)
        self.verify_crc = verify_crc
        # every pass over the dataset shuffles with its OWN generator, the n-th child of the seed: the training loop starts the
        # next epoch's reader while this epoch's shuffle buffer is still draining (train.py), and two passes drawing from one
        # generator would make the order depend on thread timing
        self._seed_seq = np.random.SeedSequence(seed)
        self._spawn_lock = threading.Lock()
        self.files_exhausted = False              # the newest pass has read its last record (its shuffle buffer may still drain)

    def get_instance(self, proto):
        """One parsed utterance (speech_dataset.py:13-45)."""
        ctx, seq = tfrecord.parse_sequence_example(proto)
        F = self.params.feat_length
        frames = seq.get("logmel", [])
        if isinstance(frames, np.ndarray):        # uniform frames, parsed in one shot (tfrecord._uniform_float_frames)
            logmel = frames.astype(np.float32, copy=False)
        else:
            logmel = np.stack(frames).astype(np.float32) if len(frames) else np.zeros((0, F), np.float32)
        if logmel.shape[1:] != (F,):
            raise ValueError("logmel frames of width %s, expected feat_length=%d" % (logmel.shape[1:], F))
        def ints(name):
            v = seq.get(name, [])
            if isinstance(v, np.ndarray):         # one value per step, parsed in one pass (tfrecord._single_int_steps)
                return v[:, 0].astype(np.int64)
            return np.asarray([int(s[0]) for s in v], dtype=np.int64)
        return {"logmel": logmel, "char": ints("cint"), "phone": ints("pint"),
                "logmel_len": int(ctx["logmel_len"][0]), "char_len": int(ctx["cint_len"][0]),
                "phone_len": int(ctx["pint_len"][0]), "utt_id": ctx["segment"][0]}

    def _instances(self):
        self.files_exhausted = False
        for fn in self.data_files:
            for rec in tfrecord.read_records(fn, verify_payload=self.verify_crc):
                yield self.get_instance(rec)
        self.files_exhausted = True

    def collate(self, insts):
        F = self.params.feat_length
        return {"logmel": _pad_stack([i["logmel"] for i in insts], (F,), np.float32),
                "char": _pad_stack([i["char"] for i in insts], (), np.int64),
                "phone": _pad_stack([i["phone"] for i in insts], (), np.int64),
                "logmel_len": np.asarray([i["logmel_len"] for i in insts], np.int64),
                "char_len": np.asarray([i["char_len"] for i in insts], np.int64),
                "phone_len": np.asarray([i["phone_len"] for i in insts], np.int64),
                "utt_id": [i["utt_id"].decode("utf-8") if isinstance(i["utt_id"], bytes) else i["utt_id"] for i in insts]}

    def __iter__(self):
        src = self._instances()
        if self.is_training:
            with self._spawn_lock:
                rng = np.random.default_rng(self._seed_seq.spawn(1)[0])
            src = shuffle_buffer(src, self.SHUFFLE_BUFFER, rng)
        batch = []
        for inst in src:
            batch.append(inst)
            if len(batch) == self.params.batch_size:
                yield self.collate(batch)
                batch = []
        if batch:
            yield self.collate(batch)


def write_speech_tfrecord(path, utterances):
    """utterances: iterable of dicts with utt_id, logmel [T,F], char [L], phone [Lp] -- writes the reference's layout."""
    def rec(u):
        lm = np.asarray(u["logmel"], np.float32)
        ch, ph = np.asarray(u["char"], np.int64), np.asarray(u.get("phone", []), np.int64)
        seg = u["utt_id"].encode("utf-8") if isinstance(u["utt_id"], str) else u["utt_id"]
        return tfrecord.make_sequence_example(
            {"segment": seg, "logmel_len": np.int64(u.get("logmel_len", len(lm))),
             "cint_len": np.int64(u.get("char_len", len(ch))), "pint_len": np.int64(u.get("phone_len", len(ph)))},
            {"logmel": [f for f in lm], "cint": [np.int64(c) for c in ch], "pint": [np.int64(c) for c in ph]})
    tfrecord.write_records(path, (rec(u) for u in utterances))
