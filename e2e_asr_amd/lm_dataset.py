"""Language-model data sets from TFRecord files -- the role of lm_dataset.py:5-46 without TensorFlow: records are
SequenceExamples with context `cint_len` and feature list `cint`; shuffle buffer 10000 (:38), padded batches (:39-40)."""
import numpy as np

from . import tfrecord
from .speech_dataset import _pad_stack, shuffle_buffer


class LMDataset(object):
    SHUFFLE_BUFFER = 10000

    def __init__(self, filenames, batch_size, seed=None, verify_crc=False):
        self.batch_size = batch_size
        self.filenames = list(filenames)
        self.verify_crc = verify_crc
        self._rng = np.random.default_rng(seed)

    def get_instance(self, proto):
        ctx, seq = tfrecord.parse_sequence_example(proto)
        return {"char": np.asarray([int(s[0]) for s in seq.get("cint", [])], np.int64), "char_len": int(ctx["cint_len"][0])}

    def __iter__(self):
        def src():
            for fn in self.filenames:
                for rec in tfrecord.read_records(fn, verify_payload=self.verify_crc):
                    yield self.get_instance(rec)
        batch = []
        for inst in shuffle_buffer(src(), self.SHUFFLE_BUFFER, self._rng):
            batch.append(inst)
            if len(batch) == self.batch_size:
                yield self._collate(batch)
                batch = []
        if batch:
            yield self._collate(batch)

    @staticmethod
    def _collate(insts):
        return {"char": _pad_stack([i["char"] for i in insts], (), np.int64),
                "char_len": np.asarray([i["char_len"] for i in insts], np.int64)}


def write_lm_tfrecord(path, sequences):
    tfrecord.write_records(path, (tfrecord.make_sequence_example(
        {"cint_len": np.int64(len(s))}, {"cint": [np.int64(c) for c in s]}) for s in sequences))
