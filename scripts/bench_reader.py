"""Host-side input pipeline throughput (TFRecord -> parsed utterances -> padded batches of 32), single thread: the train step
needs a batch every ~9.4 ms.  Diagnostic."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from e2e_asr_amd import tfrecord
from e2e_asr_amd.speech_dataset import SpeechDataset, write_speech_tfrecord
rng = np.random.default_rng(0)
utts = []
for i in range(512):
    T = int(rng.integers(400, 801))
    utts.append(dict(utt_id="u%d" % i, logmel=rng.standard_normal((T, 80)).astype(np.float32),
                     char=rng.integers(0, 1000, size=60).astype(np.int64), phone=rng.integers(3, 40, size=100).astype(np.int64)))
d = tempfile.mkdtemp(); path = os.path.join(d, "a.tfrecord")
write_speech_tfrecord(path, utts)
class P: batch_size = 32; feat_length = 80
ds = SpeechDataset(P, [path], isTraining=False)
nb = len(utts) // 32
t0 = time.perf_counter(); recs = list(tfrecord.read_records(path)); t1 = time.perf_counter()
print("read_records  %6.2f ms per batch of 32" % ((t1 - t0) / nb * 1e3))
t0 = time.perf_counter(); insts = [ds.get_instance(r) for r in recs]; t1 = time.perf_counter()
print("get_instance  %6.2f ms per batch of 32" % ((t1 - t0) / nb * 1e3))
t0 = time.perf_counter()
for k in range(nb): x = ds.collate(insts[32 * k:32 * k + 32])
t1 = time.perf_counter()
print("collate       %6.2f ms per batch" % ((t1 - t0) / nb * 1e3))
for tr in (False, True):
    ds = SpeechDataset(P, [path], isTraining=tr, seed=1)
    t0 = time.perf_counter(); n = 0
    for b in ds: n += 1
    print("whole iterator (training=%s) %6.2f ms per batch" % (tr, (time.perf_counter() - t0) / n * 1e3))
