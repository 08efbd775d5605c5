"""The REAL training loop (e2e_asr_amd.train.Train: TFRecord buckets -> reader thread -> pinned staging -> HIP train steps) on
config-2-sized synthetic utterances: ms per step of the loop as a user runs it, next to bench.py's resident-batch step.
Diagnostic.  usage: bench_train_loop.py [steps]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from e2e_asr_amd.seq2seq_model import Seq2SeqModel
from e2e_asr_amd.speech_dataset import write_speech_tfrecord
from e2e_asr_amd.train import Train

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(0)
d = tempfile.mkdtemp()
def corpus(n, tmin, tmax):
    out = []
    for i in range(n):
        T, L = int(rng.integers(tmin, tmax + 1)), int(rng.integers(60, 120))
        ch = np.concatenate([[1], rng.integers(3, 1000, L), [2]])
        out.append({"utt_id": "u%d" % i, "logmel": rng.standard_normal((T, 80)).astype(np.float32), "char": ch, "char_len": len(ch) - 1,
                    "phone": rng.integers(3, 40, L), "phone_len": L - 1})
    return out
for k in range(12):
    write_speech_tfrecord(os.path.join(d, "train_1k.0.%d" % k), corpus(32 * 8, 600, 800))     # 96 steps per epoch
write_speech_tfrecord(os.path.join(d, "dev.0"), corpus(8, 100, 200))
p = Seq2SeqModel.class_params()
p.encoder_params.use_lstm = True
tp = Train.class_params()
tp.data_dir = d; tp.train_dir = os.path.join(d, "run"); tp.best_model_dir = os.path.join(d, "run", "best")
tp.batch_size = 32; tp.buck_batch_size = [32]; tp.steps_per_checkpoint = 10 ** 6; tp.max_epochs = 10 ** 6
tr = Train(p, tp, device="cuda:0")
tr.train(max_steps=8)                       # warm-up (builds the model, first launches)
torch.cuda.synchronize()
t0 = time.perf_counter()
tr2 = Train(p, tp, device="cuda:0")
marks = {}
orig = Seq2SeqModel.step
def timed_step(self, batch=None):
    n = marks.setdefault("n", 0)
    if n == 8:
        torch.cuda.synchronize(); marks["t0"] = time.perf_counter(); marks["in_step"] = 0.0; marks["between"] = 0.0
    marks["n"] = n + 1
    t_in = time.perf_counter()
    if n > 8:
        marks["between"] += t_in - marks["t_out"]
    r = orig(self, batch)
    marks["t_out"] = time.perf_counter()
    if n >= 8:
        marks["in_step"] += marks["t_out"] - t_in
    return r
Seq2SeqModel.step = timed_step
if len(sys.argv) > 2 and sys.argv[2] == "mem":      # batches already collated in memory: the loop without the reader
    from e2e_asr_amd.speech_dataset import SpeechDataset
    from e2e_asr_amd.base_params import Bunch
    import glob
    files = sorted(glob.glob(os.path.join(d, "train_1k.0.*")))
    batches = list(SpeechDataset(Bunch(batch_size=32, feat_length=80), files, isTraining=False))
    devb = list(SpeechDataset(Bunch(batch_size=32, feat_length=80), [os.path.join(d, "dev.0")], isTraining=False))
    tr2.train(buck_train_sets=[batches], dev_set=devb, max_steps=8 + steps)
else:
    tr2.train(max_steps=8 + steps)
torch.cuda.synchronize()
dt = (time.perf_counter() - marks["t0"]) / steps
print("Train loop: %.2f ms per step over %d steps (32 utterances of 600-800 frames per batch, ragged); host: %.2f ms inside model.step, %.2f ms between steps (waiting for the reader, loop bookkeeping)" % (dt * 1e3, steps, marks["in_step"] / steps * 1e3, marks["between"] / steps * 1e3))
