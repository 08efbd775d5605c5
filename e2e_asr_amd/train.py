"""Training loop -- mirror of the reference's train.py (37-430) around the HIP train step.

`Train(model_params, train_params).train()` keeps the reference's loop policy: length-bucketed
training sets consumed smallest-bucket-first until each is exhausted (261-295, 379-383), optional
LM steps with probability lm_prob (269-291), every `steps_per_checkpoint` steps a greedy dev
decode (322), LR halving when the dev error is no better than the worst of the last three after
`min_steps` while lr > 1e-4 (334-343), stop after 10 non-improving checkpoints at floor LR
(346-349, check_progess 153-158), `best.txt` / `asr_err.txt` / keep-all + best checkpoints
(353-371).  Data sets are re-iterable batch sources yielding the speech_dataset.py:43-45 dict: `get_data_sets()`
builds them from the TFRecord buckets `train_1k.<bucket>.*` / `dev*` of `data_dir` (94-131); `train()` also takes
injected ones (tests, synthetic corpora)."""
import copy
import gc
import glob
import math
import os
import random
import sys
import time

from . import checkpoint, ops
from .base_params import BaseParams, Bunch
from .eval_model import Eval
from .lm_encoder import LMEncoder
from .lm_dataset import LMDataset
from .lm_model import LMModel
from .prefetch import DevicePrefetcher
from .seq2seq_model import Seq2SeqModel
from .speech_dataset import SpeechDataset


class Train(BaseParams):
    @classmethod
    def class_params(cls):
        # train.py:39-72
        return Bunch(batch_size=128, buck_batch_size=[128, 128, 64, 64, 32], max_epochs=30, min_steps=25000,
                     feat_length=80, data_dir="", lm_data_dir="", vocab_dir="", train_base_dir="",
                     train_dir="/tmp/asr_train", best_model_dir="/tmp/asr_train/best", lm_prob=0.0,
                     lm_params=LMModel.class_params(), lm_enc_params=LMEncoder.class_params(), run_id=1,
                     steps_per_checkpoint=500, pretrain_lm_path="", pretrain_phone_path="", chaos=False, subset_file="",
                     steps_per_epoch=3006)      # train.py:217: "For default setup it's 3006" (epoch = global_step / 3006)

    def __init__(self, model_params, train_params=None, device="cuda:0"):
        self.params = self.class_params() if train_params is None else train_params
        self.seq2seq_params = model_params
        self.eval_model = None
        self.device = device

    @staticmethod
    def load_train_subset_file(subset_file):
        """train.py:83-92: file names (one per line) the training buckets are restricted to."""
        try:
            with open(subset_file) as f:
                return dict((line.strip(), 0) for line in f.readlines())
        except (IOError, OSError):
            return {}

    def get_data_sets(self, logging=True):
        """train.py:94-131: one training set per length bucket (batch sizes buck_batch_size, files
        data_dir/train_1k.<bucket>.*, shuffled file order, optional subset file) and the dev set (data_dir/dev*)."""
        params = self.params
        subset = self.load_train_subset_file(params.subset_file) if params.subset_file else None
        buck_train_sets, total = [], 0
        for batch_id, batch_size in enumerate(params.buck_batch_size):
            files = sorted(glob.glob(os.path.join(params.data_dir, "train_1k." + str(batch_id) + ".*")))
            if subset:
                files = [f for f in files if os.path.basename(f) in subset]
            random.shuffle(files)
            total += len(files)
            buck_train_sets.append(SpeechDataset(Bunch(batch_size=batch_size, feat_length=params.feat_length), files,
                                                 isTraining=True))
        dev_files = sorted(glob.glob(os.path.join(params.data_dir, "dev*")))
        if logging:
            print("Total train files: %d" % total)
            print("Total dev files: %d" % len(dev_files))
        dev_set = SpeechDataset(Bunch(batch_size=params.batch_size, feat_length=params.feat_length), dev_files,
                                isTraining=False)
        return buck_train_sets, dev_set

    def get_lm_files(self):
        return sorted(glob.glob(os.path.join(self.params.lm_data_dir, "lm*")))          # train.py:134-137

    def get_lm_set(self):
        """The LM corpus as a batch source (train.py:197-200: LMDataset(lm_files, lm_params.batch_size))."""
        return LMDataset(self.get_lm_files(), self.params.lm_params.lm_batch_size)

    @staticmethod
    def check_progess(previous_errs, num=10):
        """train.py:153-158 (sic): False when the best error is older than the last `num` checkpoints."""
        if len(previous_errs) > num:
            if min(previous_errs) != min(previous_errs[-num:]):
                return False
        return True

    @staticmethod
    def should_decay(previous_errs, asr_err_cur, global_step, min_steps, lr):
        """train.py:334-343: decay when past min_steps, >3 checkpoints seen, the current error is no
        better than the worst of the previous three, and lr is still above 1e-4."""
        return (global_step >= min_steps and len(previous_errs) > 3 and asr_err_cur >= max(previous_errs[-3:])
                and lr > 1e-4)

    def create_eval_model(self, variables):
        """train.py:138-151: second model over the SAME variables, isTraining=False, char task only."""
        p = copy.deepcopy(self.seq2seq_params)
        p.tasks = ["char"]
        p.num_layers = {"char": p.num_layers["char"]}
        model_dev = Seq2SeqModel(None, isTraining=False, params=p, variables=variables, device=self.device,
                                 feat_length=self.params.feat_length)
        self.eval_model = Eval(model_dev, params=Bunch(best_model_dir=self.params.best_model_dir, vocab_dir=self.params.vocab_dir))
        return model_dev

    def train(self, buck_train_sets=None, dev_set=None, lm_set=None, max_steps=None):
        """train.py:160-393.  (Wrapper: whatever way the loop ends, the reader threads it started are stopped.)"""
        self._open_prefetchers = []
        try:
            model = self._train(buck_train_sets, dev_set, lm_set, max_steps)
            # whichever way the loop ended (max_steps, early stop, last epoch): a persistent-kernel time-out in the steps since
            # the last periodic check must not leave with the returned model unnoticed
            ops.check_device_flag(model.device)
            return model
        finally:
            for pf in self._open_prefetchers:
                pf.close()
            self._open_prefetchers = []

    def _train(self, buck_train_sets=None, dev_set=None, lm_set=None, max_steps=None):
        """buck_train_sets: list (shortest bucket first) of re-iterable batch sources; dev_set: a
        re-iterable of dev batches; lm_set: re-iterable of LM batches (needed when lm_prob > 0).  Left out, they are
        read from params.data_dir / params.lm_data_dir like the reference does."""
        params = self.params
        if buck_train_sets is None or dev_set is None:
            buck_train_sets, dev_set = self.get_data_sets()
        if lm_set is None and params.lm_prob > 0:
            lm_set = self.get_lm_set()
        random.seed(int(time.time()) if params.chaos else 10)                       # train.py:167-174
        os.makedirs(params.train_dir, exist_ok=True)
        os.makedirs(params.best_model_dir, exist_ok=True)
        model = Seq2SeqModel(None, True, self.seq2seq_params, device=self.device, feat_length=params.feat_length)
        self.model = model
        self.create_eval_model(model.variables)
        lm_model = None
        if params.lm_prob > 0:
            lm_model = LMModel(LMEncoder(params=params.lm_enc_params, variables=model.variables), params=params.lm_params)
        self.lm_model = lm_model
        latest = os.path.join(params.train_dir, "checkpoint.txt")
        asr_err_best = 1.0
        if os.path.isfile(latest):                                                  # resume (train.py:205-215)
            ck = open(latest).read().strip()
            model.global_step, lr = checkpoint.restore(model.variables, ck)
            if lr is not None:
                model.learning_rate = lr
            # the TF checkpoint also carries the LM's step counter (AdamLM's beta powers follow it), its learning-rate
            # variable and both epoch counters (tf.global_variables(), train.py:202): restore them too
            extra = checkpoint.load_scalars(ck, ("epoch", "lm_global_step", "lm_learning_rate", "lm_epoch"))
            model.epoch = int(extra.get("epoch", model.epoch))
            if lm_model is not None:
                lm_model.lm_global_step = int(extra.get("lm_global_step", 0))
                lm_model.learning_rate = float(extra.get("lm_learning_rate", lm_model.learning_rate))
                lm_model.epoch = int(extra.get("lm_epoch", 0))
            score_file = os.path.join(params.train_dir, "best.txt")
            if os.path.isfile(score_file):
                try:
                    asr_err_best = float(open(score_file).readline().strip("\n"))
                except ValueError:
                    pass
        else:
            if params.pretrain_lm_path:
                checkpoint.restore_common_variables(model.variables, params.pretrain_lm_path)
            if params.pretrain_phone_path:
                checkpoint.restore_common_variables(model.variables, params.pretrain_phone_path)
        print("\nBest ASR error rate - %f" % asr_err_best)
        previous_errs = []
        try:
            with open(os.path.join(params.train_dir, "asr_err.txt")) as f:
                previous_errs = [float(l.strip()) for l in f]
        except IOError:
            pass
        loss, current_step, lm_loss, lm_steps = 0.0, 0, 0.0, 0
        loss_acc = lm_loss_acc = None         # device-side running means of the step losses of this checkpoint interval
        ckpt_start = time.time()
        lm_iter = iter(lm_set) if lm_set is not None else None
        epoch = model.global_step // max(1, int(params.steps_per_epoch))              # train.py:217
        save_extra = lambda: dict(epoch=model.epoch, **(dict(
            lm_global_step=lm_model.lm_global_step, lm_learning_rate=lm_model.learning_rate, lm_epoch=lm_model.epoch)
            if lm_model is not None else {}))
        carry = None                          # the next epoch's first bucket, its reader already running
        # A train step makes no cyclic garbage, but a generation-2 pass of CPython's collector walks every container object
        # torch and the model keep alive (~170 000: 26-40 ms, five steps' worth of GPU time, and the GPU runs dry meanwhile --
        # measured, scripts/host_stall.py).  Collect once, then park the survivors in the permanent generation.
        gc.collect()
        gc.freeze()
        while epoch <= params.max_epochs:
            print("\nEpochs done: %d" % epoch)
            # (each bucket's batches are staged into HBM one batch ahead of the step: the iterator's prefetch of the reference)
            prefetchers = [DevicePrefetcher(s, self.device) for s in buck_train_sets]        # train.py:261-266
            started = {}
            if carry is not None:                 # bucket 0 of this epoch is the reader that was primed during the last one
                prefetchers[0], started[0] = carry
            self._open_prefetchers.extend(p for p in prefetchers if p not in self._open_prefetchers)
            carry = None

            def bucket_iter(k):                          # reader threads start one bucket ahead of the loop (prefetch.primed)
                if k not in started:
                    started[k] = prefetchers[k].primed()
                return started[k]
            active = list(range(len(prefetchers)))
            while active:
                if max_steps is not None and current_step >= max_steps:
                    return model
                if lm_model is not None and params.lm_prob > random.random():        # :269-291
                    try:
                        lm_batch = next(lm_iter)
                    except StopIteration:
                        lm_model.epoch_incr()
                        lm_iter = iter(lm_set)
                        continue
                    lm_step_loss = lm_model.step(lm_batch).detach() / params.steps_per_checkpoint      # (device-side, as below)
                    lm_loss_acc = lm_step_loss if lm_loss_acc is None else lm_loss_acc + lm_step_loss
                    lm_steps += 1
                    if lm_steps % 16 == 0:
                        ops.check_device_flag(model.device)
                    if lm_steps % params.steps_per_checkpoint == 0:
                        lm_loss, lm_loss_acc = float(lm_loss_acc.item()), None
                        print("LM steps: %d, Perplexity: %f" % (lm_model.lm_global_step,
                                                                math.exp(lm_loss) if lm_loss < 300 else float("inf")))
                        lm_loss = 0.0
                    continue
                try:
                    it = bucket_iter(active[0])
                    if len(active) > 1:
                        bucket_iter(active[1])
                    elif carry is None and epoch < params.max_epochs and (
                            len(buck_train_sets) > 1 or prefetchers[active[0]].reader_finished()):
                        # last bucket of the epoch: the next epoch's first one.  With a single bucket that is the dataset this
                        # epoch is still iterating (its files would be parsed twice at once), so the carry starts only when
                        # this epoch's reader has parsed its last record -- the loop then still has the shuffle buffer (4 000
                        # utterances) to consume, and every pass shuffles with its own generator (speech_dataset.py).
                        carry_pf = DevicePrefetcher(buck_train_sets[0], self.device)     # fills its shuffle buffer meanwhile
                        self._open_prefetchers.append(carry_pf)                          # (0.74 -> 0.25 s per epoch change)
                        carry = (carry_pf, carry_pf.primed())
                    batch = next(it)                                                  # smallest bucket first (:295)
                except StopIteration:
                    del active[0]                                                     # :379-383
                    continue
                step_loss = model.step(batch)["char"]
                current_step += 1
                # the loss accumulates ON THE DEVICE and the host synchronises every 16 steps only (to look at the time-out
                # flag of the persistent kernels) and at checkpoints: reading the loss after every step, as sess.run does
                # (train.py:297-303), costs the host its run-ahead -- 10.4 instead of 9.3 ms per step in scripts/bench_train_loop.py
                loss_acc = step_loss.detach() / params.steps_per_checkpoint if loss_acc is None else \
                    loss_acc + step_loss.detach() / params.steps_per_checkpoint
                if current_step % 16 == 0 or current_step % params.steps_per_checkpoint == 0:
                    ops.check_device_flag(model.device)  # a timed-out persistent kernel raises here, at most 16 updates late
                if current_step % params.steps_per_checkpoint:
                    continue
                loss, loss_acc = float(loss_acc.item()), None
                perplexity = math.exp(loss) if loss < 300 else float("inf")           # :305-312
                print("Step %d Learning rate %.4f Checkpoint time %.2f Perplexity %.2f" % (
                    model.global_step, model.learning_rate, time.time() - ckpt_start, perplexity))
                asr_err_cur = self.eval_model.greedy_decode(dev_set)                  # :322
                print("ASR error: %.4f" % asr_err_cur)
                with open(os.path.join(params.train_dir, "asr_err.txt"), "a") as f:
                    f.write(str(asr_err_cur) + "\n")
                if self.should_decay(previous_errs, asr_err_cur, model.global_step, params.min_steps, model.learning_rate):
                    model.learning_rate_decay_op()
                    print("Learning rate decreased !!")
                previous_errs.append(asr_err_cur)
                if not (model.learning_rate > 1e-4) and not self.check_progess(previous_errs):
                    print("No improvement in 10 checkpoints")
                    return model
                if asr_err_best > asr_err_cur:                                         # :353-367
                    asr_err_best = asr_err_cur
                    print("Best ASR Error rate: %.4f\nSaving the best model !!" % asr_err_best)
                    with open(os.path.join(params.train_dir, "best.txt"), "w") as f:
                        f.write(str(asr_err_best))
                    checkpoint.save(os.path.join(params.best_model_dir, "asr.ckpt-%d" % model.global_step),
                                    model.variables, model.global_step, model.learning_rate, extra=save_extra())
                ck = checkpoint.save(os.path.join(params.train_dir, "asr.ckpt-%d" % model.global_step),
                                     model.variables, model.global_step, model.learning_rate, extra=save_extra())   # :370-371
                with open(latest, "w") as f:
                    f.write(ck)
                ckpt_start, loss = time.time(), 0.0
                sys.stdout.flush()
            model.epoch_incr()
            epoch += 1
        return model

    @classmethod
    def add_parse_options(cls, parser):
        # train.py:396-429
        parser.add_argument("-lm_prob", default=0.0, type=float, help="Prob. of running the LM task")
        parser.add_argument("-run_id", "--run_id", default=0, type=int, help="Run ID")
        parser.add_argument("-data_dir", default="", type=str, help="Data directory")
        parser.add_argument("-lm_data_dir", default="", type=str, help="Data directory")
        parser.add_argument("-vocab_dir", "--vocab_dir", default="", type=str, help="Vocab directory")
        parser.add_argument("-tb_dir", "--train_base_dir", default="", type=str, help="Training directory")
        parser.add_argument("-feat_len", "--feat_length", default=80, type=int, help="Number of features per frame")
        parser.add_argument("-steps_per_checkpoint", default=500, type=int, help="Gradient steps per checkpoint")
        parser.add_argument("-min_steps", "--min_steps", default=25000, type=int, help="Min steps BEFORE DECREASING LEARNING RATE")
        parser.add_argument("-pretrain_lm_path", default="", type=str, help="Pretrain language model path")
        parser.add_argument("-pretrain_phone_path", default="", type=str, help="Pretrain phone model path")
        parser.add_argument("-chaos", default=False, action="store_true", help="Random seed is not controlled if set")
        parser.add_argument("-subset_file", default="", type=str, help="Subset file")
