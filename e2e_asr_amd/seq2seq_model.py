"""Seq2Seq model -- host-side mirror of the reference's seq2seq_model.py (50-216).

The reference builds a TF graph once and runs it with sess.run([updates, losses])
(train.py:297-299).  This counterpart is eager: `forward(batch)` runs encoder -> per-task
decoder -> loss on the HIP kernels and fills the same attributes (`outputs`, `losses`,
`total_loss`, `encoder_hidden_states`, `seq_len_encs`, ...); `step()` is the equivalent of
one sess.run of `updates`: forward, backward, [data-parallel all-reduce], global-norm clip,
Adam, global_step += 1.
"""
import os

import numpy as np
import torch

from . import ops
from .attn_decoder import AttnDecoder
from .base_params import BaseParams, Bunch
from .devcache import dev_i32
from .encoder import Encoder
from .losses import LossUtils
from .variables import VariableStore
from .weights import init_weights


def create_shifted_targets(dec_input, seq_len):
    """tf_utils.py:4-12: targets = dec_input[1:]; weights = time-major length mask, flattened."""
    targets = dec_input[1:]
    T = targets.shape[0]
    ln = dev_i32(seq_len, targets.device)
    w = (torch.arange(T, device=targets.device)[:, None] < ln[None, :]).to(torch.float32)
    return targets, w.reshape(-1)


class LazyTargetWeights(dict):
    """{task: flattened time-major length mask} (tf_utils.py:4-12), built on first access: the loss kernels mask by the
    lengths themselves, so on the hot path nobody reads the mask and its four small launches stay off the step."""

    def __init__(self, make):
        super(LazyTargetWeights, self).__init__()
        self._make = make

    def __missing__(self, task):
        v = self._make(task)
        self[task] = v
        return v


class Seq2SeqModel(BaseParams):
    """Attention-enabled encoder-decoder with optional auxiliary-task decoders."""

    @classmethod
    def class_params(cls):
        # seq2seq_model.py:28-48
        return Bunch(tasks=["char"], num_layers={"char": 4}, max_output={"char": 120},
                     learning_rate=1e-3, learning_rate_decay_factor=0.5, max_gradient_norm=5.0,
                     avg=True, encoder_params=Encoder.class_params(),
                     decoder_params={"char": AttnDecoder.class_params()})

    def __init__(self, data_iter, isTraining=True, params=None, variables=None, device="cuda:0",
                 feat_length=80, seed=10):
        self.params = self.class_params() if params is None else params
        params = self.params
        self.device = torch.device(device)
        self.data_iter = data_iter
        self.isTraining = isTraining
        if variables is None:      # variable creation = tf.global_variables_initializer (train.py:207)
            variables = self.create_variables(params, self.device, feat_length, seed)
        self.variables = variables
        self.encoder = Encoder(isTraining=isTraining, params=params.encoder_params, variables=variables)
        self.decoder = {}
        for task in params.tasks:
            self.decoder[task] = AttnDecoder(isTraining=isTraining, params=params.decoder_params[task],
                                             scope=task, variables=variables)
        self.learning_rate = float(params.learning_rate)        # seq2seq_model.py:74-77
        self.global_step = 0
        self.epoch = 0
        self.outputs, self.losses, self.total_loss = {}, {}, None
        self.encoder_hidden_states, self.time_major_states, self.seq_len_encs = {}, {}, {}
        self.dist = None           # set by parallel.DataParallel
        self.rank_seed = 0         # set by parallel.DataParallel: decorrelates dropout / sampler noise across replicas
        self._loss_ws = {}

    # ------------------------------------------------------------------ variables
    @staticmethod
    def create_variables(params, device, feat_length=80, seed=10):
        ep = params.encoder_params
        tasks = list(params.tasks)
        dp = params.decoder_params[tasks[0]]
        depth = max(params.num_layers[t] for t in tasks)
        arrays = init_weights(
            feat=feat_length * ep.stack_cons, hidden=ep.hidden_size, bi_dir=ep.bi_dir, depth=depth,
            tasks=tasks, vocab={t: params.decoder_params[t].vocab_size for t in tasks},
            emb=dp.emb_size, hidden_dec=dp.hidden_size_dec, lm_hidden=dp.lm_hidden_size,
            attn_vec=dp.attention_vec_size, seed=seed, skip_step=ep.skip_step,
            max_scaling_down=ep.max_scaling_down, initial_res_fac=ep.initial_res_fac,
            num_layers_dec=getattr(dp, "num_layers_dec", 1), use_lstm=bool(ep.use_lstm), dec_use_lstm=bool(getattr(dp, "use_lstm", True)),
            ind_softmax={t: bool(getattr(params.decoder_params[t], "ind_softmax", False)) for t in tasks})
        return VariableStore.from_arrays(arrays, device)

    def learning_rate_decay_op(self):
        self.learning_rate *= self.params.learning_rate_decay_factor
        return self.learning_rate

    def epoch_incr(self):
        self.epoch += 1
        return self.epoch

    # ------------------------------------------------------------------ batch unpack
    def get_batch(self, batch):
        """seq2seq_model.py:159-197: frame stacking, time-major decoder inputs, eval lengths
        forced to max_output."""
        dev = self.device
        x = torch.as_tensor(np.asarray(batch["logmel"], np.float32)).to(dev) \
            if not torch.is_tensor(batch["logmel"]) else batch["logmel"].to(dev, torch.float32)
        enc_len = np.asarray(batch["logmel_len"]).astype(np.int64)
        sc = self.encoder.params.stack_cons
        if sc > 1:                                              # :164-183
            B, T, F = x.shape
            parts = [x]
            for shift in range(1, sc):
                parts.append(torch.cat([x[:, shift:, :], x.new_zeros(B, shift, F)], 1))
            x = torch.cat(parts, 2)
        dec_in, dec_len = {}, {}
        for task in self.params.tasks:
            ids = batch[task]
            if torch.is_tensor(ids):
                dec_in[task] = ids.to(dev).t().contiguous().to(torch.int32)
            else:
                dec_in[task] = dev_i32(np.asarray(ids).T, dev)            # :189 (time-major)
            ln = np.asarray(batch[task + "_len"]).astype(np.int64)
            if not self.isTraining:                                       # :191-193
                ln = np.ones_like(ln) * self.params.max_output[task]
            dec_len[task] = ln
        if not self.isTraining and "utt_id" in batch:
            dec_in["utt_id"] = batch["utt_id"]
        return x.contiguous(), dec_in, enc_len, dec_len

    # ------------------------------------------------------------------ forward
    def forward(self, batch=None):
        """Encoder -> decoders -> losses (seq2seq_model.py:88-144)."""
        params = self.params
        if batch is None:
            batch = self.data_iter.get_next()
        self.encoder_inputs, self.decoder_inputs, self.seq_len, self.seq_len_target = self.get_batch(batch)
        self.targets = {task: self.decoder_inputs[task][1:] for task in params.tasks}       # tf_utils.py:4-12 (a view)
        src, lens = dict(self.decoder_inputs), dict(self.seq_len_target)
        self.target_weights = LazyTargetWeights(lambda task: create_shifted_targets(src[task], lens[task])[1])
        self.encoder.dropout_seed = (self.global_step ^ self.rank_seed) & 0x7FFFFFFF
        self.encoder_hidden_states, self.time_major_states, self.seq_len_encs = self.encoder(
            self.encoder_inputs, self.seq_len, {t: params.num_layers[t] for t in params.tasks})
        self.outputs = {}
        for task in params.tasks:
            d = params.num_layers[task]
            dec = self.decoder[task]
            dec.rng_seed = ((self.global_step * 2654435761 + sum(map(ord, task)) % 9973) ^ self.rank_seed) & 0x7FFFFFFF
            dec.coin_step = self.global_step      # NOT mixed with rank_seed: the coin is common to all replicas (attn_decoder.py:132)
            if not self.isTraining and self.decoder_inputs[task].shape[0] < params.max_output[task]:
                pad = params.max_output[task] - self.decoder_inputs[task].shape[0]
                self.decoder_inputs[task] = torch.cat(
                    [self.decoder_inputs[task],
                     self.decoder_inputs[task].new_zeros(pad, self.decoder_inputs[task].shape[1])], 0)
            self.outputs[task] = dec(self.decoder_inputs[task], self.seq_len_target[task],
                                     self.encoder_hidden_states[d], self.seq_len_encs[d])
        if self.isTraining:
            self.losses = {}
            if getattr(self, "_gscale", None) is None:          # d total_loss / d task loss: constant for the model's life
                self._gscale = torch.full((1,), 1.0 / len(params.tasks) if params.avg else 1.0, device=self.device)
            for task in params.tasks:
                T_out = self.decoder[task].saved["t_out"]
                self.losses[task], self._loss_ws[task] = LossUtils.cross_entropy_loss(
                    self.outputs[task], self.targets[task][:T_out], self.seq_len_target[task], return_ws=True,
                    grad_scale=self._gscale)          # (the logit gradient in the same pass over the logits)
            total = None
            for task in params.tasks:                               # :140-144
                total = self.losses[task] if total is None else total + self.losses[task]
            if params.avg and len(params.tasks) > 1:            # (one task: x / 1.0 == x, no launch for it on the step's critical path)
                total = total / float(len(params.tasks))
            self.total_loss = total
        return self.outputs

    # ------------------------------------------------------------------ backward + update
    def backward(self):
        """tf.gradients(total_loss, trainable_vars) (seq2seq_model.py:148) into variables.grad."""
        params = self.params
        v = self.variables
        v.ensure_grad()
        if self.dist is not None:
            self.dist.begin_step()
        v.grad.zero_()
        if getattr(self, "_gscale", None) is None:          # d total_loss / d task loss: constant for the model's life
            self._gscale = torch.full((1,), 1.0 / len(params.tasks) if params.avg else 1.0, device=self.device)
        gscale = self._gscale
        d_states = {}
        # The decoders' LM cell chains (a persistent BPTT each, 4 B workgroups) are independent of the encoder: run behind the
        # encoder's last BPTT they share the chip with the side stream's weight-gradient backlog instead of making the
        # encoder's first BPTT wait for them (round 5; ASR_LM_DEFER=0: right behind the decoder chain as before).  Not in the
        # data-parallel overlap mode, whose buckets assume the decoder's gradients are complete when the encoder's begin.
        defer_lm = os.environ.get("ASR_LM_DEFER", "1") != "0" and not (self.dist is not None and getattr(self.dist, "overlap", False))
        for k, task in enumerate(params.tasks):
            lw = self._loss_ws[task]
            dlogits = lw.pop("dlogits", None)
            if dlogits is None:
                dlogits = ops.masked_ce_bwd(self.outputs[task], lw["targets"], lw["lse"], lw["len"], gscale)
            d = params.num_layers[task]
            if d not in d_states:
                d_states[d] = torch.zeros_like(self.decoder[task].saved["enc"])
            self.decoder[task].backward(dlogits, d_states[d], defer_lm=defer_lm, side_busy=k > 0)
        # the decoders' LM-chain gradients are still in flight on the library's side stream and overlap the encoder BPTT;
        # data-parallel overlap mode all-reduces the finished buckets in the tail after the last BPTT (parallel.py)
        self.encoder.backward(d_states, on_layer_done=(
            (lambda depth: self.dist.grad_ready(depth, v.grad)) if self.dist is not None else None))
        for task in params.tasks:
            self.decoder[task].backward_lm_tail()
        ops.side_join()
        if self.dist is not None:
            self.dist.grad_ready(0, v.grad)

    def apply_gradients(self, slot="Adam", lr=None):
        """[data-parallel all-reduce ->] tf.clip_by_global_norm -> AdamOptimizer.apply_gradients
        (seq2seq_model.py:137,150-155).  The all-reduce sums shard gradients; the 1/N is folded
        into the clip+Adam kernel, so clipping sees the global-batch gradient."""
        v = self.variables
        n = 1
        if self.dist is not None:
            n = self.dist.all_reduce_grads(v.grad)
        m, vv = v.ensure_adam(slot)
        self._gnorm_sq = ops.sumsq(v.grad)
        self.global_step += 1
        t = self.global_step
        lr = self.learning_rate if lr is None else lr
        lr_t = lr * np.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t)
        ops.clip_adam(v.flat, m, vv, v.grad, self._gnorm_sq, 1.0 / n, self.params.max_gradient_norm, lr_t)

    def step(self, batch=None):
        """One sess.run([model.updates, model.losses]) (train.py:297-299).  Returns the losses
        dict (device scalars; no host synchronisation inside the step)."""
        if not self.isTraining:
            raise ValueError("step() needs a model built with isTraining=True")
        ops.ws_arena_begin(self.device)           # one zero fill for the step's exchange workspaces
        try:
            self.forward(batch)
            self.backward()
        finally:
            ops.ws_arena_end(self.device)
        self.apply_gradients()
        return self.losses

    # greedy hypotheses of the eval graph (eval_model.py:84-87)
    def greedy_ids(self, task="char"):
        logits = self.outputs[task]
        B = self.encoder_inputs.shape[0]
        return logits.argmax(1).reshape(-1, B).t()

    @classmethod
    def add_parse_options(cls, parser):
        # seq2seq_model.py:199-216
        parser.add_argument("-tasks", "--tasks", default="", type=str, help="Auxiliary task choices")
        parser.add_argument("-nlc", "--num_layers_char", default=4, type=int, help="Encoder layer used for char.")
        parser.add_argument("-nlp", "--num_layers_phone", default=3, type=int, help="Encoder layer used for phone.")
        parser.add_argument("-max_out_char", "--max_output_char", default=120, type=int,
                            help="Maximum length of char/word-piece sequence")
        parser.add_argument("-max_out_phone", "--max_output_phone", default=250, type=int,
                            help="Maximum length of phone sequence")
        parser.add_argument("-lr_decay", "--learning_rate_decay_factor", default=0.5, type=float,
                            help="Learning rate decay factor")
        parser.add_argument("-avg", "--avg", default=False, action="store_true", help="Average the loss")
