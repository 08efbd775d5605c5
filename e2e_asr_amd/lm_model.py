"""Language model -- mirror of the reference's lm_model.py (23-121): the char LM that shares
variables with the attention decoder's inner LSTM, trained with its own Adam ('AdamLM', lr 1e-4)
and interleaved with ASR steps with probability lm_prob (train.py:269-291)."""
import numpy as np
import torch

from . import ops
from .base_params import BaseParams, Bunch
from .devcache import dev_i32
from .losses import LossUtils
from .seq2seq_model import create_shifted_targets


class LMModel(BaseParams):
    @classmethod
    def class_params(cls):
        # lm_model.py:26-37
        return Bunch(lm_batch_size=128, lm_learning_rate=1e-4, lm_learning_rate_decay_factor=0.5,
                     max_gradient_norm=5.0, simple_lm=False)

    def __init__(self, encoder, data_files=None, params=None, data_iter=None):
        self.params = self.class_params() if params is None else params
        self.data_files = data_files
        self.data_iter = data_iter
        self.learning_rate = float(self.params.lm_learning_rate)
        self.lm_global_step = 0
        self.epoch = 0
        self.encoder = encoder
        self.dist = None

    def learning_rate_decay_op(self):
        self.learning_rate *= self.params.lm_learning_rate_decay_factor
        return self.learning_rate

    def epoch_incr(self):
        self.epoch += 1

    def get_batch(self, batch=None):
        """lm_model.py:108-115: ids are [B,T+1] -> time-major; the length keeps the (T+1)th symbol out."""
        if batch is None:
            batch = self.data_iter.get_next()
        dev = self.encoder.variables.device
        return dev_i32(np.asarray(batch["char"]).T, dev), np.asarray(batch["char_len"]).astype(np.int64)

    def forward(self, batch=None):
        self.encoder_inputs, self.seq_len = self.get_batch(batch)
        self.targets = self.encoder_inputs[1:]                                   # tf_utils.py:4-12 (a view)
        self._target_weights = None                                              # the mask is built when somebody reads it
        self.outputs = self.encoder(self.encoder_inputs, self.seq_len)
        self.losses, self._loss_ws = LossUtils.cross_entropy_loss(self.outputs, self.targets, self.seq_len, return_ws=True)
        return self.losses

    @property
    def target_weights(self):
        """Flattened time-major length mask (tf_utils.py:4-12); the loss kernels mask by length themselves, so it is built lazily."""
        if self._target_weights is None:
            self._target_weights = create_shifted_targets(self.encoder_inputs, self.seq_len)[1]
        return self._target_weights

    def step(self, batch=None):
        """One sess.run([lm_model.updates, lm_model.losses]) (train.py:272-273)."""
        v = self.encoder.variables
        self.encoder.dropout_seed = (self.lm_global_step * 40503 + 17) & 0x7FFFFFFF
        self.forward(batch)
        v.ensure_grad()
        v.grad.zero_()
        lw = self._loss_ws
        one = torch.ones(1, device=v.device)
        dlogits = ops.masked_ce_bwd(self.outputs, lw["targets"], lw["lse"], lw["len"], one)
        self.encoder.backward(dlogits)
        ops.side_join()
        n = self.dist.all_reduce_grads(v.grad) if self.dist is not None else 1
        m, vv = v.ensure_adam("AdamLM")                                   # lm_model.py:76
        self._gnorm_sq = ops.sumsq(v.grad)
        self.lm_global_step += 1
        t = self.lm_global_step
        lr_t = self.learning_rate * np.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t)
        ops.clip_adam(v.flat, m, vv, v.grad, self._gnorm_sq, 1.0 / n, self.params.max_gradient_norm, lr_t)
        return self.losses

    @classmethod
    def add_parse_options(cls, parser):
        parser.add_argument("-lm_learning_rate", default=0.0001, type=float, help="LM learning rate")
