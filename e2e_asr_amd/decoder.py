"""Abstract decoder -- host-side mirror of the reference's decoder.py (49-180)."""
import abc

import torch

from . import ops
from .base_params import BaseParams, Bunch


class Decoder(BaseParams):
    """Base class: cell/state conventions, embedding + feedback functions."""

    @classmethod
    def class_params(cls):
        # decoder.py:22-35
        return Bunch(out_prob_dec=0.9, hidden_size_dec=256, num_layers_dec=1, emb_size=256,
                     vocab_size=1000, samp_prob=0.1, max_output=400, use_lstm=True)

    def __init__(self, isTraining=True, params=None):
        self.params = self.class_params() if params is None else params
        self.isTraining = isTraining

    def get_cell(self, hidden_size=None):
        """decoder.py:49-72.  The cell itself is csrc/skinny.hip's fused LSTM epilogue."""
        p = self.params
        if not p.use_lstm:
            raise NotImplementedError("GRUCell decoder: not on the hot path")
        if p.num_layers_dec > 1:
            raise NotImplementedError("MultiRNNCell decoder (num_layers_dec > 1) is outside the hot path")
        return "BasicLSTMCell(%d)" % (p.hidden_size_dec if hidden_size is None else hidden_size)

    def get_state(self, state):
        """decoder.py:74-82: the attention query is the LSTM CELL state c (not h)."""
        if self.params.num_layers_dec > 1:
            state = state[-1]
        return state[0] if self.params.use_lstm else state      # state = (c, h)

    def prepare_decoder_input(self, decoder_inputs, embedding):
        """decoder.py:84-115: which feedback the loop uses.  The embedding lookup itself is
        fused into the lm-cell kernel as a row gather."""
        if self.isTraining:
            return "sample" if self.params.samp_prob > 0 else "teacher"
        return "argmax"

    @abc.abstractmethod
    def __call__(self, decoder_inp, seq_len, encoder_hidden_states, seq_len_inp):
        pass

    def _get_argmax(self, embedding):
        """decoder.py:139-154."""
        def loop_function(logits):
            return embedding[ops.next_token(logits).long()]
        return loop_function

    def _sample_argmax(self, embedding, seed=0):
        """decoder.py:156-180 (tf.multinomial -> Gumbel-max on device)."""
        def loop_function(prev, step=0):
            return embedding[ops.next_token(prev, sample=True, seed=seed, step=step).long()]
        return loop_function

    @classmethod
    def add_parse_options(cls, parser):
        # decoder.py:182-193
        parser.add_argument("-hsize_dec", "--hidden_size_dec", default=256, type=int, help="Hidden size of decoder RNN")
        parser.add_argument("-emb_size", "--emb_size", default=256, type=int, help="Embedding size")
        parser.add_argument("-num_layers_dec", "--num_layers_dec", default=1, type=int, help="Number of RNN layers")
        parser.add_argument("-out_prob_dec", "--out_prob_dec", default=0.9, type=float, help="1 - dropout_prob")
