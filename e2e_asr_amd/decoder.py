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

    def __init__(self, isTraining=True, params=None, scope=None, variables=None):
        self.params = self.class_params() if params is None else params
        self.isTraining = isTraining
        self.scope = scope              # task name: variables live under model/rnn_decoder_<scope>/
        self.variables = variables
        self.rng_seed = 0
        if not self.params.use_lstm and int(self.params.num_layers_dec) > 1:
            # decoder.py:56-59 with 66-68: GRU cells in MultiRNNCell stacks are not built (no reference flag reaches either).
            # Refused here, not inside the first call.
            raise ValueError("Decoder: GRU cells (use_lstm=False) in stacks (num_layers_dec > 1) are not built; set "
                             "params.use_lstm = True (decoder.py:34) or num_layers_dec = 1")

    def get_cell(self, hidden_size=None):
        """decoder.py:49-72.  The LSTM cell itself is csrc/skinny.hip's fused epilogue / the persistent decoder kernels; the GRU
        cell is csrc/gru.hip."""
        p = self.params
        size = p.hidden_size_dec if hidden_size is None else hidden_size
        if not p.use_lstm:             # decoder.py:58-59 (runs through e2e_asr_amd/gru_decoder.py)
            return "GRUCell(%d)" % size
        if p.num_layers_dec > 1:       # decoder.py:66-68 (runs through e2e_asr_amd/multi_decoder.py)
            return "MultiRNNCell([BasicLSTMCell(%d)] * %d)" % (size, p.num_layers_dec)
        return "BasicLSTMCell(%d)" % size

    def get_state(self, state):
        """decoder.py:74-82: the attention query is the LSTM CELL state c (not h); with GRU cells the state itself."""
        if self.params.num_layers_dec > 1:
            state = state[-1]
        return state[0] if self.params.use_lstm else state      # state = (c, h)

    def embedding(self):
        from .weights import dec_name
        return self.variables[dec_name(self.scope, "decoder/embedding")]

    def feedback_mode(self):
        """Which feedback the loop uses (decoder.py:100-113): 0 teacher forcing, 1 argmax, 2 scheduled sampling."""
        if self.isTraining:
            return 2 if self.params.samp_prob > 0 else 0
        return 1

    def prepare_decoder_input(self, decoder_inputs):
        """decoder.py:84-115: decoder_inputs [T,B] ids -> (embedded_inp [T,B,E], loop_function).  loop_function is None
        under pure teacher forcing, `_sample_argmax` with scheduled sampling, `_get_argmax` in the inference graph.
        (AttnDecoder.__call__ fuses the lookup into the LM-cell input GEMM as a row gather and the feedback into the
        persistent kernels; it only asks `feedback_mode()`.  This method is the reference's boundary and is what a
        caller composing its own loop uses.)"""
        emb = self.embedding()
        ids = decoder_inputs if torch.is_tensor(decoder_inputs) else torch.as_tensor(decoder_inputs)
        ids = ids.to(emb.device, torch.int32).contiguous()
        embedded_inp = ops.gather_rows(emb, ids.reshape(-1)).view(tuple(ids.shape) + (emb.shape[1],))
        mode = self.feedback_mode()
        if mode == 2:
            print("Scheduled sampling!")
            loop_function = self._sample_argmax(emb)
        elif mode == 0:
            loop_function = None
        else:
            loop_function = self._get_argmax(emb)
        return embedded_inp, loop_function

    @abc.abstractmethod
    def __call__(self, decoder_inp, seq_len, encoder_hidden_states, seq_len_inp):
        pass

    def _get_argmax(self, embedding):
        """decoder.py:139-154: embed the arg-max symbol (first maximum, as tf.argmax / np.argmax)."""
        def loop_function(logits):
            return ops.gather_rows(embedding, ops.next_token(logits))
        return loop_function

    def _sample_argmax(self, embedding):
        """decoder.py:156-180: embed a symbol drawn from softmax(prev) (tf.multinomial -> Gumbel-max on the device with a
        counter-based generator keyed on (rng_seed, call index, row, symbol))."""
        calls = [0]

        def loop_function(prev):
            tok = ops.next_token(prev, sample=True, seed=self.rng_seed, step=calls[0])
            calls[0] += 1
            return ops.gather_rows(embedding, tok)
        return loop_function

    @classmethod
    def add_parse_options(cls, parser):
        # decoder.py:182-193
        parser.add_argument("-hsize_dec", "--hidden_size_dec", default=256, type=int, help="Hidden size of decoder RNN")
        parser.add_argument("-emb_size", "--emb_size", default=256, type=int, help="Embedding size")
        parser.add_argument("-num_layers_dec", "--num_layers_dec", default=1, type=int, help="Number of RNN layers")
        parser.add_argument("-out_prob_dec", "--out_prob_dec", default=0.9, type=float, help="1 - dropout_prob")
