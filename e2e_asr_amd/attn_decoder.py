"""Attention decoder -- host-side mirror of the reference's attn_decoder.py (18-186).

__call__ keeps the reference signature and output convention (time-major flattened logits
[(T_out*B), V], zero rows for finished utterances); the loop body runs as the HIP step
kernels orchestrated by csrc/decoder.hip.
"""
import numpy as np
import torch

from . import ops
from .decoder import Decoder
from .devcache import dev_i32
from .weights import dec_name


def sampling_coins(seed, stream, step, n):
    """The scheduled-sampling coins of ONE decoder call: n uniforms, one per output step for the whole batch
    (attn_decoder.py:131-133 draws a single tf.random_uniform([]) per step).  A pure function of (seed, stream, step) and
    prefix-stable in n, so data-parallel ranks whose shards end at different longest targets still see the SAME coin at
    every step they share -- a per-process generator consumed t_out draws per call and drifted apart."""
    return np.random.default_rng([int(seed) & 0xFFFFFFFF, int(stream) & 0xFFFFFFFF, int(step) & 0xFFFFFFFF]).random(int(n))


class AttnDecoder(Decoder):
    @classmethod
    def class_params(cls):
        params = super(AttnDecoder, cls).class_params()       # attn_decoder.py:21-28
        params["attention_vec_size"] = 128
        params["lm_hidden_size"] = 256
        params["ind_softmax"] = False
        return params

    def __init__(self, isTraining, params=None, scope=None, variables=None):
        super(AttnDecoder, self).__init__(isTraining=isTraining, params=params, scope=scope, variables=variables)
        self.cell = self.get_cell()
        self.saved = None
        self.multi = None
        if not self.params.use_lstm:                # GRUCell decoder (decoder.py:56-59): host-composed per-step path
            from .gru_decoder import GruDecoderPath
            self.multi = GruDecoderPath(self)
        elif self.params.num_layers_dec > 1:        # MultiRNNCell stacks: host-composed per-step path
            from .multi_decoder import MultiLayerPath
            self.multi = MultiLayerPath(self)
        # scheduled-sampling coin = f(coin_seed, task, coin_step): common to all data-parallel ranks by construction.
        # Seq2SeqModel sets coin_step to its global step before every call; stand-alone use counts calls.
        self.coin_seed = 0
        self.coin_stream = sum(map(ord, self.scope or "")) % 9973
        self.coin_step = 0
        self.last_coin = None

    def draw_coins(self, t_out):
        """One uniform scalar per output step for the whole batch (attn_decoder.py:132), for THIS call."""
        coin = sampling_coins(self.coin_seed, self.coin_stream, self.coin_step, t_out)
        self.coin_step += 1
        self.last_coin = coin
        return coin

    def weight_tensors(self):
        """struct field -> tensor, by TF variable name (beam_search.py:56-98)."""
        v, task = self.variables, self.scope
        out = {}
        for field, leaf in ops.DEC_WEIGHT_LEAVES.items():
            if self.params.ind_softmax and leaf.startswith("rnn/OutputProjection"):
                leaf = leaf.replace("OutputProjection", "OutputProjection2")   # attn_decoder.py:119-122
            t = v.get(dec_name(task, leaf))
            if t is not None and field == "attn_enc_w":
                t = t.reshape(t.shape[-2], t.shape[-1])
            out[field] = t
        return out

    def __call__(self, decoder_inp, seq_len, encoder_hidden_states, seq_len_inp):
        """decoder_inp [T_dec,B] int; seq_len [B] target lengths (host); encoder_hidden_states
        [B,Te,D]; seq_len_inp [B] (host).  Returns logits [(T_out*B),V] -- attn_decoder.py:37-172."""
        p = self.params
        dev = encoder_hidden_states.device
        seq_len = np.asarray(seq_len).astype(np.int64)
        t_out = int(seq_len.max())
        if t_out > decoder_inp.shape[0]:
            raise ValueError("decoder input has %d steps, need %d" % (decoder_inp.shape[0], t_out))
        if p.lm_hidden_size != p.hidden_size_dec and dec_name(self.scope, "rnn/SimpleProjection/kernel") not in self.variables:
            raise ValueError("Could not find SimpleProjection weights for lm_hidden_size != hidden_size_dec")
        mode = self.feedback_mode()          # prepare_decoder_input's choice (decoder.py:100-113); the lookup itself is fused
        coin = self.draw_coins(t_out) if mode == 2 else None
        keep_lm = p.out_prob_dec if self.isTraining else 1.0
        tok = decoder_inp if decoder_inp.dtype == torch.int32 else decoder_inp.to(torch.int32)
        enc = encoder_hidden_states.contiguous()
        enc_len_dev = dev_i32(seq_len_inp, dev)
        if self.multi is not None:
            logits, sv = self.multi.forward(tok.to(dev), seq_len, enc, enc_len_dev, mode, coin, p.samp_prob, keep_lm,
                                            self.rng_seed, t_out)
            self.saved = dict(multi=sv, ws=dict(tok=sv["tok"]), seq_len=seq_len, t_out=t_out, keep_lm=keep_lm, seed=self.rng_seed,
                              enc=enc, enc_len=np.asarray(seq_len_inp), enc_len_dev=enc_len_dev)
            return logits
        logits, ws = ops.attn_decoder_fwd(
            self.weight_tensors(), tok.to(dev), dev_i32(seq_len, dev),
            enc, enc_len_dev, mode=mode, coin=coin, samp_prob=p.samp_prob, keep_lm=keep_lm,
            seed=self.rng_seed, t_out=t_out)
        self.saved = dict(ws=ws, seq_len=seq_len, t_out=t_out, keep_lm=keep_lm, seed=self.rng_seed,
                          enc=enc, enc_len=np.asarray(seq_len_inp), enc_len_dev=enc_len_dev)
        return logits

    def grad_tensors(self):
        """Views of the flat gradient buffer, by the same struct fields as weight_tensors()."""
        v, task = self.variables, self.scope
        v.ensure_grad()
        out = {}
        for field, leaf in ops.DEC_WEIGHT_LEAVES.items():
            if self.params.ind_softmax and leaf.startswith("rnn/OutputProjection"):
                leaf = leaf.replace("OutputProjection", "OutputProjection2")
            name = dec_name(task, leaf)
            out[field] = v.grad_of(name) if name in v else None
        return out

    def backward(self, dlogits, denc, defer_lm=False, side_busy=False):
        """Gradient of __call__: accumulates weight gradients into the flat buffer and the
        encoder-state gradient into denc [B,Te,D].  defer_lm: the LM cell chain's backward is left to backward_lm_tail(),
        which the caller runs behind the encoder's backward pass (Seq2SeqModel.backward).  side_busy: not the first decoder of
        this step (ops.attn_decoder_bwd)."""
        sv = self.saved
        self._lm_tail = None
        if self.multi is not None:
            self.variables.ensure_grad()
            self.multi.backward(sv["multi"], dlogits, denc)
            self.saved = None
            return
        bw = ops.attn_decoder_bwd(self.weight_tensors(), self.grad_tensors(), sv["ws"], sv["enc"], sv["enc_len_dev"],
                                  dlogits, denc, keep_lm=sv["keep_lm"], seed=sv["seed"], defer_lm=defer_lm,
                                  side_busy=side_busy)
        if "_lm_tail" in bw:
            self._lm_tail = bw
        self.saved = None

    def backward_lm_tail(self):
        """The part of backward() that defer_lm left out (no-op otherwise); before ops.side_join()."""
        bw, self._lm_tail = getattr(self, "_lm_tail", None), None
        if bw is not None:
            ops.attn_decoder_bwd_lm(bw)

    @classmethod
    def add_parse_options(cls, parser):
        super(AttnDecoder, cls).add_parse_options(parser)     # attn_decoder.py:174-186
        parser.add_argument("-samp_prob", "--samp_prob", default=0.1, type=float, help="Scheduled sampling probability")
        parser.add_argument("-attn_vec_size", "--attention_vec_size", default=128, type=int, help="Attention vector size")
        parser.add_argument("-lm_hsize", "--lm_hidden_size", default=256, type=int, help="Hidden Size of LM layer")
        parser.add_argument("-ind_softmax", "--ind_softmax", default=False, action="store_true",
                            help="Independent (from LM) softmax params")
